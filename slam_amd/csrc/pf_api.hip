// C ABI of include/slam_pf.h: handle, device memory, stream ordering, argument checks.
// HIP-only by design: there is no CPU path in this library.
#include "../../include/slam_pf.h"
#include "pf_kernels.h"
#include "pf_trig.h"
#include "pf_hostpool.h"
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <string>
#include <algorithm>
#include <atomic>
#include <vector>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

namespace {

std::string g_create_error;

struct EventPair { hipEvent_t a, b; };

// Pinned host staging for an upload the caller repeats every step: two slots, an event behind each
// slot's copies. A call fills the slot whose copies of two calls ago have finished (normally long
// ago) and never waits for the stream itself, so the host can prepare step i+1 while step i runs.
struct Stage {
    char* host[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    size_t cap = 0;
    int flip = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        for (int k = 0; k < 2; ++k) {
            if (ev[k]) { hipError_t e = hipEventSynchronize(ev[k]); if (e != hipSuccess) return e; }
            if (host[k]) (void)hipHostFree(host[k]);
            host[k] = nullptr;
            hipError_t e = hipHostMalloc((void**)&host[k], bytes, hipHostMallocDefault);
            if (e != hipSuccess) { cap = 0; return e; }
            if (!ev[k]) { e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming); if (e != hipSuccess) return e; }
        }
        cap = bytes;
        return hipSuccess;
    }
    // the slot to fill now (its earlier copies have left it)
    hipError_t acquire(size_t bytes, char** out) {
        hipError_t e = reserve(bytes);
        if (e != hipSuccess) return e;
        flip ^= 1;
        e = hipEventSynchronize(ev[flip]);
        *out = host[flip];
        return e;
    }
    hipError_t commit(hipStream_t st) { return hipEventRecord(ev[flip], st); }
    void release() {
        for (int k = 0; k < 2; ++k) {
            if (host[k]) (void)hipHostFree(host[k]);
            if (ev[k]) (void)hipEventDestroy(ev[k]);
            host[k] = nullptr; ev[k] = nullptr;
        }
        cap = 0;
    }
};

}  // namespace

void pfk_set_global_error(const char* text) { g_create_error = text ? text : ""; }

struct pf_handle {
    pf_config cfg;
    hipStream_t stream = nullptr;      // the stream in use (own_stream or the caller's)
    hipStream_t own_stream = nullptr;
    std::string err;
    // sizes
    int P = 0, Pg = 0, offset = 0;
    long long elements = 0;            // cells per grid
    unsigned long long slab = 0;       // bytes between per-particle slabs (0 in shared mode)
    long long local_elements = 0;
    // device state
    double* pose = nullptr;            // [9][P]
    // one GPU: pf_predict_device only remembers the caller's block; the fill stage of the next
    // resample reads the rows from there while it writes `pose` (a 720 KB copy and a launch less
    // per step). Any other reader of `pose` makes the copy first (materialize_pose).
    const double* pose_src = nullptr;
    double* w = nullptr;               // [Pg]
    float* logw = nullptr;             // [Pg]
    double* cdf = nullptr;             // [Pg+1]
    double* scratch = nullptr;         // [Pg+1]
    double* u = nullptr;               // [Pg]
    uint8_t* maps = nullptr;           // shared grid or P slabs
    uint8_t* shadow = nullptr;         // shared grid again, in K1's gather layout (PfSwz), or null
    bool shadow_dirty = true;          // maps changed by something other than K5 since the last build
    PfSwz swz{};
    double* sorted = nullptr;          // [5][P] pose rows in lane order (written with order[])
    int8_t* local = nullptr;           // local int8 map
    char* beam_blk = nullptr;          // one allocation behind the seven beam-list pointers below: pf_set_beams
                                       // packs them into one pinned block and uploads it with ONE copy
    size_t beam_blk_bytes = 0;
    PfBeam* beams = nullptr;           // [max_beams] every beam (fuse path, voxel indices)
    PfBeam* beams_occ = nullptr;       // [max_beams] occupied beams only (fast measure path)
    PfBeamF* beams_occ_f = nullptr;    // [max_beams] the same in fp32
    PfColG* beams_col = nullptr;       // [max_beams] the same grouped by local column (k_measure_col)
    PfTileG* beams_tile = nullptr;     // [max_beams] the same groups as k_measure_tile reads them
    std::vector<PfColG> h_cols, h_sort_tmp;
    std::vector<PfTileG> h_tcols;      // the column groups as k_measure_tile reads them (up to PF_TILE_GROUP beams each; h_cols then
                                       // holds two PfColG per group: the lz of its beams 0-3 and 4-7)
    struct TileGroupH { float lx, ly; int32_t n; int32_t lz[PF_TILE_GROUP]; };
    std::vector<TileGroupH> h_tg, h_tg_tmp;
    int tile_ends[PF_TILE_GROUP] = {0};   // ends of the groups with PF_TILE_GROUP, ..., 1 beams in the sorted list
    std::vector<int32_t> h_deal_perm, h_deal_table;   // pfh_deal_groups' outputs
    int tile_chunks = 1;               // chunks the list in beams_tile was dealt to (its first tile_chunks records are their table)
    int lz_min = 0, lz_max = 0;        // bounds of lz over the occupied beams
    std::vector<unsigned long long> h_sort_keys;
    // k_measure_box: the same groups ordered by cluster, the clusters ordered by blockIdx.y group
    PfColG* beams_colc = nullptr;      // [max_beams]
    PfCluster* clusters = nullptr;     // [max_beams]
    int* group_start = nullptr;        // [257]
    std::vector<PfColG> h_colc;
    std::vector<PfCluster> h_clusters;
    std::vector<int> h_group_start;
    int box_kp = 1, box_parts = 1, box_lp = 1, box_groups = 0;   // pfk_box_plan; groups 0 = no plan
    // k_measure_fuse_cols (per-particle maps): the full beam list as columns and their dense values
    PfFCol* fcols = nullptr;
    uint8_t* fvals = nullptr;
    std::vector<PfFCol> h_fcols;
    std::vector<uint8_t> h_fvals;
    PfFColBounds fbounds{};
    // the same for the occupied beams only (measure without the add)
    PfFCol* mcols = nullptr;
    uint8_t* mvals = nullptr;
    std::vector<PfFCol> h_mcols;
    std::vector<uint8_t> h_mvals;
    PfFColBounds mbounds{};
    int n_mcols = 0;
    int n_fcols = 0;                   // 0: the beam list does not qualify, k_measure_fuse_wave takes the step
    int partial_rows = 1;              // rows of `partial` (chunks of the col kernels / groups of the box kernel)
    int k1_mode = 0;                   // SLAM_PF_K1: 0 auto (= tile), 1 col (brick layout, byte gathers), 2 box (LDS), 3 tile
    int lanes = 0;                     // lanes of K1's order: P, or pfk_padded_lanes(P) with z classes (k_measure_tile)
    bool zsplit = false;               // the lane order keeps one round(z/res) per wave (PfOrderParams::zsplit)
    int n_bmin = 0;                    // entries of bmin: ceil(lanes/64)
    int32_t* order_ext = nullptr;      // [2][PF_EXT_WORDS] extent of the cloud by the last two lane orderings (PfOrderParams::ext)
    int ext_cur = -1;                  // the record the last ordering filled (-1: none yet); the other one is empty
    double yaw_hint = 0.0;             // mean yaw of the last host-side predict: aligns the clusters with the grid
    int n_cols = 0;
    int n_runs = 0;                    // beams_col[0 .. n_runs): run groups (8-cell z windows)
    int col_ends[4] = {0, 0, 0, 0};    // ends of the 4-, 3-, 2-, 1-beam groups in beams_col
    int M_occ = 0;                     // padded to a multiple of PF_BEAM_GROUP
    int M_occ_real = 0;
    double lmax_occ = 0;               // max over occupied beams of max(|lx|+|ly|, |lz|)
    int2* partial = nullptr;           // [max_chunks][P]
    int2* scores = nullptr;            // [P]
    // keyframe ring (pf_ring_*): indices on the device, bookkeeping here
    pf_ring_params ring{};
    int32_t* ring_idx = nullptr;       // [n_slots][slot_capacity]
    int32_t* ring_cnt = nullptr;       // device scalar for kernel-side counts
    double* ring_pts = nullptr;        // [slot_capacity][3] the scan being indexed
    std::vector<int32_t> ring_count, ring_life;
    double ring_pos_update[3] = {0, 0, 0}, ring_euler_update[3] = {0, 0, 0};
    int ring_first_run = 1;
    char* vote_scratch = nullptr;      // pf_scan_vote's device scratch, grown on demand
    size_t vote_cap = 0;
    int32_t* bmin = nullptr;           // [ceil(P/64)] block minima of scores (pfk_reduce_scores)
    int32_t* bmin_ext = nullptr;       // [pfk_block_minima_count(Pg)] the same for scores the caller gathered
    int32_t* draw = nullptr;           // [Pg]
    int32_t* first = nullptr;          // [Pg]
    int32_t* dup_anc = nullptr;        // [Pg]
    int32_t* free_slot = nullptr;      // [Pg]
    int32_t* slot_anc = nullptr;       // [Pg]
    int32_t* best = nullptr;           // inside `counters`' allocation: one D2H copy returns all
    // pinned host memory K4's last kernel writes {counters[3], best} to: two slots, so that one
    // resample may be in flight while the next step is enqueued (pf_resample_async / _result)
    unsigned long long* h_result = nullptr;    // [2] slots of result_slot_u64 words (pfk_result_slot_bytes)
    unsigned long long* d_result = nullptr;    // the same memory as the device sees it
    size_t result_slot_u64 = 0;
    int32_t best_prev = 0;                     // S3P:756: index_high_weight, kept when no weight is above 0
    hipEvent_t rs_event[2] = {nullptr, nullptr};   // recorded behind each slot's resample
    // The caller's thresholds are staged per result slot in pinned, device-mapped memory and K4's
    // search reads them from there (8 B per draw over PCIe, once): no copy is enqueued and the call
    // does not wait for the stream. A copy on the compute stream between two kernels cost 40 us
    // per step. A later call without thresholds first brings the last set into HBM (u), once.
    double* h_u = nullptr;                         // pinned [2][Pg]
    double* d_u_map = nullptr;                     // the same memory as the device sees it
    int u_src = -1;                                // slot of h_u that holds the current set, -1: `u` does
    // from PF_U_PREFETCH_MIN particles on, a fresh set goes to the device on a stream of its own while the step's
    // earlier kernels still run, instead of the search reading it through the mapping (8 B per particle over PCIe
    // inside the kernel: 160 us at a million particles)
    double* u_pre = nullptr;                       // device [2][Pg]
    hipStream_t copy_stream = nullptr;
    hipEvent_t u_event[2] = {nullptr, nullptr};
    unsigned long long rs_issued = 0, rs_done = 0; // resamples enqueued / results handed out
    unsigned long long* counters = nullptr;   // [3]
    PfResampleScratch rs{};            // block-level scratch of K3/K4
    void* rs_mem = nullptr;
    int32_t* vox = nullptr;            // [max_beams]
    int32_t* order = nullptr;          // [P] lane -> particle for K1
    int32_t* order_keys = nullptr;     // [P]
    int32_t* order_hist[2] = {nullptr, nullptr};   // [PF_ORDER_BUCKETS] each, used alternately
    int order_flip = 0;
    bool order_dirty = true;           // poses changed since order[] was built
    bool hist_counted = false;         // order_keys / order_hist[0] already hold the current poses' keys
                                       // (pf_resample's fill stage counted them)
    double* points = nullptr;          // [max_points][3]
    double* wpts = nullptr;            // [max_points][3]
    // the local map's ray march as count + apply (PfUpdateWork::cnt), allocated by the first pf_update_map on it
    unsigned* ray_cnt = nullptr;       // [local_elements + 3], zero between calls
    int32_t* ray_idx = nullptr;        // [max_points]
    bool full_list_resident = true, occ_lists_resident = true;   // false: still in h_beams / h_beams_occ(_f) only
    std::vector<double> h_ysum;        // pf_predict's per-chunk yaw sums and first bad indices
    std::vector<int> h_badc;
    // host staging
    Stage st_pose, st_beams, st_weights, st_scan;   // pinned, double-buffered (pf_predict, pf_set_beams, ...)
    std::unique_ptr<HostPool> pool;                 // host threads for pf_predict's trig (created on first use)
    std::vector<double> h_pose;
    std::vector<PfBeam> h_beams, h_beams_occ;
    std::vector<PfBeamF> h_beams_occ_f;
    int tune_chunks = 0;               // SLAM_PF_CHUNKS (development knob)
    int force_exact_measure = 0;       // SLAM_PF_EXACT_MEASURE=1: fp64-only kernel
    int max_chunks = 1;
    int M = 0;                         // resident beams
    int n_points = 0, n_rays = 0;
    bool measured = false, weighted = false, have_u = false;
    bool weights_given = false;        // pf_set_weights since the last resample: pf_resample may run on them alone
    bool exact = true;
    // The shared-grid update (K5) depends on nothing K3/K4 produce, and nothing but K1, K6 and the
    // map accessors read what it writes. pf_update_map(PF_GRID_SHARED) therefore only records the
    // work; the next pf_weights launches it as extra workgroups of its two kernels (pfk_weights'
    // `carry`), and any call that reads or writes the grid or the scan buffers first runs it on
    // its own (flush_update). Measured before this: the same update on a side stream lost more to
    // the cross-stream waits than the overlap gave (104.9 vs 96.5 us per step).
    PfUpdateWork upd{};
    bool upd_pending = false;
    // RCCL communicator of a sharded handle (pf_comm_init): the two all-gathers of the step are
    // then enqueued by pf_weights / pf_resample themselves, on the handle's stream
    ncclComm_t comm = nullptr;
    int comm_world = 1;
    int2* scores_all = nullptr;        // [Pg]
    int32_t* bmin_all = nullptr;       // [world][ceil(P/64)] every rank's block minima (rides in the group)
    // The pose blocks pf_resample needs are final once the caller has predicted, which the
    // reference does before the weights (S3P:911-914 precede S3P:920): pf_weights then issues both
    // all-gathers as one RCCL group (one launch, one latency) and pf_resample finds the blocks
    // there. A predict between the two calls clears the flag and pf_resample gathers itself.
    // SLAM_PF_TWO_GATHERS=1: always one collective per call, as before.
    bool poses_gathered = false;
    int comm_rank = 0;
    std::vector<int32_t> h_slots, mg_src, mg_speer, mg_dst, mg_rpeer;   // map migration (per-particle maps, sharded)
    std::vector<int64_t> mg_sseq, mg_rseq;
    bool group_gathers = true;
    int virtual_shards = 0;            // SLAM_PF_VIRTUAL_SHARDS (test switch, world 1 only): the shard's slabs move between
                                       // this many equal blocks of slots through the send/receive groups, peer = this rank
    double* poses_all = nullptr;       // [world][9][P]
    // profiling
    bool profile = false;
    unsigned profile_mask = ~0u;       // bit k: time kernel class k
    std::vector<EventPair> ev[PF_K_COUNT];
    std::vector<EventPair> ev_pool;
};

namespace {

int fail(pf_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

// particles (all ranks') from which fresh thresholds are copied ahead on their own stream. (One rank of eight, rehearsed: a
// million particles 1.35 -> 1.24 ms per step; at 80 k, where the host is barely ahead of a 0.1 ms step, the copy and its
// event came out slower than the search reading through the mapping, 0.125 against 0.108 ms.)
#define PF_U_PREFETCH_MIN 262144
#define HIP_TRY(h, expr)                                                                    \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(h, PF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                \
    } while (0)

#define NEED(h, cond, msg)                                                                  \
    do { if (!(cond)) return fail(h, PF_ERR_ARG, "%s: %s", __func__, msg); } while (0)

template <typename T>
hipError_t dalloc(T** p, size_t n) {
    return hipMalloc((void**)p, (n ? n : 1) * sizeof(T));
}

unsigned long long round_up(unsigned long long v, unsigned long long a) { return (v + a - 1) / a * a; }

PfGridParams grid_params(const pf_handle* h) {
    PfGridParams gp;
    gp.gyz = (double)h->cfg.grid_y * (double)h->cfg.grid_z;
    gp.gzd = (double)h->cfg.grid_z;
    gp.elements = (double)h->elements;
    gp.occ_threshold = h->cfg.occ_threshold;
    gp.lo = h->cfg.map_lo;
    gp.hi = h->cfg.map_hi;
    return gp;
}

// The poses the last pf_resample counted K1's order keys for are being replaced.
hipError_t drop_counted_keys(pf_handle* h) {
    if (!h->hist_counted) return hipSuccess;
    h->hist_counted = false;
    return hipMemsetAsync(h->order_hist[0], 0, sizeof(int32_t) * 4096, h->stream);   // only the 12-bit keys are counted ahead
}

// RCCL is bound at run time (dlopen), so the library loads on hosts without it and a one-GPU
// caller never touches it.
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl& rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) return r;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    r.Send = (decltype(r.Send))dlsym(r.lib, "ncclSend");
    r.Recv = (decltype(r.Recv))dlsym(r.lib, "ncclRecv");
    r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GetErrorString;
    return r;
}

struct Timed {
    pf_handle* h; int id; hipStream_t st; EventPair ep{}; bool on = false;
    Timed(pf_handle* h_, int id_, hipStream_t st_ = nullptr) : h(h_), id(id_), st(st_ ? st_ : h_->stream) {
        if (!h->profile || !((h->profile_mask >> id) & 1u)) return;
        if (!h->ev_pool.empty()) { ep = h->ev_pool.back(); h->ev_pool.pop_back(); }
        else if (hipEventCreate(&ep.a) != hipSuccess || hipEventCreate(&ep.b) != hipSuccess) return;
        on = hipEventRecord(ep.a, st) == hipSuccess;
    }
    ~Timed() {
        if (!on) return;
        (void)hipEventRecord(ep.b, st);
        h->ev[id].push_back(ep);
    }
};

hipError_t materialize_pose(pf_handle* h) {
    if (!h->pose_src) return hipSuccess;
    const double* src = h->pose_src;
    h->pose_src = nullptr;
    return hipMemcpyAsync(h->pose, src, sizeof(double) * PF_POSE_ROWS * (size_t)h->P, hipMemcpyDeviceToDevice,
                          h->stream);
}

// Runs a recorded shared-grid update now, on its own (see pf_handle::upd).
hipError_t flush_update(pf_handle* h) {
    if (!h->upd_pending) return hipSuccess;
    h->upd_pending = false;
    Timed t(h, PF_K_UPDATE);
    hipError_t e = pfk_update_hits(h->stream, h->upd);
    if (e == hipSuccess) e = pfk_update_rays(h->stream, h->upd);
    return e;
}
hipError_t wait_map_writes(pf_handle* h) { return flush_update(h); }

// Fills the pinned pose block; returns the index of the first particle whose pose is not finite or has
// |pos/res| >= 2^24 (indices would no longer be exact), or -1.
int upload_pose_rows(pf_handle* h, double* hp, const double* x, const double* y, const double* z,
                     const double* yaw) {
    const int P = h->P;
    const double res = h->cfg.resolution;
    const double lim = 16777216.0 * res;
    // chunks of 512 particles; the yaw sums are kept per chunk and added in chunk order (the same mean whoever ran what)
    const int CH = 512, nch = (P + CH - 1) / CH;
    std::vector<double>& ysum_c = h->h_ysum; ysum_c.assign((size_t)nch, 0.0);
    std::vector<int>& bad_c = h->h_badc; bad_c.assign((size_t)nch, -1);
    auto rows = [&, hp, x, y, z, yaw, P, res, lim](int c) {
        const int lo = c * CH, hi = std::min(P, lo + CH);
        double ys = 0;
        int bad = -1;
        for (int i = lo; i < hi; ++i) {
            if (bad < 0 && (!(fabs(x[i]) < lim && fabs(y[i]) < lim && fabs(z[i]) < lim) || !isfinite(yaw[i]))) bad = i;
            hp[0 * (size_t)P + i] = x[i];
            hp[1 * (size_t)P + i] = y[i];
            hp[2 * (size_t)P + i] = z[i];
            hp[3 * (size_t)P + i] = yaw[i];
            const PfSinCos t(yaw[i]);                     // S3P:860 via rot_matrix (pf_trig.h)
            hp[4 * (size_t)P + i] = t.c;
            hp[5 * (size_t)P + i] = t.s;
            hp[6 * (size_t)P + i] = x[i] * 1 / res;       // S3P:862-864: a division
            hp[7 * (size_t)P + i] = y[i] * 1 / res;
            hp[8 * (size_t)P + i] = z[i] * 1 / res;
            ys += yaw[i];
        }
        ysum_c[(size_t)c] = ys; bad_c[(size_t)c] = bad;
    };
    // the trig is ~30 ns per particle on one core: split it over a few host threads from a few
    // thousand particles on (the same libm call per element, so the same bits)
    bool done = false;
    if (P >= 4096) {
        if (!h->pool) {
            int n = (int)std::thread::hardware_concurrency();
            if (const char* e = getenv("OMP_NUM_THREADS")) { const int v = atoi(e); if (v > 0) n = v; }
            n = n < 2 ? 0 : (n > 16 ? 16 : n) - 1;
            if (const char* e = getenv("SLAM_PF_HOST_THREADS")) n = atoi(e) > 0 ? atoi(e) - 1 : 0;
            h->pool.reset(new HostPool(n));
        }
        if (h->pool->size() > 0) {
            h->pool->run_chunks(nch, rows);
            done = true;
        }
    }
    if (!done) for (int c = 0; c < nch; ++c) rows(c);
    double ysum = 0;
    int bad = -1;
    for (int c = 0; c < nch; ++c) {
        ysum += ysum_c[(size_t)c];
        if (bad < 0 && bad_c[(size_t)c] >= 0) bad = bad_c[(size_t)c];
    }
    h->yaw_hint = ysum / P;
    return bad;
}

bool box_eligible(const pf_handle* h);
// Clusters of beams for k_measure_box: recursive median splits of the occupied beams along their
// longest extent (in a frame turned by the cloud's mean yaw, so that wall pieces are thin boxes of
// the grid) until the box a cluster needs -- its extent plus the margin a particle cloud typically
// adds -- fits the LDS buffer with room to spare; inside a cluster the beams of one local column
// form groups of up to four (x and y are evaluated once per group); then the clusters are dealt
// to the blockIdx.y groups, largest first onto the least loaded one. SLAM_PF_BOX=volume,mx,my,mz
// overrides the budget (cells) and the margins (development knob). Any clustering gives the same
// results; it only decides how much a workgroup stages per evaluation.
// A beam list as local columns for k_measure_fuse_cols: maximal runs that share (lx, ly) and ascend
// strictly in lz (the order S3P:868-877 produces), cut at PF_FCOL_MAXZ cells. Returns false (outputs
// empty) when the list does not qualify: a zero value (0 means "no beam" in the values), or fewer than
// min_per_col beams per column on average. A column costs a 16-byte load (and store) per piece whatever
// it holds: on a sparse list (one beam per column: configs[3]'s 2048-beam scan, 2.1 ms against 1.5 ms
// with the add) the beam-by-beam kernels are faster, on the reference's dense local maps (35 beams per
// column) the column form is 3.4 times faster.
bool columns_of(const PfBeam* B, int n, int occ_threshold, int min_per_col, bool forced, std::vector<PfFCol>& C,
                std::vector<uint8_t>& V, PfFColBounds& bd) {
    C.clear(); V.clear();
    if (n <= 0) return false;
    std::vector<int> first;                       // first beam of each column
    bd = PfFColBounds{(float)B[0].lx, (float)B[0].lx, (float)B[0].ly, (float)B[0].ly, (int32_t)B[0].lz, (int32_t)B[0].lz};
    for (int j = 0; j < n;) {
        if (B[j].v == 0) { C.clear(); return false; }
        PfFCol c{};
        c.lx = (float)B[j].lx; c.ly = (float)B[j].ly;        // exact: |offset| <= 2^20
        c.z0 = (int32_t)B[j].lz;
        int k = j + 1;
        while (k < n && B[k].lx == B[j].lx && B[k].ly == B[j].ly && B[k].v != 0 && B[k].lz > B[k - 1].lz &&
               (int32_t)B[k].lz - c.z0 < PF_FCOL_MAXZ)
            ++k;
        c.n = (int32_t)B[k - 1].lz - c.z0 + 1;
        bd.lx0 = fminf(bd.lx0, c.lx); bd.lx1 = fmaxf(bd.lx1, c.lx); bd.ly0 = fminf(bd.ly0, c.ly); bd.ly1 = fmaxf(bd.ly1, c.ly);
        bd.z0 = std::min(bd.z0, c.z0); bd.z1 = std::max(bd.z1, c.z0 + c.n - 1);
        C.push_back(c); first.push_back(j);
        j = k;
    }
    first.push_back(n);
    if (!forced && (size_t)n < (size_t)min_per_col * C.size()) { C.clear(); return false; }
    // values, columns in list order (the kernel's election keeps the list's order per cell)
    V.assign(16, 0);
    for (size_t i = 0; i < C.size(); ++i) {
        PfFCol& c = C[i];
        const int e = c.z0 & 15;
        const size_t base = V.size();
        c.voff = (int32_t)(base / 16);
        V.resize(base + (((size_t)(e + c.n) + 15) & ~(size_t)15) + 16, 0);     // the run's units and the zero unit behind them
        for (int q = first[i]; q < first[i + 1]; ++q) {
            V[base + (size_t)(e + (int32_t)B[q].lz - c.z0)] = (uint8_t)(int8_t)B[q].v;
            if (B[q].v > occ_threshold) ++c.cnt;
        }
    }
    V.resize(V.size() + 48, 0);                                 // the last column's loads run up to 31 bytes past its zero unit
    return true;
}

// The column forms of the step's list for per-particle maps: every beam (measure + fuse) and the occupied
// beams only (measure alone). Left empty when the kernel's conditions do not hold: the step then runs
// beam by beam (k_measure_fuse_wave, k_measure_fast).
void build_fuse_columns(pf_handle* h, int n_real_occ) {
    h->h_fcols.clear(); h->h_fvals.clear(); h->n_fcols = 0;
    h->h_mcols.clear(); h->h_mvals.clear(); h->n_mcols = 0;
    // SLAM_PF_FUSE_COLS: 0 never, 1 whenever the kernel's conditions hold, unset: when the list is dense enough
    const char* knob = getenv("SLAM_PF_FUSE_COLS");
    const bool off = knob && atoi(knob) == 0, forced = knob && atoi(knob) != 0;
    if (off || h->cfg.map_mode != PF_MAP_PER_PARTICLE || h->cfg.grid_z % 16 != 0 || h->slab % 16 != 0 ||
        h->elements >= 2147483648ll)
        return;
    if (columns_of(h->h_beams.data(), (int)h->h_beams.size(), h->cfg.occ_threshold, 4, forced, h->h_fcols, h->h_fvals, h->fbounds))
        h->n_fcols = (int)h->h_fcols.size();
    if (columns_of(h->h_beams_occ.data(), n_real_occ, h->cfg.occ_threshold, 2, forced, h->h_mcols, h->h_mvals, h->mbounds))
        h->n_mcols = (int)h->h_mcols.size();
}

void build_clusters(pf_handle* h, int n_occ) {
    h->box_groups = 0;
    h->h_colc.clear(); h->h_clusters.clear(); h->h_group_start.clear();
    if (n_occ == 0 || !box_eligible(h)) return;
    double vol = 20000, mx = 12, my = 12, mz = 6;
    if (const char* e = getenv("SLAM_PF_BOX")) (void)sscanf(e, "%lf,%lf,%lf,%lf", &vol, &mx, &my, &mz);
    const PfSinCos t0(h->yaw_hint);
    const double c0 = t0.c, s0 = t0.s;
    struct Ref { float u, v, z; int idx; };
    std::vector<Ref> r((size_t)n_occ);
    for (int i = 0; i < n_occ; ++i) {
        const PfBeam& b = h->h_beams_occ[(size_t)i];
        r[(size_t)i] = Ref{(float)(c0 * b.lx - s0 * b.ly), (float)(s0 * b.lx + c0 * b.ly), (float)b.lz, i};
    }
    struct Span { int a, b; };
    std::vector<Span> todo{{0, n_occ}}, done;
    while (!todo.empty()) {
        const Span sp = todo.back(); todo.pop_back();
        float u0 = r[(size_t)sp.a].u, u1 = u0, v0 = r[(size_t)sp.a].v, v1 = v0, z0 = r[(size_t)sp.a].z, z1 = z0;
        for (int i = sp.a + 1; i < sp.b; ++i) {
            const Ref& q = r[(size_t)i];
            u0 = std::min(u0, q.u); u1 = std::max(u1, q.u); v0 = std::min(v0, q.v); v1 = std::max(v1, q.v);
            z0 = std::min(z0, q.z); z1 = std::max(z1, q.z);
        }
        const double eu = u1 - u0, ev = v1 - v0, ez = z1 - z0;
        // (at most 256 column groups per cluster, BOX_COLCAP: every beam could be a group of its own)
        if (sp.b - sp.a <= 1 || ((eu + mx) * (ev + my) * (ez + mz) <= vol && sp.b - sp.a <= 256)) { done.push_back(sp); continue; }
        const int mid = (sp.a + sp.b) / 2;
        auto first = r.begin() + sp.a, last = r.begin() + sp.b, nth = r.begin() + mid;
        if (eu >= ev && eu >= ez) std::nth_element(first, nth, last, [](const Ref& a, const Ref& b) { return a.u < b.u; });
        else if (ev >= ez) std::nth_element(first, nth, last, [](const Ref& a, const Ref& b) { return a.v < b.v; });
        else std::nth_element(first, nth, last, [](const Ref& a, const Ref& b) { return a.z < b.z; });
        todo.push_back(Span{sp.a, mid}); todo.push_back(Span{mid, sp.b});
    }
    const int nc = (int)done.size();
    pfk_box_plan(h->P, nc, &h->box_kp, &h->box_parts, &h->box_lp, &h->box_groups);
    // largest cluster first onto the least loaded group
    std::vector<int> by_size((size_t)nc);
    for (int c = 0; c < nc; ++c) by_size[(size_t)c] = c;
    auto size_of = [&](int c) { return done[(size_t)c].b - done[(size_t)c].a; };
    std::sort(by_size.begin(), by_size.end(), [&](int a, int b) { return size_of(a) != size_of(b) ? size_of(a) > size_of(b) : a < b; });
    const int G = h->box_groups;
    std::vector<long long> load((size_t)G, 0);
    std::vector<std::vector<int>> members((size_t)G);
    for (int c : by_size) {
        int best = 0;
        for (int g = 1; g < G; ++g) if (load[(size_t)g] < load[(size_t)best]) best = g;
        load[(size_t)best] += size_of(c) + 4;               // a cluster costs a staging round besides its beams
        members[(size_t)best].push_back(c);
    }
    std::vector<int> ids;
    h->h_group_start.push_back(0);
    for (int g = 0; g < G; ++g) {
        for (int c : members[(size_t)g]) {
            ids.clear();
            for (int i = done[(size_t)c].a; i < done[(size_t)c].b; ++i) ids.push_back(r[(size_t)i].idx);
            std::sort(ids.begin(), ids.end());              // the list's own order: x, y, z ascending (S3P:868)
            PfCluster pc{};
            pc.g0 = (int)h->h_colc.size();
            pc.lx0 = pc.ly0 = 3.0e38f; pc.lx1 = pc.ly1 = -3.0e38f; pc.lz0 = INT32_MAX; pc.lz1 = INT32_MIN;
            for (size_t j = 0; j < ids.size();) {
                const PfBeam& b0 = h->h_beams_occ[(size_t)ids[j]];
                PfColG q{};
                q.lx = (float)b0.lx; q.ly = (float)b0.ly; q.n = 0;
                while (j < ids.size() && q.n < 4 && h->h_beams_occ[(size_t)ids[j]].lx == b0.lx &&
                       h->h_beams_occ[(size_t)ids[j]].ly == b0.ly)
                    q.lz[q.n++] = (int32_t)h->h_beams_occ[(size_t)ids[j++]].lz;
                std::sort(q.lz, q.lz + q.n);
                for (int k = q.n; k < 4; ++k) q.lz[k] = q.lz[q.n - 1];
                h->h_colc.push_back(q);
                pc.lx0 = std::min(pc.lx0, q.lx); pc.lx1 = std::max(pc.lx1, q.lx);
                pc.ly0 = std::min(pc.ly0, q.ly); pc.ly1 = std::max(pc.ly1, q.ly);
                pc.lz0 = std::min(pc.lz0, q.lz[0]); pc.lz1 = std::max(pc.lz1, q.lz[q.n - 1]);
            }
            pc.g1 = (int)h->h_colc.size();
            h->h_clusters.push_back(pc);
        }
        h->h_group_start.push_back((int)h->h_clusters.size());
    }
    if (getenv("SLAM_PF_BOX_STATS"))
        fprintf(stderr, "build_clusters: %d occupied beams -> %d clusters, %zu column groups, %d groups\n", n_occ, nc,
                h->h_colc.size(), G);
}

}  // namespace

extern "C" {

int pf_default_config(pf_config* c) {
    if (!c) return PF_ERR_ARG;
    memset(c, 0, sizeof *c);
    c->struct_size = (int32_t)sizeof(pf_config);
    c->device = 0;
    c->n_particles = 50; c->n_particles_global = 50; c->particle_offset = 0;
    c->map_mode = PF_MAP_PER_PARTICLE;
    c->grid_x = 300; c->grid_y = 300; c->grid_z = 80;
    c->local_x = 200; c->local_y = 200; c->local_z = 80;
    c->max_beams = 1 << 17; c->max_points = 19200;
    c->occ_threshold = 20; c->map_lo = 0; c->map_hi = 100; c->map_prior = 50;
    c->order_mode = PF_ORDER_AUTO;
    c->resolution = 0.05;
    return PF_OK;
}

const char* pf_last_error(const pf_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

void pf_destroy(pf_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void* ptrs[] = {h->ring_idx, h->ring_cnt, h->ring_pts, h->vote_scratch, h->bmin, h->bmin_ext, h->bmin_all, h->shadow, h->sorted, h->pose, h->w, h->logw, h->cdf, h->scratch, h->u, h->maps, h->local, h->beam_blk,
                    h->partial, h->scores, h->draw, h->first, h->dup_anc, h->free_slot,
                    h->slot_anc, h->counters, h->rs_mem, h->vox, h->order, h->order_keys, h->order_ext, h->order_hist[0], h->order_hist[1], h->points, h->wpts, h->ray_cnt};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& v : h->ev) for (auto& e : v) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto& e : h->ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (h->comm && rccl().ok) (void)rccl().CommDestroy(h->comm);
    if (h->scores_all) (void)hipFree(h->scores_all);
    if (h->poses_all) (void)hipFree(h->poses_all);
    for (hipEvent_t e : h->rs_event) if (e) (void)hipEventDestroy(e);
    if (h->h_result) (void)hipHostFree(h->h_result);
    if (h->h_u) (void)hipHostFree(h->h_u);
    if (h->copy_stream) { (void)hipStreamSynchronize(h->copy_stream); (void)hipStreamDestroy(h->copy_stream); }
    for (hipEvent_t e : h->u_event) if (e) (void)hipEventDestroy(e);
    if (h->u_pre) (void)hipFree(h->u_pre);
    for (Stage* s : {&h->st_pose, &h->st_beams, &h->st_weights, &h->st_scan}) s->release();
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

int pf_create(const pf_config* cfg, pf_handle** out) {
    if (!cfg || !out) return fail(nullptr, PF_ERR_ARG, "pf_create: null argument");
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(pf_config))
        return fail(nullptr, PF_ERR_ARG, "pf_create: struct_size %d != %zu", cfg->struct_size,
                    sizeof(pf_config));
    const pf_config& c = *cfg;
    if (c.n_particles < 1 || c.n_particles_global < c.n_particles || c.particle_offset < 0 ||
        (long long)c.particle_offset + c.n_particles > c.n_particles_global)
        return fail(nullptr, PF_ERR_ARG, "pf_create: particle counts/offset inconsistent");
    if (c.n_particles_global != c.n_particles && c.n_particles_global % c.n_particles != 0)
        return fail(nullptr, PF_ERR_ARG, "pf_create: shards must be equal (n_global %% n_local != 0)");
    if (c.grid_x < 1 || c.grid_y < 1 || c.grid_z < 1 ||
        (long long)c.grid_x * c.grid_y * c.grid_z >= 2147483647LL ||
        (long long)c.grid_y * c.grid_z >= (1LL << 26))
        return fail(nullptr, PF_ERR_ARG, "pf_create: grid dims out of range");
    if (c.local_x < 0 || c.local_y < 0 || c.local_z < 0 ||
        (long long)c.local_x * c.local_y * c.local_z >= 2147483647LL)
        return fail(nullptr, PF_ERR_ARG, "pf_create: local dims out of range");
    if (c.map_mode != PF_MAP_SHARED && c.map_mode != PF_MAP_PER_PARTICLE)
        return fail(nullptr, PF_ERR_ARG, "pf_create: map_mode");
    if (!(c.resolution > 0.0) || c.max_beams < 1 || c.max_points < 0 || c.map_lo < 0 ||
        c.map_hi > 255 || c.map_lo > c.map_hi || c.map_prior < 0 || c.map_prior > 255)
        return fail(nullptr, PF_ERR_ARG, "pf_create: resolution/capacities/clamps");
    if (c.order_mode < PF_ORDER_AUTO || c.order_mode > PF_ORDER_FAST)
        return fail(nullptr, PF_ERR_ARG, "pf_create: order_mode");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, PF_ERR_NODEVICE, "pf_create: no HIP device (this library has no CPU path)");
    if (c.device < 0 || c.device >= ndev)
        return fail(nullptr, PF_ERR_NODEVICE, "pf_create: device %d of %d", c.device, ndev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c.device) != hipSuccess)
        return fail(nullptr, PF_ERR_HIP, "pf_create: hipGetDeviceProperties");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, PF_ERR_NODEVICE, "pf_create: device is %s, kernels are gfx950 only",
                    prop.gcnArchName);

    pf_handle* h = new (std::nothrow) pf_handle;
    if (!h) return fail(nullptr, PF_ERR_NOMEM, "pf_create: host allocation");
    h->cfg = c;
    h->P = c.n_particles; h->Pg = c.n_particles_global; h->offset = c.particle_offset;
    h->elements = (long long)c.grid_x * c.grid_y * c.grid_z;
    h->local_elements = (long long)c.local_x * c.local_y * c.local_z;
    h->slab = c.map_mode == PF_MAP_PER_PARTICLE ? round_up((unsigned long long)h->elements, 256) : 0;
    h->exact = c.order_mode == PF_ORDER_EXACT || (c.order_mode == PF_ORDER_AUTO && h->Pg <= 2048);
    if (const char* e = getenv("SLAM_PF_K1")) h->k1_mode = !strcmp(e, "col") ? 1 : !strcmp(e, "box") ? 2 : !strcmp(e, "tile") ? 3 : 0;
    if (const char* e = getenv("SLAM_PF_EXACT_MEASURE")) h->force_exact_measure = atoi(e);
    // K1's copy of the shared grid: tiles (k_measure_tile, the default) when gz*a fits 32-bit byte offsets,
    // else / on request bricks (k_measure_col)
    int layout = -1;
    unsigned long long lay_a = 0, lay_b = 0, lay_bytes = 0;
    if (c.map_mode == PF_MAP_SHARED && h->elements < (1ll << 31) && !getenv("SLAM_PF_NO_GATHER_LAYOUT")) {
        const unsigned long long gx = c.grid_x, gy = c.grid_y, gz = c.grid_z;
        const unsigned long long ntx = (gx + 7) / 8 + 2, nty = (gy + 7) / 8 + 2;
        const unsigned long long tb = ntx * 64, ta = nty * tb;
        const unsigned long long nx4 = (gx + 3) / 4, ny4 = (gy + 3) / 4, nz4 = (gz + 3) / 4;
        const unsigned long long sa = ny4 * nz4 * 64, sb = nz4 * 64, sbytes = nx4 * sa;
        const bool bricks_fit = sa < (1ull << 24) && nx4 < (1ull << 24) && ny4 < (1ull << 24) && sbytes < (1ull << 32);
        if ((h->k1_mode == 0 || h->k1_mode == 3) && gz * ta < (1ull << 32)) { layout = PF_LAYOUT_TILES; lay_a = ta; lay_b = tb; lay_bytes = gz * ta; }
        else if (bricks_fit) { layout = PF_LAYOUT_BRICKS; lay_a = sa; lay_b = sb; lay_bytes = sbytes; }
    }
    h->zsplit = layout == PF_LAYOUT_TILES && !h->force_exact_measure;
    h->lanes = h->zsplit ? pfk_padded_lanes(h->P) : h->P;
    h->n_bmin = (h->lanes + 63) / 64;
    // beam chunks: enough waves to fill 256 CUs x 4 SIMDs several times over
    {
        // at most 2048 blocks of 256 threads = one full residency of 256 CUs x 8 blocks: every
        // block does the same amount of work, so a second, partly filled round would be a tail
        const int blocks_x = (h->lanes + 255) / 256;
        int ch = 2048 / blocks_x;
        if (ch < 1) ch = 1;
        if (ch > 1024) ch = 1024;
        h->max_chunks = ch;
    }
#define CREATE_TRY(expr)                                                                     \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            int rc_ = fail(nullptr, e_ == hipErrorOutOfMemory ? PF_ERR_NOMEM : PF_ERR_HIP,   \
                           "pf_create: %s failed: %s", #expr, hipGetErrorString(e_));        \
            pf_destroy(h);                                                                   \
            return rc_;                                                                      \
        }                                                                                    \
    } while (0)
    CREATE_TRY(hipSetDevice(c.device));
    CREATE_TRY(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    const size_t P = h->P, Pg = h->Pg;
    CREATE_TRY(dalloc(&h->pose, PF_POSE_ROWS * P));
    CREATE_TRY(dalloc(&h->w, Pg));
    CREATE_TRY(dalloc(&h->logw, Pg));
    CREATE_TRY(dalloc(&h->cdf, Pg + 1 + (Pg + 255) / 256));   // + the CDF at the end of every 256-particle block (k_search_t's table)
    CREATE_TRY(dalloc(&h->scratch, Pg + 1));
    CREATE_TRY(dalloc(&h->u, Pg));
    const unsigned long long map_bytes =
        c.map_mode == PF_MAP_PER_PARTICLE ? h->slab * P : round_up((unsigned long long)h->elements, 256);
    CREATE_TRY(dalloc(&h->maps, (size_t)map_bytes));
    CREATE_TRY(dalloc(&h->sorted, 5 * (size_t)h->lanes));
    if (layout >= 0) {
        // gather-layout copy for K1 (any grid up to 1024^3 has one); otherwise K1 reads the reference order
        CREATE_TRY(dalloc(&h->shadow, (size_t)round_up(lay_bytes, 256)));
        CREATE_TRY(hipMemsetAsync(h->shadow, 0, (size_t)round_up(lay_bytes, 256), h->stream));
        h->swz = PfSwz{(uint32_t)c.grid_x, (uint32_t)c.grid_y, (uint32_t)c.grid_z, (uint32_t)lay_a, (uint32_t)lay_b, (uint32_t)layout};
    }
    if (h->local_elements)
        CREATE_TRY(dalloc(&h->local, (size_t)round_up((unsigned long long)h->local_elements, 256)));
    {
        const size_t mb = (size_t)c.max_beams;
        h->beam_blk_bytes = 64 * 9 + sizeof(PfBeam) * mb + sizeof(PfTileG) * (mb + 1 + 1024) + sizeof(PfColG) * 2 * (mb + 1) + sizeof(PfBeam) * (mb + PF_BEAM_GROUP) +
                            sizeof(PfBeamF) * (mb + PF_BEAM_GROUP) + sizeof(PfColG) * (mb + 1) + sizeof(PfCluster) * (mb + 1) +
                            sizeof(int) * 257;
        if (c.map_mode == PF_MAP_PER_PARTICLE)      // worst case, every beam a column of its own: 32 B + two units
            h->beam_blk_bytes += 2 * (64 * 2 + sizeof(PfFCol) * mb + 32 * mb + 64);
        CREATE_TRY(hipMalloc((void**)&h->beam_blk, h->beam_blk_bytes));
        h->beams = (PfBeam*)h->beam_blk;       // (real offsets are set by every pf_set_beams)
    }
    if (const char* e = getenv("SLAM_PF_TWO_GATHERS")) h->group_gathers = atoi(e) == 0;
    if (const char* e = getenv("SLAM_PF_CHUNKS")) h->tune_chunks = atoi(e);
    if (h->tune_chunks > h->max_chunks) h->max_chunks = h->tune_chunks < 1024 ? h->tune_chunks : 1024;
    h->partial_rows = std::max(h->max_chunks, 256 / (((int)P + 4095) / 4096));
    CREATE_TRY(dalloc(&h->partial, (size_t)h->partial_rows * (size_t)h->lanes));
    CREATE_TRY(dalloc(&h->scores, P));
    CREATE_TRY(dalloc(&h->bmin, (size_t)h->n_bmin));
    CREATE_TRY(dalloc(&h->bmin_ext, (size_t)pfk_block_minima_count((int)Pg)));
    CREATE_TRY(dalloc(&h->draw, Pg));
    CREATE_TRY(dalloc(&h->first, Pg));
    CREATE_TRY(dalloc(&h->dup_anc, Pg));
    CREATE_TRY(dalloc(&h->free_slot, Pg));
    CREATE_TRY(dalloc(&h->slot_anc, Pg));
    CREATE_TRY(dalloc(&h->counters, 4));            // [0..2] counters, [3] holds `best`
    h->best = (int32_t*)(h->counters + 3);
    h->result_slot_u64 = pfk_result_slot_bytes(h->Pg) / sizeof(unsigned long long);
    CREATE_TRY(hipHostMalloc((void**)&h->h_result, 2 * pfk_result_slot_bytes(h->Pg), hipHostMallocMapped));
    memset(h->h_result, 0, 2 * pfk_result_slot_bytes(h->Pg));
    CREATE_TRY(hipHostGetDevicePointer((void**)&h->d_result, h->h_result, 0));
    CREATE_TRY(hipHostMalloc((void**)&h->h_u, 2 * sizeof(double) * Pg, hipHostMallocMapped));
    CREATE_TRY(hipHostGetDevicePointer((void**)&h->d_u_map, h->h_u, 0));
    // SLAM_PF_U_PREFETCH=0: the search reads fresh thresholds through the mapping at every size (comparison)
    if (Pg >= PF_U_PREFETCH_MIN && !(getenv("SLAM_PF_U_PREFETCH") && atoi(getenv("SLAM_PF_U_PREFETCH")) == 0)) {
        CREATE_TRY(dalloc(&h->u_pre, 2 * Pg));
        CREATE_TRY(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        for (hipEvent_t& e : h->u_event) CREATE_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    CREATE_TRY(hipEventCreateWithFlags(&h->rs_event[0], hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&h->rs_event[1], hipEventDisableTiming));
    {
        // one allocation, carved: 6 double arrays of nb (+8 for prm), 2 int2 arrays, 1 int array, ndup
        const size_t nb = (Pg + 255) / 256;
        const size_t seg_bytes = 128 * ((size_t)Pg / 8192 + 6);
        const size_t bytes = seg_bytes + sizeof(double) * (8 + 5 * nb) + sizeof(int2) * 2 * nb + sizeof(int32_t) * (nb + 4);
        CREATE_TRY(hipMalloc(&h->rs_mem, bytes));
        char* q = (char*)h->rs_mem;
        h->rs.segs = (void*)q; q += seg_bytes;
        h->rs.prm = (double*)q; q += sizeof(double) * 8;
        h->rs.bsum = (double*)q; q += sizeof(double) * nb;
        h->rs.bmax = (double*)q; q += sizeof(double) * nb;
        h->rs.bse = (double*)q; q += sizeof(double) * nb;
        h->rs.boff = (double*)q; q += sizeof(double) * nb;
        h->rs.bbest_v = (double*)q; q += sizeof(double) * nb;
        h->rs.bcnt = (int2*)q; q += sizeof(int2) * nb;
        h->rs.boff2 = (int2*)q; q += sizeof(int2) * nb;
        h->rs.bbest_i = (int32_t*)q; q += sizeof(int32_t) * nb;
        h->rs.ndup = (int32_t*)q;
    }
    CREATE_TRY(dalloc(&h->vox, (size_t)c.max_beams));
    CREATE_TRY(dalloc(&h->order, (size_t)h->lanes));
    CREATE_TRY(dalloc(&h->order_keys, P));
    {
        int32_t empty[2 * PF_EXT_WORDS];
        for (int k = 0; k < 2 * PF_EXT_WORDS; ++k) empty[k] = ((k / PF_EXT_STRIDE) & 1) ? INT32_MIN : INT32_MAX;
        CREATE_TRY(dalloc(&h->order_ext, (size_t)2 * PF_EXT_WORDS));
        CREATE_TRY(hipMemcpy(h->order_ext, empty, sizeof empty, hipMemcpyHostToDevice));
    }
    CREATE_TRY(dalloc(&h->order_hist[0], (size_t)PF_ORDER_BUCKETS));
    CREATE_TRY(dalloc(&h->order_hist[1], (size_t)PF_ORDER_BUCKETS));
    CREATE_TRY(hipMemsetAsync(h->order_hist[0], 0, sizeof(int32_t) * PF_ORDER_BUCKETS, h->stream));
    CREATE_TRY(hipMemsetAsync(h->order_hist[1], 0, sizeof(int32_t) * PF_ORDER_BUCKETS, h->stream));
    // the scan's points and, behind them (256-byte aligned, placed by pf_set_scan), its ray indices: one block, one copy
    CREATE_TRY(dalloc(&h->points, 3 * (size_t)c.max_points + ((size_t)c.max_points + 1) / 2 + 64));
    CREATE_TRY(dalloc(&h->wpts, 3 * (size_t)c.max_points));
    h->ray_idx = (int32_t*)h->points;             // set by pf_set_scan
    h->h_pose.resize(PF_POSE_ROWS * P);
    // S3P:47-51 initial state, S3P:71-77 prior
    {
        std::vector<double> x(P, 7.5), y(P, 7.5), z(P, 1.5), yaw(P, 0.0), w(Pg, 1.0 / (double)Pg);
        upload_pose_rows(h, h->h_pose.data(), x.data(), y.data(), z.data(), yaw.data());
        CREATE_TRY(hipMemcpyAsync(h->pose, h->h_pose.data(), sizeof(double) * PF_POSE_ROWS * P,
                                  hipMemcpyHostToDevice, h->stream));
        CREATE_TRY(hipMemcpyAsync(h->w, w.data(), sizeof(double) * Pg, hipMemcpyHostToDevice, h->stream));
        CREATE_TRY(hipMemsetAsync(h->logw, 0, sizeof(float) * Pg, h->stream));
        CREATE_TRY(hipMemsetAsync(h->counters, 0, 4 * sizeof(unsigned long long), h->stream));
        CREATE_TRY(pfk_fill_u8(h->stream, h->maps, map_bytes, c.map_prior));
        if (h->local) CREATE_TRY(hipMemsetAsync(h->local, 0, (size_t)h->local_elements, h->stream));
        CREATE_TRY(hipStreamSynchronize(h->stream));
    }
#undef CREATE_TRY
    *out = h;
    return PF_OK;
}

int pf_predict(pf_handle* h, const double* x, const double* y, const double* z, const double* yaw) {
    if (!h) return PF_ERR_ARG;
    NEED(h, x && y && z && yaw, "null pose array");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t nb = sizeof(double) * PF_POSE_ROWS * (size_t)h->P;
    char* slot = nullptr;
    HIP_TRY(h, h->st_pose.acquire(nb, &slot));               // pinned, double-buffered: no wait for the stream
    // (the poses are checked in the same pass that fills the block; a refused call leaves the handle's
    // poses as they were: the block has not been committed)
    const int bad = upload_pose_rows(h, (double*)slot, x, y, z, yaw);
    if (bad >= 0) return fail(h, PF_ERR_ARG, "pf_predict: particle %d pose not finite or |pos/res| >= 2^24", bad);
    h->pose_src = nullptr;
    HIP_TRY(h, hipMemcpyAsync(h->pose, slot, nb, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, h->st_pose.commit(h->stream));
    HIP_TRY(h, drop_counted_keys(h));
    h->order_dirty = true;
    h->poses_gathered = false;
    return PF_OK;
}

int pf_predict_device(pf_handle* h, const void* pose_block_dev) {
    if (!h) return PF_ERR_ARG;
    NEED(h, pose_block_dev, "null pose block");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (h->P == h->Pg || h->comm) {
        // read in place by the next pf_resample (one GPU) or by the library's all-gather (sharded
        // with a communicator); copied on any other use
        h->pose_src = (const double*)pose_block_dev;
    } else {
        h->pose_src = nullptr;
        HIP_TRY(h, hipMemcpyAsync(h->pose, pose_block_dev, sizeof(double) * PF_POSE_ROWS * h->P,
                                  hipMemcpyDeviceToDevice, h->stream));
    }
    HIP_TRY(h, drop_counted_keys(h));
    h->order_dirty = true;
    h->poses_gathered = false;
    return PF_OK;
}

int pf_get_poses(pf_handle* h, double* x, double* y, double* z, double* yaw) {
    if (!h) return PF_ERR_ARG;
    NEED(h, x && y && z && yaw, "null pose array");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, materialize_pose(h));
    const size_t n = sizeof(double) * (size_t)h->P;
    double* dst[4] = {x, y, z, yaw};
    for (int k = 0; k < 4; ++k)
        HIP_TRY(h, hipMemcpyAsync(dst[k], h->pose + (size_t)k * h->P, n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return PF_OK;
}

int pf_set_weights(pf_handle* h, const double* w) {
    if (!h) return PF_ERR_ARG;
    NEED(h, w, "null weights");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    char* slot = nullptr;
    HIP_TRY(h, h->st_weights.acquire(sizeof(double) * (size_t)h->Pg, &slot));
    memcpy(slot, w, sizeof(double) * (size_t)h->Pg);
    HIP_TRY(h, hipMemcpyAsync(h->w, slot, sizeof(double) * (size_t)h->Pg, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, h->st_weights.commit(h->stream));
    h->weights_given = true;
    h->weighted = false;               // a CDF of older weights is stale now
    return PF_OK;
}

int pf_set_best_index(pf_handle* h, int32_t index) {
    if (!h) return PF_ERR_ARG;
    NEED(h, index >= 0 && index < h->Pg, "index outside [0, n_particles_global)");
    NEED(h, h->rs_issued == h->rs_done, "a pf_resample_async still waits for pf_resample_result");
    h->best_prev = index;
    return PF_OK;
}

int pf_get_weights(pf_handle* h, double* w, float* logw) {
    if (!h) return PF_ERR_ARG;
    NEED(h, w, "null weights");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipMemcpyAsync(w, h->w, sizeof(double) * (size_t)h->Pg, hipMemcpyDeviceToHost, h->stream));
    if (logw)
        HIP_TRY(h, hipMemcpyAsync(logw, h->logw, sizeof(float) * (size_t)h->Pg, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return PF_OK;
}

int pf_fill_maps(pf_handle* h, int value) {
    if (!h) return PF_ERR_ARG;
    NEED(h, value >= 0 && value <= 255, "value");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, wait_map_writes(h));
    const unsigned long long bytes = h->slab ? h->slab * (unsigned long long)h->P
                                             : round_up((unsigned long long)h->elements, 256);
    HIP_TRY(h, pfk_fill_u8(h->stream, h->maps, bytes, value));
    h->shadow_dirty = true;
    return PF_OK;
}

int pf_write_map(pf_handle* h, int32_t i, const uint8_t* src) {
    if (!h) return PF_ERR_ARG;
    NEED(h, src, "null src");
    if (h->slab) NEED(h, i >= 0 && i < h->P, "particle index"); else i = 0;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, wait_map_writes(h));
    HIP_TRY(h, hipMemcpyAsync(h->maps + h->slab * (unsigned long long)i, src, (size_t)h->elements,
                              hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->shadow_dirty = true;
    return PF_OK;
}

int pf_broadcast_map(pf_handle* h, const uint8_t* src) {
    if (!h) return PF_ERR_ARG;
    NEED(h, src, "null src");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, wait_map_writes(h));
    HIP_TRY(h, hipMemcpyAsync(h->maps, src, (size_t)h->elements, hipMemcpyHostToDevice, h->stream));
    h->shadow_dirty = true;
    if (h->slab)
        for (int i = 1; i < h->P; ++i)
            HIP_TRY(h, hipMemcpyAsync(h->maps + h->slab * (unsigned long long)i, h->maps,
                                      (size_t)h->elements, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return PF_OK;
}

int pf_read_map(pf_handle* h, int32_t i, uint8_t* dst) {
    if (!h) return PF_ERR_ARG;
    NEED(h, dst, "null dst");
    if (h->slab) NEED(h, i >= 0 && i < h->P, "particle index"); else i = 0;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, wait_map_writes(h));
    HIP_TRY(h, hipMemcpyAsync(dst, h->maps + h->slab * (unsigned long long)i, (size_t)h->elements,
                              hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return PF_OK;
}

// The beam lists pf_set_beams kept on the host (shared grid, default K1): sent when a call needs them.
static int ensure_beam_lists(pf_handle* h, bool full, bool occ) {
    bool sent = false;
    if (full && !h->full_list_resident) {
        if (h->M > 0) HIP_TRY(h, hipMemcpyAsync(h->beams, h->h_beams.data(), sizeof(PfBeam) * (size_t)h->M, hipMemcpyHostToDevice, h->stream));
        h->full_list_resident = true; sent = true;
    }
    if (occ && !h->occ_lists_resident) {
        if (h->M_occ > 0) {
            HIP_TRY(h, hipMemcpyAsync(h->beams_occ, h->h_beams_occ.data(), sizeof(PfBeam) * (size_t)h->M_occ, hipMemcpyHostToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(h->beams_occ_f, h->h_beams_occ_f.data(), sizeof(PfBeamF) * (size_t)h->M_occ, hipMemcpyHostToDevice, h->stream));
        }
        h->occ_lists_resident = true; sent = true;
    }
    // (the sources are plain host vectors that the next pf_set_beams rewrites: the copies have to be through by then)
    if (sent) HIP_TRY(h, hipStreamSynchronize(h->stream));
    return PF_OK;
}

int pf_set_beams(pf_handle* h, const int32_t* bx, const int32_t* by, const int32_t* bz,
                 const int8_t* bv, int64_t n) {
    if (!h) return PF_ERR_ARG;
    NEED(h, n >= 0 && n <= h->cfg.max_beams, "n_beams beyond max_beams");
    NEED(h, n == 0 || (bx && by && bz && bv), "null beam array");
    // the whole input is checked before any state of the handle is touched: a refused call leaves the resident
    // lists and their host copies as they were (the lazy lists were uploaded from clobbered vectors)
    for (int64_t j = 0; j < n; ++j)
        if ((uint32_t)(bx[j] + (1 << 20)) > (2u << 20) || (uint32_t)(by[j] + (1 << 20)) > (2u << 20) || (uint32_t)(bz[j] + (1 << 20)) > (2u << 20))
            return fail(h, PF_ERR_ARG, "pf_set_beams: beam %lld offset beyond +-2^20 voxels", (long long)j);
    h->h_beams.resize((size_t)n);
    h->h_beams_occ.resize((size_t)n + PF_BEAM_GROUP); h->h_beams_occ_f.resize((size_t)n + PF_BEAM_GROUP);
    int n_real_occ = 0;
    long long lmax_i = 0;
    {
        PfBeam* hb = h->h_beams.data();
        PfBeam* ho = h->h_beams_occ.data();
        PfBeamF* hf = h->h_beams_occ_f.data();
        const int thr = h->cfg.occ_threshold;
        for (int64_t j = 0; j < n; ++j) {
            const int32_t X = bx[j], Y = by[j], Z = bz[j];
            PfBeam b;
            b.lx = (double)X; b.ly = (double)Y; b.lz = (double)Z;               // S3P:745-747 int->double
            b.v = bv[j]; b.pad = 0;
            hb[j] = b;
            if (b.v > thr) {                                                    // S3P:886
                ho[n_real_occ] = b;
                hf[n_real_occ] = PfBeamF{(float)X, (float)Y, (float)Z, 0};
                ++n_real_occ;
                const long long r = std::max<long long>((long long)abs(X) + abs(Y), abs(Z));
                lmax_i = std::max(lmax_i, r);
            }
        }
    }
    const double lmax = (double)lmax_i;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    // pad the occupied lists to a multiple of PF_BEAM_GROUP with beams no pose can bring into
    // the grid (|coordinate| ~ 2^22 voxels while |pos/res| < 2^24 ... the flat index stays far
    // outside [0, elements) for every validated pose: 2^22*gy*gz dominates)
    {
        size_t m = (size_t)n_real_occ;
        while (m % PF_BEAM_GROUP) {
            PfBeam d{}; d.lx = -3.0e7; d.ly = 0; d.lz = 0; d.v = 127;
            h->h_beams_occ[m] = d;
            h->h_beams_occ_f[m] = PfBeamF{-3.0e7f, 0.f, 0.f, 0};
            ++m;
        }
        h->h_beams_occ.resize(m); h->h_beams_occ_f.resize(m);
    }
    const size_t no = h->h_beams_occ.size();
    // column groups: the occupied beams of one local column (a run of the list that shares (lx, ly)), lz
    // ascending. An 8-cell z window that holds at least PF_RUN_MIN of them becomes a RUN group (three
    // dword gathers for the whole window, k_measure_col's run phase); the others go four to a group;
    // groups with more beams first (k_measure_col's phases), and dealing a phase round-robin gives every
    // chunk the same share of it
    h->h_cols.clear();
    h->h_tg.clear();
    std::vector<PfColG> runs;
    static const bool runs_wanted = !(getenv("SLAM_PF_RUNS") && atoi(getenv("SLAM_PF_RUNS")) == 0);
    const bool tiles = h->shadow && h->swz.kind == PF_LAYOUT_TILES;
    const bool use_runs = runs_wanted && !tiles;   // (the tile kernel takes plain groups)
    {
        std::vector<int32_t> zs, rest;
        for (int j = 0; j < n_real_occ;) {
            const PfBeam& b0 = h->h_beams_occ[(size_t)j];
            zs.clear(); rest.clear();
            bool ascending = true;                                   // the local map's order (S3P:868): nothing to sort
            while (j < n_real_occ && h->h_beams_occ[(size_t)j].lx == b0.lx && h->h_beams_occ[(size_t)j].ly == b0.ly) {
                const int32_t zj = (int32_t)h->h_beams_occ[(size_t)j++].lz;
                ascending = ascending && (zs.empty() || zj > zs.back());
                zs.push_back(zj);
            }
            bool distinct = true;                                    // a cell listed twice counts twice: no mask for that
            if (!ascending) {
                std::sort(zs.begin(), zs.end());
                for (size_t i = 1; i < zs.size(); ++i) distinct = distinct && zs[i] != zs[i - 1];
            }
            for (size_t i = 0; i < zs.size();) {
                size_t k = i;
                while (k < zs.size() && zs[k] < zs[i] + 8) ++k;
                if (use_runs && distinct && k - i >= PF_RUN_MIN) {
                    PfColG g{};
                    g.lx = (float)b0.lx; g.ly = (float)b0.ly; g.n = (int32_t)(k - i); g.lz[0] = zs[i];
                    unsigned long long m = 0;
                    for (size_t q = i; q < k; ++q) m |= 1ull << (8 * (zs[q] - zs[i]));
                    g.lz[1] = (int32_t)(unsigned)m; g.lz[2] = (int32_t)(unsigned)(m >> 32); g.lz[3] = zs[k - 1];
                    runs.push_back(g);
                    i = k;
                } else {
                    rest.push_back(zs[i++]);
                }
            }
            if (tiles) {                                             // k_measure_tile: up to PF_TILE_GROUP beams of a column per group
                for (size_t i = 0; i < rest.size(); i += PF_TILE_GROUP) {
                    pf_handle::TileGroupH g{};
                    g.lx = (float)b0.lx; g.ly = (float)b0.ly; g.n = (int32_t)std::min<size_t>(PF_TILE_GROUP, rest.size() - i);
                    for (int k = 0; k < PF_TILE_GROUP; ++k) g.lz[k] = rest[std::min(i + (size_t)k, rest.size() - 1)];
                    h->h_tg.push_back(g);
                }
                continue;
            }
            for (size_t i = 0; i < rest.size(); i += 4) {
                PfColG g{};
                g.lx = (float)b0.lx; g.ly = (float)b0.ly; g.n = (int32_t)std::min<size_t>(4, rest.size() - i);
                for (int k = 0; k < g.n; ++k) g.lz[k] = rest[i + (size_t)k];
                for (int k = g.n; k < 4; ++k) g.lz[k] = g.lz[g.n - 1];
                h->h_cols.push_back(g);
            }
        }
    }
    build_clusters(h, n_real_occ);
    build_fuse_columns(h, n_real_occ);
    // ... and inside a phase by range: how far the lanes of a wave spread over cache lines grows
    // with the beam's range, so neighbours in this order cost about the same and go to different chunks
    // (sorted through 64-bit keys: count, squared range, then the position in the list)
    auto by_range = [](std::vector<PfColG>& g, bool by_count, std::vector<unsigned long long>& keys, std::vector<PfColG>& tmp) {
        if (g.size() >= (1u << 22)) {                              // (beyond the keys' index field: the plain way)
            std::stable_sort(g.begin(), g.end(), [by_count](const PfColG& a, const PfColG& b) {
                if (by_count && a.n != b.n) return a.n > b.n;
                return a.lx * a.lx + a.ly * a.ly > b.lx * b.lx + b.ly * b.ly;
            });
            return;
        }
        keys.resize(g.size());
        for (size_t i = 0; i < g.size(); ++i) {
            const long long lx = (long long)g[i].lx, ly = (long long)g[i].ly;
            const unsigned long long r2 = (unsigned long long)(lx * lx + ly * ly) >> 3;          // < 2^39
            keys[i] = ((unsigned long long)(by_count ? 4 - g[i].n : 0) << 61) | (((1ull << 39) - 1 - r2) << 22) | (unsigned long long)i;
        }
        std::sort(keys.begin(), keys.end());
        tmp.resize(g.size());
        for (size_t i = 0; i < g.size(); ++i) tmp[i] = g[(size_t)(keys[i] & ((1ull << 22) - 1))];
        g.swap(tmp);
    };
    by_range(h->h_cols, true, h->h_sort_keys, h->h_sort_tmp);
    // the run groups go in front (the longest trips first), farthest first like the others
    by_range(runs, false, h->h_sort_keys, h->h_sort_tmp);
    h->n_runs = (int)runs.size();
    h->h_cols.insert(h->h_cols.begin(), runs.begin(), runs.end());
    h->n_cols = (int)h->h_cols.size();
    {
        int e = h->n_runs;
        for (int n = 4; n >= 1; --n) {
            while (e < h->n_cols && h->h_cols[(size_t)e].n == n) ++e;
            h->col_ends[4 - n] = e;
        }
    }
    if (tiles) {
        // groups with more beams first, inside a count the farthest first (as above), through 64-bit keys
        std::vector<pf_handle::TileGroupH>& tg = h->h_tg;
        h->h_sort_keys.resize(tg.size());
        for (size_t i = 0; i < tg.size(); ++i) {
            const long long lx = (long long)tg[i].lx, ly = (long long)tg[i].ly;
            const unsigned long long r2 = std::min<unsigned long long>((unsigned long long)(lx * lx + ly * ly) >> 3, (1ull << 36) - 1);
            h->h_sort_keys[i] = ((unsigned long long)(PF_TILE_GROUP - tg[i].n) << 60) | (((1ull << 36) - 1 - r2) << 24) | (unsigned long long)(i & 0xffffff);
        }
        if (tg.size() < (1u << 24)) {
            std::sort(h->h_sort_keys.begin(), h->h_sort_keys.end());
            h->h_tg_tmp.resize(tg.size());
            for (size_t i = 0; i < tg.size(); ++i) h->h_tg_tmp[i] = tg[(size_t)(h->h_sort_keys[i] & 0xffffff)];
            tg.swap(h->h_tg_tmp);
        } else {
            std::stable_sort(tg.begin(), tg.end(), [](const pf_handle::TileGroupH& a, const pf_handle::TileGroupH& b) { return a.n > b.n; });
        }
        int zmin = INT32_MAX, zmax = INT32_MIN;
        for (const auto& g : tg) { zmin = std::min(zmin, g.lz[0]); zmax = std::max(zmax, g.lz[g.n - 1]); }
        if (tg.empty()) zmin = zmax = 0;
        h->lz_min = zmin; h->lz_max = zmax;
        h->n_cols = (int)tg.size();
        int e = 0;
        for (int n = PF_TILE_GROUP; n >= 1; --n) {
            while (e < h->n_cols && tg[(size_t)e].n == n) ++e;
            h->tile_ends[PF_TILE_GROUP - n] = e;
        }
        // The chunks (blockIdx.y) of k_measure_tile: >= 12 groups (about 32 beams) each. The groups of a count go
        // tile_trip(count) to a trip, the trips of the whole list are numbered through and trip j belongs to chunk
        // j % C, so every chunk gets the same number of trips of every size to within one (a count's last trip may
        // be short). The list the kernel reads is CHUNK-MAJOR: a chunk's groups lie together, counts descending,
        // behind a table of C records (64 B each: the ends of its eight counts, then its start), so that a wave
        // finds its share with one scalar load, walks contiguous memory and divides nothing.
        int C = h->tune_chunks > 0 ? h->tune_chunks : h->max_chunks;
        if (C > h->max_chunks) C = h->max_chunks;
        if (C > (h->n_cols + 11) / 12) C = (h->n_cols + 11) / 12;
        if (C < 1) C = 1;
        h->tile_chunks = C;
        std::vector<pf_handle::TileGroupH>& pg = h->h_tg_tmp;
        pg.clear(); pg.reserve(tg.size());
        h->h_tcols.assign((size_t)C + tg.size(), PfTileG{});
        {
            int32_t trips[PF_TILE_GROUP];
            for (int ph = 0; ph < PF_TILE_GROUP; ++ph) trips[ph] = pfk_tile_trip(PF_TILE_GROUP - ph);
            h->h_deal_perm.resize(tg.size());
            h->h_deal_table.resize(9 * (size_t)C);
            pfh_deal_groups(h->tile_ends, C, trips, h->h_deal_perm.data(), h->h_deal_table.data());
            for (size_t i = 0; i < tg.size(); ++i) pg.push_back(tg[(size_t)h->h_deal_perm[i]]);
            for (int y = 0; y < C; ++y) {
                int32_t* row = (int32_t*)&h->h_tcols[(size_t)y];
                for (int k = 0; k < 9; ++k) row[k] = h->h_deal_table[9 * (size_t)y + (size_t)k];
            }
        }
        h->h_cols.resize(2 * pg.size());
        for (size_t i = 0; i < pg.size(); ++i) {
            const auto& g = pg[i];
            PfTileG t{};
            t.lx = g.lx; t.ly = g.ly; t.n = g.n;
            for (int k = 0; k < PF_TILE_GROUP; ++k) t.lza[k] = (uint32_t)(g.lz[k] - zmin) * h->swz.a;   // (mod 2^32: used only when the layer is inside the grid)
            h->h_tcols[(size_t)C + i] = t;
            for (int half = 0; half < 2; ++half) {
                PfColG c{};
                c.lx = g.lx; c.ly = g.ly; c.n = std::max(0, std::min(4, g.n - 4 * half));
                for (int k = 0; k < 4; ++k) c.lz[k] = g.lz[4 * half + k];
                h->h_cols[2 * i + (size_t)half] = c;
            }
        }
    }
    // everything goes through one pinned, double-buffered block: the call does not wait for the stream
    {
        // The shared grid's default K1 (column groups on the gather layout) reads the groups only: the full list
        // (pf_voxel_indices) and the two occupied lists (the other K1 forms) then stay on the host until a call
        // asks for them (ensure_beam_lists): 270 KB of the 320 KB a 4096-beam scan used to send every step.
        static const bool eager = getenv("SLAM_PF_EAGER_LISTS") && atoi(getenv("SLAM_PF_EAGER_LISTS")) != 0;   // (for comparison)
        const bool lazy = !eager && h->cfg.map_mode == PF_MAP_SHARED && h->shadow && !h->force_exact_measure && !box_eligible(h);
        struct Seg { const void* src; size_t bytes; bool later; };
        const Seg segs[] = {
            {h->h_cols.data(), sizeof(PfColG) * h->h_cols.size(), false},
            {h->h_tcols.data(), tiles ? sizeof(PfTileG) * h->h_tcols.size() : 0, false},
            {h->h_colc.data(), h->box_groups > 0 ? sizeof(PfColG) * h->h_colc.size() : 0, false},
            {h->h_clusters.data(), h->box_groups > 0 ? sizeof(PfCluster) * h->h_clusters.size() : 0, false},
            {h->h_group_start.data(), h->box_groups > 0 ? sizeof(int) * h->h_group_start.size() : 0, false},
            {h->h_fcols.data(), h->n_fcols > 0 ? sizeof(PfFCol) * h->h_fcols.size() : 0, false},
            {h->h_fvals.data(), h->n_fcols > 0 ? h->h_fvals.size() : 0, false},
            {h->h_mcols.data(), h->n_mcols > 0 ? sizeof(PfFCol) * h->h_mcols.size() : 0, false},
            {h->h_mvals.data(), h->n_mcols > 0 ? h->h_mvals.size() : 0, false},
            {h->h_beams.data(), sizeof(PfBeam) * (size_t)n, lazy},                 // the lists that may wait come last,
            {h->h_beams_occ.data(), sizeof(PfBeam) * no, lazy},                    // so what goes now is one run
            {h->h_beams_occ_f.data(), sizeof(PfBeamF) * no, lazy},
        };
        size_t total = 64;
        for (const Seg& sg : segs) total += (sg.bytes + 63) & ~(size_t)63;
        if (total > h->beam_blk_bytes) return fail(h, PF_ERR_ARG, "pf_set_beams: beam lists beyond the handle's capacity");
        char* slot = nullptr;
        HIP_TRY(h, h->st_beams.acquire(total, &slot));
        size_t off = 0, now = 0;
        char* dev_at[12];
        int k = 0;
        for (const Seg& sg : segs) {
            dev_at[k++] = h->beam_blk + off;
            if (!sg.bytes) continue;
            if (!sg.later) { memcpy(slot + off, sg.src, sg.bytes); now = off + ((sg.bytes + 63) & ~(size_t)63); }
            off += (sg.bytes + 63) & ~(size_t)63;
        }
        if (now) HIP_TRY(h, hipMemcpyAsync(h->beam_blk, slot, now, hipMemcpyHostToDevice, h->stream));
        h->beams_col = (PfColG*)dev_at[0]; h->beams_tile = (PfTileG*)dev_at[1];
        h->beams_colc = (PfColG*)dev_at[2]; h->clusters = (PfCluster*)dev_at[3];
        h->group_start = (int*)dev_at[4];
        h->fcols = (PfFCol*)dev_at[5]; h->fvals = (uint8_t*)dev_at[6];
        h->mcols = (PfFCol*)dev_at[7]; h->mvals = (uint8_t*)dev_at[8];
        h->beams = (PfBeam*)dev_at[9]; h->beams_occ = (PfBeam*)dev_at[10]; h->beams_occ_f = (PfBeamF*)dev_at[11];
        h->full_list_resident = h->occ_lists_resident = !lazy;
        HIP_TRY(h, h->st_beams.commit(h->stream));
    }
    h->M = (int)n;
    h->M_occ = (int)no;
    h->M_occ_real = n_real_occ;
    h->lmax_occ = lmax;
    return PF_OK;
}

// K1 lane order: Morton key over (yaw, x/res, y/res) in cells of 1/200 rad and 1/3 voxel; 4 bits
// per axis sort 64-lane waves finely enough up to 64k particles per handle, 5 bits beyond.
// SLAM_PF_ORDER=qa,qx,qy,mode,bits overrides (development knob).
namespace {
bool box_eligible(const pf_handle* h) {
    // opt-in (SLAM_PF_K1=box): measured slower than the gather-layout kernel at every BASELINE size so
    // far (the per-cluster barrier makes every wave wait for the one with the most
    // fp64-corrected columns, and the sparse bench scan stages 15 KB per nine beams)
    return h->cfg.map_mode == PF_MAP_SHARED && h->cfg.grid_z % 4 == 0 && !h->force_exact_measure && h->k1_mode == 2 && !h->zsplit;
}
}  // namespace
static PfOrderParams order_params(const pf_handle* h) {
    PfOrderParams op{200.0, 3.0, 3.0, 1, h->P <= 65536 ? 4 : 5, 0, nullptr};
    // k_measure_tile: the top three key bits are the z class (one round(z/res) per wave), the Morton part
    // has a bit less per axis over the same extent (cells of 1/100 rad and 2/3 voxel). 12-bit keys at every
    // particle count: the cells follow the cloud's extent, so 4096 of them order 125 k particles as well for K1 as
    // 32768 did (0.693 against 0.687 ms at configs[2]'s share) and the ordering is one launch with LDS counters
    // instead of three with global ones (34 against 120 us there)
    if (h->zsplit) op = PfOrderParams{200.0, 1.5, 1.5, 1, 4, 1, nullptr};
    static const bool fixed_cells = getenv("SLAM_PF_ORDER_FIXED") != nullptr;    // (comparison: cells that do not follow the cloud)
    if (!fixed_cells && h->ext_cur >= 0) op.ext = h->order_ext + PF_EXT_WORDS * h->ext_cur;
    // k_measure_box wants its workgroups' particles close in yaw (the box of a cluster grows with
    // range x yaw span): yaw-major keys, 1/400 rad and one-voxel cells
    if (box_eligible(h)) op = PfOrderParams{400.0, 1.0, 1.0, 2, 4, 0, nullptr};
    if (const char* e = getenv("SLAM_PF_ORDER")) {
        double a, x, y; int m, b;
        if (sscanf(e, "%lf,%lf,%lf,%d,%d", &a, &x, &y, &m, &b) == 5 && (b == 4 || b == 5))
            op = PfOrderParams{a, x, y, m, b, (m == 1 && h->zsplit) ? 1 : 0, op.ext};
    }
    return op;
}

int pf_measure(pf_handle* h, int add_points) {
    if (!h) return PF_ERR_ARG;
    const bool fuse = add_points != 0 && h->cfg.map_mode == PF_MAP_PER_PARTICLE;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, materialize_pose(h));
    // Without fuse only the occupied beams matter (S3P:886) and beams may be split over
    // blockIdx.y; with fuse every beam is walked in order by one lane.
    const int Mk = fuse ? h->M : h->M_occ;
    int chunks = 1, bpc = Mk > 0 ? Mk : 1;
    if (!fuse && Mk > 0) {
        chunks = h->tune_chunks > 0 ? h->tune_chunks : h->max_chunks;
        if (chunks > h->max_chunks) chunks = h->max_chunks;
        const int min_bpc = 32;                       // do not split below 32 beams per chunk
        if (chunks > (Mk + min_bpc - 1) / min_bpc) chunks = (Mk + min_bpc - 1) / min_bpc;
        if (chunks < 1) chunks = 1;
        bpc = (Mk + chunks - 1) / chunks;
        bpc = (bpc + PF_BEAM_GROUP - 1) / PF_BEAM_GROUP * PF_BEAM_GROUP;
        chunks = (Mk + bpc - 1) / bpc;
    }
    const PfOrderParams op = order_params(h);
    const bool small_keys = op.mode != 0 && op.bits == 4;      // 4096 buckets: fixed counter pair
    // (the column form on per-particle maps is a wave per particle: no lane order to prepare)
    const bool mcols = !fuse && h->cfg.map_mode == PF_MAP_PER_PARTICLE && h->n_mcols > 0 && !h->force_exact_measure;
    if (!fuse && !mcols && !h->force_exact_measure && h->order_dirty) {
        Timed t(h, PF_K_ORDER);
        // this ordering fills the extent record that is empty now and empties the other one (which the keys of
        // these poses were made with)
        const int ext_w = h->ext_cur < 0 ? 0 : h->ext_cur ^ 1;
        if (small_keys) {
            HIP_TRY(h, pfk_order(h->stream, h->pose, h->P, h->lanes, h->order_keys, h->order_hist[0], h->order_hist[1],
                                 h->order, h->sorted, op, h->hist_counted, h->order_ext + PF_EXT_WORDS * ext_w,
                                 h->order_ext + PF_EXT_WORDS * (ext_w ^ 1)));
        } else {
            HIP_TRY(h, pfk_order(h->stream, h->pose, h->P, h->lanes, h->order_keys, h->order_hist[h->order_flip],
                                 h->order_hist[h->order_flip ^ 1], h->order, h->sorted, op, false,
                                 h->order_ext + PF_EXT_WORDS * ext_w, h->order_ext + PF_EXT_WORDS * (ext_w ^ 1)));
            h->order_flip ^= 1;
        }
        h->order_dirty = false;
        h->ext_cur = ext_w;
    }
    h->hist_counted = false;           // consumed above, or stale: pfk_reduce_scores zeroes the pair
    HIP_TRY(h, wait_map_writes(h));      // the ordering above may overlap a running K5, K1 may not
    // LDS-staged boxes of the reference-order grid on request (SLAM_PF_K1=box); else the gather layout
    const bool box = !fuse && h->box_groups > 0 && box_eligible(h);
    const bool swz = h->shadow && !fuse && !h->force_exact_measure && !box;
    if (swz && h->shadow_dirty) {
        HIP_TRY(h, pfk_swz_build(h->stream, h->maps, h->shadow, h->swz));
        h->shadow_dirty = false;
    }
    if (!box && !swz)                                  // the other forms read the occupied lists
        if (int rc = ensure_beam_lists(h, false, true)) return rc;
    {
        Timed t(h, PF_K_MEASURE);
        if (box) {
            chunks = h->box_groups;
            HIP_TRY(h, pfk_measure_box(h->stream, h->sorted, h->P, h->maps, h->cfg.grid_x, h->cfg.grid_y, h->cfg.grid_z,
                                       h->beams_colc, h->clusters, h->group_start, h->box_kp, h->box_parts, h->box_lp,
                                       h->box_groups, grid_params(h), h->lmax_occ, h->partial));
        }
        else if (swz) {
            // >= 12 groups (about 32 beams) per chunk, like min_bpc above
            int cch = h->tune_chunks > 0 ? h->tune_chunks : h->max_chunks;
            if (cch > h->max_chunks) cch = h->max_chunks;
            if (cch > (h->n_cols + 11) / 12) cch = (h->n_cols + 11) / 12;
            if (cch < 1) cch = 1;
            chunks = cch;
            if (h->swz.kind == PF_LAYOUT_TILES) {
                chunks = h->tile_chunks;               // (the list was dealt to this many chunks by pf_set_beams)
                HIP_TRY(h, pfk_measure_tile(h->stream, h->sorted, h->lanes, h->order, h->shadow, h->maps, h->swz, h->beams_tile, h->beams_col,
                                            h->lz_min, h->lz_max, chunks, grid_params(h), h->lmax_occ, h->partial));
            }
            else
            HIP_TRY(h, pfk_measure_col(h->stream, h->sorted, h->P, h->shadow, h->swz, h->beams_col, h->n_runs, h->col_ends,
                                       chunks, grid_params(h), h->lmax_occ, h->order, h->partial));
        }
        else if (fuse && h->n_fcols > 0)
            HIP_TRY(h, pfk_measure_fuse_cols(h->stream, h->pose, h->P, h->maps, h->slab, h->beams, h->M, h->fcols, h->n_fcols,
                                             h->fvals, h->fbounds, h->cfg.grid_x, h->cfg.grid_y, h->cfg.grid_z,
                                             grid_params(h), h->partial, true));
        else if (fuse)
            HIP_TRY(h, pfk_measure(h->stream, h->pose, h->P, h->maps, h->slab, h->beams, h->M, 1, bpc,
                                   grid_params(h), true, h->partial));
        else if (mcols) {
            chunks = 1;                               // a wave per particle, one partial each
            HIP_TRY(h, pfk_measure_fuse_cols(h->stream, h->pose, h->P, h->maps, h->slab, h->beams_occ, h->M_occ, h->mcols,
                                             h->n_mcols, h->mvals, h->mbounds, h->cfg.grid_x, h->cfg.grid_y, h->cfg.grid_z,
                                             grid_params(h), h->partial, false));
        }
        else if (h->force_exact_measure)
            HIP_TRY(h, pfk_measure(h->stream, h->pose, h->P, h->maps, h->slab, h->beams_occ, Mk, chunks,
                                   bpc, grid_params(h), false, h->partial));
        else
            HIP_TRY(h, pfk_measure_fast(h->stream, h->pose, h->P, h->maps, h->slab, h->beams_occ_f,
                                        h->beams_occ, Mk, chunks, bpc, grid_params(h), h->lmax_occ,
                                        h->order, h->partial));
    }
    {
        Timed t(h, PF_K_ORDER);
        if (box || swz)
            HIP_TRY(h, pfk_reduce_scores_l(h->stream, h->partial, chunks, h->lanes, h->order, h->scores, h->bmin,
                                           h->order_hist[0], h->order_hist[1], small_keys ? 4096 : 0));
        else
        HIP_TRY(h, pfk_reduce_scores(h->stream, h->partial, chunks, h->P, h->scores, h->bmin, h->order_hist[0],
                                     h->order_hist[1], small_keys ? 4096 : 0));
    }
    h->measured = true;
    return PF_OK;
}

int pf_get_scores(pf_handle* h, int32_t* sum, int32_t* cnt) {
    if (!h) return PF_ERR_ARG;
    if (!h->measured) return fail(h, PF_ERR_STATE, "pf_get_scores before pf_measure");
    NEED(h, sum && cnt, "null output");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    std::vector<int32_t> tmp(2 * (size_t)h->P);
    HIP_TRY(h, hipMemcpyAsync(tmp.data(), h->scores, sizeof(int2) * (size_t)h->P, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < h->P; ++i) { sum[i] = tmp[2 * (size_t)i]; cnt[i] = tmp[2 * (size_t)i + 1]; }
    return PF_OK;
}

void* pf_scores_device(pf_handle* h) { return h ? (void*)h->scores : nullptr; }
void* pf_poses_device(pf_handle* h) {
    if (!h) return nullptr;
    if (h->pose_src && (hipSetDevice(h->cfg.device) != hipSuccess || materialize_pose(h) != hipSuccess)) return nullptr;
    return (void*)h->pose;
}
void* pf_stream(pf_handle* h) { return h ? (void*)h->stream : nullptr; }

int pf_voxel_indices(pf_handle* h, int32_t i, int32_t* flat_out) {
    if (!h) return PF_ERR_ARG;
    NEED(h, i >= 0 && i < h->P, "particle index");
    NEED(h, flat_out || h->M == 0, "null output");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, materialize_pose(h));
    if (int rc = ensure_beam_lists(h, true, false)) return rc;
    HIP_TRY(h, pfk_voxel_indices(h->stream, h->pose, h->P, i, h->beams, h->M, grid_params(h), h->vox));
    if (h->M)
        HIP_TRY(h, hipMemcpyAsync(flat_out, h->vox, sizeof(int32_t) * (size_t)h->M, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return PF_OK;
}

int pf_weights(pf_handle* h, const void* scores_global_dev) {
    if (!h) return PF_ERR_ARG;
    if (!scores_global_dev) {
        if (h->Pg != h->P && !h->comm)
            return fail(h, PF_ERR_ARG, "pf_weights: sharded handle needs the gathered scores (or pf_comm_init)");
        if (!h->measured) return fail(h, PF_ERR_STATE, "pf_weights before pf_measure");
    }
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const int2* sc = scores_global_dev ? (const int2*)scores_global_dev : h->scores;
    bool gathered_bmin = false;
    if (!scores_global_dev && h->comm) {              // C1: 8 B per particle, rank order
        const bool both = h->group_gathers && rccl().GroupStart && rccl().GroupEnd;
        if (!both) HIP_TRY(h, materialize_pose(h));
        ncclResult_t r = both ? rccl().GroupStart() : ncclSuccess;
        if (r == ncclSuccess)
            r = rccl().AllGather(h->scores, h->scores_all, 2 * (size_t)h->P, ncclInt32, h->comm, h->stream);
        if (r == ncclSuccess && both)                 // C2 rides along: 72 B per particle, from wherever the rows are
            r = rccl().AllGather(h->pose_src ? h->pose_src : h->pose, h->poses_all, PF_POSE_ROWS * (size_t)h->P,
                                 ncclFloat64, h->comm, h->stream);
        if (r == ncclSuccess && both)                 // and every rank's block minima of its scores (K3's min)
            r = rccl().AllGather(h->bmin, h->bmin_all, (size_t)h->n_bmin, ncclInt32, h->comm, h->stream);
        if (both) { const ncclResult_t e = rccl().GroupEnd(); if (r == ncclSuccess) r = e; }
        gathered_bmin = both;
        if (r != ncclSuccess) return fail(h, PF_ERR_HIP, "pf_weights: ncclAllGather: %s", rccl().GetErrorString(r));
        h->poses_gathered = both;
        sc = h->scores_all;
    }
    {
        Timed t(h, PF_K_WEIGHTS);
        const int32_t* bmin = sc == h->scores && h->P == h->Pg ? h->bmin : gathered_bmin ? h->bmin_all : nullptr;
        int n_bmin = gathered_bmin ? h->comm_world * h->n_bmin : bmin ? h->n_bmin : 0;
        if (!bmin && !h->exact && h->Pg > 4096) {     // scores from the caller's own all-gather: a pass of block minima
            HIP_TRY(h, pfk_block_minima(h->stream, sc, h->Pg, h->bmin_ext));
            bmin = h->bmin_ext; n_bmin = pfk_block_minima_count(h->Pg);
        }
        const bool carry = h->upd_pending && pfk_weights_carries(h->Pg, h->exact, bmin);
        HIP_TRY(h, pfk_weights(h->stream, sc, h->Pg, h->w, h->logw, h->cdf, h->scratch, h->exact, h->first,
                               h->counters, h->rs, bmin, carry ? &h->upd : nullptr, n_bmin));
        if (carry) h->upd_pending = false;
    }
    h->weighted = true;
    return PF_OK;
}

int pf_resample_async(pf_handle* h, const double* u, const void* poses_global_dev, int32_t* slot_anc_out) {
    if (!h) return PF_ERR_ARG;
    if (h->rs_issued - h->rs_done >= 2)
        return fail(h, PF_ERR_STATE, "pf_resample_async: two resamples already wait for pf_resample_result");
    if (!h->weighted && !h->weights_given)
        return fail(h, PF_ERR_STATE, "pf_resample before pf_weights (or pf_set_weights)");
    if (!u && !h->have_u) return fail(h, PF_ERR_ARG, "pf_resample: no thresholds resident");
    if (h->Pg != h->P && !poses_global_dev && !h->comm)
        return fail(h, PF_ERR_ARG, "pf_resample: sharded handle needs the gathered pose blocks (or pf_comm_init)");
    const bool migrate = h->slab && (h->Pg != h->P || h->virtual_shards > 1);
    if (migrate && !(h->comm && rccl().Send && rccl().Recv && rccl().GroupStart && rccl().GroupEnd))
        return fail(h, PF_ERR_ARG, "pf_resample: per-particle maps on a sharded handle move between ranks "
                                   "inside the library: pf_comm_init first");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (!poses_global_dev && h->comm) {               // C2: the 72-B pose blocks, rank order
        if (!h->poses_gathered) {                     // (else pf_weights' group already brought them)
            const ncclResult_t r = rccl().AllGather(h->pose_src ? h->pose_src : h->pose, h->poses_all,
                                                    PF_POSE_ROWS * (size_t)h->P,
                                                    ncclFloat64, h->comm, h->stream);
            if (r != ncclSuccess) return fail(h, PF_ERR_HIP, "pf_resample: ncclAllGather: %s", rccl().GetErrorString(r));
        }
        poses_global_dev = h->poses_all;
    }
    h->poses_gathered = false;
    if (poses_global_dev && poses_global_dev != (const void*)h->poses_all)
        HIP_TRY(h, materialize_pose(h));              // the caller's gather: kept slots stay where they are in h->pose
    const int slot = (int)(h->rs_issued & 1ull);
    const double* u_dev = h->u;
    if (u) {
        // this slot's buffer is free: its previous set belongs to a resample whose results have
        // been handed out (at most two are outstanding)
        memcpy(h->h_u + (size_t)slot * (size_t)h->Pg, u, sizeof(double) * (size_t)h->Pg);
        h->u_src = slot;
        h->have_u = true;
        u_dev = h->d_u_map + (size_t)slot * (size_t)h->Pg;
        if (h->u_pre) {
            // (the device slot is free for the same reason as the pinned one; the copy starts now, on its own stream,
            // and this stream's search waits for it)
            double* dst = h->u_pre + (size_t)slot * (size_t)h->Pg;
            HIP_TRY(h, hipMemcpyAsync(dst, h->h_u + (size_t)slot * (size_t)h->Pg, sizeof(double) * (size_t)h->Pg,
                                      hipMemcpyHostToDevice, h->copy_stream));
            HIP_TRY(h, hipEventRecord(h->u_event[slot], h->copy_stream));
            HIP_TRY(h, hipStreamWaitEvent(h->stream, h->u_event[slot], 0));
            u_dev = dst;
        }
    } else if (h->u_src >= 0) {
        HIP_TRY(h, hipMemcpyAsync(h->u, h->h_u + (size_t)h->u_src * (size_t)h->Pg, sizeof(double) * (size_t)h->Pg,
                                  hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));     // once per switch: the slot may be refilled by the next call
        h->u_src = -1;
    }
    {
        Timed t(h, PF_K_RESAMPLE);
        // FAST order: boundary k of the parallel CDF is one fixed tree of at most ~50 roundings on
        // the raw weights, the sequential one k roundings on the normalised weights plus one per
        // weight from the division, all of partial sums <= cdf[k]: the two differ by at most
        // (k + 52) * 2^-53 * cdf[k] * (1 + 1e-9). k_search takes that with 12 % to spare as the band
        // of boundary k (`bound` = the unit roundoff with the factor); it decides how often a step
        // pays for the sequential-order CDF: about once per 1/(8e-17 Pg^3) steps, and half as often since
        // draw_is_ambiguous tests a draw inside that band again with the partial sums bounded block by block.
        // resample() on its own (S3P:757-821 on the weights pf_set_weights put there): the CDF is
        // built in the reference's sequential order, so no draw is ambiguous
        const bool alone = !h->weighted;
        if (alone) HIP_TRY(h, pfk_resample_prepare(h->stream, h->w, h->Pg, h->cdf, h->first, h->counters));
        const double bound = h->exact || alone ? 0.0 : 1.125 * 1.1102230246251565e-16;
        // one GPU, shared-key ordering: the fill stage leaves K1's order keys of the new poses
        const PfOrderParams op = order_params(h);
        // (sharded: the kernel that writes the local shard's rows counts them)
        const bool count_keys = op.mode != 0 && op.bits == 4;
        // a second resample with no pf_measure in between: the first one's key counts were never
        // consumed (pfk_reduce_scores zeroes them) and the fill stage below would add to them
        if (count_keys) HIP_TRY(h, drop_counted_keys(h));
        HIP_TRY(h, pfk_resample(h->stream, h->cdf, u_dev, h->Pg, h->P, h->offset, bound, h->w, h->logw,
                                h->pose, h->pose_src, (const double*)poses_global_dev, h->slab ? h->maps : nullptr,
                                h->slab, (unsigned long long)h->elements, h->draw, h->first, h->dup_anc,
                                h->free_slot, h->slot_anc, h->best, h->counters, h->rs, h->d_result + h->result_slot_u64 * slot,
                                count_keys ? h->order_hist[0] : nullptr, h->order_keys, op,
                                h->weighted && !h->exact,     // the FAST-order weight kernels left the block-end table behind the CDF
                                h->virtual_shards > 1 ? h->P / h->virtual_shards : 0));
        h->hist_counted = count_keys;
        h->pose_src = nullptr;                        // every row is in h->pose now
    }
    if (slot_anc_out)
        HIP_TRY(h, hipMemcpyAsync(slot_anc_out, h->slot_anc, sizeof(int32_t) * (size_t)h->Pg,
                                  hipMemcpyDeviceToHost, h->stream));
    if (migrate) {
        // S3P:810 across ranks: the slabs whose ancestor lives on another GPU (local ones were
        // copied by the kernels above). Sources are surviving slots and destinations never-drawn
        // ones, so no transfer reads what another writes.
        const size_t Pg = (size_t)h->Pg;
        h->h_slots.resize(Pg);
        HIP_TRY(h, hipMemcpyAsync(h->h_slots.data(), h->slot_anc, sizeof(int32_t) * Pg, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        // The transfers this process takes part in, in transfer-number order: a rank's own sends and receives, or
        // (SLAM_PF_VIRTUAL_SHARDS, one rank) those of every virtual shard, whose peer is this rank itself; a
        // transfer between two virtual shards then has its send and its receive in the same group.
        struct Xfer { int64_t seq; int32_t slot, peer; bool send; };
        std::vector<Xfer> xf;
        const int V = h->virtual_shards > 1 ? h->virtual_shards : 1;
        const int Pv = h->P / V;
        for (auto* v : {&h->mg_src, &h->mg_speer, &h->mg_dst, &h->mg_rpeer}) v->resize(Pg);
        h->mg_sseq.resize(Pg); h->mg_rseq.resize(Pg);
        for (int v = 0; v < V; ++v) {
            int64_t ns = 0, nr = 0;
            pfh_migration_plan(h->h_slots.data(), h->Pg, Pv, V > 1 ? v : h->comm_rank, h->mg_src.data(), h->mg_speer.data(),
                               h->mg_sseq.data(), &ns, h->mg_dst.data(), h->mg_rpeer.data(), h->mg_rseq.data(), &nr,
                               (int64_t)Pg);
            const int32_t base = V > 1 ? v * Pv : 0;                 // a virtual shard's slots inside this rank's maps
            for (int64_t k = 0; k < ns; ++k)
                xf.push_back(Xfer{h->mg_sseq[(size_t)k], base + h->mg_src[(size_t)k], V > 1 ? h->comm_rank : h->mg_speer[(size_t)k], true});
            for (int64_t k = 0; k < nr; ++k)
                xf.push_back(Xfer{h->mg_rseq[(size_t)k], base + h->mg_dst[(size_t)k], V > 1 ? h->comm_rank : h->mg_rpeer[(size_t)k], false});
        }
        // (a rank's two lists ascend in the transfer number; merged: by number, a number's send before its receive)
        std::stable_sort(xf.begin(), xf.end(), [](const Xfer& a, const Xfer& b) { return a.seq != b.seq ? a.seq < b.seq : a.send && !b.send; });
        // one RCCL group per kGroup transfer numbers: the same cut on every rank
        const int64_t kGroup = 512;
        size_t ix = 0;
        while (ix < xf.size()) {
            const int64_t end = (xf[ix].seq / kGroup + 1) * kGroup;
            ncclResult_t r = rccl().GroupStart();
            for (; r == ncclSuccess && ix < xf.size() && xf[ix].seq < end; ++ix) {
                uint8_t* slab = h->maps + h->slab * (unsigned long long)xf[ix].slot;
                r = xf[ix].send ? rccl().Send(slab, (size_t)h->elements, ncclUint8, xf[ix].peer, h->comm, h->stream)
                                : rccl().Recv(slab, (size_t)h->elements, ncclUint8, xf[ix].peer, h->comm, h->stream);
            }
            const ncclResult_t e = rccl().GroupEnd();
            if (r == ncclSuccess) r = e;
            if (r != ncclSuccess) return fail(h, PF_ERR_HIP, "pf_resample: map migration: %s", rccl().GetErrorString(r));
        }
    }
    HIP_TRY(h, hipEventRecord(h->rs_event[slot], h->stream));
    ++h->rs_issued;
    h->order_dirty = true;
    h->weighted = false;
    h->weights_given = false;
    h->measured = false;
    return PF_OK;
}

int pf_resample_result(pf_handle* h, int32_t* best_out, int64_t* nohit_out, int64_t* ambiguous_out) {
    if (!h) return PF_ERR_ARG;
    if (h->rs_done == h->rs_issued) return fail(h, PF_ERR_STATE, "pf_resample_result: no resample outstanding");
    const int slot = (int)(h->rs_done & 1ull);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipEventSynchronize(h->rs_event[slot]));
    const unsigned long long* r = h->h_result + h->result_slot_u64 * slot;
    const unsigned long long cnt[3] = {r[0], r[1], r[2]};
    h->best_prev = pfk_result_best(r, h->best_prev);
    if (best_out) *best_out = h->best_prev;
    if (nohit_out) *nohit_out = (int64_t)(cnt[1] ? cnt[2] : cnt[0]);
    if (ambiguous_out) *ambiguous_out = (int64_t)cnt[1];
    ++h->rs_done;
    return PF_OK;
}

int pf_resample(pf_handle* h, const double* u, const void* poses_global_dev, int32_t* slot_anc_out,
                int32_t* best_out, int64_t* nohit_out, int64_t* ambiguous_out) {
    if (!h) return PF_ERR_ARG;
    if (h->rs_issued != h->rs_done)
        return fail(h, PF_ERR_STATE, "pf_resample: a pf_resample_async still waits for pf_resample_result");
    const int rc = pf_resample_async(h, u, poses_global_dev, slot_anc_out);
    if (rc != PF_OK) return rc;
    return pf_resample_result(h, best_out, nohit_out, ambiguous_out);
}

static size_t up256(size_t n) { return (n + 255) & ~(size_t)255; }
int pf_set_scan(pf_handle* h, const double* points, int64_t n_points, const int32_t* ray_idx,
                int64_t n_rays) {
    if (!h) return PF_ERR_ARG;
    NEED(h, n_points >= 0 && n_points <= h->cfg.max_points, "n_points beyond max_points");
    NEED(h, n_rays >= 0 && n_rays <= h->cfg.max_points, "n_rays beyond max_points");
    NEED(h, (n_points == 0 || points) && (n_rays == 0 || ray_idx), "null array");
    // checked in passes without an exit (they vectorise); the index is looked for only when something is wrong
    {
        bool ok = true;
        for (int64_t i = 0; i < 3 * n_points; ++i) ok &= fabs(points[i]) < 1.0e6;
        if (!ok)
            for (int64_t i = 0; i < 3 * n_points; ++i)
                if (!(fabs(points[i]) < 1.0e6)) return fail(h, PF_ERR_ARG, "pf_set_scan: point %lld not finite/too far", (long long)(i / 3));
        ok = true;
        for (int64_t r = 0; r < n_rays; ++r) ok &= (uint64_t)(int64_t)ray_idx[r] < (uint64_t)n_points;
        if (!ok)
            for (int64_t r = 0; r < n_rays; ++r)
                if (ray_idx[r] < 0 || ray_idx[r] >= n_points)
                    return fail(h, PF_ERR_ARG, "pf_set_scan: ray_idx[%lld] outside [0, n_points)", (long long)r);
    }
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, wait_map_writes(h));
    const size_t nb_p = sizeof(double) * 3 * (size_t)n_points, nb_r = sizeof(int32_t) * (size_t)n_rays;
    const size_t off_r = up256(nb_p);                  // the ray indices' place in the block, on the host and the device
    char* slot = nullptr;
    HIP_TRY(h, h->st_scan.acquire(off_r + nb_r + 8, &slot));
    if (n_points) memcpy(slot, points, nb_p);
    if (n_rays) memcpy(slot + off_r, ray_idx, nb_r);
    h->ray_idx = (int32_t*)((char*)h->points + off_r);
    if (n_points || n_rays)
        HIP_TRY(h, hipMemcpyAsync(h->points, slot, n_rays ? off_r + nb_r : nb_p, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, h->st_scan.commit(h->stream));
    h->n_points = (int)n_points; h->n_rays = (int)n_rays;
    return PF_OK;
}

int pf_set_scan_device(pf_handle* h, const double* cloud_dev, int64_t n_points, const int32_t* ray_idx,
                       int64_t n_rays) {
    if (!h) return PF_ERR_ARG;
    NEED(h, n_points >= 0 && n_points <= h->cfg.max_points, "n_points beyond max_points");
    NEED(h, n_rays >= 0 && n_rays <= h->cfg.max_points, "n_rays beyond max_points");
    NEED(h, (n_points == 0 || cloud_dev) && (n_rays == 0 || ray_idx), "null array");
    {
        bool ok = true;
        for (int64_t r = 0; r < n_rays; ++r) ok &= (uint64_t)(int64_t)ray_idx[r] < (uint64_t)n_points;
        if (!ok)
            for (int64_t r = 0; r < n_rays; ++r)
                if (ray_idx[r] < 0 || ray_idx[r] >= n_points)
                    return fail(h, PF_ERR_ARG, "pf_set_scan_device: ray_idx[%lld] outside [0, n_points)", (long long)r);
    }
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, wait_map_writes(h));
    const size_t nb_p = sizeof(double) * 3 * (size_t)n_points, nb_r = sizeof(int32_t) * (size_t)n_rays;
    const size_t off_r = up256(nb_p);                  // the same place as in pf_set_scan
    h->ray_idx = (int32_t*)((char*)h->points + off_r);
    HIP_TRY(h, pfk_copy_points(h->stream, cloud_dev, (int)n_points, h->points));
    if (n_rays) {
        char* slot = nullptr;
        HIP_TRY(h, h->st_scan.acquire(nb_r + 8, &slot));
        memcpy(slot, ray_idx, nb_r);
        HIP_TRY(h, hipMemcpyAsync(h->ray_idx, slot, nb_r, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, h->st_scan.commit(h->stream));
    }
    h->n_points = (int)n_points; h->n_rays = (int)n_rays;
    return PF_OK;
}
int pf_update_map(pf_handle* h, int target, const pf_update_params* p, const double pos[3],
                  const double zyx[3]) {
    if (!h) return PF_ERR_ARG;
    NEED(h, p && pos && zyx, "null argument");
    PfUpdateParams up{};
    void* grid = nullptr;
    if (target == PF_GRID_SHARED) {
        NEED(h, h->cfg.map_mode == PF_MAP_SHARED, "PF_GRID_SHARED needs map_mode PF_MAP_SHARED");
        grid = h->maps; up.elements = h->elements; up.gy = h->cfg.grid_y; up.gz = h->cfg.grid_z;
        up.is_signed = 0;
        if (h->shadow && !h->shadow_dirty) {        // keep K1's gather-layout copy in step
            up.shadow = h->shadow; up.swz_a = h->swz.a; up.swz_b = h->swz.b; up.swz_kind = h->swz.kind;
        }
        NEED(h, p->hit_clamp >= 0 && p->hit_clamp <= 255 && p->miss_clamp >= 0 && p->miss_clamp <= 255, "clamps outside uint8");
    } else if (target == PF_GRID_LOCAL) {
        NEED(h, h->local != nullptr, "no local map configured");
        grid = h->local; up.elements = h->local_elements; up.gy = h->cfg.local_y; up.gz = h->cfg.local_z;
        up.is_signed = 1;
        NEED(h, p->hit_clamp >= -128 && p->hit_clamp <= 127 && p->miss_clamp >= -128 && p->miss_clamp <= 127, "clamps outside int8");
    } else {
        return fail(h, PF_ERR_ARG, "pf_update_map: target");
    }
    NEED(h, p->hit_inc >= 0 && p->hit_inc <= 127 && p->miss_inc <= 0 && p->miss_inc >= -127, "increments");
    NEED(h, isfinite(pos[0]) && isfinite(pos[1]) && isfinite(pos[2]) && isfinite(zyx[0]) &&
            isfinite(zyx[1]) && isfinite(zyx[2]) && fabs(pos[0]) < 1e6 && fabs(pos[1]) < 1e6 && fabs(pos[2]) < 1e6,
         "pose not finite");
    const double z = zyx[0], y = zyx[1], x = zyx[2];
    pf_rot_zyx(up.rot, z, y, x);    // S3P:536-544
    up.pos[0] = pos[0]; up.pos[1] = pos[1]; up.pos[2] = pos[2];
    up.conv = 1 / h->cfg.resolution;                               // S3P:284
    up.max_range = p->max_range;
    up.hit_inc = p->hit_inc; up.hit_clamp = p->hit_clamp;
    up.miss_inc = p->miss_inc; up.miss_clamp = p->miss_clamp;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, flush_update(h));                      // an earlier update that nobody picked up
    PfUpdateWork uw{};
    uw.grid = grid; uw.points = h->points; uw.wpts = h->wpts; uw.ray_idx = h->ray_idx;
    uw.M = h->n_points; uw.R = h->n_rays; uw.up = up;
    if (target == PF_GRID_SHARED) {                   // rides with the next pf_weights, or runs at the next map access
        h->upd = uw;
        h->upd_pending = true;
        return PF_OK;
    }
    // SLAM_PF_COUNT=0: the march by compare-and-swap per step (comparison)
    static const bool count_wanted = !(getenv("SLAM_PF_COUNT") && atoi(getenv("SLAM_PF_COUNT")) == 0);
    if (count_wanted && h->n_rays > 0) {
        if (!h->ray_cnt) {
            HIP_TRY(h, dalloc(&h->ray_cnt, (size_t)h->local_elements + 3));
            HIP_TRY(h, hipMemsetAsync(h->ray_cnt, 0, sizeof(unsigned) * ((size_t)h->local_elements + 3), h->stream));
        }
        uw.cnt = h->ray_cnt;
    }
    Timed t(h, PF_K_UPDATE);
    HIP_TRY(h, pfk_update_hits(h->stream, uw));
    HIP_TRY(h, pfk_update_rays(h->stream, uw));
    return PF_OK;
}

static int grow_scratch(pf_handle* h, size_t need);
// S3P:315-376. points: host rows (cloud_dev == nullptr), or cloud_dev: device rows of which the vote takes
// the n_points rows pick[] names (S3P:320-322 done by the rotation kernel's loads).
static int scan_vote(pf_handle* h, const double* points, const double* cloud_dev, const int32_t* pick,
                     int64_t n_points, const double pos[3], const double zyx[3], int32_t n_yaw, int32_t yaw_k0,
                     int32_t yaw_kstep, int32_t n_t, double length, int32_t* sums_out, int64_t* best_out) {
    const double pi = 2 * acos(0.0);                              // S3P:24
    const double res = h->cfg.resolution;
    const int per = n_t * n_t * n_t, nh = n_yaw * per;
    std::vector<double> mats(9 * (size_t)n_yaw), trans(3 * (size_t)nh);
    for (int r = 0; r < n_yaw; ++r) {
        const double z = zyx[0] + (yaw_k0 + yaw_kstep * r) * pi / 180, y = zyx[1], x = zyx[2];   // S3P:329
        double* m = &mats[9 * (size_t)r];
        pf_rot_zyx(m, z, y, x);    // S3P:536-544
        int k = r * per;
        for (int a = 0; a < n_t; ++a)
            for (int b = 0; b < n_t; ++b)
                for (int d = 0; d < n_t; ++d, ++k) {              // S3P:335, 341, 348
                    trans[3 * (size_t)k + 0] = pos[0] + a * res - (length - res) / 2;
                    trans[3 * (size_t)k + 1] = pos[1] + b * res - (length - res) / 2;
                    trans[3 * (size_t)k + 2] = pos[2] + d * res - (length - res) / 2;
                }
    }
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, wait_map_writes(h));
    // scratch: matrices, translations, rotated points, sums; kept by the handle and grown on demand
    const size_t nb_m = sizeof(double) * mats.size(), nb_t = sizeof(double) * trans.size();
    const size_t nb_r = sizeof(double) * 3 * (size_t)n_yaw * (size_t)(n_points ? n_points : 1);
    const size_t nb_s = up256(sizeof(int32_t) * (size_t)nh);
    // the sampled points are staged here too, not in the scan buffers of pf_set_scan: a later
    // pf_update_map on this handle still marches over the scan it was given
    const size_t nb_p = cloud_dev ? up256(sizeof(int32_t) * (size_t)(n_points ? n_points : 1))
                                  : sizeof(double) * 3 * (size_t)(n_points ? n_points : 1);
    if (int rc = grow_scratch(h, nb_m + nb_t + nb_r + nb_s + nb_p + 64)) return rc;
    char* dev = h->vote_scratch;
    double* d_m = (double*)dev; double* d_t = (double*)(dev + nb_m);
    double* d_r = (double*)(dev + nb_m + nb_t); int32_t* d_s = (int32_t*)(dev + nb_m + nb_t + nb_r);
    double* d_p = (double*)(dev + nb_m + nb_t + nb_r + nb_s);
    std::vector<int32_t> sums((size_t)nh);
    hipError_t e = hipMemcpyAsync(d_m, mats.data(), nb_m, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_t, trans.data(), nb_t, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess && n_points)
        e = cloud_dev ? hipMemcpyAsync(d_p, pick, sizeof(int32_t) * (size_t)n_points, hipMemcpyHostToDevice, h->stream)
                      : hipMemcpyAsync(d_p, points, sizeof(double) * 3 * (size_t)n_points, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        Timed t(h, PF_K_VOTE);
        e = pfk_scan_vote(h->stream, h->maps, cloud_dev ? cloud_dev : d_p, cloud_dev ? (const int32_t*)d_p : nullptr,
                          (int)n_points, d_m, n_yaw, d_t, per, 1 / res, h->cfg.grid_y, h->cfg.grid_z, h->elements, d_r, d_s);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(sums.data(), d_s, sizeof(int32_t) * (size_t)nh, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(h, PF_ERR_HIP, "pf_scan_vote: %s", hipGetErrorString(e));
    int64_t best = -1; int32_t best_sum = 0;                      // S3P:326, 365
    for (int k = 0; k < nh; ++k)
        if (sums[(size_t)k] > best_sum) { best_sum = sums[(size_t)k]; best = k; }
    *best_out = best;
    if (sums_out) memcpy(sums_out, sums.data(), sizeof(int32_t) * (size_t)nh);
    return PF_OK;
}

// Device scratch shared by the calls that are not part of the per-step loop (pf_scan_vote,
// pf_depth_to_cloud, pf_read_occupied): one allocation, grown on demand, kept by the handle.
static int grow_scratch(pf_handle* h, size_t need) {
    if (need <= h->vote_cap) return PF_OK;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->vote_scratch) (void)hipFree(h->vote_scratch);
    h->vote_scratch = nullptr; h->vote_cap = 0;
    HIP_TRY(h, hipMalloc((void**)&h->vote_scratch, need));
    h->vote_cap = need;
    return PF_OK;
}


int pf_scan_vote(pf_handle* h, const double* points, int64_t n_points, const double pos[3],
                 const double zyx[3], int32_t n_yaw, int32_t yaw_k0, int32_t yaw_kstep,
                 int32_t n_t, double length, int32_t* sums_out, int64_t* best_out) {
    if (!h) return PF_ERR_ARG;
    NEED(h, h->cfg.map_mode == PF_MAP_SHARED, "needs a PF_MAP_SHARED handle (the grid voted on)");
    NEED(h, pos && zyx && best_out, "null argument");
    NEED(h, n_points >= 0 && n_points <= h->cfg.max_points && (n_points == 0 || points), "n_points beyond max_points");
    NEED(h, n_yaw >= 1 && n_yaw <= 64 && n_t >= 1 && n_t <= 16, "lattice size");
    for (int64_t i = 0; i < 3 * n_points; ++i)
        if (!(fabs(points[i]) < 1.0e6)) return fail(h, PF_ERR_ARG, "pf_scan_vote: point %lld not finite/too far", (long long)(i / 3));
    return scan_vote(h, points, nullptr, nullptr, n_points, pos, zyx, n_yaw, yaw_k0, yaw_kstep, n_t, length, sums_out, best_out);
}

int pf_scan_vote_device(pf_handle* h, const double* cloud_dev, int64_t n_cloud, const int32_t* pick, int64_t n_points,
                        const double pos[3], const double zyx[3], int32_t n_yaw, int32_t yaw_k0, int32_t yaw_kstep,
                        int32_t n_t, double length, int32_t* sums_out, int64_t* best_out) {
    if (!h) return PF_ERR_ARG;
    NEED(h, h->cfg.map_mode == PF_MAP_SHARED, "needs a PF_MAP_SHARED handle (the grid voted on)");
    NEED(h, pos && zyx && best_out, "null argument");
    NEED(h, n_cloud >= 0 && n_cloud < (1ll << 31) && (n_cloud == 0 || cloud_dev), "n_cloud");
    NEED(h, n_points >= 0 && n_points <= h->cfg.max_points && (n_points == 0 || pick), "n_points beyond max_points");
    NEED(h, n_yaw >= 1 && n_yaw <= 64 && n_t >= 1 && n_t <= 16, "lattice size");
    for (int64_t i = 0; i < n_points; ++i)
        if (pick[i] < 0 || pick[i] >= n_cloud)
            return fail(h, PF_ERR_ARG, "pf_scan_vote_device: pick[%lld] outside [0, n_cloud)", (long long)i);
    return scan_vote(h, nullptr, n_points ? cloud_dev : nullptr, pick, n_points, pos, zyx, n_yaw, yaw_k0, yaw_kstep, n_t,
                     length, sums_out, best_out);
}

int pf_sequential_cdf(pf_handle* h, const double* w, int64_t n, int32_t how, double* cdf_out) {
    if (!h) return PF_ERR_ARG;
    NEED(h, w && cdf_out && n >= 1 && n <= (int64_t)1 << 30, "null argument / n outside 1..2^30");
    NEED(h, how == 0 || how == 1 || how == 2, "how");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t nb_w = sizeof(double) * (size_t)n, nb_c = sizeof(double) * ((size_t)n + 1);
    const size_t nb_s = how == 2 ? pfk_cdf_scratch_bytes((int)n) : 0;
    if (int rc = grow_scratch(h, up256(nb_w) + up256(nb_c) + nb_s)) return rc;
    double* d_w = (double*)h->vote_scratch;
    double* d_c = (double*)(h->vote_scratch + up256(nb_w));
    void* d_s = (void*)(h->vote_scratch + up256(nb_w) + up256(nb_c));
    HIP_TRY(h, hipMemcpyAsync(d_w, w, nb_w, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, pfk_cdf_of(h->stream, d_w, (int)n, d_c, how, d_s));
    HIP_TRY(h, hipMemcpyAsync(cdf_out, d_c, nb_c, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return PF_OK;
}

// S3P:562-612. cloud_dev == nullptr: the cloud lands in the handle's scratch (*cloud_at says where);
// the count is the one value that comes back to the host here.
static int depth_cloud(pf_handle* h, const char* who, const uint8_t* image, int32_t width, int32_t height, int32_t pool,
                       double focal, double* cloud_dev, int64_t capacity, double** cloud_at, int32_t* n_out) {
    if (!(pool >= 1 && pool <= 16 && width >= pool && height >= pool && width <= 16384 && height <= 16384))
        return fail(h, PF_ERR_ARG, "%s: image size / pool factor", who);
    if (!(focal > 0 && focal < 1.0e9)) return fail(h, PF_ERR_ARG, "%s: focal length", who);
    const int64_t npix = (int64_t)(width / pool) * (height / pool);
    if (capacity < npix) return fail(h, PF_ERR_ARG, "%s: the cloud holds fewer than (width/pool)*(height/pool) points", who);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t nb_i = up256((size_t)width * height * 2), nb_d = up256(sizeof(int4) * (size_t)npix);
    const size_t nb_c = cloud_dev ? 0 : up256(sizeof(double) * 3 * (size_t)npix);
    const size_t nb_b = up256(sizeof(int32_t) * (size_t)((npix + 255) / 256));
    if (int rc = grow_scratch(h, nb_i + nb_d + nb_c + 256 + nb_b)) return rc;
    char* dev = h->vote_scratch;
    uint8_t* d_img = (uint8_t*)dev; int4* d_pix = (int4*)(dev + nb_i);
    double* d_cloud = cloud_dev ? cloud_dev : (double*)(dev + nb_i + nb_d);
    int32_t* d_n = (int32_t*)(dev + nb_i + nb_d + nb_c);
    int32_t* d_b = (int32_t*)(dev + nb_i + nb_d + nb_c + 256);
    int32_t n = 0;
    HIP_TRY(h, hipMemcpyAsync(d_img, image, (size_t)width * height * 2, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, pfk_depth_to_cloud(h->stream, d_img, width, height, pool, focal, d_pix, d_cloud, d_n, d_b));
    HIP_TRY(h, hipMemcpyAsync(&n, d_n, sizeof n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    *cloud_at = d_cloud; *n_out = n;
    return PF_OK;
}

int pf_depth_to_cloud(pf_handle* h, const uint8_t* image, int32_t width, int32_t height, int32_t pool,
                      double focal, double* cloud_out, int64_t capacity, int64_t* n_out) {
    if (!h) return PF_ERR_ARG;
    NEED(h, image && cloud_out && n_out, "null argument");
    double* d_cloud = nullptr; int32_t n = 0;
    if (int rc = depth_cloud(h, "pf_depth_to_cloud", image, width, height, pool, focal, nullptr, capacity, &d_cloud, &n)) return rc;
    if (n > 0) {
        HIP_TRY(h, hipMemcpyAsync(cloud_out, d_cloud, sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    *n_out = n;
    return PF_OK;
}

int pf_depth_to_cloud_device(pf_handle* h, const uint8_t* image, int32_t width, int32_t height, int32_t pool,
                             double focal, double* cloud_dev, int64_t capacity, int64_t* n_out) {
    if (!h) return PF_ERR_ARG;
    NEED(h, image && cloud_dev && n_out, "null argument");
    double* at = nullptr; int32_t n = 0;
    if (int rc = depth_cloud(h, "pf_depth_to_cloud_device", image, width, height, pool, focal, cloud_dev, capacity, &at, &n)) return rc;
    *n_out = n;
    return PF_OK;
}

int pf_read_occupied(pf_handle* h, int32_t particle, int32_t threshold, int32_t stride, float* xyz_out,
                     int64_t capacity, int64_t* n_out, int64_t* total_out) {
    if (!h) return PF_ERR_ARG;
    NEED(h, n_out && capacity >= 0 && (capacity == 0 || xyz_out), "null argument");
    NEED(h, stride >= 1, "stride");
    NEED(h, capacity <= (int64_t)1 << 28, "capacity beyond 2^28 points");
    const uint8_t* grid = h->maps;
    if (h->slab) {                                    // like pf_read_map: slab i of the local shard
        NEED(h, particle >= 0 && particle < h->P, "particle index");
        grid += h->slab * (unsigned long long)particle;
    }
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, wait_map_writes(h));
    const long long cand = (h->elements + stride - 1) / stride;
    const int nb = (int)((cand + 1023) / 1024);
    const size_t nb_b = up256(sizeof(int32_t) * (size_t)nb), nb_x = up256(sizeof(float) * 3 * (size_t)capacity);
    if (int rc = grow_scratch(h, nb_b + nb_x + 256)) return rc;
    char* dev = h->vote_scratch;
    int32_t* d_b = (int32_t*)dev; float* d_x = (float*)(dev + nb_b); int32_t* d_n = (int32_t*)(dev + nb_b + nb_x);
    int32_t total = 0;
    HIP_TRY(h, pfk_read_occupied(h->stream, grid, h->elements, stride, threshold, h->cfg.grid_y, h->cfg.grid_z,
                                 h->cfg.resolution, d_b, nb, d_n, d_x, capacity));
    HIP_TRY(h, hipMemcpyAsync(&total, d_n, sizeof total, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const int64_t n = total < capacity ? total : capacity;
    if (n > 0) {
        HIP_TRY(h, hipMemcpyAsync(xyz_out, d_x, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    *n_out = n;
    if (total_out) *total_out = total;
    return PF_OK;
}

int pf_write_local_map(pf_handle* h, const int8_t* src) {
    if (!h) return PF_ERR_ARG;
    NEED(h, src && h->local, "no local map / null src");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipMemcpyAsync(h->local, src, (size_t)h->local_elements, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return PF_OK;
}
int pf_read_local_map(pf_handle* h, int8_t* dst) {
    if (!h) return PF_ERR_ARG;
    NEED(h, dst && h->local, "no local map / null dst");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipMemcpyAsync(dst, h->local, (size_t)h->local_elements, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return PF_OK;
}
int pf_reset_local_map(pf_handle* h) {
    if (!h) return PF_ERR_ARG;
    NEED(h, h->local, "no local map");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipMemsetAsync(h->local, 0, (size_t)h->local_elements, h->stream));
    return PF_OK;
}

int pf_ring_default_params(pf_ring_params* p) {
    if (!p) return PF_ERR_ARG;
    p->n_slots = 5; p->slot_capacity = 19200; p->life_new = 6; p->reserved = 0;   // S3P:286, 463
    p->move_threshold = 0.15; p->recentre_xy = 2.0; p->recentre_z = 1.0;          // S3P:428, 485-486
    return PF_OK;
}

int pf_ring_init(pf_handle* h, const pf_ring_params* p) {
    if (!h) return PF_ERR_ARG;
    NEED(h, p, "null params");
    NEED(h, h->cfg.map_mode == PF_MAP_SHARED, "needs a PF_MAP_SHARED handle (the localisation grid)");
    NEED(h, p->n_slots >= 1 && p->n_slots <= 64 && p->slot_capacity >= 1 && p->slot_capacity <= (1 << 24),
         "n_slots in 1..64, slot_capacity in 1..2^24");
    NEED(h, h->elements < (1ll << 31), "grid too large for int32 indices");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->ring_idx) { (void)hipFree(h->ring_idx); h->ring_idx = nullptr; }
    if (h->ring_pts) { (void)hipFree(h->ring_pts); h->ring_pts = nullptr; }
    HIP_TRY(h, dalloc(&h->ring_pts, 3 * (size_t)p->slot_capacity));
    if (!h->ring_cnt) HIP_TRY(h, dalloc(&h->ring_cnt, 1));
    const size_t n = (size_t)p->n_slots * (size_t)p->slot_capacity;
    HIP_TRY(h, dalloc(&h->ring_idx, n));
    HIP_TRY(h, hipMemsetAsync(h->ring_idx, 0, sizeof(int32_t) * n, h->stream));
    h->ring = *p;
    h->ring_count.assign((size_t)p->n_slots, 0);
    h->ring_life.assign((size_t)p->n_slots, 0);                                   // S3P:289
    const double res = h->cfg.resolution;
    h->ring_pos_update[0] = h->cfg.grid_x * res / 2;                              // S3P:416
    h->ring_pos_update[1] = h->cfg.grid_y * res / 2;
    h->ring_pos_update[2] = h->cfg.grid_z * res / 2;
    h->ring_euler_update[0] = h->ring_euler_update[1] = h->ring_euler_update[2] = 0;   // S3P:415
    h->ring_first_run = 1;                                                        // S3P:296
    return PF_OK;
}

int pf_ring_paint(pf_handle* h) {
    if (!h) return PF_ERR_ARG;
    NEED(h, h->ring_idx, "pf_ring_init first");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, wait_map_writes(h));
    PfRingPaint rp{};
    rp.cap = h->ring.slot_capacity;
    for (int i = 0; i < h->ring.n_slots; ++i)                                     // S3P:305-312
        if (h->ring_life[(size_t)i] > 0 && h->ring_count[(size_t)i] > 0) {
            rp.slot[rp.n] = i; rp.count[rp.n] = h->ring_count[(size_t)i]; ++rp.n;
        }
    HIP_TRY(h, pfk_ring_paint(h->stream, h->maps, h->ring_idx, rp));
    h->shadow_dirty = true;
    return PF_OK;
}

// S3P:419-529. cloud: host rows, or (on_device) device rows read where they are.
static int ring_update(pf_handle* h, const double* cloud, bool on_device, int64_t n_points, double pos_scan[3],
                       const double zyx[3], int32_t* flags_out) {
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, wait_map_writes(h));
    const pf_ring_params& rp = h->ring;
    const double res = h->cfg.resolution, conv = 1 / res;                         // S3P:284
    const double cx = h->cfg.grid_x * res / 2, cy = h->cfg.grid_y * res / 2, cz = h->cfg.grid_z * res / 2;
    const double rot_y = zyx[1], rot_x = zyx[2];
    int flags = 0;
    const double update_value =                                                   // S3P:423-425
        fabs(pos_scan[0] - h->ring_pos_update[0]) + fabs(pos_scan[1] - h->ring_pos_update[1])
        + fabs(pos_scan[2] - h->ring_pos_update[2]) + 0.5 * fabs(zyx[0] - h->ring_euler_update[0])
        + 0.5 * fabs(rot_y - h->ring_euler_update[1]) + 0.5 * fabs(rot_x - h->ring_euler_update[2]);
    if (update_value > rp.move_threshold || h->ring_first_run == 1) {             // S3P:428
        flags |= 1;
        for (int i = 0; i < rp.n_slots; ++i) {
            int32_t& life = h->ring_life[(size_t)i];
            if (life == 1 || life <= 0) {                                         // S3P:432
                int32_t* slot = h->ring_idx + (size_t)i * rp.slot_capacity;
                if (life != 0)                                                    // S3P:433-437
                    HIP_TRY(h, pfk_ring_set(h->stream, h->maps, slot, h->ring_count[(size_t)i], 0));
                PfUpdateParams up{};
                const double z = zyx[0], y = rot_y, x = rot_x;
                pf_rot_zyx(up.rot, z, y, x);    // S3P:536-544
                up.pos[0] = pos_scan[0]; up.pos[1] = pos_scan[1]; up.pos[2] = pos_scan[2];
                up.conv = conv; up.elements = h->elements; up.gy = h->cfg.grid_y; up.gz = h->cfg.grid_z;
                int32_t cnt = 0;
                if (n_points) {
                    if (!on_device)
                        HIP_TRY(h, hipMemcpyAsync(h->ring_pts, cloud, sizeof(double) * 3 * (size_t)n_points,
                                                  hipMemcpyHostToDevice, h->stream));
                    HIP_TRY(h, hipMemsetAsync(h->ring_cnt, 0, sizeof(int32_t), h->stream));
                    HIP_TRY(h, pfk_ring_fill(h->stream, on_device ? cloud : h->ring_pts, (int)n_points, up, slot, h->ring_cnt));
                    HIP_TRY(h, hipMemcpyAsync(&cnt, h->ring_cnt, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
                    HIP_TRY(h, hipStreamSynchronize(h->stream));
                }
                h->ring_count[(size_t)i] = cnt;                                   // S3P:444-455
                life = rp.life_new;                                               // S3P:463
                break;
            }
        }
        for (int i = 0; i < rp.n_slots; ++i) h->ring_life[(size_t)i] += -1;       // S3P:470-472
        h->ring_first_run = 0;
        for (int k = 0; k < 3; ++k) h->ring_pos_update[k] = pos_scan[k];          // S3P:475-480
        h->ring_euler_update[0] = zyx[0]; h->ring_euler_update[1] = rot_y; h->ring_euler_update[2] = rot_x;
        h->shadow_dirty = true;
    }
    if (fabs(pos_scan[0] - cx) > rp.recentre_xy || fabs(pos_scan[1] - cy) > rp.recentre_xy ||
        fabs(pos_scan[2] - cz) > rp.recentre_z) {                                 // S3P:485-486
        flags |= 2;
        const long long ox = (long long)round((cx - pos_scan[0]) * conv);         // S3P:489-491
        const long long oy = (long long)round((cy - pos_scan[1]) * conv);
        const long long oz = (long long)round((cz - pos_scan[2]) * conv);
        const long long corr = ox * h->cfg.grid_y * h->cfg.grid_z + oy * h->cfg.grid_z + oz;   // S3P:494
        for (int i = 0; i < rp.n_slots; ++i) {
            int32_t* slot = h->ring_idx + (size_t)i * rp.slot_capacity;
            const int cnt_old = h->ring_count[(size_t)i];
            HIP_TRY(h, pfk_ring_set(h->stream, h->maps, slot, cnt_old, 0));       // S3P:498-500
            int32_t cnt = 0;
            if (cnt_old > 0) {
                HIP_TRY(h, pfk_ring_shift(h->stream, slot, cnt_old, corr, h->elements, h->ring_cnt));   // S3P:502-516
                HIP_TRY(h, hipMemcpyAsync(&cnt, h->ring_cnt, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
                HIP_TRY(h, hipStreamSynchronize(h->stream));
            }
            h->ring_count[(size_t)i] = cnt;
        }
        h->ring_pos_update[0] += cx - pos_scan[0]; pos_scan[0] = cx;              // S3P:519-524
        h->ring_pos_update[1] += cy - pos_scan[1]; pos_scan[1] = cy;
        h->ring_pos_update[2] += cz - pos_scan[2]; pos_scan[2] = cz;
        h->shadow_dirty = true;
    }
    if (flags_out) *flags_out = flags;
    return PF_OK;
}

int pf_ring_update(pf_handle* h, const double* cloud, int64_t n_points, double pos_scan[3],
                   const double zyx[3], int32_t* flags_out) {
    if (!h) return PF_ERR_ARG;
    NEED(h, h->ring_idx, "pf_ring_init first");
    NEED(h, pos_scan && zyx && (n_points == 0 || cloud), "null argument");
    NEED(h, n_points >= 0 && n_points <= h->ring.slot_capacity, "n_points beyond slot_capacity");
    for (int k = 0; k < 3; ++k)
        NEED(h, isfinite(pos_scan[k]) && fabs(pos_scan[k]) < 1e6 && isfinite(zyx[k]), "pose not finite");
    for (int64_t i = 0; i < 3 * n_points; ++i)
        if (!(fabs(cloud[i]) < 1.0e6)) return fail(h, PF_ERR_ARG, "pf_ring_update: point %lld not finite/too far", (long long)(i / 3));
    return ring_update(h, cloud, false, n_points, pos_scan, zyx, flags_out);
}

int pf_ring_update_device(pf_handle* h, const double* cloud_dev, int64_t n_points, double pos_scan[3],
                          const double zyx[3], int32_t* flags_out) {
    if (!h) return PF_ERR_ARG;
    NEED(h, h->ring_idx, "pf_ring_init first");
    NEED(h, pos_scan && zyx && (n_points == 0 || cloud_dev), "null argument");
    NEED(h, n_points >= 0 && n_points <= h->ring.slot_capacity, "n_points beyond slot_capacity");
    for (int k = 0; k < 3; ++k)
        NEED(h, isfinite(pos_scan[k]) && fabs(pos_scan[k]) < 1e6 && isfinite(zyx[k]), "pose not finite");
    return ring_update(h, cloud_dev, true, n_points, pos_scan, zyx, flags_out);
}

int pf_ring_state(pf_handle* h, int32_t* counts, int32_t* lives, double pos_update[3],
                  double euler_update[3], int32_t* idx_out) {
    if (!h) return PF_ERR_ARG;
    NEED(h, h->ring_idx, "pf_ring_init first");
    for (int i = 0; i < h->ring.n_slots; ++i) {
        if (counts) counts[i] = h->ring_count[(size_t)i];
        if (lives) lives[i] = h->ring_life[(size_t)i];
    }
    for (int k = 0; k < 3; ++k) {
        if (pos_update) pos_update[k] = h->ring_pos_update[k];
        if (euler_update) euler_update[k] = h->ring_euler_update[k];
    }
    if (idx_out) {
        HIP_TRY(h, hipSetDevice(h->cfg.device));
        HIP_TRY(h, hipMemcpyAsync(idx_out, h->ring_idx, sizeof(int32_t) * (size_t)h->ring.n_slots * h->ring.slot_capacity,
                                  hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    return PF_OK;
}

int pf_comm_unique_id(void* id_out, int64_t capacity) {
    if (!id_out || capacity < (int64_t)sizeof(ncclUniqueId))
        return fail(nullptr, PF_ERR_ARG, "pf_comm_unique_id: needs %d bytes", (int)sizeof(ncclUniqueId));
    if (!rccl().ok) return fail(nullptr, PF_ERR_NODEVICE, "pf_comm_unique_id: librccl not loadable");
    ncclUniqueId id;
    const ncclResult_t r = rccl().GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, PF_ERR_HIP, "ncclGetUniqueId: %s", rccl().GetErrorString(r));
    memcpy(id_out, &id, sizeof id);
    return PF_OK;
}

int pf_comm_init(pf_handle* h, const void* id, int32_t rank, int32_t world) {
    if (!h) return PF_ERR_ARG;
    NEED(h, id && world >= 1 && rank >= 0 && rank < world, "id / rank / world");
    NEED(h, (long long)h->P * world == h->Pg && h->offset == rank * h->P,
         "shards must be equal contiguous blocks: n_particles * world == n_particles_global, particle_offset == rank * n_particles");
    NEED(h, !h->comm, "communicator already set");
    if (!rccl().ok) return fail(h, PF_ERR_NODEVICE, "pf_comm_init: librccl not loadable");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    const ncclResult_t r = rccl().CommInitRank(&h->comm, world, uid, rank);
    if (r != ncclSuccess) {
        h->comm = nullptr;
        return fail(h, PF_ERR_HIP, "ncclCommInitRank: %s", rccl().GetErrorString(r));
    }
    h->comm_world = world;
    h->comm_rank = rank;
    // test switch: on one rank, per-particle maps move between virtual shards through the same send/receive groups
    if (const char* e = getenv("SLAM_PF_VIRTUAL_SHARDS")) {
        const int v = atoi(e);
        if (world == 1 && h->slab && v > 1 && h->P % v == 0) h->virtual_shards = v;
    }
    HIP_TRY(h, materialize_pose(h));
    HIP_TRY(h, dalloc(&h->scores_all, (size_t)h->Pg));
    HIP_TRY(h, dalloc(&h->bmin_all, (size_t)world * (size_t)h->n_bmin));
    HIP_TRY(h, dalloc(&h->poses_all, PF_POSE_ROWS * (size_t)h->Pg));
    return PF_OK;
}

int pf_set_stream(pf_handle* h, void* hip_stream) {
    if (!h) return PF_ERR_ARG;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, flush_update(h));
    HIP_TRY(h, hipStreamSynchronize(h->stream));            // nothing of ours left on the old one
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return PF_OK;
}

int pf_synchronize(pf_handle* h) {
    if (!h) return PF_ERR_ARG;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, wait_map_writes(h));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return PF_OK;
}

int pf_profile(pf_handle* h, int enable) {
    if (!h) return PF_ERR_ARG;
    h->profile = enable != 0;
    h->profile_mask = enable == 1 ? ~0u : (unsigned)enable >> 1;
    return PF_OK;
}

int pf_kernel_time(pf_handle* h, int id, double* total_ms, int64_t* launches) {
    if (!h) return PF_ERR_ARG;
    NEED(h, id >= 0 && id < PF_K_COUNT && total_ms && launches, "kernel id / null output");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    double tot = 0;
    for (auto& e : h->ev[id]) {
        float ms = 0;
        HIP_TRY(h, hipEventElapsedTime(&ms, e.a, e.b));
        tot += ms;
        h->ev_pool.push_back(e);
    }
    *total_ms = tot;
    *launches = (int64_t)h->ev[id].size();
    h->ev[id].clear();
    return PF_OK;
}

}  // extern "C"
