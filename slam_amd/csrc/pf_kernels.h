// Internal launch interface between the C ABI (pf_api.hip) and the gfx950 kernels
// (pf_kernels.hip). Not part of the public boundary (include/slam_pf.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// One beam of the step's beam list as the kernels read it (32 B, one s_load_dwordx8):
// the integer voxel offsets of S3P:875-877 already promoted to double (the promotion
// S3P:745-747 performs per evaluation is done once per beam on the host).
struct PfBeam {
    double lx, ly, lz;
    int32_t v;       // int8 local-map value (S3P:886, 895)
    int32_t pad;
};

// The same beam for the fp32 filter of the fast measure kernels (16 B, one s_load_dwordx4).
struct PfBeamF {
    float lx, ly, lz;
    int32_t v;
};
// Gather layout of the shared grid ("bricks"): 4 x 4 x 4 cells per 64-B line, byte address of cell
// (X, Y, Z) = ((X/4 * ny4 + Y/4) * nz4 + Z/4) * 64 + (X%4)*16 + (Y%4)*4 + Z%4 with ny4 = ceil(gy/4),
// nz4 = ceil(gz/4); bytes = ceil(gx/4)*ny4*nz4*64. a = ny4*nz4*64, b = nz4*64.
// kind 1 ("tiles", k_measure_tile): 8 x 8 cells of ONE z layer per 64-B line, z outermost, one tile of margin
// around every layer (a 2 x 2-tile window that holds a cell of the grid never leaves the array): byte address
// of cell (X, Y, Z) = Z*a + (Y/8 + 1)*b + (X/8 + 1)*64 + (X%8)*8 + Y%8 with b = ntx*64, a = nty*b,
// ntx = ceil(gx/8) + 2, nty = ceil(gy/8) + 2; bytes = gz*a. The tiles (TX, TY), (TX+1, TY) are 128 contiguous bytes.
#define PF_LAYOUT_BRICKS 0
#define PF_LAYOUT_TILES 1
struct PfSwz {
    uint32_t gx, gy, gz, a, b;
    uint32_t kind;
};
// Up to four occupied beams of one local column (same lx, ly) for k_measure_col (32 B, one scalar
// load): X, Y, the rounding test, the axis checks and the column's part of the address are per
// column, only Z and the gather per beam. lz ascending; lz[k >= n] unused.
struct __attribute__((aligned(32))) PfColG {
    float lx, ly;
    int32_t n, pad;
    int32_t lz[4];
};

// A spatially compact cluster of column groups for k_measure_box (32 B, one scalar load): its groups
// cg[g0 .. g1) and the bounds of their local coordinates.
struct __attribute__((aligned(32))) PfCluster {
    int32_t g0, g1;
    float lx0, lx1, ly0, ly1;
    int32_t lz0, lz1;
};

// One local column of the FULL beam list for k_measure_fuse_cols (32 B): a run of beams that share
// (lx, ly) and ascend strictly in lz from z0 over n <= PF_FCOL_MAXZ cells (cells of the run without a
// beam hold the value 0); cnt of them count as occupied (S3P:886). Its int8 values lie in the values
// array in 16-byte units aligned to the local z grid: unit voff covers lz in [floor16(z0), +16); a zero
// unit lies in front of and behind every column.
#define PF_FCOL_MAXZ 65
struct __attribute__((aligned(32))) PfFCol {
    float lx, ly;
    int32_t z0, n, voff, cnt, pad[2];
};
// bounds of the whole beam list (z1 inclusive)
struct PfFColBounds {
    float lx0, lx1, ly0, ly1;
    int32_t z0, z1;
};

// Pose block: double [PF_POSE_ROWS][n] = x, y, z, yaw, cos, sin, x/res, y/res, z/res.
#define PF_POSE_ROWS 9

struct PfGridParams {
    double gyz;         // gy*gz as double (S3P:85: x*(15/0.05)*(4/0.05))
    double gzd;         // gz as double
    double elements;    // gx*gy*gz as double
    int32_t occ_threshold;
    int32_t lo, hi;
};

// K1 (+K2): S3P:859-908. partial: int2 [n_chunks][P].
hipError_t pfk_measure(hipStream_t st, const double* pose, int P, uint8_t* maps,
                       unsigned long long map_stride, const PfBeam* beams, int M, int n_chunks,
                       int beams_per_chunk, PfGridParams gp, bool fuse, int2* partial);
// K1+K2 column-wise on per-particle maps (pf_fuse_cols.hip). Needs gz % 16 == 0, map_stride % 16 == 0,
// elements < 2^31, every beam value non-zero. partial: int2 [P]. fuse = false: measure only (S3P:859-890),
// beams/cols/vals then hold the occupied beams only and nothing is written.
hipError_t pfk_measure_fuse_cols(hipStream_t st, const double* pose, int P, uint8_t* maps, unsigned long long map_stride,
                                 const PfBeam* beams, int M, const PfFCol* cols, int n_cols, const void* vals,
                                 PfFColBounds bd, int gx, int gy, int gz, PfGridParams gp, int2* partial, bool fuse);
// The occupied-beam lists are padded to a multiple of PF_BEAM_GROUP with out-of-bounds beams;
// M and bpc passed to pfk_measure_fast are multiples of it.
#define PF_BEAM_GROUP 4
// K1 fast form (no fuse): bf/bd hold the OCCUPIED beams only (v > occ_threshold); lmax =
// max over them of max(|lx|+|ly|, |lz|). Exactness argument in pf_kernels.hip / DESIGN.md.
hipError_t pfk_measure_fast(hipStream_t st, const double* pose, int P, const uint8_t* maps,
                            unsigned long long map_stride, const PfBeamF* bf, const PfBeam* bd,
                            int M, int n_chunks, int bpc, PfGridParams gp, double lmax,
                            const int32_t* order, int2* partial);
// Lane order for K1: order[lane] = particle, bucketed by (yaw, x, y). hist/next_hist: two
// int32 [PF_ORDER_BUCKETS] buffers used alternately (hist must be zero on entry; the call
// leaves next_hist zeroed). keys: int32 [P] scratch.
#define PF_ORDER_BUCKETS 32768
struct PfOrderParams {
    double qa, qx, qy;   // cells per radian of yaw / per voxel of x / per voxel of y
    int32_t mode;        // 0: lexicographic yaw,x,y (6/4/5 bits)  1: Morton  2: yaw-major 8/2/2 bits
    int32_t bits;        // Morton: bits per axis, 4 (4096 buckets) or 5 (32768 buckets)
    int32_t zsplit;      // 1 (Morton only): the top two key bits are round(z/res) & 3 (the z class), below them yaw with
                         // `bits` bits and x, y with bits - 1 each; every class starts at a multiple of 64 lanes (holes:
                         // order[] = -1), so a wave of k_measure_tile holds ONE value of round(z/res) unless the cloud
                         // spans more than four of them
    const int32_t* ext;  // (device, may be null) the extent of the cloud the last lane ordering saw: PF_EXT_WORDS ordered
                         // ints = min, max of yaw, x/res, y/res (pf_ext_encode), word k at ext[k * PF_EXT_STRIDE]. When set and not empty, the cells are
                         // sized so that the key's range covers 1.125 x that extent: the keys wrap, which is harmless
                         // while the cloud is narrower than the range and puts far-apart particles into one wave when
                         // it is not (fixed cells of 1/200 rad wrapped at 0.08 rad on the bench's cloud)
};
#define PF_EXT_STRIDE 32            // every word of the record on a 128-B line of its own (the atomics of a launch meet there)
#define PF_EXT_WORDS (6 * PF_EXT_STRIDE)
#define PF_ZCLASSES 4
#ifdef __HIPCC__
__device__ inline int pf_spread3(int v) {       // 5 bits -> every third bit
    return (v & 1) | ((v & 2) << 2) | ((v & 4) << 4) | ((v & 8) << 6) | ((v & 16) << 8);
}
// floats as ints whose signed order is the floats' (for atomicMin / atomicMax), and back
__device__ inline int pf_ext_encode(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ inline float pf_ext_decode(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }
// K1's lane-order key of particle i of a pose block with n columns
__device__ inline int order_key(const double* __restrict__ pose, int n, int i, PfOrderParams op) {
    if (op.ext) {                                   // cells from the last ordering's extent (wave-uniform loads)
        const float a0 = pf_ext_decode(op.ext[0]), a1 = pf_ext_decode(op.ext[PF_EXT_STRIDE]);
        const float x0 = pf_ext_decode(op.ext[2 * PF_EXT_STRIDE]), x1 = pf_ext_decode(op.ext[3 * PF_EXT_STRIDE]);
        const float y0 = pf_ext_decode(op.ext[4 * PF_EXT_STRIDE]), y1 = pf_ext_decode(op.ext[5 * PF_EXT_STRIDE]);
        if (a1 >= a0 && x1 >= x0 && y1 >= y0) {     // (not the empty record; NaN-free: the poses are validated)
            const double na = (double)(1 << op.bits), nxy = (double)(1 << (op.zsplit ? op.bits - 1 : op.bits));
            op.qa = na / fmax(1.125 * (double)(a1 - a0), na / op.qa * 0.25);     // never finer than a quarter of the
            op.qx = nxy / fmax(1.125 * (double)(x1 - x0), nxy / op.qx * 0.25);   // default cells
            op.qy = nxy / fmax(1.125 * (double)(y1 - y0), nxy / op.qy * 0.25);
        }
    }
    const int a = (int)floor(pose[3 * (size_t)n + i] * op.qa);
    const int x = (int)floor(pose[6 * (size_t)n + i] * op.qx);
    const int y = (int)floor(pose[7 * (size_t)n + i] * op.qy);
    if (op.mode == 0) return ((a & 63) << 9) | ((x & 15) << 5) | (y & 31);
    if (op.mode == 2) return ((a & 255) << 4) | ((x & 3) << 2) | (y & 3);      // yaw-major, 12 bits (k_measure_box)
    if (op.zsplit) {
        const int zc = (int)round(pose[8 * (size_t)n + i]) & (PF_ZCLASSES - 1);   // K1's zp (S3P:880 for z), wrapping
        const int m = (1 << (op.bits - 1)) - 1;
        return (zc << (3 * op.bits - 2)) | (((a >> (op.bits - 1)) & 1) << (3 * (op.bits - 1))) |
               (pf_spread3(a & m) << 2) | (pf_spread3(x & m) << 1) | pf_spread3(y & m);
    }
    const int m = (1 << op.bits) - 1;                                          // Morton
    return (pf_spread3(a & m) << 2) | (pf_spread3(x & m) << 1) | pf_spread3(y & m);
}
#endif
// sorted: double [5][lanes] = cos, sin, x/res, y/res, z/res in lane order (K1's prologue).
// hist_counted: keys/hist were already filled for these poses (pfk_resample's fill stage).
// lanes: P, or with op.zsplit pfk_padded_lanes(P): order[] then has lanes entries, -1 in the holes.
static inline int pfk_padded_lanes(int P) { return (P + 63) / 64 * 64 + PF_ZCLASSES * 64; }
// ext_out / ext_reset (device, may be null): PF_EXT_WORDS ints each; the scatter stage folds the extent of these
// poses into ext_out (which must hold the empty record on entry) and leaves the empty record in ext_reset.
hipError_t pfk_order(hipStream_t st, const double* pose, int P, int lanes, int32_t* keys, int32_t* hist,
                     int32_t* next_hist, int32_t* order, double* sorted, PfOrderParams op,
                     bool hist_counted, int32_t* ext_out = nullptr, int32_t* ext_reset = nullptr);
// A column group for k_measure_tile (64 B): up to PF_TILE_GROUP occupied beams of one local column, and instead
// of their lz the byte offset of each beam's z layer above the list's lowest one in the tile layout, lza[k] =
// (lz[k] - lzmin) * a, which the kernel hands to the load as its scalar offset as it comes. The paths that need
// lz itself read it from a PfColG list with two entries per group (beams 0-3 and 4-7).
#define PF_TILE_GROUP 8
struct __attribute__((aligned(64))) PfTileG {
    float lx, ly;
    int32_t n, pad;
    uint32_t lza[PF_TILE_GROUP];
    int32_t pad1[4];
};
// K1 on the tile layout (shared grid only): one coalesced window load per beam and wave, every lane's byte
// handed over by ds_bpermute (pf_measure_tile.hip). cg: a table of n_chunks records of 64 B (ints: [k] = end of
// the chunk's groups with PF_TILE_GROUP - k beams, [PF_TILE_GROUP] = its start; indices into the list behind the
// table), then the column groups chunk by chunk, inside a chunk by their number of beams (descending); cgz: the
// groups' lz in the same order (two PfColG per group); lzmin, lzmax: the bounds of lz over the list.
// lin: the same grid in the reference's order (the beams that leave the grid read it through the flat index).
// lanes = pfk_padded_lanes(P), order/sorted from pfk_order with op.zsplit; partial: int2 [n_chunks][lanes]
// in lane order.
hipError_t pfk_measure_tile(hipStream_t st, const double* sorted, int lanes, const int32_t* order,
                            const uint8_t* shadow, const uint8_t* lin, PfSwz sw, const PfTileG* cg, const PfColG* cgz,
                            int lzmin, int lzmax, int n_chunks, PfGridParams gp, double lmax, int2* partial);
int pfk_tile_trip(int n_beams);     // column groups per trip of k_measure_tile for groups of n_beams beams
// K1 on the gather layout (shared grid only), the beams grouped by column (PfColG): cg[0 .. n_runs) are
// RUN groups (the occupied cells of one column inside an 8-cell z window: lz[0] = first z, lz[1], lz[2] =
// a byte of 1 per occupied cell 0-3, 4-7, n = their number), then the plain groups sorted by their number
// of beams: group_ends = ends of the 4-, 3-, 2- and 1-beam groups in cg[].
#define PF_RUN_MIN 4            // beams an 8-cell window must hold to become a run group (three gathers)
hipError_t pfk_measure_col(hipStream_t st, const double* sorted, int P, const uint8_t* shadow,
                           PfSwz sw, const PfColG* cg, int n_runs, const int group_ends[4], int n_chunks,
                           PfGridParams gp, double lmax, const int32_t* order, int2* partial);
// K1 on LDS-staged boxes of the reference-order shared grid (pf_measure_box.hip). cl: the clusters,
// ordered by blockIdx.y group; group_start: int [groups + 1] into cl. partial_l: int2 [groups][P] in
// LANE order (pfk_reduce_scores_l un-permutes through order[]). kp, parts, lp, groups: pfk_box_plan.
// Needs gz % 4 == 0 (rows are staged as aligned dwords).
void pfk_box_plan(int P, int n_clusters, int* kp_out, int* parts_out, int* lp_out, int* groups_out);
hipError_t pfk_measure_box(hipStream_t st, const double* sorted, int P, const uint8_t* maps, int gx, int gy, int gz,
                           const PfColG* cg, const PfCluster* cl, const int* group_start, int kp, int parts, int lp,
                           int groups, PfGridParams gp, double lmax, int2* partial_l);
hipError_t pfk_reduce_scores_l(hipStream_t st, const int2* partial_l, int n_groups, int P, const int32_t* order,
                               int2* scores, int32_t* bmin, int32_t* zero_a, int32_t* zero_b, int n_zero);
hipError_t pfk_swz_build(hipStream_t st, const uint8_t* lin, uint8_t* shadow, PfSwz sw);
// bmin: int32 [ceil(P/64)], the smallest sum_match of each 64-particle block
// zero_a / zero_b: n_zero int32 each set to zero (the ordering's counters), n_zero may be 0
hipError_t pfk_reduce_scores(hipStream_t st, const int2* partial, int n_chunks, int P, int2* scores,
                             int32_t* bmin, int32_t* zero_a, int32_t* zero_b, int n_zero);
hipError_t pfk_voxel_indices(hipStream_t st, const double* pose, int P, int i, const PfBeam* beams,
                             int M, PfGridParams gp, int32_t* flat_out);

// Scratch of the multi-block weight/resample kernels; nb = ceil(Pg / 256).
struct PfResampleScratch {
    double* prm;        // [8]
    double* bsum;       // [nb]
    double* bmax;       // [nb]
    double* bse;        // [nb]
    double* boff;       // [nb]
    double* bbest_v;    // [nb]
    int2* bcnt;         // [nb]
    int2* boff2;        // [nb]
    int32_t* bbest_i;   // [nb]
    int32_t* ndup;      // [1]
    void* segs;         // [Pg / 8192 + 2] records of 128 B: the segments of the many-workgroup sequential-order CDF (k_cdfx_*)
};

struct PfUpdateWork;
// K3: S3P:920-937. w in/out [Pg]; logw out [Pg]; cdf out [Pg+1] (the FAST-order kernels also leave the CDF at
// the end of every 256-particle block in cdf[Pg+1 .. Pg+1+ceil(Pg/256)) for K4's search); scratch double [Pg+1].
// Also initialises K4's first[Pg] table and counters[0..2] for this step.
// bmin (may be null): n_bmin (0: ceil(Pg/64)) block minima of exactly these scores, as
// pfk_reduce_scores writes them (gathered rank by rank on a sharded handle); saves the min pass.
// carry (may be null; only when pfk_weights_carries(...)): a map update that does not depend on
// the weights and that nothing else reads until the next map reader: its hits ride as extra
// workgroups of the raw-weight kernel, its rays as extra workgroups of the normalise kernel
// (both pairs are latency-bound and small, side by side they take the time of the longer one).
hipError_t pfk_weights(hipStream_t st, const int2* scores, int Pg, double* w, float* logw,
                       double* cdf, double* scratch, bool exact, int32_t* first,
                       unsigned long long* counters, PfResampleScratch sc, const int32_t* bmin, const PfUpdateWork* carry = nullptr, int n_bmin = 0);

// K4: S3P:757-821, the whole chain on the stream, no host round trip: search (+ re-search of
// ambiguous draws on the sequential-order CDF when bound > 0), slot assignment, permutation of
// weights / pose rows / map slabs, arg-max. counters: unsigned long long [3] = {no-hit draws,
// ambiguous draws, no-hit draws after the re-search}. host_out: device-visible pinned host
// memory, one result slot (PF_RESULT_HEADER unsigned long long, then double [nb] and int32 [nb],
// nb = ceil(Pg/256)). The default form leaves the last stage of the arg-max to the host: the
// fill kernel writes every 256-particle block's best weight and index into the slot and
// {counters[0..2], PF_RESULT_BLOCKS | nb} into its header (pfk_result_best reads them); the
// one-stage-per-launch and one-workgroup forms end with the arg-max kernel, which writes
// {counters[0..2], best}.
// order_hist/order_keys (may be null): the one-GPU fill stage also counts K1's lane-order keys of
// the poses it leaves behind (pfk_order's first stage, saved for the next pf_measure).
// pose_src (may be null = pose): where the predicted pose rows of this handle still are when
// pf_predict_device left them in the caller's block: every slot's row is then written into pose
// (one GPU: read from pose_src; sharded: from pose_gathered, which was gathered from pose_src).
hipError_t pfk_resample(hipStream_t st, const double* cdf, const double* u, int Pg, int P,
                        int offset, double bound, double* w, float* logw, double* pose,
                        const double* pose_src, const double* pose_gathered, uint8_t* maps, unsigned long long map_stride,
                        unsigned long long map_bytes, int32_t* draw, int32_t* first,
                        int32_t* dup_anc, int32_t* free_slot, int32_t* slot_anc, int32_t* best,
                        unsigned long long* counters, PfResampleScratch sc,
                        unsigned long long* host_out, int32_t* order_hist, int32_t* order_keys,
                        PfOrderParams op, bool tops_valid = false, int map_vshard = 0);

#define PF_RESULT_HEADER 4
#define PF_RESULT_BLOCKS (1ull << 63)
static inline size_t pfk_result_slot_bytes(int Pg) {
    const size_t nb = ((size_t)Pg + 255) / 256;
    return (PF_RESULT_HEADER * sizeof(unsigned long long) + nb * (sizeof(double) + sizeof(int32_t)) + 63) & ~(size_t)63;
}
// S3P:814-820 from a finished result slot (host memory): first strict maximum above 0, else prev.
int32_t pfk_result_best(const unsigned long long* slot, int32_t prev);
// cdf[0] = 0, cdf[i+1] = cdf[i] + w[i] in index order (S3P:763-767), one workgroup: how = 0 the
// one-lane chain, how = 1 the workgroup scan that gives the same bits (k_cdf_exact_if's form)
// text pf_last_error(NULL) returns (errors of calls that have no handle)
void pfk_set_global_error(const char* text);
hipError_t pfk_cdf_of(hipStream_t st, const double* w, int n, double* cdf, int how, void* scratch = nullptr);
size_t pfk_cdf_scratch_bytes(int n);     // device scratch of how = 2 (the many-workgroup form)
// resample() on its own (weights given by the caller): sequential-order CDF + K4's initial state
hipError_t pfk_resample_prepare(hipStream_t st, const double* w, int Pg, double* cdf, int32_t* first,
                                unsigned long long* counters);

// K5: S3P:628-704. rot9: row-major ZYX matrix from the host libm. wpts: double [M][3] scratch.
struct PfUpdateParams {
    double rot[9];
    double pos[3];
    double conv;          // 1/resolution (S3P:284)
    double max_range;
    long long elements;
    int32_t gy, gz;
    int32_t hit_inc, hit_clamp, miss_inc, miss_clamp;
    int32_t is_signed;
    void* shadow;         // gather-layout copy of the shared grid to keep in step, or null
    uint32_t swz_a, swz_b, swz_kind;
};
// One map update: M points (sensor frame) -> wpts (grid frame) and hits, then R rays.
struct PfUpdateWork {
    void* grid;
    const double* points;   // [M][3]
    double* wpts;           // [M][3]
    const int32_t* ray_idx; // [R]
    int M, R;
    PfUpdateParams up;
    int near_steps;         // > 0: the rays' first near_steps steps are counted per cell in LDS by d5_rays_near's workgroups
                            // and d5_rays starts behind them; 0: d5_rays marches every step
    // pfk_update_rays' counting form (cnt != null; a grid of its own, no gather-layout copy): the march only COUNTS
    // its touches per cell, a second launch applies count * miss_inc with one clamp per counted cell
    unsigned* cnt;          // [elements rounded up to 4], 16-byte aligned, all zero between calls
    int near_wgs;
};
// Every ray starts at the sensor: its first steps land in the same few hundred cells as every other ray's, and a
// compare-and-swap per touch on those words serialises (2400 rays on the reference's empty local map: 217 us of
// ray march, profiles/r3_frame_*). The first PF_NEAR_STEPS steps are therefore counted per cell in an LDS table
// over the box of PF_NEAR_R cells around the sensor's cell, and each cell gets ONE touch of count * increment
// (k touches of v -> max(v + inc, clamp) are one touch of k * inc: same result, any order).
#define PF_NEAR_STEPS 14
#define PF_NEAR_R (PF_NEAR_STEPS + 2)
#define PF_NEAR_W (2 * PF_NEAR_R + 1)
#define PF_NEAR_WGS 8       // workgroups that share the rays' near steps (each applies its own counts)
bool pfk_weights_carries(int Pg, bool exact, const int32_t* bmin);
// block minima of sum_match for scores from outside (pf_weights with gathered scores): bmin[pfk_block_minima_count(Pg)]
int pfk_block_minima_count(int Pg);
hipError_t pfk_block_minima(hipStream_t st, const int2* scores, int Pg, int32_t* bmin);
hipError_t pfk_update_hits(hipStream_t st, const PfUpdateWork& uw);
hipError_t pfk_update_rays(hipStream_t st, const PfUpdateWork& uw);
hipError_t pfk_fill_u8(hipStream_t st, uint8_t* dst, unsigned long long n, int value);

// K6: S3P:328-376. mats: double [n_yaw][9]; trans: double [n_yaw*n_per_yaw][3]; rot: scratch
// double [n_yaw][N][3]; sums: int32 [n_yaw*n_per_yaw]. pick (may be null): the N rows of points voted with.
hipError_t pfk_scan_vote(hipStream_t st, const uint8_t* grid, const double* points, const int32_t* pick, int N,
                         const double* mats, int n_yaw, const double* trans, int n_per_yaw,
                         double conv, int gy, int gz, long long elements, double* rot,
                         int32_t* sums);

// Keyframe ring (S3P:305-312, 419-529). slot: int32 [cap] flat indices of one ring slot.
hipError_t pfk_ring_set(hipStream_t st, uint8_t* grid, const int32_t* slot, int count, int value);
// the live slots of a ring int32 [n_slots][cap] painted with 1 in one launch: slot[k], count[k] for k < n <= 64
struct PfRingPaint { int32_t n, cap; int32_t slot[64]; int32_t count[64]; };
hipError_t pfk_ring_paint(hipStream_t st, uint8_t* grid, const int32_t* ring, const PfRingPaint& rp);
// *count_out (zero on entry) += in-bounds points; up: rot, pos, conv, gy, gz, elements.
hipError_t pfk_ring_fill(hipStream_t st, const double* points, int n, PfUpdateParams up, int32_t* slot,
                         int32_t* count_out);
hipError_t pfk_ring_shift(hipStream_t st, int32_t* slot, int count, long long corr, long long elements,
                          int32_t* count_out);

// Depth front-end (S3P:562-612). image: device copy of the uint16 little-endian depth bytes;
// dpix: int4 [npix] scratch; cloud: double [npix][3]; *count_out = points written; bcnt (may be null): int32
// [ceil(npix/256)] scratch for the many-workgroup compaction (images of up to 262 144 pooled pixels).
hipError_t pfk_depth_to_cloud(hipStream_t st, const uint8_t* image, int width, int height, int pool,
                              double focal, int4* dpix, double* cloud, int32_t* count_out, int32_t* bcnt);
// dst[k] = src[k] for the 3n coordinates of n points, NaN where |src[k]| < 1e6 does not hold (pf_set_scan_device).
hipError_t pfk_copy_points(hipStream_t st, const double* src, int n, double* dst);
// Map read-out (S3P:1031-1051). bcnt: int32 [nb], nb = ceil(ceil(elements/stride)/1024);
// *total_out = qualifying cells (may exceed cap; only the first cap are written).
hipError_t pfk_read_occupied(hipStream_t st, const uint8_t* map, long long elements, int stride,
                             int threshold, int gy, int gz, double res, int32_t* bcnt, int nb,
                             int32_t* total_out, float* xyz, long long cap);
