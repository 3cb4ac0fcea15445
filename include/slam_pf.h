/* slam_pf.h — C ABI of the MI355X (gfx950) particle-filter hot path.
 *
 * Drop-in seam for Gabrieleenx/SLAM's Slam_3d_particle.cpp ("S3P"). The reference has no FFI:
 * main() calls particle_map() directly (S3P:1106-1108) and everything travels through globals.
 * This header is the seam INSIDE particle_map() / build_local_map(): the caller (the host
 * driver: snapshot under its mutex, motion model + noise, add_points policy, correction
 * feedback) keeps what S3P:833-855, 911-914, 941-953, 956-959 do and hands the loops
 * S3P:859-908, 920-937, 757-821 and 628-704 to the GPU, in the reference's own per-step order
 *     measure (uses the poses of the previous step)  ->  predict  ->  weights  ->  resample
 * plus update_map once per depth frame. INTEGRATION.md shows the edit to the reference.
 *
 * Conventions: every function returns PF_OK (0) or a negative PF_ERR_*; pf_last_error() gives
 * the text. No exceptions cross the boundary, no torch types, plain pointers and sizes. Host
 * pointers are the caller's and may be reused on return. One caller thread per handle (the
 * reference calls particle_map() from main() only). Work is ordered on the handle's own HIP
 * stream; functions that fill host memory return after the data has arrived, the others are
 * asynchronous on that stream. The library is HIP-only: there is no CPU fallback, and
 * pf_create fails when no gfx950 device is usable.
 */
#ifndef SLAM_PF_H
#define SLAM_PF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PF_OK            0
#define PF_ERR_ARG      -1   /* bad argument / size beyond a configured capacity */
#define PF_ERR_HIP      -2   /* a HIP runtime call failed */
#define PF_ERR_STATE    -3   /* call order violated (e.g. weights before measure) */
#define PF_ERR_NOMEM    -4
#define PF_ERR_NODEVICE -5

#define PF_MAP_SHARED        0   /* one uint8 grid read by every particle (S3P:64 style)   */
#define PF_MAP_PER_PARTICLE  1   /* one uint8 slab per particle (S3P:52, FastSLAM style)   */

#define PF_ORDER_AUTO   0   /* exact for n_particles_global <= 2048, else fast             */
#define PF_ORDER_EXACT  1   /* weight sum and CDF added in the reference's sequential order */
#define PF_ORDER_FAST   2   /* fixed-shape parallel sum/scan; ambiguous draws are counted   */

#define PF_GRID_SHARED  0   /* pf_update_map target: the shared uint8 grid                 */
#define PF_GRID_LOCAL   1   /* pf_update_map target: the int8 local map (S3P:65)           */

/* kernel ids for pf_kernel_time */
#define PF_K_MEASURE   0
#define PF_K_WEIGHTS   1
#define PF_K_RESAMPLE  2
#define PF_K_UPDATE    3
#define PF_K_VOTE      4
#define PF_K_ORDER     5   /* K1's lane ordering + score reduction (two launches per pf_measure) */
#define PF_K_COUNT     6

typedef struct pf_handle pf_handle;

/* Replaces the compile-time constants and globals of S3P:23-69. */
typedef struct pf_config {
    int32_t struct_size;          /* = sizeof(pf_config) */
    int32_t device;               /* HIP device ordinal */
    int32_t n_particles;          /* particles held by THIS handle (S3P:68: 50) */
    int32_t n_particles_global;   /* over all handles/ranks; == n_particles on one GPU */
    int32_t particle_offset;      /* global slot number of local slot 0 */
    int32_t map_mode;             /* PF_MAP_* */
    int32_t grid_x, grid_y, grid_z;   /* S3P:40-42: 300,300,80; flat = x*gy*gz + y*gz + z */
    int32_t local_x, local_y, local_z; /* S3P:30-32: 200,200,80; 0 = no local map */
    int32_t max_beams;            /* capacity of pf_set_beams */
    int32_t max_points;           /* capacity of pf_set_scan (points and rays) */
    int32_t occ_threshold;        /* S3P:886: 20 */
    int32_t map_lo, map_hi;       /* S3P:896-901: 0, 100 */
    int32_t map_prior;            /* S3P:74: 50 */
    int32_t order_mode;           /* PF_ORDER_* */
    double  resolution;           /* S3P:23: 0.05 */
} pf_config;

/* Replaces the literals of S3P:637, 650-655, 695-700. */
typedef struct pf_update_params {
    int32_t hit_inc, hit_clamp;     /* +10, 50 (local int8)  */
    int32_t miss_inc, miss_clamp;   /* -3, -50 */
    double  max_range;              /* 3.0 m: endpoints farther than this are not hits */
} pf_update_params;

/* Fills *cfg with the reference's values (S3P:23-69), one GPU, per-particle maps. */
int pf_default_config(pf_config* cfg);

/* Replaces `new map_particles[50]` + initilize_particles_map() (S3P:69-77, S3P:1087): allocates
 * all device state; poses start at S3P:47-50 (7.5,7.5,1.5,0), weights at 1/n_particles_global,
 * maps at map_prior. */
int pf_create(const pf_config* cfg, pf_handle** out);
/* Replaces the delete[]s of S3P:1065-1068. */
void pf_destroy(pf_handle* h);
/* Text of the last error on this handle (h == NULL: of the last failed pf_create). */
const char* pf_last_error(const pf_handle* h);

/* ---- particle state ------------------------------------------------------------------ */

/* Replaces the result of S3P:911-914 (the caller keeps the motion model and its RNG): uploads
 * the local shard's poses. The library computes cos/sin(yaw) with the host libm (as S3P:860
 * does through rot_matrix) and pos/resolution (S3P:862-864) on the host. */
int pf_predict(pf_handle* h, const double* x, const double* y, const double* z, const double* yaw);
/* The same for a caller whose poses already live on this GPU: pose_block_dev is a device
 * pointer to a block laid out like pf_poses_device() (double [9][n_particles]: x, y, z, yaw,
 * cos, sin, x/res, y/res, z/res), e.g. one that an earlier pf_predict produced. No host work.
 * On a handle that holds all particles the block is not copied: the next pf_resample reads
 * the rows from it while it writes the handle's own poses (any other call that needs the
 * poses copies them first), so the block must stay unchanged until that work has run on the
 * handle's stream; a sharded handle copies it at once. Rows 4..8 must be the host-libm values
 * of rows 0..3 for the results to match the reference bit for bit. */
int pf_predict_device(pf_handle* h, const void* pose_block_dev);
int pf_get_poses(pf_handle* h, double* x, double* y, double* z, double* yaw);
/* weight vector of ALL n_particles_global particles (S3P:51; every rank holds the whole vector) */
int pf_set_weights(pf_handle* h, const double* w_global);
int pf_get_weights(pf_handle* h, double* w_global, float* logw_global /* may be NULL */);
/* S3P:756: index_high_weight, the global resample() keeps when no weight is above 0 (S3P:815-820).
 * The handle starts it at 0 like the reference; a host that restores a saved filter sets it here. */
int pf_set_best_index(pf_handle* h, int32_t index);

/* maps: PER_PARTICLE: slab `i` of the local shard; SHARED: i is ignored. Replaces the reads at
 * S3P:1031-1051 and test set-up. */
int pf_fill_maps(pf_handle* h, int value);
int pf_write_map(pf_handle* h, int32_t i, const uint8_t* src);
int pf_broadcast_map(pf_handle* h, const uint8_t* src);   /* every local particle := src */
int pf_read_map(pf_handle* h, int32_t i, uint8_t* dst);

/* ---- measurement: S3P:859-908 ---------------------------------------------------------- */

/* The step's beam list = the caller's snapshot (S3P:833-846, 868-877): non-zero local voxels
 * in ascending flat order as offsets from the local origin, with their int8 value. */
int pf_set_beams(pf_handle* h, const int32_t* bx, const int32_t* by, const int32_t* bz,
                 const int8_t* bv, int64_t n_beams);
/* K1 (+K2 when add_points and PER_PARTICLE): per local particle sum_match / num_match.
 * Any beam list, in any order, gives the reference's results (a beam reads its cell after every earlier
 * beam's add, S3P:886-904). The fast forms apply to lists in the local map's own order (x, y, z ascending,
 * S3P:868): SHARED grids gather the occupied beams of a column together (8-cell z windows with four or
 * more of them as three loads); PER_PARTICLE maps whose grid_z is a multiple of 16 walk the list as whole
 * columns when it holds four or more beams per column, as the reference's dense local maps do. */
int pf_measure(pf_handle* h, int add_points);
int pf_get_scores(pf_handle* h, int32_t* sum_match, int32_t* num_match);
/* device pointer to the local scores, int32 [n_particles][2] = (sum, count): what the caller
 * all-gathers across ranks (RCCL) before pf_weights. */
void* pf_scores_device(pf_handle* h);
/* parity object "voxel indices": flat index (or -1) of every beam for local particle i. */
int pf_voxel_indices(pf_handle* h, int32_t i, int32_t* flat_out);

/* ---- weights: S3P:920-937 ---------------------------------------------------------------- */

/* scores_global_dev: device pointer to int32 [n_particles_global][2] gathered over the ranks
 * in rank order, or NULL on one GPU (the local scores are used). */
int pf_weights(pf_handle* h, const void* scores_global_dev);

/* ---- resample: S3P:757-821 --------------------------------------------------------------- */

/* Normally called after pf_weights. Called after pf_set_weights instead (no pf_weights in between)
 * it is the reference's resample() on its own: the CDF is the sequential prefix sum of the weights
 * as given (S3P:763-767; they need not sum to 1), in the reference's order whatever order_mode says.
 * u_global: n_particles_global thresholds in draw order (reference parity: the mt19937
 * uniforms of S3P:770-773; systematic resampling: (u0+i)/N), or NULL to reuse the last set.
 * The array is copied before the call returns (into pinned memory the search kernel reads: no
 * device copy is enqueued and the call does not wait for the stream). poses_global_dev: device pointer to the ranks' pose blocks gathered in rank order
 * (each block = what pf_poses_device points to), or NULL on one GPU.
 * slot_anc_out[n_particles_global] (may be NULL) = S3P's resample_index_org; *best_out =
 * S3P's index_high_weight; *nohit_out = draws that matched no interval (kept own index);
 * *ambiguous_out = draws closer to a CDF boundary than the FAST-order error bound (they are
 * re-searched on the sequential-order CDF before use; 0 in EXACT order). */
int pf_resample(pf_handle* h, const double* u_global, const void* poses_global_dev,
                int32_t* slot_anc_out, int32_t* best_out, int64_t* nohit_out,
                int64_t* ambiguous_out);
/* The same in two halves, for callers that do not need a step's results before they enqueue the
 * next one (the reference applies the filter's correction asynchronously itself: S3P:956-959
 * writes globals the camera callback consumes later, S3P:712-716). pf_resample_async enqueues
 * everything pf_resample does and returns; slot_anc_out (may be NULL) is filled by the time the
 * matching pf_resample_result returns. pf_resample_result blocks until the OLDEST outstanding
 * resample has finished and hands out its results. At most two may be outstanding (PF_ERR_STATE
 * otherwise); pf_resample itself refuses while one is. */
int pf_resample_async(pf_handle* h, const double* u, const void* poses_global_dev, int32_t* slot_anc_out);
int pf_resample_result(pf_handle* h, int32_t* best_out, int64_t* nohit_out, int64_t* ambiguous_out);
/* device pointer to the local pose block, double [9][n_particles]:
 * x, y, z, yaw, cos, sin, x/res, y/res, z/res. */
void* pf_poses_device(pf_handle* h);

/* ---- map update: S3P:628-704 -------------------------------------------------------------- */

/* points: n_points rows of sensor-frame xyz (S3P:126 cloudpoints); ray_idx: n_rays rows to cast
 * to (S3P:668-672: 2400 host-drawn indices; north-star: every beam). */
int pf_set_scan(pf_handle* h, const double* points, int64_t n_points,
                const int32_t* ray_idx, int64_t n_rays);
/* K5: hits then ray march on `target`. pos: sensor position in the grid frame (metres),
 * zyx: angles (S3P:628-631). On PF_GRID_SHARED the work is recorded and enqueued by whichever
 * comes first: the next pf_weights (it rides in that call's launches) or the next call that
 * reads or writes the grid or the scan (pf_measure, pf_read_map, pf_read_occupied,
 * pf_scan_vote, pf_ring_*, pf_set_scan, pf_write_map ..., pf_synchronize): every such call
 * sees the updated grid, as if the update had run here. */
int pf_update_map(pf_handle* h, int target, const pf_update_params* p,
                  const double pos[3], const double zyx[3]);
int pf_write_local_map(pf_handle* h, const int8_t* src);
int pf_read_local_map(pf_handle* h, int8_t* dst);
int pf_reset_local_map(pf_handle* h);                    /* S3P:750-754 */

/* ---- scan-to-map vote: S3P:328-376 (the per-frame localiser) ------------ */

/* Counts, for every pose hypothesis, the scan points that land on set cells of the handle's
 * SHARED uint8 grid (the reference's map_localization, S3P:64: create the handle with
 * PF_MAP_SHARED and that grid's dims). points: n_points rows of sensor-frame xyz (the 500
 * sampled points of S3P:318-323, drawn by the caller). Yaw hypothesis r = zyx[0] +
 * (yaw_k0 + yaw_kstep*r)*pi/180 (S3P:329: -6, 3, n_yaw 5); translations pos + a*resolution -
 * (length - resolution)/2 per axis, a < n_t (S3P:335: length 0.25, n_t 5). sums_out (may be
 * NULL): int32 [n_yaw*n_t^3] in the reference's loop order r, x, y, z; *best_out = index of the
 * first strict maximum above 0 (S3P:365) or -1. The epilogue S3P:379-404 stays with the caller. */
int pf_scan_vote(pf_handle* h, const double* points, int64_t n_points, const double pos[3],
                 const double zyx[3], int32_t n_yaw, int32_t yaw_k0, int32_t yaw_kstep,
                 int32_t n_t, double length, int32_t* sums_out, int64_t* best_out);

/* ---- the scan vote for callers that cannot allocate ------------------------------------------
 * No counterpart in the reference (its scan_localization() works on class members, S3P:293-295).
 * For callers that may not call hipMalloc or wait: code under HIP graph capture, tight serving
 * loops, harnesses that forbid allocation. Same arithmetic and kernels as pf_scan_vote, no handle:
 * the caller passes the inputs and the output as DEVICE pointers -- grid_dev: uint8 [gx*gy*gz] in the
 * reference's order (S3P:81-83); points_dev: double [n_points][3] sensor-frame points; sums_dev:
 * int32 [n_yaw*n_t^3], the per-hypothesis counts in loop order r, x, y, z (first strict maximum above
 * 0 = the reference's choice, S3P:365) -- and the kernels' scratch (rotation matrices, translations,
 * rotated points) is a fixed 1.7 MiB device array inside the library, one per GPU, held until the
 * process ends. Hence the limits: n_points <= PF_STATIC_MAX_POINTS, n_yaw <= PF_STATIC_MAX_YAW,
 * n_t <= PF_STATIC_MAX_T; beyond them PF_ERR_ARG (text via pf_last_error(NULL)). The call allocates
 * nothing, copies nothing between host and device and does not synchronise: it enqueues three
 * kernels on hip_stream (NULL = the default stream) of `device` and returns. Every call on a GPU
 * uses that GPU's one array: calls that overlap on one GPU (two streams, two threads) race; calls
 * ordered on one stream do not. Nothing in the array needs clearing between calls (each call
 * rewrites what it reads). */
#define PF_STATIC_MAX_POINTS 4096
#define PF_STATIC_MAX_YAW    16
#define PF_STATIC_MAX_T      8
int pf_scan_vote_static(int32_t device, void* hip_stream, const uint8_t* grid_dev, int32_t gx, int32_t gy,
                        int32_t gz, double resolution, const double* points_dev, int64_t n_points,
                        const double pos[3], const double zyx[3], int32_t n_yaw, int32_t yaw_k0,
                        int32_t yaw_kstep, int32_t n_t, double length, int32_t* sums_dev);

/* Set-up half of pf_scan_vote_static (no counterpart in the reference either): takes the address of `device`'s
 * copy of the scratch array (hipGetSymbolAddress: a code-object load and a symbol look-up, which may allocate
 * inside the runtime and must not run under stream capture). Call it once per device before the first
 * pf_scan_vote_static there; without it that first call does the look-up itself. Both calls leave the caller's
 * current device as it was. */
int pf_scan_vote_static_warmup(int32_t device);

/* ---- plumbing ------------------------------------------------------------------------------ */
void* pf_stream(pf_handle* h);                           /* the hipStream_t the handle works on */
/* ---- keyframe ring of the localisation grid: S3P:286-289, 305-312, 419-529 -------------------
 * (the grid pf_scan_vote reads.) On a PF_MAP_SHARED handle holding the binary
 * localisation grid (S3P:64). The ring keeps the flat indices of the last n_slots keyframe scans
 * on the device; live slots are painted before a vote, a slot is replaced when the pose moved by
 * more than move_threshold, and everything shifts by one flat-index offset when the sensor leaves
 * the middle of the grid. The non-compacted write of S3P:454-455 is kept (an in-bounds point j is
 * stored at position j of its slot, only the in-bounds count is kept). Deviation: slot contents
 * and counts start at zero (the reference reads points_scan_map_index uninitialised, S3P:495). */
typedef struct pf_ring_params {
    int32_t n_slots;          /* S3P:286: 5 */
    int32_t slot_capacity;    /* res_h*res_v = 19200 */
    int32_t life_new;         /* S3P:463: 6 */
    int32_t reserved;
    double move_threshold;    /* S3P:428: 0.15 */
    double recentre_xy;       /* S3P:485: 2 (metres from the grid's middle) */
    double recentre_z;        /* S3P:486: 1 */
} pf_ring_params;
int pf_ring_default_params(pf_ring_params* p);
/* Allocates the ring (zeroed) and sets pos_update to the grid's middle, first_run = 1 (S3P:296, 415-416). */
int pf_ring_init(pf_handle* h, const pf_ring_params* p);
/* S3P:305-312: grid[idx] = 1 for the entries of every live slot. */
int pf_ring_paint(pf_handle* h);
/* S3P:419-529, "after sensor fusion". cloud: n_points rows of sensor-frame xyz (cloudpoints);
 * pos_scan: in/out (the recentring resets it to the grid's middle, S3P:520-524); zyx =
 * {euler_zyx_scan[0], rot_y, rot_x}. *flags_out: bit 0 a slot was replaced, bit 1 recentred. */
int pf_ring_update(pf_handle* h, const double* cloud, int64_t n_points, double pos_scan[3],
                   const double zyx[3], int32_t* flags_out);
/* Test/debug read-out: counts[n_slots], lives[n_slots], pos_update[3], euler_update[3], and (when
 * idx_out != NULL) the n_slots*slot_capacity indices. */
int pf_ring_state(pf_handle* h, int32_t* counts, int32_t* lives, double pos_update[3],
                  double euler_update[3], int32_t* idx_out);

/* ---- depth front-end: S3P:562-612 with the tables S3P:116-124, 155-162 -----
 * Replaces callback()'s pooling and projection. image: host pointer to height rows of width
 * little-endian uint16 depths in millimetres (sensor_msgs/Image::data, S3P:587); pooled pool x pool
 * with the integer mean that ignores zeros (S3P:580-594); focal = the pooled focal length
 * (S3P:124: 385.0/4.0); millimetre products truncated toward zero (S3P:567-568), metres
 * (S3P:603-605); pixels whose pooled depth is 0 dropped, row-major order kept. cloud_out: host
 * double [capacity][3], capacity >= (width/pool)*(height/pool); *n_out = points written
 * (the reference's cloudpoints.size()). */
int pf_depth_to_cloud(pf_handle* h, const uint8_t* image, int32_t width, int32_t height, int32_t pool,
                      double focal, double* cloud_out, int64_t capacity, int64_t* n_out);

/* ---- the cloud kept on the GPU between the stages of callback(): S3P:126, 168-205 --------------
 * The reference's cloudpoints (S3P:126) is one member array that img_to_cloud() fills and
 * scan_localization(), scan_localization_map_update() and build_local_map() read (S3P:320-322,
 * 439-440, 628-631). These four calls are the same stages with that array in HBM: cloud_dev is the
 * CALLER's device buffer, double [capacity][3] on the handle's GPU, so no point crosses PCIe: what
 * comes back per frame is the point count (the caller draws its indices from it, S3P:317, 666), the
 * vote's sums and the ring's in-bounds count. Results are those of the host forms bit for bit
 * (tests/test_gpu_frontend.py). Each call reads or writes cloud_dev on its handle's stream: handles on
 * different streams (the localisation grid's and the local map's, say) are ordered by the caller,
 * most simply by pf_set_stream(mapper, pf_stream(loc)). The host forms reject a coordinate that is
 * not finite or not below 1e6 in magnitude; here nobody on the host sees the points, so such a
 * coordinate makes its point contribute nothing (no vote, no ring entry, no hit, no ray) -- clouds
 * from pf_depth_to_cloud_device (uint16 millimetres) never hold one.
 *
 * S3P:562-612 as pf_depth_to_cloud, the cloud left in cloud_dev; *n_out = its rows. */
int pf_depth_to_cloud_device(pf_handle* h, const uint8_t* image, int32_t width, int32_t height, int32_t pool,
                             double focal, double* cloud_dev, int64_t capacity, int64_t* n_out);
/* S3P:315-376 as pf_scan_vote with the points the rows pick[0..n_points) of cloud_dev (S3P:320-322:
 * points_scan[i] = cloudpoints[rand_idx]); pick: host, every entry in [0, n_cloud). */
int pf_scan_vote_device(pf_handle* h, const double* cloud_dev, int64_t n_cloud, const int32_t* pick,
                        int64_t n_points, const double pos[3], const double zyx[3], int32_t n_yaw,
                        int32_t yaw_k0, int32_t yaw_kstep, int32_t n_t, double length, int32_t* sums_out,
                        int64_t* best_out);
/* S3P:419-529 as pf_ring_update, the cloud read where it is. */
int pf_ring_update_device(pf_handle* h, const double* cloud_dev, int64_t n_points, double pos_scan[3],
                          const double zyx[3], int32_t* flags_out);
/* S3P:628-631, 668-672 as pf_set_scan: the handle takes its own copy of the rows (device to device, on
 * its stream), so cloud_dev may be rewritten once that copy has run; ray_idx: host. */
int pf_set_scan_device(pf_handle* h, const double* cloud_dev, int64_t n_points, const int32_t* ray_idx,
                       int64_t n_rays);

/* ---- map read-out for publishing: S3P:1031-1051 with reverse_index_conv_p S3P:100-110 ---------
 *  Every stride-th cell (S3P:1031: 4) of a grid (slab `particle` of the local
 * shard, ignored on a PF_MAP_SHARED handle, like pf_read_map) above threshold (S3P:1033: 80), in index
 * order, as the float32 triples the reference puts into sensor_msgs/PointCloud: (X*res, -Z*res,
 * Y*res), the index split in the reference's double arithmetic. xyz_out: host float [capacity][3].
 * *n_out = triples written (<= capacity); *total_out (may be NULL) = cells that qualified. */
int pf_read_occupied(pf_handle* h, int32_t particle, int32_t threshold, int32_t stride, float* xyz_out,
                     int64_t capacity, int64_t* n_out, int64_t* total_out);

/* ---- the resampler's CDF on its own: S3P:763-767 ------------------------------------------------
 * cdf_out[0] = 0, cdf_out[i+1] = cdf_out[i] + w[i] in index order, fp64, for any host vector w[n]
 * (cdf_out: host double [n+1]). how = 0: one lane adds in order, the loop as the reference wrote it;
 * how = 1: the workgroup-wide form pf_resample uses to settle draws on CDF boundaries at particle
 * counts where the chain would cost milliseconds; how = 2: the same chain by many workgroups (from 262 144
 * particles on: segments of 8192 weights summarised as functions of the running sum's parity, one workgroup
 * walking the segments with the exact sum, the others filling them in). All return the same bits for every
 * input (zeros, ties, negative and non-finite weights included); exported so that this can be checked directly. */
int pf_sequential_cdf(pf_handle* h, const double* w, int64_t n, int32_t how, double* cdf_out);

/* ---- multi-GPU: the exchange step inside the library ----------------------------
 * No counterpart in the reference (it is one process on one CPU thread, S3P:1107). One process per
 * GPU, one sharded handle per process. Rank 0 calls pf_comm_unique_id (128 bytes) and hands the id
 * to the other ranks by whatever channel the host program has; every rank then calls pf_comm_init.
 * After that pf_weights(h, NULL) all-gathers the scores (8 B per particle) and pf_resample(h, u,
 * NULL, ...) all-gathers the pose blocks (72 B per particle) itself, with RCCL on the handle's
 * stream. RCCL is loaded at run time (librccl.so.1); PF_ERR_NODEVICE when it is not there.
 * EXPERIMENTAL: the grouped all-gathers and the send/receive migration of per-particle maps have run
 * on one rank only (the development box has one GPU); the same host logic is covered at world sizes 2
 * and 4 over gloo with a test double (tests/test_sharded_gloo.py). Until a multi-GPU run is recorded,
 * the documented default for sharded use is to do the two all-gathers yourself (torch.distributed or
 * your own RCCL calls on pf_scores_device() / pf_poses_device()) and pass the gathered pointers. */
int pf_comm_unique_id(void* id_out, int64_t capacity);
int pf_comm_init(pf_handle* h, const void* id, int32_t rank, int32_t world);

/* Make the handle enqueue on the CALLER's hipStream_t (e.g. the stream its RCCL collectives
 * are ordered against), so that hand-offs need no host synchronisation; NULL returns to the
 * handle's own stream. The caller keeps ownership of its stream. */
int pf_set_stream(pf_handle* h, void* hip_stream);
int pf_synchronize(pf_handle* h);
/* HIP-event timing of the kernels, per class; off by default. enable: 0 off, 1 every class,
 * otherwise 2 * (bit mask of PF_K_* classes), e.g. 2 * (1 << PF_K_MEASURE) times K1 only (two
 * event records per step instead of eight). */
int pf_profile(pf_handle* h, int enable);
int pf_kernel_time(pf_handle* h, int kernel_id, double* total_ms, int64_t* launches);

/* ---- host-driver helpers (no GPU): the caller-side lines of the reference ------------------- */

/* S3P:868-877: beams of an int8 local map, ascending; returns the count (or -needed if > cap). */
int64_t pfh_snapshot_beams(const int8_t* local, int32_t lgx, int32_t lgy, int32_t lgz,
                           int32_t ox, int32_t oy, int32_t oz,
                           int32_t* bx, int32_t* by, int32_t* bz, int8_t* bv, int64_t cap);
/* S3P:911-914 with the noise given (4 floats per particle: pos_d, pos_d, rot_d, rot_d). */
void pfh_predict(int64_t n, double* x, double* y, double* z, double* yaw,
                 double dlx, double dly, double dlz, double dyaw, const float* noise);
/* the libstdc++ draws of S3P:852-855 / S3P:768-773 from one seed */
void pfh_draw_noise(uint32_t seed, int64_t n, float sigma_pos, float sigma_rot, float* noise);
void pfh_draw_uniform(uint32_t seed, int64_t n, double* u);
/* S3P:941-953: returns the new add_points and updates last[4] when the pose moved > 0.4. */
int pfh_add_points_policy(const double pose[4], double last[4], double threshold);
/* The pose hypothesis behind an index pf_scan_vote returns (loop order r, x, y, z of S3P:328-348):
 * out = {Tx_loc, Ty_loc, Tz_loc, rot_z_loc} as S3P:366-371 would have stored them. Returns 0, or -1
 * when `index` is outside the lattice (pf_scan_vote's "no vote" -1 included): out = the input pose
 * then. The reference leaves Tx_loc ... rot_z_loc at the previous frame's values in that case
 * (members, S3P:365-372 not entered), which equal the input pose unless the ring recentred in
 * between (S3P:520-524); a caller that wants exactly that keeps its previous `out` on -1. */
int pfh_vote_hypothesis(int64_t index, const double pos[3], const double zyx[3], double resolution,
                        int32_t n_yaw, int32_t yaw_k0, int32_t yaw_kstep, int32_t n_t, double length,
                        double out[4]);
/* S3P:315-319 and S3P:666-669: n draws of uniform_int_distribution<>(0, hi), hi INCLUSIVE as the
 * reference writes it (index hi of the cloud is one past the last point: the caller decides what
 * lives there), from one mt19937(seed). libstdc++'s algorithm, like the other pfh_draw_*. */
void pfh_draw_indices(uint32_t seed, int64_t n, int32_t hi, int32_t* idx);
/* S3P:379-404, the epilogue of scan_localization(): loc = {Tx_loc, Ty_loc, Tz_loc, rot_z_loc}
 * (pfh_vote_hypothesis). pos_scan[3] and *yaw_scan (euler_zyx_scan[0]) are in/out; gyr_yaw =
 * gyr_acc_filter_zyx[0], *gyr_yaw_old = euler_zyx_diff_gyr_old (in/out). Outputs:
 * pos_xyz_diff_scan[3], euler_zyx_diff_scan[0], euler_zyx_diff. */
void pfh_vote_epilogue(double pos_scan[3], double* yaw_scan, const double loc[4], double gyr_yaw,
                       double* gyr_yaw_old, double pos_diff_out[3], double* yaw_diff_scan_out,
                       double* yaw_diff_out);
/* S3P:712-728, update_realtime_pos(): the dead-reckoned global pose advances by the frame's scan
 * increment (rotated by the current yaw) plus the filter's pending correction feedback[4] =
 * {pos_xyz_diff[3], rot_z_diff} (S3P:956-959), yaw wrapped to (-pi, pi]; the correction is consumed
 * (zeroed). pos[3] and *yaw are in/out. */
void pfh_update_realtime_pos(double pos[3], double* yaw, const double pos_diff_scan[3], double yaw_diff,
                             double feedback[4]);
/* S3P:207-222, the pose message convention: position (x, -z, y), orientation (w, x, y, z) =
 * (cos(a), 0, sin(a), 0) with a = (-yaw - pi/2)/2. */
void pfh_pose_message(const double pos[3], double yaw, double position_out[3],
                      double orientation_wxyz_out[4]);
/* S3P:805-812 across shards (per-particle maps on several GPUs; what pf_resample does inside when
 * the handle has a communicator, exported for hosts that move the slabs themselves). slot_anc:
 * the n_global ancestors pf_resample returned; shards of n_local slots. Every replaced slot whose
 * ancestor lives on another rank is one transfer, numbered in slot order (the same number on
 * every rank). Out, for `rank`: the slabs it sends (local index of the ancestor, destination rank,
 * transfer number) and receives (local index of the slot, source rank, number), each in
 * ascending number, at most cap entries written per list. Returns the number of transfers of the
 * whole job. */
int64_t pfh_migration_plan(const int32_t* slot_anc, int64_t n_global, int64_t n_local, int32_t rank,
                           int32_t* send_src, int32_t* send_peer, int64_t* send_seq, int64_t* n_send,
                           int32_t* recv_dst, int32_t* recv_peer, int64_t* recv_seq, int64_t* n_recv,
                           int64_t cap);
/* S3P:805-812 on the caller's own per-particle records. */
void pfh_permute(int64_t n, const int32_t* slot_anc, void* records, size_t width);
/* How pf_set_beams deals K1's column groups to the chunks of its launch (no counterpart in the reference, whose loop
 * S3P:868-890 walks one list; exported so that the dealing can be checked without a GPU). The groups are sorted by
 * their number of beams, group_ends[k] = end of the groups with 8 - k beams; the groups of a count go trips[k] to a trip,
 * the trips of the whole list are numbered through and trip j belongs to chunk j % n_chunks. Out: perm[i] = index in the
 * sorted list of the group at position i of the chunk-major list (a chunk's groups together, counts descending);
 * table[9 * y + k] = end of chunk y's groups with 8 - k beams in that list, table[9 * y + 8] = its start. */
void pfh_deal_groups(const int32_t group_ends[8], int32_t n_chunks, const int32_t trips[8], int32_t* perm,
                     int32_t* table);

#ifdef __cplusplus
}
#endif
#endif
