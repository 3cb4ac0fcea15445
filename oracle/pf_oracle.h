/* TEST INFRASTRUCTURE — CPU restatement ("oracle") of the particle-filter hot path of
 * Gabrieleenx/SLAM's Slam_3d_particle.cpp (S3P). Plain C, all sizes run-time parameters, all
 * random numbers taken as INPUTS. Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product path (slam_amd/, include/) never does.
 *
 * Pinning status:
 *   pfo_measure / pfo_predict / pfo_weights / pfo_resample  — PINNED against the reference's
 *     own particle_map() (S3P:827-962) and resample() (S3P:757-821), compiled from the
 *     reference's text by oracle/build_ref.sh and run in the build container; vectors in
 *     tests/golden/ (generator: oracle/make_golden.py).
 *   pfo_update_map / pfo_scan_vote / pfo_ring_* / pfo_depth_to_cloud / pfo_read_occupied — PINNED
 *     against the reference's class Slam: build_local_map (S3P:621-710), scan_localization
 *     (S3P:300-410), scan_localization_map_update (S3P:412-530), the body of callback() (S3P:562-612)
 *     and update_realtime_pos (S3P:712-728), compiled from the class's own text without its ROS
 *     members by oracle/build_ref.sh (+ oracle/ref_harness_slam.cpp: one Slam object and accessors) and
 *     run in the build container; vectors in tests/golden/{helpers_p,depth,vote,local_map,ring,frames}.npz
 *     (generator: oracle/make_golden_frontend.py). frames.npz is 48 consecutive frames of the whole
 *     callback() + particle_map() loop.
 *
 * Deliberate, documented deviations from the reference's undefined behaviour:
 *   D1  S3P:882 accepts flat == p_map_elements (one byte past the map). Here it is out of
 *       bounds.
 *   D2  S3P:772-780 leaves resample_index[i] uninitialised when no interval matches (draw on
 *       a boundary, draw above the last boundary, or draw == 0). Here that draw keeps its own
 *       index i, and the count of such draws is returned.
 *   D3  a rounded coordinate or flat index outside int32 is out of bounds (the reference's
 *       double->int conversion is undefined there).
 */
#ifndef PF_ORACLE_H
#define PF_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pfo_grid {
    int32_t gx, gy, gz;      /* particle/shared grid dims; flat = x*gy*gz + y*gz + z (S3P:84-86) */
    double  resolution;      /* S3P:23 */
    int32_t occ_threshold;   /* beam counts as "occupied" when v > threshold (S3P:886: 20) */
    int32_t map_lo, map_hi;  /* fuse clamp (S3P:896-901: 0, 100) */
} pfo_grid;

/* S3P:868-877: enumerate the non-zero voxels of a local int8 map in ascending flat order as
 * beams (lx,ly,lz) = voxel index minus origin offset (ox,oy,oz) (S3P:875-877: 100,100,30).
 * Returns the number of beams written (at most cap). */
int64_t pfo_snapshot_beams(const int8_t* local, int32_t lgx, int32_t lgy, int32_t lgz,
                           int32_t ox, int32_t oy, int32_t oz,
                           int32_t* bx, int32_t* by, int32_t* bz, int8_t* bv, int64_t cap);

/* S3P:859-908: measurement (+ fuse when add_points). maps: per-particle slabs at
 * maps + i*map_stride (map_stride = 0: one shared grid, fuse must be off).
 * sum_out[i], cnt_out[i]: S3P:887-888 (cnt = num_match of particle i). */
void pfo_measure(const pfo_grid* g, int64_t P,
                 const double* x, const double* y, const double* z, const double* yaw,
                 uint8_t* maps, size_t map_stride,
                 int64_t M, const int32_t* bx, const int32_t* by, const int32_t* bz,
                 const int8_t* bv, int add_points,
                 int32_t* sum_out, int32_t* cnt_out);

/* Same loop, but records the flat index of every beam for ONE particle (-1 = out of bounds):
 * the "voxel indices" parity object. */
void pfo_voxel_indices(const pfo_grid* g, double x, double y, double z, double yaw,
                       int64_t M, const int32_t* bx, const int32_t* by, const int32_t* bz,
                       int32_t* flat_out);

/* S3P:911-914: predict. dlocal = local pose minus its reset value (S3P:911: p_pos_local-5 etc.,
 * computed by the caller), dyaw = p_rot_local_z, noise[4*i..] = pos_d,pos_d,rot_d,rot_d. */
void pfo_predict(int64_t P, double* x, double* y, double* z, double* yaw,
                 double dlx, double dly, double dlz, double dyaw, const float* noise);

/* S3P:920-937: w_i <- normalised weight (sequential sums, reference order).
 * num_match = cnt of the LAST particle (S3P:865, S3P:928). logw (may be NULL) = log(w_i) as
 * float. Returns the pre-normalisation sum. */
double pfo_weights(int64_t P, double* w, const int32_t* sum, int32_t num_match, float* logw);

/* S3P:757-821. u[P] thresholds in draw order. draw_anc[i] = S3P's resample_index[i];
 * slot_anc[j] = S3P's resample_index_org[j]; *best in/out = index_high_weight, evaluated on
 * the permuted weights w (permuted in place). Returns the number of no-hit draws (D2). */
int64_t pfo_resample(int64_t P, double* w, const double* u,
                     int32_t* draw_anc, int32_t* slot_anc, int32_t* best);

/* S3P:805-812 applied to an array of `width`-byte records. */
void pfo_permute(int64_t P, const int32_t* slot_anc, void* records, size_t width);

/* S3P:534-559: full ZYX rotation matrix (row-major 9) from glibc trig. */
void pfo_rot_zyx(double m[9], double z, double y, double x);

typedef struct pfo_update_cfg {
    int32_t gx, gy, gz;       /* grid updated */
    double  resolution;
    int32_t is_signed;        /* 1: int8 cells (S3P:65), 0: uint8 cells */
    int32_t hit_inc, hit_clamp;    /* +10, 50  (S3P:650-655) */
    int32_t miss_inc, miss_clamp;  /* -3, -50  (S3P:695-700) */
    double  max_range;        /* 3.0 (S3P:637) */
} pfo_update_cfg;

/* S3P:628-704. points: M rows of sensor-frame xyz; pose = sensor position in the grid frame
 * (metres) and ZYX angles; ray_idx[R] rows of `points` to ray-cast to (host-drawn). */
void pfo_update_map(const pfo_update_cfg* c, void* grid,
                    const double pos[3], const double zyx[3],
                    int64_t M, const double* points, int64_t R, const int32_t* ray_idx);

/* ---- keyframe ring of the localisation grid: S3P:286-289, 305-312, 419-529 --
 * Pinned since round 2 to the reference's own class Slam (tests/golden/ring.npz, frames.npz). The ring keeps, for n_slots past
 * scans, the flat indices their points fell on; live slots are painted into the binary grid
 * before every vote, a slot is replaced when the pose has moved, and everything is shifted by a
 * flat-index offset when the sensor leaves the middle of the grid.
 * Kept on purpose: the non-compacted write of S3P:454-455 (an in-bounds point j of a new scan is
 * stored at position j of its slot while only the in-bounds COUNT is kept, so later passes over
 * the first `count` entries see holes with older contents and miss the tail).
 * Deviations (defined behaviour for the reference's UB): idx[] and count[] start at zero (the
 * reference leaves points_scan_map_index uninitialised and S3P:495-517 reads it for slots never
 * filled); rounded coordinates outside int32 are out of bounds. */
typedef struct pfo_ring {
    int32_t gx, gy, gz;
    double resolution;
    int32_t n_slots, cap;          /* S3P:286: 5; res_h*res_v = 19200 */
    int32_t* idx;                  /* [n_slots*cap] points_scan_map, caller-allocated, zeroed */
    int32_t* count;                /* [n_slots] points_scan_map_index, zeroed */
    int32_t* life;                 /* [n_slots] points_scan_map_life, zeroed (S3P:289) */
    double pos_update[3];          /* S3P:416: the grid's centre in metres at start */
    double euler_update[3];        /* S3P:415 */
    int32_t first_run;             /* S3P:296: 1 */
    double move_threshold;         /* S3P:428: 0.15 */
    double recentre_xy, recentre_z;/* S3P:485-486: 2, 1 */
    int32_t life_new;              /* S3P:463: 6 */
} pfo_ring;
void pfo_ring_defaults(pfo_ring* r, int32_t gx, int32_t gy, int32_t gz, double resolution);
/* S3P:305-312 */
void pfo_ring_paint(const pfo_ring* r, uint8_t* grid);
/* S3P:419-529. cloud: n sensor-frame points; pos_scan (in/out, S3P:520-524) and zyx =
 * {euler_zyx_scan[0], rot_y, rot_x}. Returns bit 0: a slot was replaced, bit 1: recentred. */
int pfo_ring_update(pfo_ring* r, uint8_t* grid, int64_t n, const double* cloud, double pos_scan[3],
                    const double zyx[3]);

/* ---- depth front-end: S3P:116-124, 155-162, 562-612. Pinned since round 2 to the
 * reference's own callback() body (tests/golden/depth.npz). image: height rows of width little-endian uint16 depths in millimetres (the bytes of
 * sensor_msgs/Image, S3P:587); pooled pool x pool (S3P:580-594: integer mean that ignores zeros),
 * pinhole tables from focal = the pooled focal length (S3P:124: 385/4), integer millimetre
 * products truncated toward zero (S3P:567-568 assign a double to an int), then metres; pixels
 * whose pooled depth is 0 are dropped, row-major order kept (S3P:598-611).
 * cloud_out: [(width/pool)*(height/pool)][3]; returns the number of points. */
int64_t pfo_depth_to_cloud(const uint8_t* image, int32_t width, int32_t height, int32_t pool,
                           double focal, double* cloud_out);

/* ---- map read-out for publishing: S3P:1031-1051 with reverse_index_conv_p S3P:100-110 (SURVEY
 * 8f rank 4). Every stride-th cell above `threshold`, in index order, as the float32 triples the
 * reference puts into sensor_msgs/PointCloud: (X*res, -Z*res, Y*res). The index is split with
 * the reference's double arithmetic truncated to int. Returns the number of points written
 * (at most cap). */
int64_t pfo_read_occupied(const uint8_t* map, int32_t gx, int32_t gy, int32_t gz, double resolution,
                          int32_t threshold, int32_t stride, float* xyz_out, int64_t cap);

/* S3P:328-376: pose-lattice vote on a binary uint8 grid. Yaw hypothesis r is
 * zyx[0] + (yaw_k0 + yaw_kstep*r)*pi/180 (S3P:329: k0 = -6, kstep = 3, n_yaw = 5); translation
 * lattice n_t per axis over `length` (S3P:275, 279: 0.25, 5). sums[n_yaw*n_t^3] per hypothesis
 * in loop order r,x,y,z; returns the arg-max (first strict max over 0, S3P:365) or -1. */
int64_t pfo_scan_vote(const uint8_t* grid, int32_t gx, int32_t gy, int32_t gz, double resolution,
                      int64_t N, const double* points, const double pos[3], const double zyx[3],
                      int32_t n_yaw, int32_t yaw_k0, int32_t yaw_kstep,
                      int32_t n_t, double length, int32_t* sums);

#ifdef __cplusplus
}
#endif
#endif
