/* TEST INFRASTRUCTURE — see pf_oracle.h for status, pinning and deviations.
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared pf_oracle.c -lm  (no FMA contraction: the
 * reference is plain x86-64 g++ -O2 code, every product and sum rounds separately). */
#define _GNU_SOURCE                                 /* sincos() */
#include "pf_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define I32_LIM 2147483648.0

/* The sine and cosine of one angle, as the reference's g++ -O2 build takes them: one call of glibc's
 * sincos(), into which g++ merges sin(a) and cos(a). Written out so that the oracle does not depend
 * on its own optimisation level: glibc's sincos() differs from sin() and cos() in the last place for
 * about one argument in a thousand, enough to move a ray length off a rounding tie (S3P:676). */
#define SINCOS(a, s, c) double s, c; sincos((a), &s, &c)

/* round() result -> int32 with D3: returns 0 and sets *bad when not representable. */
static inline int32_t to_i32(double r, int* bad) {
    if (!(r > -I32_LIM && r < I32_LIM)) { *bad = 1; return 0; }
    return (int32_t)r;
}

int64_t pfo_snapshot_beams(const int8_t* local, int32_t lgx, int32_t lgy, int32_t lgz,
                           int32_t ox, int32_t oy, int32_t oz,
                           int32_t* bx, int32_t* by, int32_t* bz, int8_t* bv, int64_t cap) {
    int64_t n = 0;
    int64_t yz = (int64_t)lgy * lgz;
    int64_t total = (int64_t)lgx * yz;
    for (int64_t j = 0; j < total; j++) {           /* S3P:868 */
        if (local[j] == 0) continue;                /* S3P:870 */
        if (n >= cap) break;
        int64_t xi = j / yz;                        /* S3P:92-94 */
        int64_t yi = (j - xi * yz) / lgz;
        int64_t zi = j - xi * yz - yi * lgz;
        bx[n] = (int32_t)xi - ox;                   /* S3P:875-877 */
        by[n] = (int32_t)yi - oy;
        bz[n] = (int32_t)zi - oz;
        bv[n] = local[j];
        n++;
    }
    return n;
}

/* One beam of S3P:874-884 for a particle with rotation (c,s) and translation T (voxels).
 * Returns the flat index or -1 (out of bounds, with D1/D3). */
static inline int64_t beam_flat(const pfo_grid* g, double c, double s, double tx, double ty,
                                double tz, int32_t lx, int32_t ly, int32_t lz) {
    /* rot_matrix(z,0,0) (S3P:733-741) is exactly [[c,-s,0],[s,c,0],[-0,0,1]]; rotate()
     * (S3P:745-747) evaluates (R0*x + R1*y) + R2*z. The R2 terms are +-0 and drop out
     * without changing any rounded value; row 2 gives (double)lz exactly. */
    double ms = -s;
    double rx = (c * (double)lx + ms * (double)ly) + 0.0 * (double)lz;
    double ry = (s * (double)lx + c * (double)ly) + 0.0 * (double)lz;
    double rz = (double)lz;
    int bad = 0;
    int32_t xi = to_i32(round(rx + tx), &bad);      /* S3P:880 */
    int32_t yi = to_i32(round(ry + ty), &bad);
    int32_t zi = to_i32(round(rz + tz), &bad);
    if (bad) return -1;
    int64_t flat = (int64_t)xi * g->gy * g->gz + (int64_t)yi * g->gz + zi;  /* S3P:84-86 */
    int64_t elements = (int64_t)g->gx * g->gy * g->gz;
    if (flat < 0 || flat >= elements) return -1;    /* S3P:882 with D1 */
    return flat;
}

void pfo_measure(const pfo_grid* g, int64_t P,
                 const double* x, const double* y, const double* z, const double* yaw,
                 uint8_t* maps, size_t map_stride,
                 int64_t M, const int32_t* bx, const int32_t* by, const int32_t* bz,
                 const int8_t* bv, int add_points,
                 int32_t* sum_out, int32_t* cnt_out) {
    for (int64_t i = 0; i < P; i++) {               /* S3P:859 */
        SINCOS(yaw[i], s, c);                       /* S3P:860 */
        double tx = x[i] * 1 / g->resolution;       /* S3P:862-864: a division */
        double ty = y[i] * 1 / g->resolution;
        double tz = z[i] * 1 / g->resolution;
        uint8_t* map = maps + (size_t)i * map_stride;
        int32_t sum = 0, cnt = 0;
        for (int64_t j = 0; j < M; j++) {
            int64_t flat = beam_flat(g, c, s, tx, ty, tz, bx[j], by[j], bz[j]);
            if (flat < 0) continue;                 /* S3P:882-884 */
            int v = bv[j];
            if (v > g->occ_threshold) {             /* S3P:886-889 */
                sum += map[flat];
                cnt += 1;
            }
            if (!add_points) continue;              /* S3P:892-894 */
            int nv = (int)map[flat] + v;            /* S3P:895-904 */
            if (nv < g->map_lo) nv = g->map_lo;
            else if (nv > g->map_hi) nv = g->map_hi;
            map[flat] = (uint8_t)nv;
        }
        sum_out[i] = sum;
        cnt_out[i] = cnt;
    }
}

void pfo_voxel_indices(const pfo_grid* g, double x, double y, double z, double yaw,
                       int64_t M, const int32_t* bx, const int32_t* by, const int32_t* bz,
                       int32_t* flat_out) {
    SINCOS(yaw, s, c);
    double tx = x * 1 / g->resolution, ty = y * 1 / g->resolution, tz = z * 1 / g->resolution;
    for (int64_t j = 0; j < M; j++)
        flat_out[j] = (int32_t)beam_flat(g, c, s, tx, ty, tz, bx[j], by[j], bz[j]);
}

void pfo_predict(int64_t P, double* x, double* y, double* z, double* yaw,
                 double dlx, double dly, double dlz, double dyaw, const float* noise) {
    for (int64_t i = 0; i < P; i++) {
        SINCOS(yaw[i], s, c);
        x[i] += c * dlx - s * dly + noise[4 * i + 0];   /* S3P:911 */
        y[i] += s * dlx + c * dly + noise[4 * i + 1];   /* S3P:912 */
        z[i] += dlz + noise[4 * i + 2];                 /* S3P:913 */
        yaw[i] += dyaw + noise[4 * i + 3];              /* S3P:914 */
    }
}

double pfo_weights(int64_t P, double* w, const int32_t* sum, int32_t num_match, float* logw) {
    double smallest = 1000000000000.0;              /* S3P:920 */
    for (int64_t i = 0; i < P; i++)
        if (smallest > sum[i]) smallest = sum[i];   /* S3P:921-925 */
    double total = 0;
    for (int64_t i = 0; i < P; i++) {               /* S3P:927-930 */
        w[i] = w[i] * (sum[i] - 0.99 * smallest) * (sum[i] - 0.99 * smallest) / num_match;
        total += w[i];
    }
    for (int64_t i = 0; i < P; i++) {               /* S3P:935-937 */
        w[i] = w[i] / total;
        if (logw) logw[i] = (float)log(w[i]);
    }
    return total;
}

int64_t pfo_resample(int64_t P, double* w, const double* u,
                     int32_t* draw_anc, int32_t* slot_anc, int32_t* best) {
    double* cdf = (double*)malloc(sizeof(double) * (size_t)(P + 1));
    int32_t* taken = (int32_t*)calloc((size_t)P, sizeof(int32_t));
    double run = 0;
    cdf[0] = 0;
    for (int64_t i = 0; i < P; i++) { run += w[i]; cdf[i + 1] = run; }   /* S3P:763-767 */
    int64_t nohit = 0;
    for (int64_t i = 0; i < P; i++) {               /* S3P:772-780 */
        int64_t hit = -1;
        for (int64_t j = 1; j < P + 1; j++)
            if (u[i] > cdf[j - 1] && u[i] < cdf[j]) { hit = j - 1; break; }
        if (hit < 0) { hit = i; nohit++; }          /* D2 */
        draw_anc[i] = (int32_t)hit;
    }
    /* S3P:782-802. The initialiser {1} never equals its own index, so "slot a already holds
     * a" is simply "a was drawn before". */
    for (int64_t i = 0; i < P; i++) slot_anc[i] = (i == 0) ? 1 : 0;
    for (int64_t i = 0; i < P; i++) taken[draw_anc[i]] = 1;
    for (int64_t i = 0; i < P; i++) {
        int32_t a = draw_anc[i];
        if (a == slot_anc[a]) {
            for (int64_t j = 0; j < P; j++)
                if (taken[j] == 0) { slot_anc[j] = a; taken[j] = 1; break; }
        } else {
            slot_anc[a] = a;
        }
    }
    pfo_permute(P, slot_anc, w, sizeof(double));   /* S3P:805-812 (weight travels) */
    double high = 0;                                /* S3P:758, 815-820 */
    for (int64_t i = 0; i < P; i++)
        if (w[i] > high) { *best = (int32_t)i; high = w[i]; }
    free(cdf);
    free(taken);
    return nohit;
}

void pfo_permute(int64_t P, const int32_t* slot_anc, void* records, size_t width) {
    /* In-place ascending copy exactly as S3P:805-812 (sources are never overwritten). */
    char* r = (char*)records;
    for (int64_t i = 0; i < P; i++) {
        if (slot_anc[i] == i) continue;
        memcpy(r + (size_t)i * width, r + (size_t)slot_anc[i] * width, width);
    }
}

void pfo_rot_zyx(double m[9], double z, double y, double x) {   /* S3P:536-544 */
    SINCOS(z, sz, cz); SINCOS(y, sy, cy); SINCOS(x, sx, cx);
    m[0] = cy * cz;
    m[1] = cz * sx * sy - cx * sz;
    m[2] = sx * sz + cx * cz * sy;
    m[3] = cy * sz;
    m[4] = cx * cz + sx * sy * sz;
    m[5] = cx * sy * sz - cz * sx;
    m[6] = -sy;
    m[7] = cy * sx;
    m[8] = cx * cy;
}

static inline int64_t flat_int(int32_t gy, int32_t gz, double rx, double ry, double rz,
                               int64_t elements) {
    /* S3P:641-645 / 687-691: round per axis, int index_conv, flat-only bounds (>=). D3. */
    int bad = 0;
    int32_t xi = to_i32(round(rx), &bad), yi = to_i32(round(ry), &bad), zi = to_i32(round(rz), &bad);
    if (bad) return -1;
    int64_t flat = (int64_t)xi * gy * gz + (int64_t)yi * gz + zi;
    if (flat < 0 || flat >= elements) return -1;
    return flat;
}

static inline int cell_get(const pfo_update_cfg* c, const void* grid, int64_t f) {
    return c->is_signed ? (int)((const int8_t*)grid)[f] : (int)((const uint8_t*)grid)[f];
}
static inline void cell_set(const pfo_update_cfg* c, void* grid, int64_t f, int v) {
    if (c->is_signed) ((int8_t*)grid)[f] = (int8_t)v; else ((uint8_t*)grid)[f] = (uint8_t)v;
}

void pfo_update_map(const pfo_update_cfg* c, void* grid,
                    const double pos[3], const double zyx[3],
                    int64_t M, const double* points, int64_t R, const int32_t* ray_idx) {
    double m[9];
    double conv = 1 / c->resolution;                /* S3P:284 */
    int64_t elements = (int64_t)c->gx * c->gy * c->gz;
    pfo_rot_zyx(m, zyx[0], zyx[1], zyx[2]);
    double* w = (double*)malloc(sizeof(double) * 3 * (size_t)(M > 0 ? M : 1));
    for (int64_t i = 0; i < M; i++) {               /* S3P:546-558 */
        const double* p = points + 3 * i;
        w[3 * i + 0] = (m[0] * p[0] + m[1] * p[1] + m[2] * p[2]) + pos[0];
        w[3 * i + 1] = (m[3] * p[0] + m[4] * p[1] + m[5] * p[2]) + pos[1];
        w[3 * i + 2] = (m[6] * p[0] + m[7] * p[1] + m[8] * p[2]) + pos[2];
    }
    for (int64_t i = 0; i < M; i++) {               /* S3P:632-661 */
        double dx = w[3 * i] - pos[0], dy = w[3 * i + 1] - pos[1], dz = w[3 * i + 2] - pos[2];
        double norm = sqrt(dx * dx + dy * dy + dz * dz);
        if (norm > c->max_range) continue;
        int64_t f = flat_int(c->gy, c->gz, w[3 * i] * conv, w[3 * i + 1] * conv,
                             w[3 * i + 2] * conv, elements);
        if (f < 0) continue;
        int v = cell_get(c, grid, f) + c->hit_inc;
        if (v > c->hit_clamp) v = c->hit_clamp;
        cell_set(c, grid, f, v);
    }
    for (int64_t r = 0; r < R; r++) {               /* S3P:668-704 */
        const double* e = w + 3 * (int64_t)ray_idx[r];
        double dx = e[0] - pos[0], dy = e[1] - pos[1], dz = e[2] - pos[2];
        double norm = sqrt(dx * dx + dy * dy + dz * dz);
        double nc = norm * conv;
        double rr = round(nc);
        int bad = 0;
        int32_t n_it = to_i32(rr, &bad);            /* S3P:676; the >3 m cap is dead code */
        if (bad) continue;
        double sx = (e[0] - pos[0]) / nc, sy = (e[1] - pos[1]) / nc, sz = (e[2] - pos[2]) / nc;
        for (int32_t j = 1; j < n_it; j++) {        /* S3P:685-702 */
            int64_t f = flat_int(c->gy, c->gz, (pos[0] + sx * j) * conv, (pos[1] + sy * j) * conv,
                                 (pos[2] + sz * j) * conv, elements);
            if (f < 0) continue;
            int v = cell_get(c, grid, f) + c->miss_inc;
            if (v < c->miss_clamp) v = c->miss_clamp;
            cell_set(c, grid, f, v);
        }
    }
    free(w);
}

int64_t pfo_scan_vote(const uint8_t* grid, int32_t gx, int32_t gy, int32_t gz, double resolution,
                      int64_t N, const double* points, const double pos[3], const double zyx[3],
                      int32_t n_yaw, int32_t yaw_k0, int32_t yaw_kstep,
                      int32_t n_t, double length, int32_t* sums) {
    const double pi = 2 * acos(0.0);                /* S3P:24 */
    double conv = 1 / resolution;
    int64_t elements = (int64_t)gx * gy * gz;
    double* rot = (double*)malloc(sizeof(double) * 3 * (size_t)(N > 0 ? N : 1));
    int64_t best = -1, h = 0;
    int32_t best_sum = 0;                           /* S3P:326 */
    for (int32_t r = 0; r < n_yaw; r++) {           /* S3P:328 */
        double rz = zyx[0] + (yaw_k0 + yaw_kstep * r) * pi / 180;   /* S3P:329 */
        double m[9];
        pfo_rot_zyx(m, rz, zyx[1], zyx[2]);
        for (int64_t i = 0; i < N; i++) {
            const double* p = points + 3 * i;
            rot[3 * i + 0] = m[0] * p[0] + m[1] * p[1] + m[2] * p[2];
            rot[3 * i + 1] = m[3] * p[0] + m[4] * p[1] + m[5] * p[2];
            rot[3 * i + 2] = m[6] * p[0] + m[7] * p[1] + m[8] * p[2];
        }
        for (int32_t a = 0; a < n_t; a++) {         /* S3P:334-335 */
            double tx = pos[0] + a * resolution - (length - resolution) / 2;
            for (int32_t b = 0; b < n_t; b++) {
                double ty = pos[1] + b * resolution - (length - resolution) / 2;
                for (int32_t d = 0; d < n_t; d++, h++) {
                    double tz = pos[2] + d * resolution - (length - resolution) / 2;
                    int32_t sum = 0;
                    for (int64_t i = 0; i < N; i++) {   /* S3P:350-363 */
                        int64_t f = flat_int(gy, gz, (rot[3 * i] + tx) * conv,
                                             (rot[3 * i + 1] + ty) * conv,
                                             (rot[3 * i + 2] + tz) * conv, elements);
                        if (f < 0) continue;
                        sum += grid[f];
                    }
                    sums[h] = sum;
                    if (sum > best_sum) { best_sum = sum; best = h; }   /* S3P:365 */
                }
            }
        }
    }
    free(rot);
    return best;
}

/* ---- keyframe ring: S3P:305-312, 419-529 -------------------------------------------------- */
void pfo_ring_defaults(pfo_ring* r, int32_t gx, int32_t gy, int32_t gz, double resolution) {
    r->gx = gx; r->gy = gy; r->gz = gz; r->resolution = resolution;
    r->n_slots = 5; r->cap = 19200;
    r->pos_update[0] = gx * resolution / 2; r->pos_update[1] = gy * resolution / 2;   /* S3P:416 */
    r->pos_update[2] = gz * resolution / 2;
    r->euler_update[0] = r->euler_update[1] = r->euler_update[2] = 0;                /* S3P:415 */
    r->first_run = 1;
    r->move_threshold = 0.15; r->recentre_xy = 2; r->recentre_z = 1; r->life_new = 6;
}

void pfo_ring_paint(const pfo_ring* r, uint8_t* grid) {
    for (int32_t i = 0; i < r->n_slots; i++)                          /* S3P:305-312 */
        if (r->life[i] > 0)
            for (int64_t j = (int64_t)i * r->cap; j < (int64_t)i * r->cap + r->count[i]; j++)
                grid[r->idx[j]] = 1;
}

int pfo_ring_update(pfo_ring* r, uint8_t* grid, int64_t n, const double* cloud, double pos_scan[3],
                    const double zyx[3]) {
    int flags = 0;
    const double conv = 1 / r->resolution;                            /* S3P:284 */
    const int64_t elements = (int64_t)r->gx * r->gy * r->gz;
    const double cx = r->gx * r->resolution / 2, cy = r->gy * r->resolution / 2,
                 cz = r->gz * r->resolution / 2;                      /* map_size/2 */
    const double rot_y = zyx[1], rot_x = zyx[2];
    const double update_value =                                       /* S3P:423-425 */
        fabs(pos_scan[0] - r->pos_update[0]) + fabs(pos_scan[1] - r->pos_update[1])
        + fabs(pos_scan[2] - r->pos_update[2]) + 0.5 * fabs(zyx[0] - r->euler_update[0])
        + 0.5 * fabs(rot_y - r->euler_update[1]) + 0.5 * fabs(rot_x - r->euler_update[2]);
    if (update_value > r->move_threshold || r->first_run == 1) {      /* S3P:428 */
        flags |= 1;
        for (int32_t i = 0; i < r->n_slots; i++) {
            if (r->life[i] == 1 || r->life[i] <= 0) {                 /* S3P:432 */
                int32_t* slot = r->idx + (int64_t)i * r->cap;
                if (r->life[i] != 0)                                  /* S3P:433-437 */
                    for (int32_t j = 0; j < r->count[i]; j++) grid[slot[j]] = 0;
                double m[9];
                pfo_rot_zyx(m, zyx[0], rot_y, rot_x);                 /* S3P:439 */
                r->count[i] = 0;
                for (int64_t j = 0; j < n && j < r->cap; j++) {       /* S3P:445-459 */
                    const double* p = cloud + 3 * j;
                    const double wx = (m[0] * p[0] + m[1] * p[1] + m[2] * p[2]) + pos_scan[0];
                    const double wy = (m[3] * p[0] + m[4] * p[1] + m[5] * p[2]) + pos_scan[1];
                    const double wz = (m[6] * p[0] + m[7] * p[1] + m[8] * p[2]) + pos_scan[2];
                    const int64_t f = flat_int(r->gy, r->gz, wx * conv, wy * conv, wz * conv, elements);
                    if (f < 0) continue;
                    slot[j] = (int32_t)f;                             /* S3P:454: at j, not at count */
                    r->count[i] += 1;
                }
                r->life[i] = r->life_new;                             /* S3P:463 */
                break;
            }
        }
        for (int32_t i = 0; i < r->n_slots; i++) r->life[i] += -1;    /* S3P:470-472 */
        r->first_run = 0;
        r->pos_update[0] = pos_scan[0]; r->pos_update[1] = pos_scan[1]; r->pos_update[2] = pos_scan[2];
        r->euler_update[0] = zyx[0]; r->euler_update[1] = rot_y; r->euler_update[2] = rot_x;
    }
    if (fabs(pos_scan[0] - cx) > r->recentre_xy || fabs(pos_scan[1] - cy) > r->recentre_xy ||
        fabs(pos_scan[2] - cz) > r->recentre_z) {                     /* S3P:485-486 */
        flags |= 2;
        int bad = 0;
        const int64_t ox = to_i32(round((cx - pos_scan[0]) * conv), &bad);   /* S3P:489-491 */
        const int64_t oy = to_i32(round((cy - pos_scan[1]) * conv), &bad);
        const int64_t oz = to_i32(round((cz - pos_scan[2]) * conv), &bad);
        const int64_t corr = ox * r->gy * r->gz + oy * r->gz + oz;    /* S3P:494 */
        for (int32_t i = 0; i < r->n_slots; i++) {
            int32_t* slot = r->idx + (int64_t)i * r->cap;
            for (int32_t j = 0; j < r->count[i]; j++) grid[slot[j]] = 0;      /* S3P:498-500 */
            int32_t off = 0;
            for (int32_t j = 0; j < r->count[i]; j++) {               /* S3P:502-515 */
                const int64_t ni = (int64_t)slot[j] + corr;
                if (bad || ni < 0 || ni >= elements) off += 1;
                else slot[j - off] = (int32_t)ni;
            }
            r->count[i] += -off;                                      /* S3P:516 */
        }
        r->pos_update[0] += cx - pos_scan[0]; pos_scan[0] = cx;       /* S3P:519-524 */
        r->pos_update[1] += cy - pos_scan[1]; pos_scan[1] = cy;
        r->pos_update[2] += cz - pos_scan[2]; pos_scan[2] = cz;
    }
    return flags;
}

/* ---- depth front-end: S3P:562-612 -------------------------------------------------------- */
int64_t pfo_depth_to_cloud(const uint8_t* image, int32_t width, int32_t height, int32_t pool,
                           double focal, double* cloud_out) {
    const int32_t res_h = width / pool, res_v = height / pool;         /* S3P:116-117 */
    int64_t n = 0;
    for (int32_t i = 0; i < res_v; i++) {
        const double vertical = -1 * ((i + 1) - (res_v / 2)) / focal;  /* S3P:161 */
        for (int32_t j = 0; j < res_h; j++) {
            const double horizontal = ((j + 1) - (res_h / 2)) / focal; /* S3P:158 */
            int int_sum = 0, devide_sum = 0;                           /* S3P:580-594 */
            for (int32_t a = 0; a < pool; a++) {
                int64_t k = ((int64_t)(i * pool + a) * width + (int64_t)j * pool) * 2;
                for (int32_t b = 0; b < pool; b++) {
                    const int v = (image[k + 1] << 8) | image[k];
                    k += 2;
                    int_sum += v;
                    devide_sum += (v != 0) ? 1 : 0;
                }
            }
            const int depth_y = devide_sum != 0 ? int_sum / devide_sum : 0;
            const int depth_x = (int)(depth_y * horizontal);           /* S3P:567: double -> int */
            const int depth_z = (int)(depth_y * vertical);             /* S3P:568 */
            if (depth_y != 0) {                                        /* S3P:602-607 */
                cloud_out[3 * n + 0] = depth_x / 1000.0;
                cloud_out[3 * n + 1] = depth_y / 1000.0;
                cloud_out[3 * n + 2] = depth_z / 1000.0;
                n += 1;
            }
        }
    }
    return n;
}

/* ---- map read-out: S3P:1031-1051 --------------------------------------------------------- */
int64_t pfo_read_occupied(const uint8_t* map, int32_t gx, int32_t gy, int32_t gz, double resolution,
                          int32_t threshold, int32_t stride, float* xyz_out, int64_t cap) {
    const int64_t elements = (int64_t)gx * gy * gz;
    const double gyz = (double)gy * (double)gz, gzd = (double)gz;      /* S3P:104: (15/0.05)*(4/0.05) */
    int64_t n = 0;
    for (int64_t i = 0; i < elements; i += stride) {                   /* S3P:1031 */
        if (map[i] > threshold) {                                      /* S3P:1033 */
            const int x = (int)(i / gyz);                              /* S3P:104-106 */
            const int y = (int)((i - x * gyz) / gzd);
            const int z = (int)(i - x * gyz - y * gzd);
            if (n < cap) {
                xyz_out[3 * n + 0] = (float)(x * resolution);          /* S3P:1036-1038 */
                xyz_out[3 * n + 1] = (float)(-z * resolution);
                xyz_out[3 * n + 2] = (float)(y * resolution);
            }
            n += 1;
        }
    }
    return n < cap ? n : cap;
}
