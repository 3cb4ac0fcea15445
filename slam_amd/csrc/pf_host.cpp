// Host-driver helpers of include/slam_pf.h (pfh_*): the caller-side lines of the reference,
// for drivers that want them ready-made (driver/step_loop.cpp, slam_amd/filter.py). No GPU.
// Built without FMA contraction, like the reference.
#include "../../include/slam_pf.h"

#include "pf_trig.h"
#include <math.h>
#include <string.h>
#include <random>

extern "C" {

int64_t pfh_snapshot_beams(const int8_t* local, int32_t lgx, int32_t lgy, int32_t lgz, int32_t ox,
                           int32_t oy, int32_t oz, int32_t* bx, int32_t* by, int32_t* bz,
                           int8_t* bv, int64_t cap) {
    // S3P:868-877: every non-zero voxel, ascending flat index, as offsets from the local origin.
    const int64_t yz = (int64_t)lgy * lgz;
    int64_t n = 0;
    for (int32_t xi = 0; xi < lgx; ++xi)
        for (int32_t yi = 0; yi < lgy; ++yi) {
            const int8_t* col = local + xi * yz + (int64_t)yi * lgz;
            for (int32_t zi = 0; zi < lgz; ++zi) {
                if (col[zi] == 0) continue;
                if (n < cap) {
                    bx[n] = xi - ox; by[n] = yi - oy; bz[n] = zi - oz; bv[n] = col[zi];
                }
                ++n;
            }
        }
    return n <= cap ? n : -n;
}

void pfh_predict(int64_t n, double* x, double* y, double* z, double* yaw, double dlx, double dly,
                 double dlz, double dyaw, const float* noise) {
    for (int64_t i = 0; i < n; ++i) {
        const PfSinCos t(yaw[i]);
        const double c = t.c, s = t.s;
        x[i] += c * dlx - s * dly + noise[4 * i + 0];      // S3P:911
        y[i] += s * dlx + c * dly + noise[4 * i + 1];      // S3P:912
        z[i] += dlz + noise[4 * i + 2];                    // S3P:913
        yaw[i] += dyaw + noise[4 * i + 3];                 // S3P:914
    }
}

void pfh_draw_noise(uint32_t seed, int64_t n, float sigma_pos, float sigma_rot, float* noise) {
    // S3P:853-855 + the call pattern of S3P:911-914 (each distribution keeps its spare variate)
    std::mt19937 gen(seed);
    std::normal_distribution<float> pos_d(0, sigma_pos);
    std::normal_distribution<float> rot_d(0, sigma_rot);
    for (int64_t i = 0; i < n; ++i) {
        noise[4 * i + 0] = pos_d(gen);
        noise[4 * i + 1] = pos_d(gen);
        noise[4 * i + 2] = rot_d(gen);
        noise[4 * i + 3] = rot_d(gen);
    }
}

void pfh_draw_uniform(uint32_t seed, int64_t n, double* u) {
    std::mt19937 gen(seed);                                // S3P:769-773
    std::uniform_real_distribution<double> distr(0.0, 1.0);
    for (int64_t i = 0; i < n; ++i) u[i] = distr(gen);
}

int pfh_add_points_policy(const double pose[4], double last[4], double threshold) {
    // S3P:941-953
    const double moved = fabs(pose[0] - last[0]) + fabs(pose[1] - last[1]) +
                         fabs(pose[2] - last[2]) + fabs(pose[3] - last[3]);
    if (moved > threshold) {
        last[0] = pose[0]; last[1] = pose[1]; last[2] = pose[2]; last[3] = pose[3];
        return 1;
    }
    return 0;
}

int pfh_vote_hypothesis(int64_t index, const double pos[3], const double zyx[3], double resolution,
                        int32_t n_yaw, int32_t yaw_k0, int32_t yaw_kstep, int32_t n_t, double length,
                        double out[4]) {
    out[0] = pos[0]; out[1] = pos[1]; out[2] = pos[2]; out[3] = zyx[0];      // S3P:326-327
    const int64_t per = (int64_t)n_t * n_t * n_t;
    if (index < 0 || n_yaw < 1 || n_t < 1 || index >= per * n_yaw) return -1;
    const double pi = 2 * acos(0.0);                                         // S3P:24
    const int r = (int)(index / per);
    const int64_t k = index - (int64_t)r * per;
    const int a = (int)(k / ((int64_t)n_t * n_t)), b = (int)((k / n_t) % n_t), d = (int)(k % n_t);
    out[0] = pos[0] + a * resolution - (length - resolution) / 2;           // S3P:335
    out[1] = pos[1] + b * resolution - (length - resolution) / 2;           // S3P:341
    out[2] = pos[2] + d * resolution - (length - resolution) / 2;           // S3P:348
    out[3] = zyx[0] + (yaw_k0 + yaw_kstep * r) * pi / 180;                  // S3P:329
    return 0;
}

void pfh_draw_indices(uint32_t seed, int64_t n, int32_t hi, int32_t* idx) {
    // S3P:315-319 (the 500 scan points of the vote) and S3P:666-669 (the 2400 ray targets): one
    // mt19937, uniform_int_distribution<>(0, hi) with hi INCLUSIVE as the reference writes it.
    std::mt19937 gen(seed);
    std::uniform_int_distribution<> distr(0, hi);
    for (int64_t i = 0; i < n; ++i) idx[i] = distr(gen);
}

void pfh_vote_epilogue(double pos_scan[3], double* yaw_scan, const double loc[4], double gyr_yaw,
                       double* gyr_yaw_old, double pos_diff_out[3], double* yaw_diff_scan_out,
                       double* yaw_diff_out) {
    const double pi = 2 * acos(0.0);                                          // S3P:24
    const double e0 = *yaw_scan;
    // g++ -O2 turns cos(-a) into cos(a) and takes line 379's pair from one sincos(a); line 380's sine
    // stays a call of sin(-a) (the store between them keeps it apart) and its cosine is reused. glibc's
    // sin() and sincos() differ in the last place for about one argument in a thousand (pf_trig.h).
    const PfSinCos t(e0);
    pos_diff_out[0] = t.c * (loc[0] - pos_scan[0]) - (-t.s) * (loc[1] - pos_scan[1]);   // S3P:379
    pos_diff_out[1] = sin(-e0) * (loc[0] - pos_scan[0]) + t.c * (loc[1] - pos_scan[1]);   // S3P:380
    pos_diff_out[2] = loc[2] - pos_scan[2];                                   // S3P:381
    const double diff_scan = loc[3] - e0;                                     // S3P:382
    *yaw_scan = loc[3];                                                       // S3P:383-386
    pos_scan[0] = loc[0]; pos_scan[1] = loc[1]; pos_scan[2] = loc[2];
    double diff_gyr;
    if (fabs(*gyr_yaw_old - gyr_yaw) < pi) diff_gyr = gyr_yaw - *gyr_yaw_old;                  // S3P:393-402
    else if (gyr_yaw < 0) diff_gyr = 2 * pi + gyr_yaw - *gyr_yaw_old;
    else diff_gyr = -2 * pi + gyr_yaw - *gyr_yaw_old;
    *gyr_yaw_old = gyr_yaw;                                                   // S3P:403
    *yaw_diff_scan_out = diff_scan;
    *yaw_diff_out = (diff_gyr + diff_scan) / 2;                               // S3P:404
}

void pfh_update_realtime_pos(double pos[3], double* yaw, const double pos_diff_scan[3], double yaw_diff,
                             double feedback[4]) {
    const double pi = 2 * acos(0.0);                                          // S3P:24
    const double e0 = *yaw;
    const PfSinCos t(e0);
    pos[0] += t.c * pos_diff_scan[0] - t.s * pos_diff_scan[1] + feedback[0];   // S3P:713
    pos[1] += t.s * pos_diff_scan[0] + t.c * pos_diff_scan[1] + feedback[1];   // S3P:714
    pos[2] += pos_diff_scan[2] + feedback[2];                                 // S3P:715
    *yaw += yaw_diff + feedback[3];                                           // S3P:716
    if (*yaw > pi) *yaw += -2 * pi;                                           // S3P:717-722
    else if (*yaw < -pi) *yaw += 2 * pi;
    feedback[0] = feedback[1] = feedback[2] = feedback[3] = 0;                // S3P:723-726
}

void pfh_pose_message(const double pos[3], double yaw, double position_out[3], double orientation_wxyz_out[4]) {
    const double pi = 2 * acos(0.0);
    const PfSinCos t((-yaw - pi / 2) / 2);
    orientation_wxyz_out[0] = t.c;                                            // S3P:212-215
    orientation_wxyz_out[1] = 0;
    orientation_wxyz_out[2] = t.s;
    orientation_wxyz_out[3] = 0;
    position_out[0] = pos[0];                                                 // S3P:216-218
    position_out[1] = -pos[2];
    position_out[2] = pos[1];
}

int64_t pfh_migration_plan(const int32_t* slot_anc, int64_t n_global, int64_t n_local, int32_t rank,
                           int32_t* send_src, int32_t* send_peer, int64_t* send_seq, int64_t* n_send,
                           int32_t* recv_dst, int32_t* recv_peer, int64_t* recv_seq, int64_t* n_recv,
                           int64_t cap) {
    // S3P:805-812 across shards: slot g takes its ancestor's map; when the ancestor lives on another
    // rank the slab crosses the link. Every rank walks the same slot_anc[] in the same order, so the
    // sequence numbers agree everywhere and sends meet their receives in order per pair of ranks.
    int64_t seq = 0, ns = 0, nr = 0;
    for (int64_t g = 0; g < n_global; ++g) {
        const int64_t a = slot_anc[g];
        if (a == g) continue;
        const int64_t rg = g / n_local, ra = a / n_local;
        if (rg == ra) continue;
        if (ra == rank) {
            if (ns < cap) { send_src[ns] = (int32_t)(a - ra * n_local); send_peer[ns] = (int32_t)rg; send_seq[ns] = seq; }
            ++ns;
        }
        if (rg == rank) {
            if (nr < cap) { recv_dst[nr] = (int32_t)(g - rg * n_local); recv_peer[nr] = (int32_t)ra; recv_seq[nr] = seq; }
            ++nr;
        }
        ++seq;
    }
    *n_send = ns; *n_recv = nr;
    return seq;
}

void pfh_permute(int64_t n, const int32_t* slot_anc, void* records, size_t width) {
    char* r = (char*)records;                              // S3P:805-812
    for (int64_t i = 0; i < n; ++i)
        if (slot_anc[i] != i) memcpy(r + (size_t)i * width, r + (size_t)slot_anc[i] * width, width);
}

}  // extern "C"

// K1's column groups dealt to the chunks of its launch (declaration in slam_pf.h).
extern "C" void pfh_deal_groups(const int32_t group_ends[8], int32_t n_chunks, const int32_t trips[8], int32_t* perm,
                                int32_t* table) {
    const int C = n_chunks < 1 ? 1 : n_chunks;
    int32_t at = 0;
    for (int y = 0; y < C; ++y) {
        table[9 * y + 8] = at;
        long long slot0 = 0;
        int g0 = 0;
        for (int ph = 0; ph < 8; ++ph) {
            const int g1 = group_ends[ph], T = trips[ph] < 1 ? 1 : trips[ph];
            const long long nslots = (g1 - g0 + T - 1) / T;
            for (long long sl = ((y - slot0) % C + C) % C; sl < nslots; sl += C)
                for (int g = g0 + (int)sl * T; g < g0 + ((int)sl + 1) * T && g < g1; ++g) perm[at++] = g;
            table[9 * y + ph] = at;
            slot0 += nslots; g0 = g1;
        }
    }
}
