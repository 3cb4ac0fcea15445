// Host-driver helpers of include/slam_pf.h (pfh_*): the caller-side lines of the reference,
// for drivers that want them ready-made (driver/step_loop.cpp, slam_amd/filter.py). No GPU.
// Built without FMA contraction, like the reference.
#include "../../include/slam_pf.h"

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
        const double c = cos(yaw[i]), s = sin(yaw[i]);
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

void pfh_permute(int64_t n, const int32_t* slot_anc, void* records, size_t width) {
    char* r = (char*)records;                              // S3P:805-812
    for (int64_t i = 0; i < n; ++i)
        if (slot_anc[i] != i) memcpy(r + (size_t)i * width, r + (size_t)slot_anc[i] * width, width);
}

}  // extern "C"
