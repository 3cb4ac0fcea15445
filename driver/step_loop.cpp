// Plain C++ host driver: the stand-in for the reference's main() + Slam::callback() loop
// (S3P:1080-1118, S3P:168-205) without ROS. It keeps exactly what the reference keeps on the
// host — the local-map snapshot under a mutex, the motion model with its std::mt19937 noise,
// the add_points policy, the correction feedback — and reaches the GPU through the C ABI of
// include/slam_pf.h in the reference's per-step order. Synthetic room, synthetic depth scans.
//
//   ./step_loop [steps] [particles]        (needs an MI355X; the library has no CPU path)
#include "../include/slam_pf.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <random>
#include <vector>

#define CK(call)                                                                        \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ != PF_OK) {                                                             \
            std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, pf_last_error(h));       \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

namespace {
const double kRes = 0.05;
std::mutex mtx;                                  // S3P:27

// a box room (walls at +-2.5 m around the start pose, floor/ceiling) as a depth "scan": points
// in the sensor frame for a sensor at world pose (x, y, z, yaw)
std::vector<double> synth_scan(double x, double y, double z, double yaw, int n, std::mt19937& g) {
    std::uniform_real_distribution<double> az(-0.6, 0.6), el(-0.45, 0.45);
    std::vector<double> pts;
    pts.reserve(3 * (size_t)n);
    const double x0 = 5.0, x1 = 10.0, y0 = 5.0, y1 = 10.0, z0 = 0.3, z1 = 3.2;
    for (int i = 0; i < n; ++i) {
        const double a = az(g), e = el(g);
        // sensor looks along +y (the reference's depth axis, S3P:115)
        const double dx = std::sin(a) * std::cos(e), dy = std::cos(a) * std::cos(e), dz = std::sin(e);
        const double wx = std::cos(yaw) * dx - std::sin(yaw) * dy, wy = std::sin(yaw) * dx + std::cos(yaw) * dy;
        double t = 1e9;
        if (wx > 0) t = std::fmin(t, (x1 - x) / wx); else if (wx < 0) t = std::fmin(t, (x0 - x) / wx);
        if (wy > 0) t = std::fmin(t, (y1 - y) / wy); else if (wy < 0) t = std::fmin(t, (y0 - y) / wy);
        if (dz > 0) t = std::fmin(t, (z1 - z) / dz); else if (dz < 0) t = std::fmin(t, (z0 - z) / dz);
        if (t > 8.0) continue;
        pts.push_back(dx * t); pts.push_back(dy * t); pts.push_back(dz * t);
    }
    return pts;
}
}  // namespace

int main(int argc, char** argv) {
    const int steps = argc > 1 ? std::atoi(argv[1]) : 20;
    const int P = argc > 2 ? std::atoi(argv[2]) : 50;
    pf_config cfg;
    pf_default_config(&cfg);                     // S3P:23-69
    cfg.n_particles = cfg.n_particles_global = P;
    pf_handle* h = nullptr;
    if (pf_create(&cfg, &h) != PF_OK) {
        std::fprintf(stderr, "pf_create: %s\n", pf_last_error(nullptr));
        return 1;
    }
    const int L = cfg.local_x * cfg.local_y * cfg.local_z;
    std::vector<int8_t> local((size_t)L);
    std::vector<int32_t> bx((size_t)L), by((size_t)L), bz((size_t)L), slots((size_t)P);
    std::vector<int8_t> bv((size_t)L);
    std::vector<double> x((size_t)P, 7.5), y((size_t)P, 7.5), z((size_t)P, 1.5), yaw((size_t)P, 0.0), u((size_t)P);
    std::vector<float> noise(4 * (size_t)P);
    // host state of the reference (S3P:56-61, 823-825)
    double local_pos[3] = {5, 5, 1.5}, local_yaw = 0;
    double pos[4] = {7.5, 7.5, 1.5, 0}, last_update[4] = {7.5, 7.5, 1.5, 0}, diff[4] = {0, 0, 0, 0};
    int add_points = 1, best = 0;
    double truth[4] = {7.5, 7.5, 1.5, 0};
    std::mt19937 scan_rng(1), ray_rng(4);
    std::random_device rd;
    const pf_update_params upd = {10, 50, -3, -50, 3.0};   // S3P:637, 650-655, 695-700
    const int frames_per_step = 8;

    for (int t = 0; t < steps; ++t) {
        // ---- camera callback side (S3P:168-205), `frames_per_step` depth frames ----
        for (int f = 0; f < frames_per_step; ++f) {
            const double d[4] = {0.012 * std::cos(truth[3]), 0.012 * std::sin(truth[3]), 0.0, 0.004};
            for (int k = 0; k < 4; ++k) truth[k] += d[k];
            std::vector<double> pts = synth_scan(truth[0], truth[1], truth[2], truth[3], 4000, scan_rng);
            const int n = (int)pts.size() / 3;
            std::vector<int32_t> ray_idx(2400);
            std::uniform_int_distribution<> pick(0, n > 0 ? n - 1 : 0);
            for (auto& r : ray_idx) r = pick(ray_rng);     // S3P:666-669 (host-drawn)
            std::lock_guard<std::mutex> lk(mtx);           // S3P:622
            // odometry stand-in: the true increment in the local frame (S3P:623-626)
            local_pos[0] += std::cos(-pos[3]) * d[0] - std::sin(-pos[3]) * d[1];
            local_pos[1] += std::sin(-pos[3]) * d[0] + std::cos(-pos[3]) * d[1];
            local_pos[2] += d[2]; local_yaw += d[3];
            const double zyx[3] = {local_yaw, 0.0, 0.0};
            CK(pf_set_scan(h, pts.data(), n, ray_idx.data(), n > 0 ? (int64_t)ray_idx.size() : 0));
            CK(pf_update_map(h, PF_GRID_LOCAL, &upd, local_pos, zyx));   // S3P:628-704
            for (int k = 0; k < 4; ++k) pos[k] += d[k] + diff[k];        // S3P:712-716
            for (double& v : diff) v = 0;
        }
        // ---- particle_map() (S3P:827-962) ----
        double p_local[4], p_pos[4];
        int64_t nb;
        {
            std::lock_guard<std::mutex> lk(mtx);           // S3P:833-846
            CK(pf_read_local_map(h, local.data()));
            for (int k = 0; k < 3; ++k) p_local[k] = local_pos[k];
            p_local[3] = local_yaw;
            for (int k = 0; k < 4; ++k) p_pos[k] = pos[k];
            CK(pf_reset_local_map(h));
            local_pos[0] = 5; local_pos[1] = 5; local_pos[2] = 1.5; local_yaw = 0;
        }
        nb = pfh_snapshot_beams(local.data(), cfg.local_x, cfg.local_y, cfg.local_z, 100, 100, 30,
                                bx.data(), by.data(), bz.data(), bv.data(), (int64_t)L);   // S3P:868-877
        if (nb > cfg.max_beams) nb = cfg.max_beams;
        CK(pf_set_beams(h, bx.data(), by.data(), bz.data(), bv.data(), nb));
        CK(pf_measure(h, add_points));                                             // S3P:859-908
        pfh_draw_noise(rd(), P, 0.03f, 0.02f, noise.data());                       // S3P:852-855
        pfh_predict(P, x.data(), y.data(), z.data(), yaw.data(), p_local[0] - 5, p_local[1] - 5,
                    p_local[2] - 1.5, p_local[3], noise.data());                   // S3P:911-914
        CK(pf_predict(h, x.data(), y.data(), z.data(), yaw.data()));
        CK(pf_weights(h, nullptr));                                                // S3P:920-937
        add_points = pfh_add_points_policy(p_pos, last_update, 0.4);               // S3P:941-953
        pfh_draw_uniform(rd(), P, u.data());                                       // S3P:768-773
        int64_t nohit = 0, amb = 0;
        CK(pf_resample(h, u.data(), nullptr, slots.data(), &best, &nohit, &amb));  // S3P:954
        for (auto* v : {&x, &y, &z, &yaw}) pfh_permute(P, slots.data(), v->data(), sizeof(double));
        diff[0] = x[(size_t)best] - p_pos[0]; diff[1] = y[(size_t)best] - p_pos[1];  // S3P:956-959
        diff[2] = z[(size_t)best] - p_pos[2]; diff[3] = yaw[(size_t)best] - p_pos[3];
        std::printf("step %3d beams %6lld best %4d pose %.3f %.3f %.3f %.3f  truth %.3f %.3f %.3f %.3f  add %d\n",
                    t, (long long)nb, best, x[(size_t)best], y[(size_t)best], z[(size_t)best], yaw[(size_t)best],
                    truth[0], truth[1], truth[2], truth[3], add_points);
    }
    pf_destroy(h);
    return 0;
}
