// Host-sanitizer exerciser (development aid): random shared-grid steps through the C ABI, several handle sizes,
// no result checks (the parity suites do those): what matters here is what ASan/UBSan say about the host code.
#include "../include/slam_pf.h"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define OK(x) do { int rc_ = (x); if (rc_ != PF_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, pf_last_error(h)); return 1; } } while (0)
int main() {
    std::mt19937 gen(7);
    const int sizes[] = {1, 63, 257, 2049, 5000, 12000, 70000};
    for (int P : sizes) {
        for (int rep = 0; rep < 2; ++rep) {
            pf_config c; pf_default_config(&c);
            const int gx = 40 + gen() % 90, gy = 40 + gen() % 90, gz = (rep ? 32 : 30) + gen() % 40;
            c.n_particles = c.n_particles_global = P; c.map_mode = PF_MAP_SHARED;
            c.grid_x = gx; c.grid_y = gy; c.grid_z = gz; c.local_x = c.local_y = c.local_z = 0;
            c.max_beams = 6000; c.max_points = 3000;
            c.order_mode = rep ? PF_ORDER_FAST : PF_ORDER_AUTO;
            pf_handle* h = nullptr;
            if (pf_create(&c, &h) != PF_OK) { std::fprintf(stderr, "create: %s\n", pf_last_error(nullptr)); return 1; }
            std::vector<uint8_t> grid((size_t)gx * gy * gz);
            for (auto& v : grid) v = gen() % 4 ? 0 : gen() % 101;
            OK(pf_write_map(h, 0, grid.data()));
            std::vector<double> x(P), y(P), z(P), yaw(P), w(P), u(P);
            std::normal_distribution<double> n01(0, 1); std::uniform_real_distribution<double> u01(0, 1);
            for (int i = 0; i < P; ++i) { x[i] = gx * 0.025 + 0.05 * n01(gen); y[i] = gy * 0.025 + 0.05 * n01(gen); z[i] = gz * 0.025 + 0.02 * n01(gen); yaw[i] = 0.3 * n01(gen); w[i] = u01(gen); }
            OK(pf_predict(h, x.data(), y.data(), z.data(), yaw.data()));
            OK(pf_set_weights(h, w.data()));
            for (int step = 0; step < 4; ++step) {
                const int M = step == 3 ? 0 : 1 + gen() % 6000;
                std::vector<int32_t> bx(M), by(M), bz(M); std::vector<int8_t> bv(M);
                for (int j = 0; j < M; ++j) { bx[j] = (int)(gen() % 61) - 30; by[j] = (int)(gen() % 61) - 30; bz[j] = (int)(gen() % 25) - 12; bv[j] = (int8_t)((int)(gen() % 120) - 40); }
                if (step == 1) for (int j = 0; j < M; ++j) { bx[j] = j / 40 % 50 - 25; by[j] = 3; bz[j] = j % 40 - 20; }   // columns and runs
                OK(pf_set_beams(h, bx.data(), by.data(), bz.data(), bv.data(), M));
                OK(pf_measure(h, 0));
                std::vector<int32_t> s(P), cnt(P);
                OK(pf_get_scores(h, s.data(), cnt.data()));
                const int N = gen() % 3000;
                std::vector<double> pts(3 * (size_t)N); std::vector<int32_t> rays(N);
                for (int j = 0; j < N; ++j) { pts[3 * j] = n01(gen); pts[3 * j + 1] = n01(gen); pts[3 * j + 2] = 0.4 * n01(gen); rays[j] = N ? gen() % N : 0; }
                OK(pf_set_scan(h, pts.data(), N, rays.data(), N));
                pf_update_params up{10, 100, -3, 0, 3.0};
                const double pos[3] = {gx * 0.025, gy * 0.025, gz * 0.025}, zyx[3] = {0.2 * step, 0.01, -0.02};
                OK(pf_update_map(h, PF_GRID_SHARED, &up, pos, zyx));
                for (int i = 0; i < P; ++i) { x[i] += 0.01 * n01(gen); yaw[i] += 0.01 * n01(gen); u[i] = u01(gen); }
                OK(pf_predict(h, x.data(), y.data(), z.data(), yaw.data()));
                OK(pf_weights(h, nullptr));
                std::vector<int32_t> anc(P);
                int32_t best; int64_t nohit, amb;
                if (step & 1) { OK(pf_resample(h, u.data(), nullptr, anc.data(), &best, &nohit, &amb)); }
                else { OK(pf_resample_async(h, u.data(), nullptr, anc.data())); OK(pf_resample_result(h, &best, &nohit, &amb)); }
                OK(pf_get_poses(h, x.data(), y.data(), z.data(), yaw.data()));
                std::vector<float> lw(P);
                OK(pf_get_weights(h, w.data(), lw.data()));
                for (int i = 0; i < P; ++i) if (!(w[i] >= 0)) w[i] = 1.0 / P;
                if (N > 0) {
                    std::vector<int32_t> sums(625); int64_t vb;
                    OK(pf_scan_vote(h, pts.data(), N < 500 ? N : 500, pos, zyx, 5, -6, 3, 5, 0.25, sums.data(), &vb));
                }
                std::vector<float> occ(3 * 5000); int64_t n_out, total;
                OK(pf_read_occupied(h, 0, 80, 4, occ.data(), 5000, &n_out, &total));
            }
            std::vector<double> cdf(P + 1);
            OK(pf_sequential_cdf(h, w.data(), P, P > 2048, cdf.data()));
            OK(pf_read_map(h, 0, grid.data()));
            pf_destroy(h);
        }
        std::printf("P=%d done\n", P); std::fflush(stdout);
    }
    return 0;
}
