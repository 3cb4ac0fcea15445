// Plain C++ host driver: the stand-in for the reference's main() + Slam::callback() loop
// (S3P:1080-1118, S3P:168-205) without ROS. It keeps exactly what the reference keeps on the
// host — the member state of class Slam, the local-map snapshot under a mutex, the motion model
// with its std::mt19937 noise, the add_points policy, the correction feedback — and reaches the
// GPU through the C ABI of include/slam_pf.h in the reference's order:
//
//   per depth frame (callback(), S3P:168-205)
//     pf_depth_to_cloud                  resize_depth_to_array + img_to_cloud   S3P:574-611
//     pf_ring_paint, pf_scan_vote        scan_localization                      S3P:300-376
//     pfh_vote_hypothesis / _epilogue                                           S3P:366-404
//     pf_ring_update                     scan_localization_map_update           S3P:419-529
//     pf_set_scan, pf_update_map         build_local_map                        S3P:621-710
//     pfh_pose_message                   pub_pose_func                          S3P:207-222
//   per filter step (particle_map(), S3P:827-962)
//     pf_read_local_map ... pf_resample
//   at the end (publish_cloud, S3P:1031-1051)
//     pf_read_occupied
//
// Synthetic box room with obstacles, synthetic 640x480 uint16 depth frames of it. Two handles: the filter
// (per-particle maps, the reference's sizes) and the localisation grid (200x200x80, shared).
//
//   ./step_loop [steps] [particles] [truth]     (needs an MI355X; the library has no CPU path)
//   "truth": feed the true increment as odometry instead of the scan vote (the first form of
//   this driver; keeps the filter's input independent of the front end)
//   ./step_loop replay <file>                   recorded inputs instead of the synthetic room and
//   std::random_device: per frame the pooled depth image, the IMU angles and the seeds of the two
//   index draws, per step the seed of the motion noise (the thresholds use seed + 1); prints every
//   frame's global pose and every step's result at 17 digits. tests/test_gpu_driver.py writes such a
//   file from tests/golden/frames.npz, the reference's own 48-frame run, and compares.
//   ./step_loop replay-device <file>            the same with cloudpoints (S3P:126) kept in a device buffer
//   between the stages (pf_depth_to_cloud_device, pf_scan_vote_device, pf_ring_update_device,
//   pf_set_scan_device; both handles on one stream): same printed numbers.
//   Every mode ends with the wall time per callback() on stderr.
#include "../include/slam_pf.h"

#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <random>
#include <vector>

#define CK(hh, call)                                                                    \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ != PF_OK) {                                                             \
            std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, pf_last_error(hh));      \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

namespace {
std::mutex mtx;                                  // S3P:27
const int kW = 640, kH = 480, kPool = 4;         // S3P:113-117
const double kFocal = 385.0 / 4.0;               // S3P:124

// One depth frame of a box room with four box obstacles (the scene of oracle/make_golden_frontend.py, on
// which the reference's own run tracks the truth to about 0.1 m) seen from (x, y, z, yaw) in the room's
// frame: the camera looks along +y of its own frame (S3P:115), pixel (i, j) of the pooled image has the
// direction (horizontal[j], 1, vertical[i]) of the reference's pinhole tables (S3P:155-162); each pooled
// depth is written to its 4x4 block of uint16 millimetres, little-endian (S3P:587).
void synth_depth(double x, double y, double z, double yaw, std::vector<uint8_t>& img) {
    static const double room[6] = {2.23, 7.91, 2.62, 8.37, 0.71, 3.43};            // x0 x1 y0 y1 z0 z1
    static const double boxes[4][6] = {{6.1, 6.9, 6.4, 7.5, 0.71, 2.2}, {3.0, 3.8, 7.2, 8.37, 0.71, 1.6},
                                       {2.23, 2.9, 4.1, 5.3, 1.2, 2.6}, {5.2, 5.6, 7.6, 7.9, 0.71, 3.43}};
    const int rh = kW / kPool, rv = kH / kPool;
    const double o[3] = {x, y, z};
    img.assign((size_t)kW * kH * 2, 0);
    for (int i = 0; i < rv; ++i)
        for (int j = 0; j < rh; ++j) {
            const double hx = ((j + 1) - rh / 2) / kFocal, vz = -1 * ((i + 1) - rv / 2) / kFocal;
            const double d[3] = {std::cos(yaw) * hx - std::sin(yaw), std::sin(yaw) * hx + std::cos(yaw), vz};
            double t = 1e9;
            for (int ax = 0; ax < 3; ++ax) {                     // the room's walls, from inside
                if (d[ax] > 0) t = std::fmin(t, (room[2 * ax + 1] - o[ax]) / d[ax]);
                else if (d[ax] < 0) t = std::fmin(t, (room[2 * ax] - o[ax]) / d[ax]);
            }
            for (const auto& bx : boxes) {                       // the obstacles, from outside (slab test)
                double tn = -1e9, tf = 1e9;
                for (int ax = 0; ax < 3; ++ax) {
                    if (d[ax] == 0) { if (o[ax] < bx[2 * ax] || o[ax] > bx[2 * ax + 1]) { tn = 1e9; } continue; }
                    const double t1 = (bx[2 * ax] - o[ax]) / d[ax], t2 = (bx[2 * ax + 1] - o[ax]) / d[ax];
                    tn = std::fmax(tn, std::fmin(t1, t2)); tf = std::fmin(tf, std::fmax(t1, t2));
                }
                if (tn <= tf && tn > 0) t = std::fmin(t, tn);
            }
            const long mm = std::lround(t * 1000.0);
            if (mm < 150 || mm > 6000) continue;                 // out of the sensor's range: no depth
            for (int a = 0; a < kPool; ++a)
                for (int b = 0; b < kPool; ++b) {
                    const size_t k = ((size_t)(i * kPool + a) * kW + (size_t)j * kPool + b) * 2;
                    img[k] = (uint8_t)(mm & 0xff); img[k + 1] = (uint8_t)(mm >> 8);
                }
        }
}
// The sensor's true pose at frame f: a circle of 0.9 m radius through (5, 5) inside the room, one turn in
// about 520 frames, the heading turning half as fast, 1.1 cm per frame.
void truth_at(int f, double out[4]) {
    const double ph = 0.012 * f;
    out[0] = 5.0 + 0.9 * std::sin(ph); out[1] = 5.0 + 0.9 * (1.0 - std::cos(ph)); out[2] = 2.0; out[3] = 0.5 * ph;
}
struct Replay {
    int32_t frames = 0, per_step = 0, P = 0;
    std::vector<uint16_t> pooled;                // [frames][120][160]
    std::vector<double> gyr;                     // [frames][3]
    std::vector<uint32_t> seed_vote, seed_rays;  // [frames]
    std::vector<uint32_t> step_seed;             // [frames / per_step]
};
bool load_replay(const char* path, Replay& r) {
    FILE* fp = std::fopen(path, "rb");
    if (!fp) return false;
    bool ok = std::fread(&r.frames, 4, 1, fp) == 1 && std::fread(&r.per_step, 4, 1, fp) == 1 && std::fread(&r.P, 4, 1, fp) == 1 &&
              r.frames > 0 && r.per_step > 0 && r.P > 0;
    const size_t npx = (size_t)(kW / kPool) * (kH / kPool);
    if (ok) {
        r.pooled.resize((size_t)r.frames * npx); r.gyr.resize(3 * (size_t)r.frames);
        r.seed_vote.resize((size_t)r.frames); r.seed_rays.resize((size_t)r.frames);
        r.step_seed.resize((size_t)(r.frames / r.per_step));
        ok = std::fread(r.pooled.data(), 2, r.pooled.size(), fp) == r.pooled.size() &&
             std::fread(r.gyr.data(), 8, r.gyr.size(), fp) == r.gyr.size() &&
             std::fread(r.seed_vote.data(), 4, r.seed_vote.size(), fp) == r.seed_vote.size() &&
             std::fread(r.seed_rays.data(), 4, r.seed_rays.size(), fp) == r.seed_rays.size() &&
             std::fread(r.step_seed.data(), 4, r.step_seed.size(), fp) == r.step_seed.size();
    }
    std::fclose(fp);
    return ok;
}
// the 640 x 480 little-endian uint16 image whose 4 x 4 blocks hold the pooled depths
void expand_pooled(const uint16_t* pooled, std::vector<uint8_t>& img) {
    const int rh = kW / kPool, rv = kH / kPool;
    img.assign((size_t)kW * kH * 2, 0);
    for (int i = 0; i < rv; ++i)
        for (int j = 0; j < rh; ++j) {
            const unsigned mm = pooled[(size_t)i * rh + j];
            for (int a = 0; a < kPool; ++a)
                for (int b = 0; b < kPool; ++b) {
                    const size_t k = ((size_t)(i * kPool + a) * kW + (size_t)j * kPool + b) * 2;
                    img[k] = (uint8_t)(mm & 0xff); img[k + 1] = (uint8_t)(mm >> 8);
                }
        }
}
}  // namespace

int main(int argc, char** argv) {
    Replay rep;
    const bool device_cloud = argc > 2 && std::strcmp(argv[1], "replay-device") == 0;
    const bool replay = argc > 2 && (device_cloud || std::strcmp(argv[1], "replay") == 0);
    if (replay && !load_replay(argv[2], rep)) {
        std::fprintf(stderr, "cannot read the replay file %s\n", argv[2]);
        return 1;
    }
    const int steps = replay ? rep.frames / rep.per_step : argc > 1 ? std::atoi(argv[1]) : 20;
    const int P = replay ? rep.P : argc > 2 ? std::atoi(argv[2]) : 50;
    const bool truth_odometry = !replay && argc > 3 && std::strcmp(argv[3], "truth") == 0;

    // ---- the filter: S3P:23-69 ----
    pf_config cfg;
    pf_default_config(&cfg);
    cfg.n_particles = cfg.n_particles_global = P;
    cfg.max_beams = 1 << 19;                     // the local map's non-zero cells: 1.4e5 after 8 dense frames
    pf_handle* h = nullptr;
    if (pf_create(&cfg, &h) != PF_OK) {
        std::fprintf(stderr, "pf_create: %s\n", pf_last_error(nullptr));
        return 1;
    }
    // ---- the localisation grid and its keyframe ring: S3P:64, 286-289 ----
    pf_config lcfg;
    pf_default_config(&lcfg);
    lcfg.n_particles = lcfg.n_particles_global = 1;
    lcfg.map_mode = PF_MAP_SHARED;
    lcfg.grid_x = 200; lcfg.grid_y = 200; lcfg.grid_z = 80;
    lcfg.local_x = lcfg.local_y = lcfg.local_z = 0;
    lcfg.max_beams = 64;
    pf_handle* loc = nullptr;
    if (pf_create(&lcfg, &loc) != PF_OK) {
        std::fprintf(stderr, "pf_create (localisation grid): %s\n", pf_last_error(nullptr));
        return 1;
    }
    CK(loc, pf_fill_maps(loc, 0));
    pf_ring_params rp;
    pf_ring_default_params(&rp);
    CK(loc, pf_ring_init(loc, &rp));

    const int L = cfg.local_x * cfg.local_y * cfg.local_z;
    const int npix = (kW / kPool) * (kH / kPool);
    double* cloud_dev = nullptr;                 // cloudpoints (S3P:126) in HBM
    if (device_cloud) {
        if (hipSetDevice(cfg.device) != hipSuccess || hipMalloc((void**)&cloud_dev, sizeof(double) * 3 * ((size_t)npix + 1)) != hipSuccess) {
            std::fprintf(stderr, "hipMalloc of the cloud failed\n");
            return 1;
        }
        CK(h, pf_set_stream(h, pf_stream(loc)));   // one stream orders the cloud's writer and its readers
    }
    double callback_s = 0, warm_s = 0;           // all frames; the frames after the first filter step (code objects loaded,
    long callbacks = 0, warm = 0;                //  scratch allocated)
    std::vector<int8_t> local((size_t)L);
    std::vector<int32_t> bx((size_t)L), by((size_t)L), bz((size_t)L), slots((size_t)P);
    std::vector<int8_t> bv((size_t)L);
    std::vector<double> x((size_t)P, 7.5), y((size_t)P, 7.5), z((size_t)P, 1.5), yaw((size_t)P, 0.0), u((size_t)P);
    std::vector<float> noise(4 * (size_t)P);
    std::vector<uint8_t> img;
    std::vector<double> cloud(3 * ((size_t)npix + 1)), sample(3 * 500);   // one row more: the reference's draws are inclusive of n (S3P:317, 666) and a replay may name it
    std::vector<int32_t> pick(500), ray_idx(2400);
    // members of class Slam (S3P:56-61, 265-283, 823-825)
    double local_pos[3] = {5, 5, 1.5}, local_yaw = 0;
    double pos[4] = {7.5, 7.5, 1.5, 0}, last_update[4] = {7.5, 7.5, 1.5, 0}, diff[4] = {0, 0, 0, 0};
    double pos_scan[3] = {5, 5, 2}, yaw_scan = 0, vote_loc[4] = {5, 5, 2, 0};
    double gyr_yaw_old = 0, yaw_diff_scan = 0, rot_y = 0, rot_x = 0;   // no IMU here: rot_y = rot_x = 0 (S3P:301-302)
    int add_points = 1, best = 0;
    double truth[4] = {5, 5, 2, 0};               // the sensor in the room's frame = the localisation frame
    std::random_device rd;
    const pf_update_params upd = {10, 50, -3, -50, 3.0};   // S3P:637, 650-655, 695-700
    const int frames_per_step = replay ? rep.per_step : 8;
    long votes = 0, no_votes = 0, keyframes = 0;

    for (int t = 0; t < steps; ++t) {
        // ---- camera callback (S3P:168-205), `frames_per_step` depth frames ----
        for (int f = 0; f < frames_per_step; ++f) {
            const int fno = t * frames_per_step + f;
            double d[4] = {0, 0, 0, 0};
            if (replay) {
                expand_pooled(rep.pooled.data() + (size_t)fno * npix, img);
            } else {
                double nt[4];
                truth_at(fno + 1, nt);
                for (int k = 0; k < 4; ++k) { d[k] = nt[k] - truth[k]; truth[k] = nt[k]; }
                synth_depth(truth[0], truth[1], truth[2], truth[3], img);
            }
            int64_t n = 0;
            const auto t_cb = std::chrono::steady_clock::now();
            if (device_cloud) CK(loc, pf_depth_to_cloud_device(loc, img.data(), kW, kH, kPool, kFocal, cloud_dev, npix, &n));
            else CK(loc, pf_depth_to_cloud(loc, img.data(), kW, kH, kPool, kFocal, cloud.data(), npix, &n));   // S3P:173-175
            double pos_diff_scan[3] = {0, 0, 0}, yaw_diff = 0;
            if (truth_odometry) {
                pos_diff_scan[0] = std::cos(-truth[3] + d[3]) * d[0] - std::sin(-truth[3] + d[3]) * d[1];
                pos_diff_scan[1] = std::sin(-truth[3] + d[3]) * d[0] + std::cos(-truth[3] + d[3]) * d[1];
                pos_diff_scan[2] = d[2]; yaw_diff = d[3];
            } else if (n > 0) {
                // scan_localization(): S3P:300-410
                double gyr_yaw = pos[3] + yaw_diff_scan;         // S3P:184 (no gyro message)
                if (replay) { gyr_yaw = rep.gyr[3 * (size_t)fno]; rot_y = rep.gyr[3 * (size_t)fno + 1]; rot_x = rep.gyr[3 * (size_t)fno + 2]; }
                const double zyx[3] = {yaw_scan, rot_y, rot_x};  // S3P:301-302
                CK(loc, pf_ring_paint(loc));                                                    // S3P:305-312
                // S3P:315-319 draws in [0, n] (one past the cloud); the driver's own seeds draw in [0, n-1],
                // the recorded seeds were screened to stay below n
                pfh_draw_indices(replay ? rep.seed_vote[(size_t)fno] : rd(), 500, (int32_t)(replay ? n : n - 1), pick.data());
                int64_t hyp = -1;
                if (device_cloud) {                              // S3P:320-322 happens in the vote's rotation kernel
                    CK(loc, pf_scan_vote_device(loc, cloud_dev, n, pick.data(), 500, pos_scan, zyx, 5, -6, 3, 5, 0.25, nullptr, &hyp));
                } else {
                    for (int k = 0; k < 500; ++k)
                        for (int c = 0; c < 3; ++c) sample[3 * (size_t)k + c] = cloud[3 * (size_t)pick[(size_t)k] + c];
                    CK(loc, pf_scan_vote(loc, sample.data(), 500, pos_scan, zyx, 5, -6, 3, 5, 0.25, nullptr, &hyp));
                }
                double out[4];
                if (pfh_vote_hypothesis(hyp, pos_scan, zyx, 0.05, 5, -6, 3, 5, 0.25, out) == 0) {
                    std::memcpy(vote_loc, out, sizeof out);      // S3P:366-371; kept on "no vote"
                    ++votes;
                } else {
                    ++no_votes;
                }
                pfh_vote_epilogue(pos_scan, &yaw_scan, vote_loc, gyr_yaw, &gyr_yaw_old, pos_diff_scan,
                                  &yaw_diff_scan, &yaw_diff);                                   // S3P:379-404
                int32_t flags = 0;
                const double zyx_now[3] = {yaw_scan, rot_y, rot_x};
                if (device_cloud) CK(loc, pf_ring_update_device(loc, cloud_dev, n, pos_scan, zyx_now, &flags));
                else CK(loc, pf_ring_update(loc, cloud.data(), n, pos_scan, zyx_now, &flags));  // S3P:419-529
                keyframes += flags & 1;
            }
            // build_local_map(): S3P:621-710
            pfh_draw_indices(replay ? rep.seed_rays[(size_t)fno] : rd(), (int64_t)ray_idx.size(),
                             (int32_t)(replay ? n : n > 0 ? n - 1 : 0), ray_idx.data());                  // S3P:666-669
            std::lock_guard<std::mutex> lk(mtx);                 // S3P:622
            for (int k = 0; k < 3; ++k) local_pos[k] += pos_diff_scan[k];                       // S3P:623-626
            local_yaw += yaw_diff;
            const double lzyx[3] = {local_yaw, rot_y, rot_x};
            if (device_cloud) CK(h, pf_set_scan_device(h, cloud_dev, n, ray_idx.data(), n > 0 ? (int64_t)ray_idx.size() : 0));
            else CK(h, pf_set_scan(h, cloud.data(), n, ray_idx.data(), n > 0 ? (int64_t)ray_idx.size() : 0));
            CK(h, pf_update_map(h, PF_GRID_LOCAL, &upd, local_pos, lzyx));                      // S3P:628-704
            pfh_update_realtime_pos(pos, &pos[3], pos_diff_scan, yaw_diff, diff);              // S3P:712-728
            const double dt_cb = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_cb).count();
            callback_s += dt_cb; ++callbacks;
            if (t > 0) { warm_s += dt_cb; ++warm; }
            if (replay) std::printf("frame %d gpose %.17g %.17g %.17g %.17g\n", fno, pos[0], pos[1], pos[2], pos[3]);
        }
        // ---- particle_map() (S3P:827-962) ----
        double p_local[4], p_pos[4];
        int64_t nb;
        {
            std::lock_guard<std::mutex> lk(mtx);           // S3P:833-846
            CK(h, pf_read_local_map(h, local.data()));
            for (int k = 0; k < 3; ++k) p_local[k] = local_pos[k];
            p_local[3] = local_yaw;
            for (int k = 0; k < 4; ++k) p_pos[k] = pos[k];
            CK(h, pf_reset_local_map(h));
            local_pos[0] = 5; local_pos[1] = 5; local_pos[2] = 1.5; local_yaw = 0;
        }
        nb = pfh_snapshot_beams(local.data(), cfg.local_x, cfg.local_y, cfg.local_z, 100, 100, 30,
                                bx.data(), by.data(), bz.data(), bv.data(), (int64_t)L);   // S3P:868-877
        if (nb > cfg.max_beams) { std::fprintf(stderr, "%lld beams beyond max_beams\n", (long long)nb); return 1; }
        CK(h, pf_set_beams(h, bx.data(), by.data(), bz.data(), bv.data(), nb));
        CK(h, pf_measure(h, add_points));                                             // S3P:859-908
        pfh_draw_noise(replay ? rep.step_seed[(size_t)t] : rd(), P, 0.03f, 0.02f, noise.data());   // S3P:852-855
        pfh_predict(P, x.data(), y.data(), z.data(), yaw.data(), p_local[0] - 5, p_local[1] - 5,
                    p_local[2] - 1.5, p_local[3], noise.data());                      // S3P:911-914
        CK(h, pf_predict(h, x.data(), y.data(), z.data(), yaw.data()));
        CK(h, pf_weights(h, nullptr));                                                // S3P:920-937
        add_points = pfh_add_points_policy(p_pos, last_update, 0.4);                  // S3P:941-953
        pfh_draw_uniform(replay ? rep.step_seed[(size_t)t] + 1 : rd(), P, u.data());  // S3P:768-773
        int64_t nohit = 0, amb = 0;
        CK(h, pf_resample(h, u.data(), nullptr, slots.data(), &best, &nohit, &amb));  // S3P:954
        for (auto* v : {&x, &y, &z, &yaw}) pfh_permute(P, slots.data(), v->data(), sizeof(double));
        diff[0] = x[(size_t)best] - p_pos[0]; diff[1] = y[(size_t)best] - p_pos[1];  // S3P:956-959
        diff[2] = z[(size_t)best] - p_pos[2]; diff[3] = yaw[(size_t)best] - p_pos[3];
        if (replay) std::printf("step %d best %d add %d feedback %.17g %.17g %.17g %.17g\n", t, best, add_points, diff[0], diff[1], diff[2], diff[3]);
        double msg_pos[3], msg_q[4];
        pfh_pose_message(pos, pos[3], msg_pos, msg_q);                                // S3P:207-222
        // the filter's frame starts at (7.5, 7.5, 1.5), the room's at (5, 5, 2)
        std::printf("step %3d beams %6lld best %4d pose %.3f %.3f %.3f %.3f  truth %.3f %.3f %.3f %.3f  add %d  msg (%.2f %.2f %.2f)\n",
                    t, (long long)nb, best, x[(size_t)best], y[(size_t)best], z[(size_t)best], yaw[(size_t)best],
                    truth[0] + 2.5, truth[1] + 2.5, truth[2] - 0.5, truth[3], add_points, msg_pos[0], msg_pos[1], msg_pos[2]);
    }
    // ---- publish_cloud (S3P:1031-1051): the best particle's occupied cells ----
    std::vector<float> occ(3 * (size_t)(1 << 20));
    int64_t n_occ = 0, total_occ = 0;
    CK(h, pf_read_occupied(h, best, 80, 4, occ.data(), 1 << 20, &n_occ, &total_occ));
    std::printf("front end: %ld votes, %ld frames without a vote, %ld keyframes; published cloud: %lld points\n",
                votes, no_votes, keyframes, (long long)total_occ);
    std::fprintf(stderr, "callback(): %ld frames, %.1f us each, %.1f us after the first filter step (wall, host-inclusive%s)\n",
                 callbacks, callbacks ? callback_s / callbacks * 1e6 : 0.0, warm ? warm_s / warm * 1e6 : 0.0,
                 device_cloud ? ", cloud kept in HBM" : "");
    if (device_cloud) { CK(h, pf_set_stream(h, nullptr)); (void)hipFree(cloud_dev); }
    pf_destroy(loc);
    pf_destroy(h);
    return 0;
}
