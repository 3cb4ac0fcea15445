// TEST INFRASTRUCTURE (oracle/): appended behind ref_harness_tail.cpp by oracle/build_ref.sh, in
// the same translation unit as the reference's own text. The reference's class Slam is compiled
// from its own lines (members S3P:115-146, 150-151; table loops S3P:157-162; member functions
// S3P:258-729); this file holds ONE instance of it, accessors to its members, and thin callers of
// its member functions in the order callback() calls them (S3P:173-191). Nothing here restates the
// reference. Runs only in the build container; its outputs are the fixtures under tests/golden/.
#include <new>
#include <vector>

static Slam* ref_slam_ptr = nullptr;
static Slam& S() {
    if (!ref_slam_ptr) {
        void* mem = ::operator new(sizeof(Slam));
        memset(mem, 0, sizeof(Slam));       // the reference leaves points_scan_map_index etc.
        ref_slam_ptr = new (mem) Slam();    // uninitialised (S3P:288); zero is the defined choice
    }
    return *ref_slam_ptr;
}

// Runs `f` with std::cout captured ("update", "reset localization", S3P:482, 488).
template <class F> static void quiet(F f) {
    std::ostringstream sink;
    std::streambuf* old = std::cout.rdbuf(sink.rdbuf());
    f();
    std::cout.rdbuf(old);
}

extern "C" {

// A fresh Slam object and fresh globals of S3P:56-67 (what a newly started node holds).
void ref_slam_reset(void) {
    if (ref_slam_ptr) { ref_slam_ptr->~Slam(); ::operator delete((void*)ref_slam_ptr); ref_slam_ptr = nullptr; }
    S();
    memset(map_localization, 0, map_elements);
    memset(map_local, 0, map_elements);
    memset(map_local_copy, 0, map_elements);
    local_pos_xyz[0] = 10 / 2; local_pos_xyz[1] = 10 / 2; local_pos_xyz[2] = 1.5;   // S3P:56
    for (int i = 0; i < 3; i++) { local_euler_zyx[i] = 0; euler_zyx[i] = 0; pos_xyz_diff[i] = 0; }
    pos_xyz[0] = 7.5; pos_xyz[1] = 7.5; pos_xyz[2] = 1.5;                            // S3P:59
    rot_z_diff = 0;
}
int ref_res_h(void) { return Slam::res_h; }
int ref_res_v(void) { return Slam::res_v; }

// S3P:173-175: the depth image (bytes of sensor_msgs/Image::data) to the cloud.
int ref_depth_to_cloud(const uint8_t* image, int n_bytes) {
    std::vector<uint8_t> v(image, image + n_bytes);
    S().resize_depth_to_array(v);
    S().img_to_cloud();
    return S().cloudpoints_index;
}
// A synthetic cloud put where img_to_cloud() would have left it (S3P:126-127).
void ref_set_cloud(const double* pts, int n) {
    for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) S().cloudpoints[i][k] = pts[3 * i + k];
    S().cloudpoints_index = n;
}
int ref_get_cloud(double* out, int cap) {
    int n = S().cloudpoints_index;
    for (int i = 0; i < n && i < cap; i++) for (int k = 0; k < 3; k++) out[3 * i + k] = S().cloudpoints[i][k];
    return n;
}
void ref_get_depth_arrays(int* dx, int* dy, int* dz) {
    memcpy(dx, S().depth_x, sizeof(S().depth_x));
    memcpy(dy, S().depth_y, sizeof(S().depth_y));
    memcpy(dz, S().depth_z, sizeof(S().depth_z));
}
// The IMU filter's angles (S3P:150), which sync_filter() (S3P:244-246) would have filled.
void ref_set_gyr(double z, double y, double x) {
    S().gyr_acc_filter_zyx[0] = z; S().gyr_acc_filter_zyx[1] = y; S().gyr_acc_filter_zyx[2] = x;
}
// S3P:184, the branch callback() takes while no IMU sample is in reach.
void ref_gyr_fallback(void) { S().gyr_acc_filter_zyx[0] = euler_zyx[0] + S().euler_zyx_diff_scan[0]; }

void ref_scan_localization(void) { quiet([] { S().scan_localization(); }); }
void ref_scan_localization_map_update(void) { quiet([] { S().scan_localization_map_update(); }); }
void ref_build_local_map(void) { quiet([] { S().build_local_map(); }); }

// scan state: out[0..2] pos_xyz_scan, [3] euler_zyx_scan[0], [4..6] Tx_loc,Ty_loc,Tz_loc,
// [7] rot_z_loc, [8..10] pos_xyz_diff_scan, [11] euler_zyx_diff_scan[0], [12] euler_zyx_diff,
// [13] euler_zyx_diff_gyr, [14] euler_zyx_diff_gyr_old, [15] rot_y, [16] rot_x, [17] sum_ (last)
void ref_get_scan_state(double out[18]) {
    Slam& s = S();
    for (int i = 0; i < 3; i++) { out[i] = s.pos_xyz_scan[i]; out[8 + i] = s.pos_xyz_diff_scan[i]; }
    out[3] = s.euler_zyx_scan[0];
    out[4] = s.Tx_loc; out[5] = s.Ty_loc; out[6] = s.Tz_loc; out[7] = s.rot_z_loc;
    out[11] = s.euler_zyx_diff_scan[0]; out[12] = s.euler_zyx_diff; out[13] = s.euler_zyx_diff_gyr;
    out[14] = s.euler_zyx_diff_gyr_old; out[15] = s.rot_y; out[16] = s.rot_x; out[17] = s.sum_;
}
void ref_set_scan_pose(double x, double y, double z, double yaw) {
    Slam& s = S();
    s.pos_xyz_scan[0] = x; s.pos_xyz_scan[1] = y; s.pos_xyz_scan[2] = z; s.euler_zyx_scan[0] = yaw;
}
void ref_set_diffs(double dx, double dy, double dz, double dyaw) {   // inputs of build_local_map
    Slam& s = S();
    s.pos_xyz_diff_scan[0] = dx; s.pos_xyz_diff_scan[1] = dy; s.pos_xyz_diff_scan[2] = dz;
    s.euler_zyx_diff = dyaw;
}
void ref_set_rot_yx(double y, double x) { S().rot_y = y; S().rot_x = x; }
void ref_get_cloud_test(double* out) { memcpy(out, S().cloud_test, sizeof(S().cloud_test)); }

// keyframe ring (S3P:286-289, 415-416, 296)
int* ref_ring_idx(void) { return S().points_scan_map; }
int* ref_ring_count(void) { return S().points_scan_map_index; }
int* ref_ring_life(void) { return S().points_scan_map_life; }
void ref_get_ring_state(double out[7]) {
    Slam& s = S();
    for (int i = 0; i < 3; i++) { out[i] = s.pos_xyz_update[i]; out[3 + i] = s.euler_zyx_update[i]; }
    out[6] = s.first_run;
}
uint8_t* ref_map_localization_ptr(void) { return map_localization; }

// globals of S3P:56-61
void ref_get_global_pose(double out[4]) {
    out[0] = pos_xyz[0]; out[1] = pos_xyz[1]; out[2] = pos_xyz[2]; out[3] = euler_zyx[0];
}
void ref_set_feedback(double dx, double dy, double dz, double drot) {
    pos_xyz_diff[0] = dx; pos_xyz_diff[1] = dy; pos_xyz_diff[2] = dz; rot_z_diff = drot;
}

void ref_reverse_index_conv_p(int out[3], int j) { reverse_index_conv_p(out, j); }

// libstdc++ draws in the pattern of S3P:315-319 and S3P:664-669: n draws of
// uniform_int_distribution<>(0, hi) from one mt19937(seed).
void ref_index_stream(unsigned seed, int n, int hi, int* out) {
    std::mt19937 gen(seed);
    std::uniform_int_distribution<> distr(0, hi);
    for (int i = 0; i < n; i++) out[i] = distr(gen);
}

}  // extern "C"
