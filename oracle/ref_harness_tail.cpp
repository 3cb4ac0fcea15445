// TEST INFRASTRUCTURE (oracle/): appended BEHIND the reference's own text in one translation
// unit by oracle/build_ref.sh (the reference's struct map_particles and its globals are only
// visible inside that unit). Nothing here restates the reference: these are accessors to the
// reference's globals (S3P:56-69, S3P:756, S3P:823-825) and thin callers of its functions
// particle_map() (S3P:827) and resample() (S3P:757), plus two helpers that draw from the
// system's libstdc++ <random> in the pattern the reference uses, so that a fixture can hold
// the noise and threshold values as plain data.
//
// Runs ONLY in the build container (the reference never travels); its outputs are the
// committed fixtures under tests/golden/.
#include <sstream>
#include <string>

unsigned pf_ref_next_seed = 1;

extern "C" {

void ref_set_seed(unsigned s) { pf_ref_next_seed = s; }
unsigned ref_get_seed(void) { return pf_ref_next_seed; }

// Re-point the reference's mutable globals (S3P:68-69) at an array of n particles.
int ref_alloc_particles(int n) {
    delete[] Particles;
    Particles = new map_particles[n];
    num_particles = n;
    return (int)sizeof(map_particles);
}
int ref_p_map_elements(void) { return map_particles::p_map_elements; }
int ref_map_elements(void) { return map_elements; }
uint8_t* ref_particle_map_ptr(int i) { return Particles[i].map; }
int8_t* ref_map_local_ptr(void) { return map_local; }
void ref_init_maps(void) { initilize_particles_map(); }   // S3P:71-77 (prior 50)

void ref_set_particle(int i, double x, double y, double z, double rot, double w) {
    Particles[i].pos_x = x; Particles[i].pos_y = y; Particles[i].pos_z = z;
    Particles[i].rot_z = rot; Particles[i].weight = w;
}
void ref_get_particle(int i, double out[5]) {
    out[0] = Particles[i].pos_x; out[1] = Particles[i].pos_y; out[2] = Particles[i].pos_z;
    out[3] = Particles[i].rot_z; out[4] = Particles[i].weight;
}
void ref_set_local_pose(double x, double y, double z, double rot) {
    local_pos_xyz[0] = x; local_pos_xyz[1] = y; local_pos_xyz[2] = z; local_euler_zyx[0] = rot;
}
void ref_get_local_pose(double out[4]) {
    out[0] = local_pos_xyz[0]; out[1] = local_pos_xyz[1]; out[2] = local_pos_xyz[2];
    out[3] = local_euler_zyx[0];
}
void ref_set_global_pose(double x, double y, double z, double rot) {
    pos_xyz[0] = x; pos_xyz[1] = y; pos_xyz[2] = z; euler_zyx[0] = rot;
}
void ref_set_policy(int add, double lx, double ly, double lz, double lrot, int best) {
    add_points = add; pos_last_update[0] = lx; pos_last_update[1] = ly; pos_last_update[2] = lz;
    rot_last_update = lrot; index_high_weight = best;
}
void ref_get_policy(int* add, double last[4], int* best) {
    *add = add_points; last[0] = pos_last_update[0]; last[1] = pos_last_update[1];
    last[2] = pos_last_update[2]; last[3] = rot_last_update; *best = index_high_weight;
}
void ref_get_feedback(double out[4]) {
    out[0] = pos_xyz_diff[0]; out[1] = pos_xyz_diff[1]; out[2] = pos_xyz_diff[2]; out[3] = rot_z_diff;
}

// Calls the reference's particle_map() with std::cout captured at 17 significant digits (the
// reference streams the pre-resample normalised weights there, S3P:934-939). Returns the
// number of bytes of captured text copied into `text` (NUL-terminated, truncated to cap-1).
int ref_particle_map(char* text, int cap) {
    std::ostringstream cap_stream;
    cap_stream.precision(17);
    std::streambuf* old = std::cout.rdbuf(cap_stream.rdbuf());
    std::streamsize oldp = std::cout.precision(17);
    particle_map();
    std::cout.rdbuf(old);
    std::cout.precision(oldp);
    std::string s = cap_stream.str();
    int n = (int)s.size() < cap - 1 ? (int)s.size() : cap - 1;
    memcpy(text, s.data(), n);
    text[n] = 0;
    return n;
}

void ref_resample(void) { resample(); }

int ref_index_conv(int x, int y, int z) { return index_conv(x, y, z); }
int ref_index_conv_p(int x, int y, int z) { return index_conv_p(x, y, z); }
void ref_reverse_index_conv(int out[3], int j) { reverse_index_conv(out, j); }
void ref_rot_matrix(double out[9], double z, double y, double x) {
    double m[3][3];
    rot_matrix(m, z, y, x);
    for (int i = 0; i < 9; i++) out[i] = m[i / 3][i % 3];
}
void ref_rotate(double out[3], int p[3], double m9[9]) {
    double m[3][3];
    for (int i = 0; i < 9; i++) m[i / 3][i % 3] = m9[i];
    rotate(out, p, m);
}

// libstdc++ draws, in the pattern of S3P:852-855 + S3P:911-914: one mt19937, two
// normal_distribution<float> objects, per particle pos_d, pos_d, rot_d, rot_d.
void ref_noise_stream(unsigned seed, int n, float* out) {
    std::mt19937 gen(seed);
    std::normal_distribution<float> pos_d(0, 0.03);
    std::normal_distribution<float> rot_d(0, 0.02);
    for (int i = 0; i < n; i++) {
        out[4 * i + 0] = pos_d(gen);
        out[4 * i + 1] = pos_d(gen);
        out[4 * i + 2] = rot_d(gen);
        out[4 * i + 3] = rot_d(gen);
    }
}
// ... and of S3P:768-773: n uniform_real_distribution<double>(0,1) draws from one mt19937.
void ref_uniform_stream(unsigned seed, int n, double* out) {
    std::mt19937 gen(seed);
    std::uniform_real_distribution<double> distr(0.0, 1.0);
    for (int i = 0; i < n; i++) out[i] = distr(gen);
}

}  // extern "C"
