// pf_scan_vote_static: the scan vote (S3P:328-376) for callers that cannot allocate. Same kernels
// as pf_scan_vote (pfk_scan_vote: k_vote_rotate + k_vote); their scratch -- the yaw hypotheses'
// rotation matrices, the lattice translations and the rotated points -- lives in a file-scope
// device array instead of the handle's grown-on-demand allocation. No hipMalloc/hipFree, no host
// heap use, no host<->device copy, no synchronisation: two kernel launches and one small set-up
// launch on the caller's stream.
#include "../../include/slam_pf.h"
#include "pf_kernels.h"
#include "pf_trig.h"

#include <math.h>
#include <mutex>

#define PF_STATIC_MATS_BYTES   (((PF_STATIC_MAX_YAW * 9 * 8) + 255) / 256 * 256)
#define PF_STATIC_TRANS_BYTES  (PF_STATIC_MAX_YAW * PF_STATIC_MAX_T * PF_STATIC_MAX_T * PF_STATIC_MAX_T * 3 * 8)
#define PF_STATIC_ROT_BYTES    (PF_STATIC_MAX_YAW * PF_STATIC_MAX_POINTS * 3 * 8)
#define PF_STATIC_BYTES        (PF_STATIC_MATS_BYTES + PF_STATIC_TRANS_BYTES + PF_STATIC_ROT_BYTES)

// One copy per GPU the library runs on, held until the process ends (1.7 MiB). In host code this
// name is only a handle for hipGetSymbolAddress: the address kernels and copies use is scratch_ptr().
static __device__ __align__(256) unsigned char scratch[PF_STATIC_BYTES];

namespace {

constexpr int kMaxDevices = 64;
void* g_ptr[kMaxDevices];          // device address of `scratch` on each GPU, 0 = not asked yet
std::mutex g_mu;

// With `device` current: the address of this GPU's copy of the array.
hipError_t scratch_ptr(int device, unsigned char** out) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_ptr[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_ptr[device] = p;
    }
    *out = (unsigned char*)g_ptr[device];
    return hipSuccess;
}

struct VoteSetup {
    double mats[PF_STATIC_MAX_YAW * 9];   // host-libm rotation matrices (S3P:536-544), one per yaw hypothesis
    double pos[3];
    double res, length;
    int n_yaw, n_t;
};

// Writes the set-up into the scratch the vote kernels read: the matrices as given, the translations
// by the reference's own expression (S3P:335, 341, 348; compiled without contraction like the host's).
__global__ __launch_bounds__(256) void k_vote_setup(VoteSetup s, double* __restrict__ mats, double* __restrict__ trans) {
    const int per = s.n_t * s.n_t * s.n_t, nh = s.n_yaw * per;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < s.n_yaw * 9) mats[k] = s.mats[k];
    if (k >= nh) return;
    const int q = k % per;
    const int a = q / (s.n_t * s.n_t), b = (q / s.n_t) % s.n_t, d = q % s.n_t;
    trans[3 * (size_t)k + 0] = s.pos[0] + a * s.res - (s.length - s.res) / 2;
    trans[3 * (size_t)k + 1] = s.pos[1] + b * s.res - (s.length - s.res) / 2;
    trans[3 * (size_t)k + 2] = s.pos[2] + d * s.res - (s.length - s.res) / 2;
}

}  // namespace

// Resolves this GPU's copy of the scratch array (a code-object load and a symbol look-up on the first call per
// device, which may allocate inside the runtime and is not allowed under stream capture): call it once per device
// at set-up, outside any capture; pf_scan_vote_static then only launches kernels.
extern "C" int pf_scan_vote_static_warmup(int32_t device) {
    if (device < 0 || device >= kMaxDevices) { pfk_set_global_error("pf_scan_vote_static_warmup: device ordinal outside [0, 64)"); return PF_ERR_ARG; }
    int prev_device = -1;
    (void)hipGetDevice(&prev_device);
    hipError_t e = prev_device == device ? hipSuccess : hipSetDevice(device);
    unsigned char* base = nullptr;
    if (e == hipSuccess) e = scratch_ptr(device, &base);
    if (prev_device >= 0 && prev_device != device) (void)hipSetDevice(prev_device);
    if (e != hipSuccess) {
        pfk_set_global_error(hipGetErrorString(e));
        return e == hipErrorInvalidDevice || e == hipErrorNoDevice ? PF_ERR_NODEVICE : PF_ERR_HIP;
    }
    return PF_OK;
}

extern "C" int pf_scan_vote_static(int32_t device, void* hip_stream, const uint8_t* grid_dev, int32_t gx,
                                   int32_t gy, int32_t gz, double resolution, const double* points_dev,
                                   int64_t n_points, const double pos[3], const double zyx[3], int32_t n_yaw,
                                   int32_t yaw_k0, int32_t yaw_kstep, int32_t n_t, double length,
                                   int32_t* sums_dev) {
#define REJECT(cond, msg) do { if (cond) { pfk_set_global_error("pf_scan_vote_static: " msg); return PF_ERR_ARG; } } while (0)
    REJECT(!grid_dev || !sums_dev || !pos || !zyx, "null argument");
    REJECT(device < 0 || device >= kMaxDevices, "device ordinal outside [0, 64)");
    REJECT(n_points < 0 || n_points > PF_STATIC_MAX_POINTS, "n_points beyond PF_STATIC_MAX_POINTS (4096)");
    REJECT(n_points > 0 && !points_dev, "null points");
    REJECT(n_yaw < 1 || n_yaw > PF_STATIC_MAX_YAW, "n_yaw outside 1..PF_STATIC_MAX_YAW (16)");
    REJECT(n_t < 1 || n_t > PF_STATIC_MAX_T, "n_t outside 1..PF_STATIC_MAX_T (8)");
    REJECT(gx < 1 || gy < 1 || gz < 1 || (long long)gx * gy * gz >= 2147483647LL || (long long)gy * gz >= (1LL << 26),
           "grid dims out of range");
    REJECT(!(resolution > 0.0) || !isfinite(resolution) || !isfinite(length), "resolution / length");
    for (int k = 0; k < 3; ++k) REJECT(!isfinite(pos[k]) || fabs(pos[k]) >= 1e6 || !isfinite(zyx[k]), "pose not finite");
#undef REJECT
    // the caller's current device is put back before returning
    int prev_device = -1;
    (void)hipGetDevice(&prev_device);
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{prev_device == device ? -1 : prev_device};
    hipError_t e = prev_device == device ? hipSuccess : hipSetDevice(device);
    unsigned char* base = nullptr;
    if (e == hipSuccess) e = scratch_ptr(device, &base);
    if (e != hipSuccess) {
        pfk_set_global_error(hipGetErrorString(e));
        return e == hipErrorInvalidDevice || e == hipErrorNoDevice ? PF_ERR_NODEVICE : PF_ERR_HIP;
    }
    double* d_m = (double*)base;
    double* d_t = (double*)(base + PF_STATIC_MATS_BYTES);
    double* d_r = (double*)(base + PF_STATIC_MATS_BYTES + PF_STATIC_TRANS_BYTES);
    VoteSetup s;
    const double pi = 2 * acos(0.0);                              // S3P:24
    for (int r = 0; r < n_yaw; ++r) {
        const double z = zyx[0] + (yaw_k0 + yaw_kstep * r) * pi / 180, y = zyx[1], x = zyx[2];   // S3P:329
        double* m = &s.mats[9 * r];
        pf_rot_zyx(m, z, y, x);    // S3P:536-544
    }
    for (int k = 9 * n_yaw; k < PF_STATIC_MAX_YAW * 9; ++k) s.mats[k] = 0;
    s.pos[0] = pos[0]; s.pos[1] = pos[1]; s.pos[2] = pos[2];
    s.res = resolution; s.length = length; s.n_yaw = n_yaw; s.n_t = n_t;
    const int per = n_t * n_t * n_t, nh = n_yaw * per;
    hipStream_t st = (hipStream_t)hip_stream;
    const int setup_items = nh > 9 * n_yaw ? nh : 9 * n_yaw;
    hipLaunchKernelGGL(k_vote_setup, dim3((setup_items + 255) / 256), dim3(256), 0, st, s, d_m, d_t);
    e = hipGetLastError();
    if (e == hipSuccess)
        e = pfk_scan_vote(st, grid_dev, points_dev, nullptr, (int)n_points, d_m, n_yaw, d_t, per, 1 / resolution, gy, gz,
                          (long long)gx * gy * gz, d_r, sums_dev);
    if (e != hipSuccess) { pfk_set_global_error(hipGetErrorString(e)); return PF_ERR_HIP; }
    return PF_OK;
}
