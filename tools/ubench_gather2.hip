// Micro-benchmark 2: do VALU work and divergent byte gathers overlap on a CU? Lane->line mapping
// contiguous (sorted lanes, like K1 after the particle ordering) or interleaved.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_gather2.hip -o tools/ubench_gather2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int VALU, bool LOAD>
__global__ __launch_bounds__(256) void k(const unsigned char* __restrict__ tab, unsigned mask, int L,
                                         int contiguous, int line_stride, int iters,
                                         unsigned* __restrict__ out) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const unsigned line = contiguous ? lane / (64 / L) : lane % L;
    const unsigned lane_off = line * line_stride + (lane * 7u) % 61u;      // bytes within the line
    unsigned acc = 0;
    float fa = (float)lane * 1e-3f, fb = 1.0001f;
    unsigned base = wave * 2654435761u;
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            base = base * 1664525u + 1013904223u;
            const unsigned a = (((base >> 6) << 6) + lane_off) & mask;
#pragma unroll
            for (int v = 0; v < VALU; ++v) fa = __builtin_fmaf(fa, fb, 0.5f);
            if (LOAD) acc += tab[a];
        }
    }
    if (acc == 0xdeadbeef || fa == 123.0f) out[wave] = acc;
}

int main() {
    const size_t TAB = 64u << 20;
    unsigned char* tab; unsigned* out;
    CK(hipMalloc(&tab, TAB)); CK(hipMalloc(&out, 1 << 22));
    CK(hipMemset(tab, 1, TAB));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 2048, iters = 2048;
    const double wave_instr = (double)blocks * 4 * iters;
    auto time = [&](auto launch, const char* name, int L, int contiguous) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%-22s lines %2d %-11s : %7.3f ms  %6.1f CU-cycles per iteration (2.4 GHz)\n", name, L,
               contiguous ? "contiguous" : "interleaved", best, best * 1e-3 * 2.4e9 * 256 / wave_instr);
    };
    const unsigned mask = (2u << 20) - 1;   // 2 MiB table: L2 resident
#define RUN(V, LD, NAME, L, C) time([&] { hipLaunchKernelGGL((k<V, LD>), dim3(blocks), dim3(256), 0, 0, tab, mask, L, C, 256, iters, out); }, NAME, L, C)
    RUN(24, false, "valu24 only", 1, 1);
    RUN(12, false, "valu12 only", 1, 1);
    for (int c = 1; c >= 0; --c)
        for (int L : {1, 2, 4, 8, 16}) {
            RUN(0, true, "load only", L, c);
            RUN(12, true, "load + valu12", L, c);
            RUN(24, true, "load + valu24", L, c);
        }
    return 0;
}
