// Micro-benchmark 4: what does a wave-wide gather cost as a function of the bytes each lane
// loads (1, 4, 8, 16) at a given number of 64-B lines per instruction? Decides whether K1 may
// fetch a whole z-octet (or two) per lane instead of one cell.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_gather4.hip -o tools/ubench_gather4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int W>
__global__ __launch_bounds__(256) void k(const unsigned char* __restrict__ tab, unsigned mask, int L,
                                         int line_stride, int iters, unsigned* __restrict__ out) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const unsigned line = lane / (64 / L);
    const unsigned lane_off = line * line_stride + (((lane * 7u) % 61u) & ~(unsigned)(W - 1)) % (64 - W + 1);
    unsigned acc = 0;
    unsigned base = wave * 2654435761u;
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            base = base * 1664525u + 1013904223u;
            const unsigned a = ((((base >> 6) << 6) + lane_off) & mask) & ~(unsigned)(W - 1);
            if (W == 1) acc += tab[a];
            if (W == 4) acc += *(const unsigned*)(tab + a);
            if (W == 8) { const uint2 v = *(const uint2*)(tab + a); acc += v.x ^ v.y; }
            if (W == 16) { const uint4 v = *(const uint4*)(tab + a); acc += v.x ^ v.y ^ v.z ^ v.w; }
        }
    }
    if (acc == 0xdeadbeef) out[wave] = acc;
}

int main() {
    const size_t TAB = 64u << 20;
    unsigned char* tab; unsigned* out;
    CK(hipMalloc(&tab, TAB)); CK(hipMalloc(&out, 1 << 22));
    CK(hipMemset(tab, 1, TAB));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 2048, iters = 2048;
    const double wave_instr = (double)blocks * 4 * iters;
    const unsigned mask = (2u << 20) - 1;   // 2 MiB table: L2 resident
    auto run = [&](auto kern, int W, int L) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, tab, mask, L, 256, iters, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("bytes/lane %2d lines %2d : %7.3f ms  %6.1f CU-cycles per wave-gather (2.4 GHz)\n", W, L, best,
               best * 1e-3 * 2.4e9 * 256 / wave_instr);
    };
    for (int L : {1, 2, 4, 8, 16}) {
        run(k<1>, 1, L); run(k<4>, 4, L); run(k<8>, 8, L); run(k<16>, 16, L);
    }
    return 0;
}
