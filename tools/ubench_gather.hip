// Micro-benchmark: cost of a wave-wide byte / dword gather on gfx950 as a function of how the 64
// lane addresses spread over dwords and 64-B lines, from a table that fits L1 / L2 / MALL.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_gather.hip -o tools/ubench_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int WIDTH>   // 1 = ubyte, 4 = dword
__global__ __launch_bounds__(256) void k_gather(const unsigned char* __restrict__ tab, unsigned mask,
                                                int L, int D, int line_stride, int iters,
                                                unsigned* __restrict__ out) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const unsigned lane_off = (lane % L) * line_stride + ((lane / L) % D) * 4 + (WIDTH == 1 ? (lane / (L * D)) & 3 : 0);
    unsigned acc = 0;
    unsigned base = wave * 2654435761u;
    for (int it = 0; it < iters; it += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            base = base * 1664525u + 1013904223u;
            const unsigned a = (((base >> 6) << 6) + lane_off) & mask;
            if (WIDTH == 1) acc += tab[a];
            else acc += *(const unsigned*)(tab + (a & ~3u));
        }
    }
    if (acc == 0xdeadbeef) out[wave] = acc;
}

__global__ __launch_bounds__(256) void k_lds(const unsigned char* __restrict__ tab, int L, int iters,
                                             unsigned* __restrict__ out) {
    __shared__ unsigned char sh[65536];
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) ((unsigned*)sh)[i] = ((const unsigned*)tab)[i];
    __syncthreads();
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    unsigned acc = 0, base = wave * 2654435761u + lane * (L ? 40503u : 0u);
    for (int it = 0; it < iters; it += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            base = base * 1664525u + 1013904223u;
            acc += sh[(base >> 8) & 65535u];
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
    auto run = [&](int width, unsigned bytes, int L, int D, int ls) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (width == 1) hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(256), 0, 0, tab, bytes - 1, L, D, ls, iters, out);
            else hipLaunchKernelGGL(k_gather<4>, dim3(blocks), dim3(256), 0, 0, tab, bytes - 1, L, D, ls, iters, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        // CU-cycles per wave instruction at 2.4 GHz, 256 CUs
        const double cyc = best * 1e-3 * 2.4e9 * 256 / wave_instr;
        printf("width %d table %6u KiB  lines %2d dwords/line %2d stride %5d : %8.3f ms  %6.1f CU-cycles/wave-instr  %7.1f Glane/s\n",
               width, bytes >> 10, L, D, ls, best, cyc, wave_instr * 64 / (best * 1e-3) / 1e9);
    };
    const unsigned sizes[3] = {16u << 10, 2u << 20, 16u << 20};
    for (unsigned sz : sizes)
        for (int width : {1, 4}) {
            run(width, sz, 1, 1, 64);      // all lanes one dword
            run(width, sz, 1, 4, 64);      // 4 dwords of one line
            run(width, sz, 1, 16, 64);     // 16 dwords of one line
            run(width, sz, 2, 8, 64);
            run(width, sz, 4, 4, 64);
            run(width, sz, 4, 4, 256);     // 4 lines a grid row apart
            run(width, sz, 8, 2, 256);
            run(width, sz, 16, 1, 256);
            run(width, sz, 16, 4, 256);
            run(width, sz, 64, 1, 256);
        }
    for (int L : {0, 1}) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), 0, 0, tab, L, iters, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("LDS ds_read_u8 %s : %8.3f ms  %6.1f CU-cycles/wave-instr\n", L ? "random per lane" : "wave-uniform address",
               best, best * 1e-3 * 2.4e9 * 256 / wave_instr);
    }
    return 0;
}
