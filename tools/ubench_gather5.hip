// Micro-benchmark 5: when all 64 lanes of a wave want bytes of ONE 64-B line (K1's common case: the lanes of a
// wave are neighbours in pose, and a beam lands them in the same 4x4x4 brick), is it cheaper to load the line
// once, cooperatively, and hand each lane its byte through the LDS crossbar (ds_bpermute) than to let every
// lane gather its own byte?
//   A  byte gather, 64 lanes, random bytes of one line           (what K1 does)
//   B  dword load, address = line + (lane & 15) * 4              (the line's 16 dwords, four lanes each)
//   C  dword load, address = base + lane * 4                     (256 contiguous bytes)
//   E  dword load by lanes 0..15 only (the others masked off)    (one line, one lane per dword)
//   D  B + ds_bpermute + byte extract                            (the candidate)
//   F  E + ds_bpermute + byte extract
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_gather5.hip -o tools/ubench_gather5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned char* __restrict__ tab, unsigned mask, int iters,
                                         unsigned* __restrict__ out) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    unsigned acc = 0;
    unsigned base = wave * 2654435761u;
    const unsigned mybyte = (lane * 7u) % 61u;               // this lane's byte in the line
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            base = base * 1664525u + 1013904223u;
            const unsigned line = ((base >> 6) << 6) & mask;
            if (MODE == 0) acc += tab[line + mybyte];
            if (MODE == 1) acc += *(const unsigned*)(tab + line + (lane & 15u) * 4u);
            if (MODE == 2) acc += *(const unsigned*)(tab + ((line & ~255u) + lane * 4u));
            if (MODE == 4) { unsigned v = 0; if (lane < 16) v = *(const unsigned*)(tab + line + lane * 4u); acc += v; }
            if (MODE == 3) {
                const unsigned v = *(const unsigned*)(tab + line + (lane & 15u) * 4u);
                const unsigned d = (unsigned)__builtin_amdgcn_ds_bpermute((int)((mybyte >> 2) << 2), (int)v);
                acc += (d >> ((mybyte & 3u) * 8u)) & 0xffu;
            }
            if (MODE == 5) {
                unsigned v = 0; if (lane < 16) v = *(const unsigned*)(tab + line + lane * 4u);
                const unsigned d = (unsigned)__builtin_amdgcn_ds_bpermute((int)((mybyte >> 2) << 2), (int)v);
                acc += (d >> ((mybyte & 3u) * 8u)) & 0xffu;
            }
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
    auto run = [&](auto kern, const char* name) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, tab, mask, iters, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%-58s %7.3f ms  %6.1f CU-cycles per wave-instruction (2.4 GHz)\n", name, best,
               best * 1e-3 * 2.4e9 * 256 / wave_instr);
    };
    run(k<0>, "A byte gather, random bytes of one line");
    run(k<1>, "B dword load, line + (lane&15)*4");
    run(k<2>, "C dword load, 256 contiguous bytes");
    run(k<4>, "E dword load by lanes 0..15 only");
    run(k<3>, "D = B + ds_bpermute + byte extract");
    run(k<5>, "F = E + ds_bpermute + byte extract");
    return 0;
}
