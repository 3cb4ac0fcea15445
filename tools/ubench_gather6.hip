// Micro-benchmark 6: what does the texture path charge for a WINDOW load (64 lanes, one dword each, covering four
// 64-B lines of two 128-B segments), by instruction form and lane-to-address mapping? (k_measure_tile's load.)
//   0  global_load_dword, 256 contiguous bytes, 256-B aligned                       (ubench 5's C)
//   1  global_load_dword, two 128-B segments (64-B aligned, far apart), lanes linear within a segment
//   2  the same segments, lanes permuted inside a segment (lane 4j+q -> bytes 8j + 4(q&1) + 64(q>>1))
//   3  buffer_load_dword offen + soffset, mapping of 1
//   4  buffer_load_dword offen + soffset, mapping of 2
//   5  global_load_dword, four 64-B lines far apart, 16 lanes each, linear
//   6  global_load_dwordx4 by lanes 0..15 (the four lines, 16 B per lane), other lanes masked off
//   7  global_load_dwordx2 by lanes 0..31
//   8  byte gather, random bytes of the four lines                                    (what k_measure_col does)
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_gather6.hip -o tools/ubench_gather6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned char* __restrict__ tab, unsigned mask, int iters, unsigned segstride,
                                         unsigned* __restrict__ out) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    unsigned acc = 0;
    unsigned base = __builtin_amdgcn_readfirstlane(wave * 2654435761u);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)tab, 0, (int)(mask + 1u + 2u * segstride), 0x00020000);
    const unsigned lin = (lane >> 5) * segstride + (lane & 31u) * 4u;
    const unsigned q = lane & 3u, j = (lane >> 2) & 7u;
    const unsigned perm = (lane >> 5) * segstride + 8u * j + 4u * (q & 1u) + 64u * (q >> 1);
    const unsigned four = (lane >> 4) * (segstride / 2u + 64u) + (lane & 15u) * 4u;
    const unsigned x4 = (lane & 3u) * (segstride / 2u + 64u) + ((lane >> 2) & 3u) * 16u;
    const unsigned x2 = ((lane >> 3) & 3u) * (segstride / 2u + 64u) + (lane & 7u) * 8u;
    const unsigned rnd = ((lane * 7u) % 4u) * (segstride / 2u + 64u) + (lane * 13u) % 61u;
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            base = base * 1664525u + 1013904223u;                     // scalar
            const unsigned line = ((base >> 6) << 6) & mask;
            if (MODE == 0) acc += *(const unsigned*)(tab + ((line & ~255u) + lane * 4u));
            if (MODE == 1) acc += *(const unsigned*)(tab + line + lin);
            if (MODE == 2) acc += *(const unsigned*)(tab + line + perm);
            if (MODE == 3) acc += (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)lin, (int)line, 0);
            if (MODE == 4) acc += (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)perm, (int)line, 0);
            if (MODE == 5) acc += *(const unsigned*)(tab + line + four);
            if (MODE == 6) { uint4 v = make_uint4(0, 0, 0, 0); if (lane < 16) v = *(const uint4*)(tab + line + x4); acc += v.x + v.y + v.z + v.w; }
            if (MODE == 7) { uint2 v = make_uint2(0, 0); if (lane < 32) v = *(const uint2*)(tab + line + x2); acc += v.x + v.y; }
            if (MODE == 8) acc += tab[line + rnd];
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
    const unsigned mask = (2u << 20) - 1;   // 2 MiB of line starts: L2 resident
    const unsigned segstride = 4160;        // the tile layout's row of tiles at 256^3 plus margin: (32 + 2) * 64 ... a few KB
    auto run = [&](auto kern, const char* name) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, tab, mask, iters, segstride, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%-72s %7.3f ms  %6.1f CU-cycles per wave-instruction (2.4 GHz)\n", name, best,
               best * 1e-3 * 2.4e9 * 256 / wave_instr);
    };
    run(k<0>, "0 global dword, 256 B contiguous, aligned");
    run(k<1>, "1 global dword, 2 x 128 B segments, linear");
    run(k<2>, "2 global dword, 2 x 128 B segments, permuted in the segment");
    run(k<3>, "3 buffer dword offen+soffset, 2 x 128 B, linear");
    run(k<4>, "4 buffer dword offen+soffset, 2 x 128 B, permuted");
    run(k<5>, "5 global dword, 4 x 64 B lines apart, 16 lanes each");
    run(k<6>, "6 global dwordx4 by 16 lanes (4 lines)");
    run(k<7>, "7 global dwordx2 by 32 lanes (4 lines)");
    run(k<8>, "8 byte gather over the 4 lines");
    return 0;
}
