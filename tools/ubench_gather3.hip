// Micro-benchmark 3: replay K1's real gather address vectors (tools/k1_addr.bin, uint32
// [156][512][64]) as bare byte gathers, 4 in flight per wave, to price the memory side of K1 alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int W = 156, B = 512;

// mode 0: real addresses; 1: every lane uses lane 0's address; 2: addresses & ~63 | lane (one line per distinct line, bytes spread);
// 3: real addresses, dword loads
template <int VALU>
__global__ __launch_bounds__(256) void k(const unsigned char* __restrict__ tab, const unsigned* __restrict__ vec,
                                         int mode, int reps, unsigned* __restrict__ out) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const unsigned* my = vec + (size_t)(wave % W) * B * 64 + lane;
    const int j0 = (wave / W) * 37 % (B - 16);
    unsigned a[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        a[u] = my[(size_t)(j0 + u) * 64];
        if (mode == 1) a[u] = __shfl(a[u], 0, 64);
        if (mode == 2) a[u] = (a[u] & ~63u) | lane;
    }
    unsigned acc = 0; float fa = lane * 1e-3f;
    unsigned sh = wave * 2654435761u;
    for (int r = 0; r < reps; ++r) {
        sh = sh * 1664525u + 1013904223u;
        const unsigned shift = (sh >> 8) & 0xffffc0u;     // wave-uniform, line-aligned translation
#pragma unroll
        for (int g = 0; g < 16; g += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int v = 0; v < VALU; ++v) fa = __builtin_fmaf(fa, 1.0001f, 0.5f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned ad = (a[g + u] + shift) & 0xffffffu;
                acc += mode == 3 ? *(const unsigned*)(tab + (ad & ~3u)) : tab[ad];
            }
        }
    }
    if (acc == 0xdeadbeef || fa == 123.f) out[wave] = acc;
}

int main() {
    std::vector<unsigned> h((size_t)W * B * 64);
    FILE* f = fopen("tools/k1_addr.bin", "rb");
    if (!f || fread(h.data(), 4, h.size(), f) != h.size()) { printf("no tools/k1_addr.bin\n"); return 1; }
    fclose(f);
    unsigned char* tab; unsigned* vec; unsigned* out;
    CK(hipMalloc(&tab, 16u << 20)); CK(hipMemset(tab, 1, 16u << 20));
    CK(hipMalloc(&vec, h.size() * 4)); CK(hipMemcpy(vec, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 1 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 2028 / 4 * 4 / 4 * 4;   // 156*13 waves = 2028 waves... use 13 replicas of the 156 waves
    const int nblocks = W * 13 * 4 / 4;         // W*13 waves*... each block 4 waves
    (void)blocks;
    const int reps = 64;
    for (int mode = 0; mode < 4; ++mode)
        for (int valu : {0, 20}) {
            float best = 1e9f;
            const int nb = W * 52 / 4;          // 52 wave-replicas per address set: 8112 waves = 32 per CU
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                if (valu) hipLaunchKernelGGL(k<20>, dim3(nb), dim3(256), 0, 0, tab, vec, mode, reps, out);
                else hipLaunchKernelGGL(k<0>, dim3(nb), dim3(256), 0, 0, tab, vec, mode, reps, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const double wi = (double)nb * 4 * 16 * reps;
            printf("mode %d valu %2d : %7.3f ms  %6.1f CU-cycles per wave-gather\n", mode, valu, best,
                   best * 1e-3 * 2.4e9 * 256 / wi);
        }
    (void)nblocks;
    return 0;
}
