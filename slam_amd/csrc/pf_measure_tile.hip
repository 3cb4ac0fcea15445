// K1 measure (S3P:859-890) on the shared grid's TILE layout: one texture instruction per beam and WAVE.
//
// k_measure_col (pf_kernels.hip) gathers one byte per lane and beam; the texture unit prices such a
// wave-wide byte gather at ~17 CU-cycles wherever the bytes lie (profiles/r2_pmc_k1_col.txt: 17.6 TA-cycles
// per gather, a floor of 27 % of the contract's roofline for one gather per evaluation), while a dword load
// whose 64 lanes read 256 contiguous bytes costs 5.7 and handing every lane its byte through the LDS
// crossbar 2.8 more (tools/ubench_gather5.hip). The 64 lanes of a wave are neighbouring hypotheses
// (pfk_order): for one beam they land within a few cells of each other in X and Y, and (this file's lane
// order, PfOrderParams::zsplit) they share ONE value of round(z/res), so the beam's Z is the same cell layer
// for the whole wave. The grid copy K1 reads is therefore kept z-outermost in 8 x 8 tiles of one layer
// (PF_LAYOUT_TILES: a 64-B line = one tile, X-neighbour tiles contiguous), and per column group:
//   * x, y in fp32 with the exactness filter of k_measure_col (E <= fract <= 1-E on both axes proves
//     floor(x + 0.5) is the reference's round(); a group with a lane outside the band takes X, Y of EVERY
//     lane from the reference's fp64 sequence S3P:745-747, 880 instead, once per group);
//   * a 16 x 16-cell window = 2 x 2 tiles is anchored at the tile of (the anchor lane's cell - 4); when
//     every lane's cell is inside the window and inside the grid, each beam is ONE buffer_load_dword (lane l
//     reads a fixed dword of the window: vector offset = per-lane constant + the window's base, scalar
//     offset = the beam's z layer straight from the group's scalar load), one ds_bpermute_b32 from the lane
//     that holds my cell's dword (the same lane for every beam of the group) and one v_dot4 against a
//     one-hot byte mask that extracts and accumulates: no scalar-ALU work per beam at all;
//   * a group whose lanes spread beyond one window but stay inside the grid is a byte gather per lane and
//     beam on the same layout; anything else (a lane off the grid on any axis, a wave whose lanes differ in
//     round(z/res) or sit on a z rounding tie) goes beam by beam through the reference's flat index S3P:85
//     and its flat-only bounds test S3P:882 (the cell the flat index names, aliasing included).
// What bounds it: instruction issue, the scalar unit first. The first version of this file did the window
// arithmetic on the scalar ALU (25 scalar and 17 vector instructions per wave-beam, 16.0 M + 10.7 M per
// launch at configs[1]: 34.8 us, and the same 37 us with the loads and the bpermutes compiled out,
// profiles/r3_a_*), so everything per group that can run on the vector ALU does, on wave-uniform values.
// Integer sums: bit-exact whatever the path. Compiled with -ffp-contract=off (Makefile).
#include "pf_kernels.h"
#include "pf_device.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#ifndef TILE_EXP
#define TILE_EXP 0      // development: 1 no window loads, 2 no ds_bpermute, 3 neither, 4 prologue and epilogue only, 5 no work per beam, 6 nothing at all
                        // (wrong results, timing only)
#endif
#ifndef TILE_T
#define TILE_T 2        // column groups per trip for groups of up to four beams (their loads are in flight together). With the
#endif                  // chunk-major list (3, 2): 23.4 us at configs[1], (3, 1): 22.8, (2, 1): 22.4, (4, 2): 28.6; the kernel sits
#ifndef TILE_T8         // at its register limits (64 vector, 78 scalar at eight waves per SIMD) and every wider trip spills more
#define TILE_T8 1       // ... for groups of five to eight beams
#endif
namespace {

typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int t_cvt_floor_i32(float x) {       // (int)floorf(x) in one instruction
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ int t_mad_i24(int a, int b, int c) {  // a*b + c, |a|, |b| < 2^23; the sum wraps mod 2^32
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
// round(fl64(lz + tz)) == lz + round(tz) unless tz is within the rounding error of the sum of a half-integer
// (pf_kernels.hip, z_is_ambiguous: |lz + tz| < 2^25, error <= 2^-28, window 2^-20)
__device__ __forceinline__ bool t_z_is_ambiguous(double tz) {
    const double frac = fabs(tz - trunc(tz));
    return !(fabs(frac - 0.5) > 9.5367431640625e-07);
}
// every lane is active in this kernel (holes included), so "all lanes" is "no lane's ballot bit for the negation"
__device__ __forceinline__ bool t_all(bool pred) { return __builtin_amdgcn_ballot_w64(!pred) == 0ull; }

struct TileLane {
    v2f_t rotA, rotB, trh;       // (cos, sin), (-sin, cos), (x/res + 0.5, y/res + 0.5) in fp32
    float thr;                   // 0.5 - E: a lane whose x + 0.5 and y + 0.5 lie at most this far from a cell's middle is exact
    int lanez;                   // byte offset of this lane's dword of a window at tile (0, 0) of the wave's lowest layer
    int rd;                      // where this lane's fp64 pose rows are (sorted[k * lanes + rd]), for the rare paths
};
struct TileGrid {
    unsigned gx, gy, gz, a, b;
    int b8;                      // b / 8: bytes per cell step of the window's base in X
    int zoff;                    // (zp0 + lzmin) * a + the layout's origin: what lanez holds besides the lane's own part
};

// The reference's sequence for one beam and every lane (S3P:745-747 done: X, Y are exact; S3P:880 for z,
// S3P:85, S3P:882) on the reference-order copy of the grid, where the flat index is the address (a cell off an
// axis aliases into the neighbouring row exactly as in the reference).
__device__ __forceinline__ void tile_beam_generic(int X, int Y, int lz, double tz, const PfGridParams& gp,
                                                  const uint8_t* __restrict__ lin, int& sum, int& cnt) {
    const double Zd = round((double)lz + tz);
    const double fd = fma((double)X, gp.gyz, fma((double)Y, gp.gzd, Zd));      // exact (integers < 2^53)
    const bool ok = (fd >= 0.0) & (fd < gp.elements);                        // S3P:882
    const unsigned vv = lin[ok ? (unsigned)fd : 0u];
    sum += ok ? (int)vv : 0;                                                 // S3P:887-888
    cnt += ok ? 1 : 0;
}

// One trip: T groups of N beams each. al: the anchor lane (scalar); coop_ok: every lane of the wave has the
// same round(z/res), none sits on a z rounding tie, and every beam's z layer is inside the grid for it.
template <int N, int T, bool STATS>
__device__ __forceinline__ void tile_trip(const PfTileG* __restrict__ cg, const PfColG* __restrict__ cgz, int g, const TileLane& L,
                                          const TileGrid& G, const PfGridParams& gp,
                                          const uint8_t* __restrict__ lin, __amdgpu_buffer_rsrc_t rsrc,
                                          const double* __restrict__ sorted, int lanes,
                                          int al, bool coop_ok, int& sum, int& cnt, unsigned (&st)[4],
                                          unsigned long long* __restrict__ stats) {
    PfTileG col[T];
#pragma unroll
    for (int q = 0; q < T; ++q) col[q] = cg[g + q];              // wave-uniform: scalar loads
    int X[T], Y[T];
    unsigned src4[T], onehot[T];
    int voff[T];
    bool coop[T];
    bool all_coop = coop_ok;
#pragma unroll
    for (int q = 0; q < T; ++q) {
        const v2f_t lxx = {col[q].lx, col[q].lx}, lyy = {col[q].ly, col[q].ly};
        const v2f_t h = __builtin_elementwise_fma(L.rotA, lxx, __builtin_elementwise_fma(L.rotB, lyy, L.trh));
        X[q] = t_cvt_floor_i32(h.x); Y[q] = t_cvt_floor_i32(h.y);
        // E <= fract <= 1 - E on both axes, as |fract - 0.5| <= thr (the subtraction rounds by at most 2^-26, inside thr's margin)
        const v2f_t fc = v2f_t{__builtin_amdgcn_fractf(h.x), __builtin_amdgcn_fractf(h.y)} - v2f_t{0.5f, 0.5f};
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(__builtin_fmaxf(__builtin_fabsf(fc.x), __builtin_fabsf(fc.y)) <= L.thr)) != 0ull, 0)) {
            if (STATS) st[2] += N;
            // a lane is near a rounding boundary: the reference's fp64 sequence for every lane (S3P:745-747, 880)
            const double c = sorted[L.rd], s = sorted[(size_t)lanes + L.rd];
            const double tx = sorted[2 * (size_t)lanes + L.rd], ty = sorted[3 * (size_t)lanes + L.rd];
            const double ms = -s;
            const double dlx = (double)col[q].lx, dly = (double)col[q].ly;   // exact: integers below 2^24
            X[q] = (int)round((c * dlx + ms * dly) + tx);
            Y[q] = (int)round((s * dlx + c * dly) + ty);
        }
        // the window: 2 x 2 tiles from the tile of (the anchor lane's cell - 4); vector ALU on wave-uniform values
        const int Xa = __builtin_amdgcn_readlane(X[q], al), Ya = __builtin_amdgcn_readlane(Y[q], al);
        int tX, tY;
        asm("v_add_u32 %0, -4, %1" : "=v"(tX) : "s"(Xa));
        asm("v_add_u32 %0, -4, %1" : "=v"(tY) : "s"(Ya));
        tX &= ~7; tY &= ~7;
        const unsigned xr = (unsigned)(X[q] - tX), yr = (unsigned)(Y[q] - tY);
        // inside the window, and inside the grid on x and y (z: coop_ok): the window loads serve every lane
        // (one ballot per comparison: each is the comparison's own lane mask)
        coop[q] = (__builtin_amdgcn_ballot_w64((xr | yr) >= 16u) | __builtin_amdgcn_ballot_w64((unsigned)X[q] >= G.gx) |
                   __builtin_amdgcn_ballot_w64((unsigned)Y[q] >= G.gy)) == 0ull;
        all_coop = all_coop && coop[q];
        // lane 32 (yr >> 3) + 2 xr + ((yr >> 2) & 1) loads my cell's dword; my byte in it: yr & 3
        src4[q] = (((yr & 8u) << 4) | (xr << 3)) | (yr & 4u);
        onehot[q] = 1u << ((yr & 3u) << 3);
        voff[q] = t_mad_i24(tY, G.b8, L.lanez) + (tX << 3);
        if (STATS && (T == 1 || !(coop_ok && coop[q]))) {      // (a trip that falls back is counted by its T = 1 calls)
            if (T == 1) { if (coop_ok && coop[q]) st[0] += N; else st[1] += N; }
        }
        if (STATS && T == 1) {
            int x0 = X[q], x1 = X[q], y0 = Y[q], y1 = Y[q];
            for (int o = 32; o > 0; o >>= 1) {
                x0 = min(x0, __shfl_xor(x0, o, 64)); x1 = max(x1, __shfl_xor(x1, o, 64));
                y0 = min(y0, __shfl_xor(y0, o, 64)); y1 = max(y1, __shfl_xor(y1, o, 64));
            }
            if ((threadIdx.x & 63) == 0) atomicAdd(&stats[8 + min(max(x1 - x0, y1 - y0), 23)], (unsigned long long)N);
        }
    }
#if TILE_EXP == 5
    if (all_coop) {
#pragma unroll
        for (int q = 0; q < T; ++q) sum += (int)(src4[q] ^ onehot[q]) + voff[q] + (int)col[q].lza[N - 1];
        cnt += T * N;
        return;
    }
#endif
    if (__builtin_expect(all_coop, 1)) {
        unsigned v[T][N];
#pragma unroll
        for (int q = 0; q < T; ++q)
#pragma unroll
            for (int k = 0; k < N; ++k) {
#if TILE_EXP == 1 || TILE_EXP == 3
                v[q][k] = (unsigned)voff[q] + col[q].lza[k];
#else
                // buffer_load_dword v, voff, rsrc, lza offen: no address arithmetic per beam
                v[q][k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff[q], (int)col[q].lza[k], 0);
#endif
            }
#pragma unroll
        for (int q = 0; q < T; ++q)
#pragma unroll
            for (int k = 0; k < N; ++k) {
#if TILE_EXP == 2 || TILE_EXP == 3
                const unsigned d = v[q][k] ^ src4[q];
#else
                const unsigned d = (unsigned)__builtin_amdgcn_ds_bpermute((int)src4[q], (int)v[q][k]);
#endif
                sum = (int)__builtin_amdgcn_udot4(d, onehot[q], (unsigned)sum, false);        // S3P:887
            }
        cnt += T * N;                                                                        // S3P:888
        if (STATS && T > 1) st[0] += T * N;
        return;
    }
    if (T > 1) {
        // some group of the trip cannot take the window loads: one group at a time, from scratch (short live ranges
        // in the common path matter more than the recomputation here)
#pragma unroll
        for (int q = 0; q < T; ++q)
            tile_trip<N, 1, STATS>(cg, cgz, g + q, L, G, gp, lin, rsrc, sorted, lanes, al, coop_ok, sum, cnt, st, stats);
        return;
    }
    if (coop_ok && t_all(((unsigned)X[0] < G.gx) & ((unsigned)Y[0] < G.gy))) {
        // the lanes spread beyond one window: a byte gather per lane (the z layer is still the wave's)
        const int cb = (int)(((unsigned)Y[0] >> 3) * G.b + ((((unsigned)X[0] >> 3) << 6) | (((unsigned)X[0] & 7u) << 3) | ((unsigned)Y[0] & 7u))) + G.zoff;
        unsigned v[N];
#pragma unroll
        for (int k = 0; k < N; ++k)
            v[k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rsrc, cb, (int)col[0].lza[k], 0);
#pragma unroll
        for (int k = 0; k < N; ++k) sum += (int)v[k];
        cnt += N;
    } else {
        const double tz = sorted[4 * (size_t)lanes + L.rd];
#pragma unroll
        for (int k = 0; k < N; ++k) tile_beam_generic(X[0], Y[0], cgz[2 * g + (k >> 2)].lz[k & 3], tz, gp, lin, sum, cnt);
    }
}

// One phase: the chunk's groups [g0, g1) all have N beams and go T = tile_t(N) to a trip; what is left at the end
// (the short last trip of a count, which belongs to one chunk) is walked group by group.
__host__ __device__ constexpr int tile_t(int N) { return N > 4 ? TILE_T8 : TILE_T; }
template <int N, bool STATS>
__device__ __forceinline__ void tile_phase(const PfTileG* __restrict__ cg, const PfColG* __restrict__ cgz, int g0, int g1,
                                           const TileLane& L, const TileGrid& G, const PfGridParams& gp,
                                           const uint8_t* __restrict__ lin, __amdgpu_buffer_rsrc_t rsrc,
                                           const double* __restrict__ sorted, int lanes,
                                           int al, bool coop_ok, int& sum, int& cnt, unsigned (&st)[4],
                                           unsigned long long* __restrict__ stats) {
    constexpr int T = tile_t(N);
    int g = g0;
    for (; g + T <= g1; g += T) tile_trip<N, T, STATS>(cg, cgz, g, L, G, gp, lin, rsrc, sorted, lanes, al, coop_ok, sum, cnt, st, stats);
    if (T > 1)
        for (; g < g1; ++g) tile_trip<N, 1, STATS>(cg, cgz, g, L, G, gp, lin, rsrc, sorted, lanes, al, coop_ok, sum, cnt, st, stats);
}

// STATS (SLAM_PF_TILE_STATS=1, development aid): stats[0..3] += wave-beams by window load / the other ways / of groups
// re-done in fp64; stats[4] += waves without coop_ok; stats[8..32): wave-beams by the wave's spread in cells
#ifndef TILE_WG
#define TILE_WG 256
#endif
template <bool STATS>
__global__ __launch_bounds__(TILE_WG) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_measure_tile(
        const double* __restrict__ sorted, int lanes, const int32_t* __restrict__ order,
        const uint8_t* __restrict__ shadow, const uint8_t* __restrict__ lin, PfSwz sw, const PfTileG* __restrict__ tab, const PfColG* __restrict__ cgz, int lzmin, int lzmax,
        PfGridParams gp, double lmax, int2* __restrict__ partial, unsigned long long* __restrict__ stats) {
#if TILE_EXP == 6
    return;                                          // (an empty launch of the same grid)
#endif
    __builtin_amdgcn_s_setprio(3);
    // this chunk's record of the table in front of the list (wave-uniform: scalar loads): the ends of its groups with
    // 8, 7, ..., 1 beams and its start. The list is chunk-major (pf_set_beams), so a wave walks contiguous memory and
    // divides nothing (dealing the trips to the chunks in the kernel cost sixteen scalar divisions per wave)
    const int32_t* __restrict__ row = (const int32_t*)(tab + blockIdx.y);
    const int4 ends = make_int4(row[0], row[1], row[2], row[3]), ends2 = make_int4(row[4], row[5], row[6], row[7]);
    const int gstart = row[PF_TILE_GROUP];
    const PfTileG* __restrict__ cg = tab + gridDim.y;
    const int lane_id = blockIdx.x * TILE_WG + threadIdx.x;
    // a hole of the lane order (order[] = -1: the padding that starts every z class on a wave boundary)
    // stands in for the wave's first real lane, so that all 64 lanes take part in the window loads; a wave
    // of holes only has nothing to do
    const int who = lane_id < lanes ? order[lane_id] : -1;
    TileLane L;
    L.rd = min(lane_id, lanes - 1);
    // the rows are loaded before order[] is known (one memory round trip for the prologue, not two); only a wave with
    // holes loads them again, from its first real lane
    double c = sorted[L.rd], s = sorted[(size_t)lanes + L.rd];
    double tx = sorted[2 * (size_t)lanes + L.rd], ty = sorted[3 * (size_t)lanes + L.rd];
    double tz = sorted[4 * (size_t)lanes + L.rd];
    const unsigned long long vm = __ballot(who >= 0);
    if (vm == 0ull) return;
    if (vm != ~0ull) {
        L.rd = who >= 0 ? lane_id : (lane_id & ~63) + (__ffsll((long long)vm) - 1);
        c = sorted[L.rd]; s = sorted[(size_t)lanes + L.rd];
        tx = sorted[2 * (size_t)lanes + L.rd]; ty = sorted[3 * (size_t)lanes + L.rd];
        tz = sorted[4 * (size_t)lanes + L.rd];
    }
    const float cf = (float)c, sf = (float)s;
    L.rotA = v2f_t{cf, sf}; L.rotB = v2f_t{-sf, cf};
    L.trh = v2f_t{(float)(tx + 0.5), (float)(ty + 0.5)};
    const int zp = (int)round(tz);                                // exact: |tz| < 2^24
    const double tmax = fmax(fmax(fabs(tx), fabs(ty)), fabs(tz));
    L.thr = (float)(0.5 - (2.5e-7 * (lmax + tmax + 1.0) + 1.0e-7) - 2.0e-7);   // k_measure_col's bound E, and the margin above
    const unsigned l = threadIdx.x & 63u;
    const int zp0 = __builtin_amdgcn_readfirstlane(zp);
    // one z layer per beam for the whole wave, and every layer of the list inside the grid for it
    const bool coop_ok = t_all(zp == zp0 && !t_z_is_ambiguous(tz)) && zp0 + lzmin >= 0 && (unsigned)(zp0 + lzmax) < sw.gz;
    TileGrid G{sw.gx, sw.gy, sw.gz, sw.a, sw.b, (int)(sw.b / 8u), 0};
    G.zoff = (int)((unsigned)(zp0 + lzmin) * sw.a + tile_addr(0, 0, 0, sw.a, sw.b));   // (used with coop_ok only)
    // lane l loads dword l & 31 of the 128 contiguous bytes of the window's tile row l >> 5 (two x-neighbour tiles):
    // the texture path takes a linear 128-B run at half the cycles of a permuted one (tools/ubench_gather6.hip: 7.8-9.0
    // CU-cycles per wave-instruction against 17)
    L.lanez = (int)((l >> 5) * sw.b + (l & 31u) * 4u) + G.zoff;
    // the anchor lane: the middle one of the wave in the lane order's curve (TILE_ANCHOR_MEAN: the lane nearest the
    // wave's mean pose instead, four wave reductions: no more groups take the window loads that way)
#ifdef TILE_ANCHOR_MEAN
    int al;
    {
        float mx = (float)tx, my = (float)ty, mq = sf;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mx += __shfl_xor(mx, o, 64); my += __shfl_xor(my, o, 64); mq += __shfl_xor(mq, o, 64); }
        const float d = fabsf((float)tx - mx * (1.0f / 64)) + fabsf((float)ty - my * (1.0f / 64)) +
                        0.5f * (float)lmax * fabsf(sf - mq * (1.0f / 64));
        unsigned key = (min((unsigned)(d * 64.0f), 0x3ffffffu) << 6) | l;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) key = min(key, (unsigned)__shfl_xor((int)key, o, 64));
        al = __builtin_amdgcn_readfirstlane((int)(key & 63u));
    }
#else
    const int al = 31;
#endif
    // raw buffer over the whole copy (stride 0)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)shadow, 0, (int)(sw.gz * sw.a), 0x00020000);
    int sum = 0, cnt = 0;
    unsigned st[4] = {0, 0, 0, 0};
    // priority follows progress (k_measure_col: the issue arbiter favours the oldest wave, which staggers the
    // finish times of a launch that is one residency deep)
#if TILE_EXP == 4
#define TILE_PHASE(N, G0, G1) sum += (G1) - (G0)
#else
#define TILE_PHASE(N, G0, G1) tile_phase<N, STATS>(cg, cgz, G0, G1, L, G, gp, lin, rsrc, sorted, lanes, al, coop_ok, sum, cnt, st, stats)
#endif
    TILE_PHASE(8, gstart, ends.x);
    TILE_PHASE(7, ends.x, ends.y);
    TILE_PHASE(6, ends.y, ends.z);
    TILE_PHASE(5, ends.z, ends.w);
    __builtin_amdgcn_s_setprio(2);
    TILE_PHASE(4, ends.w, ends2.x);
    TILE_PHASE(3, ends2.x, ends2.y);
    __builtin_amdgcn_s_setprio(1);
    TILE_PHASE(2, ends2.y, ends2.z);
    __builtin_amdgcn_s_setprio(0);
    TILE_PHASE(1, ends2.z, ends2.w);
#undef TILE_PHASE
    // in lane order (coalesced); pfk_reduce_scores_l sends the sums to the particles through order[]
    if (who >= 0) partial[(size_t)blockIdx.y * lanes + lane_id] = make_int2(sum, cnt);
    if (STATS && (threadIdx.x & 63) == 0) {
        for (int k = 0; k < 4; ++k) atomicAdd(&stats[k], (unsigned long long)st[k]);
        if (!coop_ok && blockIdx.y == 0) atomicAdd(&stats[4], 1ull);
    }
}

}  // namespace

int pfk_tile_trip(int n_beams) { return tile_t(n_beams); }
static_assert(PF_TILE_GROUP == 8, "k_measure_tile has eight phases");

hipError_t pfk_measure_tile(hipStream_t st, const double* sorted, int lanes, const int32_t* order,
                            const uint8_t* shadow, const uint8_t* lin, PfSwz sw, const PfTileG* cg, const PfColG* cgz,
                            int lzmin, int lzmax, int n_chunks, PfGridParams gp, double lmax, int2* partial) {
    static const bool want_stats = getenv("SLAM_PF_TILE_STATS") != nullptr;
    const dim3 grid((lanes + TILE_WG - 1) / TILE_WG, n_chunks);
    if (want_stats) {
        unsigned long long* stats = nullptr;
        if (hipMalloc((void**)&stats, 256) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemsetAsync(stats, 0, 256, st);
        hipLaunchKernelGGL(k_measure_tile<true>, grid, dim3(TILE_WG), 0, st, sorted, lanes, order, shadow, lin, sw, cg, cgz, lzmin,
                           lzmax, gp, lmax, partial, stats);
        unsigned long long hs[32];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(hs, stats, 256, hipMemcpyDeviceToHost);
        (void)hipFree(stats);
        fprintf(stderr, "k_measure_tile: wave-beams by window load %llu, the other ways %llu, in groups re-done in fp64 %llu; "
                        "waves without one z layer %llu of %d\n", hs[0], hs[1], hs[2], hs[4], (lanes + 63) / 64);
        fprintf(stderr, "  wave-beams by the wave's spread in cells (max of x, y; last = 23 and more):");
        for (int k = 8; k < 32; ++k) fprintf(stderr, " %llu", hs[k]);
        fprintf(stderr, "\n");
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_measure_tile<false>, grid, dim3(TILE_WG), 0, st, sorted, lanes, order, shadow, lin, sw, cg, cgz, lzmin,
                       lzmax, gp, lmax, partial, (unsigned long long*)nullptr);
    return hipGetLastError();
}
