// K1 + K2 on per-particle maps, column-wise: S3P:859-908 with add_points on.
//
// The beams of a step are the non-zero cells of the local map (S3P:868-877) in x, y, z order, so
// they come as columns: runs that share (lx, ly) and ascend in lz (the reference's own local maps
// are dense: ~35 cells per column, 94 % filled, most of them the free cells its ray casts leave).
// The particle transform is a yaw rotation plus a translation, so a local column lands on ONE
// column (X, Y) of the particle's map, and since z is the fastest axis of the map (S3P:84-86) on
// one contiguous run of bytes. k_measure_fuse_wave touches those bytes one beam at a time (a
// 4-byte load and a 1-byte store per beam, an LDS election per beam); here the run is read as
// 16-byte pieces, a lane per piece, every beam is applied to the bytes in registers (two cells per
// packed 16-bit instruction, the occupied cells' sum as packed 16-bit multiply-adds), and the
// pieces are written back: one load and one store per 16 cells instead of 16 + 16, and the five
// lanes of a column (and the columns next to it in y) share their cache lines.
//
// Order. The reference's sequential semantics (a beam reads its cell after every earlier beam's
// add, S3P:886-904) only matter between beams that reach the same cell. With every column of the
// particle inside the map on all three axes (checked first; otherwise the flat-only bounds test
// S3P:882 aliases cells into neighbouring rows and the particle is walked beam by beam in list
// order, exactly as k_measure_fuse_wave does), two columns can share cells only if they round to
// the same (X, Y), which takes |dx| < 1 and |dy| < 1 after the rotation: local columns at distance
// 0 or 1, i.e. the same (lx, ly) or 4-neighbours: a few per cent of the columns at most yaws.
// One wave walks the columns of one particle in list order as a ROLLING batch: every round each
// free lane takes the next column of the list (x, y, the key and the addresses are computed a lane
// per column), all lanes that hold a column enter an LDS election keyed by (X, Y) (the table
// k_measure_fuse_wave uses per cell; the lowest column index wins), the winners' runs are
// read-modify-written in parallel and their lanes become free, the few losers stay for the next
// round, where they meet the newly taken columns (all later in the list) and win against them.
// So every cell sees its columns in list order, and a collision costs its loser one round of
// waiting instead of costing the whole wave a round.
// A round's stores are acknowledged before the next round's loads. Map rows are 16-byte aligned
// (gz % 16 == 0 is required), so columns with different (X, Y) never share a 16-byte piece.
//
// Values. Each column's int8 values lie in 16-byte units aligned to the LOCAL z grid (unit u of a
// column covers lz in [floor16(z0) + 16 u, +16)), cells without a beam hold 0, a zero unit lies in
// front of and behind every column. The map piece q of a column then needs the 16 value bytes that
// start 16 (q + carry) - r bytes into the column's units, with r = (z shift of the particle) mod 16
// the same for the whole wave and carry = 0/1 per column: dword loads plus v_alignbyte with a
// wave-uniform shift.
#include "pf_kernels.h"
#include "pf_device.h"
#include <stdio.h>
#include <stdlib.h>

#define FC_TBL 256
#define FC_EMPTY 0xffffffffu
#define FC_LPC 5                              // lanes (16-byte pieces) per column in the piece phase
#define FC_CPP (64 / FC_LPC)                  // columns per pass
#define FC_PASSES 5
#define FC_DEPTH 2                            // passes whose loads are in flight together
#define FC_SLOTS (FC_CPP * FC_PASSES)         // columns a wave holds at a time (one per lane 0 .. FC_SLOTS-1)
static_assert(PF_FCOL_MAXZ + 15 <= 16 * FC_LPC, "a column spans at most FC_LPC 16-byte pieces");
static_assert(FC_SLOTS <= 64, "one column slot per lane");

namespace {

typedef short s2 __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef unsigned u4a __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ bool zamb(double tz) {
    const double frac = fabs(tz - trunc(tz));
    return !(fabs(frac - 0.5) > 9.5367431640625e-07);
}

// beams [j0, j1) of one particle, beam by beam, in the reference's order (the election of
// k_measure_fuse_wave per cell): the route of particles the column form cannot take.
template <bool FUSE>
__device__ __forceinline__ void fuse_beam_range(int j0, int j1, const PfBeam* __restrict__ beams, double c, double s,
                                             double tx, double ty, double tz, uint8_t* map, const PfGridParams& gp,
                                             unsigned* keys, unsigned* minl, int lane, int& sum, int& cnt) {
    const double ms = -s;
    for (int jb = j0; jb < j1; jb += 64) {
        const int j = jb + lane;
        bool pending = false;
        unsigned f = 0, slot = 0;
        int v = 0;
        if (j < j1) {
            const PfBeam b = beams[j];
            const double X = round((c * b.lx + ms * b.ly) + tx);   // S3P:745-747, 880, 85
            const double Y = round((s * b.lx + c * b.ly) + ty);
            const double Z = round(b.lz + tz);
            const double fd = fma(X, gp.gyz, fma(Y, gp.gzd, Z));
            pending = fd >= 0.0 && fd < gp.elements;               // S3P:882
            f = pending ? (unsigned)(int)fd : 0u;
            v = b.v;
        }
        while (__any(pending)) {
            if (pending) {
                unsigned sl = (f * 2654435761u) >> 24;
                while (true) {
                    const unsigned old = atomicCAS(&keys[sl], FC_EMPTY, f);
                    if (old == FC_EMPTY || old == f) break;
                    sl = (sl + 1u) & (FC_TBL - 1u);
                }
                atomicMin(&minl[sl], (unsigned)lane);
                slot = sl;
            }
            __builtin_amdgcn_wave_barrier();
            const bool winner = pending && __hip_atomic_load(&minl[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) == (unsigned)lane;
            __builtin_amdgcn_wave_barrier();
            if (pending) {
                __hip_atomic_store(&keys[slot], FC_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_store(&minl[slot], 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            __builtin_amdgcn_wave_barrier();
            if (winner) {
                const unsigned w = __hip_atomic_load((const unsigned*)(map + (f & ~3u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int cell = (int)((w >> (8u * (f & 3u))) & 0xffu);
                if (v > gp.occ_threshold) { sum += cell; cnt += 1; }   // S3P:886-889
                int nv = cell + v;                                      // S3P:895-904
                nv = nv < gp.lo ? gp.lo : (nv > gp.hi ? gp.hi : nv);
                if (FUSE && nv != cell) map[f] = (uint8_t)nv;
                pending = false;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
}

// four map cells and the four beam values that meet them (0: no beam on this cell), S3P:886-904 on
// 16-bit pairs: cells 0, 2 and cells 1, 3. acc gathers cell * [value > occ_threshold] per 16-bit half
// (at most 2 * 4 * 255 per 16-byte piece, read out by the caller before it can overflow).
__device__ __forceinline__ unsigned fuse_dword(unsigned cells, unsigned vals, s2 lo2, s2 hi2, s2 thr2, us2& acc) {
    const s2 c02 = __builtin_bit_cast(s2, cells & 0x00ff00ffu), c13 = __builtin_bit_cast(s2, (cells >> 8) & 0x00ff00ffu);
    const s2 v02 = __builtin_bit_cast(s2, __builtin_bit_cast(us2, vals) << (us2)(8)) >> (s2)(8);
    const s2 v13 = __builtin_bit_cast(s2, vals) >> (s2)(8);
    // S3P:886-889: [v > threshold] = clamp(v - threshold, 0, 1); the local map holds no zero beams, a cell
    // without a beam (value 0) must not count whatever the threshold
    const s2 zero = (s2)((short)0);
    const s2 z02 = __builtin_bit_cast(s2, __builtin_elementwise_min(__builtin_bit_cast(us2, v02), (us2)((unsigned short)1)));
    const s2 z13 = __builtin_bit_cast(s2, __builtin_elementwise_min(__builtin_bit_cast(us2, v13), (us2)((unsigned short)1)));
    const s2 o02 = __builtin_elementwise_min(__builtin_elementwise_max(v02 - thr2, zero), z02);
    const s2 o13 = __builtin_elementwise_min(__builtin_elementwise_max(v13 - thr2, zero), z13);
    acc += __builtin_bit_cast(us2, c02 * o02) + __builtin_bit_cast(us2, c13 * o13);
    const s2 n02 = __builtin_elementwise_min(__builtin_elementwise_max(c02 + v02, lo2), hi2);  // S3P:895-904
    const s2 n13 = __builtin_elementwise_min(__builtin_elementwise_max(c13 + v13, lo2), hi2);
    // S3P:870: a zero cell of the local map is no beam, the map cell stays as it is (whatever it holds)
    const s2 r02 = c02 + z02 * (n02 - c02), r13 = c13 + z13 * (n13 - c13);
    return __builtin_bit_cast(unsigned, r02) | (__builtin_bit_cast(unsigned, r13) << 8);
}

// FUSE = false: the same walk without the add (S3P:892 add_points == 0): the list then holds the occupied
// beams only, nothing is written, so no two columns can interfere and there is no election.
template <bool FUSE>
__global__ __launch_bounds__(256) void k_measure_fuse_cols(
        const double* __restrict__ pose, int P, uint8_t* maps, unsigned long long map_stride,
        const PfBeam* __restrict__ beams, int M, const PfFCol* __restrict__ cols, int n_cols,
        const unsigned char* __restrict__ vals, PfFColBounds bd,
        int gx, int gy, int gz, PfGridParams gp, int2* __restrict__ partial, unsigned* stats) {
    __shared__ unsigned keys[4][FC_TBL];
    __shared__ unsigned minl[4][FC_TBL];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int p = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv);
    if (p >= P) return;                                   // wave-uniform; no block barrier below
    for (int k = lane; k < FC_TBL; k += 64) {
        __hip_atomic_store(&keys[wv][k], FC_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_store(&minl[wv][k], 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    __builtin_amdgcn_wave_barrier();
    const double c = pose[4 * (size_t)P + p], s = pose[5 * (size_t)P + p];
    const double tx = pose[6 * (size_t)P + p], ty = pose[7 * (size_t)P + p];
    const double tz = pose[8 * (size_t)P + p];
    const double ms = -s;
    uint8_t* map = maps + map_stride * (unsigned long long)p;
    int sum = 0, cnt = 0;

    // every column inside the map on every axis? z: exact integers once tz is not near a tie
    // (round(lz + tz) = lz + round(tz) then). x, y: first the rotated bounding rectangle with a margin
    // of a cell, then column by column.
    bool inside = !zamb(tz) && fabs(tz) < 1048576.0;
    const int zp = inside ? (int)round(tz) : 0;
    inside = inside && bd.z0 + zp >= 0 && bd.z1 + zp < gz;
    if (inside) {
        const double ax0 = c * (double)bd.lx0, ax1 = c * (double)bd.lx1, bx0 = ms * (double)bd.ly0, bx1 = ms * (double)bd.ly1;
        const double ay0 = s * (double)bd.lx0, ay1 = s * (double)bd.lx1, by0 = c * (double)bd.ly0, by1 = c * (double)bd.ly1;
        const double xlo = fmin(ax0, ax1) + fmin(bx0, bx1) + tx, xhi = fmax(ax0, ax1) + fmax(bx0, bx1) + tx;
        const double ylo = fmin(ay0, ay1) + fmin(by0, by1) + ty, yhi = fmax(ay0, ay1) + fmax(by0, by1) + ty;
        const bool roomy = xlo >= 1.0 && xhi <= (double)(gx - 2) && ylo >= 1.0 && yhi <= (double)(gy - 2);
        if (!roomy) {
            for (int b0 = 0; b0 < n_cols && inside; b0 += 64) {
                const PfFCol col = cols[min(b0 + lane, n_cols - 1)];
                const double Xd = round((c * (double)col.lx + ms * (double)col.ly) + tx);
                const double Yd = round((s * (double)col.lx + c * (double)col.ly) + ty);
                const bool in = Xd >= 0.0 && Xd < (double)gx && Yd >= 0.0 && Yd < (double)gy;
                inside = !__any(!in);
            }
        }
    }
    if (!inside) {
        if (stats && lane == 0) atomicAdd(&stats[0], 1u);
        fuse_beam_range<FUSE>(0, M, beams, c, s, tx, ty, tz, map, gp, keys[wv], minl[wv], lane, sum, cnt);
    } else {
        const unsigned r = (unsigned)zp & 15u;                                 // wave-uniform
        const unsigned bsh = (0u - r) & 3u;                                    // byte shift of the value windows
        const s2 lo2 = (s2)((short)gp.lo), hi2 = (s2)((short)gp.hi);
        const s2 thr2 = (s2)((short)max(-200, min(200, gp.occ_threshold)));     // int8 values: any threshold beyond +-128 acts alike
        int next = 0;                                                           // wave-uniform: first column not yet taken
        bool pending = false;
        unsigned key = 0, seg0 = 0, s0 = 0, myci = 0, slot = 0;
        int nseg = 0;
        // S3P:745-747, 880 for a column (x and y do not depend on lz), its key, addresses and piece count
        auto take = [&](const PfFCol& col, int ci) {
            const int X = (int)round((c * (double)col.lx + ms * (double)col.ly) + tx);
            const int Y = (int)round((s * (double)col.lx + c * (double)col.ly) + ty);
            key = (unsigned)X * (unsigned)gy + (unsigned)Y;
            const unsigned a0 = key * (unsigned)gz + (unsigned)(col.z0 + zp);   // first cell of the run (rows are 16-aligned)
            seg0 = a0 & ~15u;
            nseg = (int)(((a0 & 15u) + (unsigned)col.n + 15u) >> 4);
            const unsigned carry = (((unsigned)col.z0 & 15u) + r) >> 4;
            s0 = 16u * ((unsigned)col.voff + carry) - r;                        // byte offset of piece 0's values
            cnt += col.cnt;                                                     // S3P:888: every beam is in range here
            myci = (unsigned)ci;
            pending = true;
        };
        if (lane < FC_SLOTS && lane < n_cols) take(cols[lane], lane);
        next = min(n_cols, FC_SLOTS);
        while (__any(pending)) {
            bool winner = pending;
            if (FUSE) {
                if (pending) {
                    unsigned sl = (key * 2654435761u) >> 24;
                    while (true) {
                        const unsigned old = atomicCAS(&keys[wv][sl], FC_EMPTY, key);
                        if (old == FC_EMPTY || old == key) break;
                        sl = (sl + 1u) & (FC_TBL - 1u);
                    }
                    atomicMin(&minl[wv][sl], myci);
                    slot = sl;
                }
                __builtin_amdgcn_wave_barrier();
                winner = pending && __hip_atomic_load(&minl[wv][slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) == myci;
                __builtin_amdgcn_wave_barrier();
                if (pending) {
                    __hip_atomic_store(&keys[wv][slot], FC_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __hip_atomic_store(&minl[wv][slot], 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (stats && lane == 0) atomicAdd(&stats[1], 1u);
            // the winners' lanes are free after this round: their next columns of the list, asked for now and
            // set up while this round's stores are on their way
            const unsigned long long freem = __ballot(winner);
            const int nci = next + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(freem >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)freem, 0u));
            const bool more = winner && nci < n_cols;
            PfFCol ncol{};
            if (more) ncol = cols[nci];
            next = min(n_cols, next + (int)__popcll(freem));
            // The winners' runs, a lane per 16-byte piece: pass k gives the FC_LPC lanes FC_LPC j .. of the wave
            // the pieces of the column held by lane FC_CPP k + j, so the lanes of a wave instruction read whole
            // 64-byte lines where the map columns are neighbours (z is the fastest axis, then y; with 80 cells
            // per column, the reference's shape, five pieces are exactly one column). Plain loads: this wave is
            // the only one that touches this particle's map, and a CU's L1 is coherent for accesses made through
            // it once the earlier stores are acknowledged (the wait below).
            const int sub = lane / FC_LPC, qn = lane - sub * FC_LPC;
            const int wn = winner ? nseg : 0;
            // FC_DEPTH passes in flight: pass k + FC_DEPTH is asked for when pass k has been written back
            uint4 mc[FC_DEPTH];
            u4a vq[FC_DEPTH];
            unsigned v5[FC_DEPTH], at[FC_DEPTH];
            bool act[FC_DEPTH];
            auto ask = [&](int k, int b) {
                const int from = FC_CPP * k + sub;
                const int n_k = __shfl(wn, from, 64);
                const unsigned sv = __shfl(s0, from, 64) + 16u * (unsigned)qn;
                at[b] = __shfl(seg0, from, 64) + 16u * (unsigned)qn;
                act[b] = sub < FC_CPP && qn < n_k;
                if (act[b]) {
                    const unsigned* vw = (const unsigned*)(vals + (sv & ~3u));
                    mc[b] = *(const uint4*)(map + at[b]);
                    vq[b] = *(const u4a*)vw; v5[b] = vw[4];
                }
            };
            auto finish = [&](int b) {
                if (act[b]) {
                    const unsigned cw[4] = {mc[b].x, mc[b].y, mc[b].z, mc[b].w};
                    const unsigned vd[5] = {vq[b].x, vq[b].y, vq[b].z, vq[b].w, v5[b]};
                    unsigned o[4];
                    us2 acc = (us2)((unsigned short)0);
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        o[d] = fuse_dword(cw[d], __builtin_amdgcn_alignbyte(vd[d + 1], vd[d], bsh), lo2, hi2, thr2, acc);
                    sum += (int)acc.x + (int)acc.y;
                    uint4 ov; ov.x = o[0]; ov.y = o[1]; ov.z = o[2]; ov.w = o[3];
                    // a piece no beam changed (saturated cells, S3P:896-901) is not written back
                    const bool same = ((o[0] ^ cw[0]) | (o[1] ^ cw[1]) | (o[2] ^ cw[2]) | (o[3] ^ cw[3])) == 0u;
                    if (FUSE && !same) *(uint4*)(map + at[b]) = ov;
                }
            };
#pragma unroll
            for (int k = 0; k < FC_DEPTH; ++k) ask(k, k);
#pragma unroll
            for (int k = 0; k < FC_PASSES; ++k) {
                finish(k % FC_DEPTH);
                if (k + FC_DEPTH < FC_PASSES) ask(k + FC_DEPTH, k % FC_DEPTH);
            }
            if (winner) pending = false;
            if (more) take(ncol, nci);
            // the stores of this round reach L2 before any later load of this wave is issued
            if (FUSE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o, 64); cnt += __shfl_xor(cnt, o, 64); }
    if (lane == 0) partial[p] = make_int2(sum, cnt);
}

}  // namespace

hipError_t pfk_measure_fuse_cols(hipStream_t st, const double* pose, int P, uint8_t* maps, unsigned long long map_stride,
                                 const PfBeam* beams, int M, const PfFCol* cols, int n_cols, const void* vals,
                                 PfFColBounds bd, int gx, int gy, int gz, PfGridParams gp, int2* partial, bool fuse) {
    // SLAM_PF_FCOL_STATS=1 (development, tests): particles walked beam by beam and rounds, on stderr
    const bool want_stats = getenv("SLAM_PF_FCOL_STATS") && atoi(getenv("SLAM_PF_FCOL_STATS")) != 0;
    unsigned* stats = nullptr;
    if (want_stats) {
        if (hipMalloc((void**)&stats, 16) != hipSuccess) return hipGetLastError();
        (void)hipMemsetAsync(stats, 0, 16, st);
    }
    if (fuse)
        hipLaunchKernelGGL(k_measure_fuse_cols<true>, dim3((P + 3) / 4), dim3(256), 0, st, pose, P, maps, map_stride, beams, M,
                           cols, n_cols, (const unsigned char*)vals, bd, gx, gy, gz, gp, partial, stats);
    else
        hipLaunchKernelGGL(k_measure_fuse_cols<false>, dim3((P + 3) / 4), dim3(256), 0, st, pose, P, maps, map_stride, beams, M,
                           cols, n_cols, (const unsigned char*)vals, bd, gx, gy, gz, gp, partial, stats);
    const hipError_t e = hipGetLastError();
    if (want_stats) {
        unsigned hs[4] = {0, 0, 0, 0};
        (void)hipMemcpyAsync(hs, stats, 16, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        (void)hipFree(stats);
        fprintf(stderr, "[fuse_cols] P=%d columns=%d: %u particles beam by beam, %u rounds (%.2f per 64 columns)\n", P,
                n_cols, hs[0], hs[1], (double)hs[1] / (((double)(P - hs[0]) + 1e-9) * ((n_cols + 63) / 64)));
    }
    return e;
}
