// K1 on LDS-staged boxes (shared grid): S3P:859-890 without fuse.
//
// The texture path prices a wave-wide byte gather at 16 CU-cycles whatever it fetches, which caps a
// one-gather-per-evaluation kernel at about 0.31 of the contract's roofline. Here
// the gathers go to LDS instead. The occupied beams are clustered on the host into spatially
// compact groups of columns; for one cluster, ALL particles of a workgroup land inside a small box
// of the grid (the cluster's extent plus the spread of the particle cloud), which the workgroup
// copies from the reference-order grid into LDS once (dword loads, double-buffered: the next
// cluster's box is in flight while the current one is evaluated) and then serves every
// evaluation from there with ds_read_u8. One workgroup = 1024 threads x KP particles each (pose
// constants stay in registers for the whole launch) x the clusters of one blockIdx.y group.
//
// Exactness is the same construction as k_measure_col: X = floor(fp32(x + 0.5)) is the reference's
// round() whenever the fraction is farther than E from 0 and 1; a column for which some lane is
// closer is redone for those lanes in the reference's fp64 sequence (x and y only: z is an
// integer shift decided once per lane); lanes that leave an axis range, or whose z shift is
// ambiguous, take the reference's whole sequence per beam, flat-only bounds test included, from
// global memory. A box that does not fit LDS (a widely spread cloud) is read from global memory
// with the same arithmetic.
#include "pf_kernels.h"
#include "pf_device.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define BOX_WG 1024
#define BOX_PRE 8                              // dwords staged per thread and cluster
#define BOX_CAP_DW (BOX_WG * BOX_PRE)          // 32 KB per buffer, two buffers
#define BOX_COLCAP 256                         // column groups per cluster (the host splits beyond), staged with the box

typedef float v2f_b __attribute__((ext_vector_type(2)));
// LDS pointers carry their address space, so that the gathers are ds_read_u8 and never flat loads
typedef __attribute__((address_space(3))) const uint8_t lds_u8;
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u4v lds_u4;

namespace {

__device__ __forceinline__ int flr_i32(float x) {
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ bool z_ambiguous(double tz) {          // see pf_kernels.hip
    const double frac = fabs(tz - trunc(tz));
    return !(fabs(frac - 0.5) > 9.5367431640625e-07);
}
__device__ __forceinline__ float uni(float v) {                    // wave-uniform value into an SGPR
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

struct Box {                // wave-uniform description of a staged box
    int X0, Y0, Z0;         // first cell (Z0 a multiple of 4)
    int nx, ny, nzq;        // cells in x and y, dwords in z
    int dwords;             // nx*ny*nzq, 0 = nothing of the cluster can be inside the grid
    int lds;                // 1: fits a buffer
    int inside;             // 1: even the unclipped box lies inside the grid: no lane can leave an axis
};

struct Bounds { float a0, a1, x0, x1, y0, y1; int z0, z1; };   // yaw interval, x/res + 0.5, y/res + 0.5, round(z/res)

__device__ __forceinline__ void prod_range(float a0, float a1, float b0, float b1, float& lo, float& hi) {
    const float p0 = a0 * b0, p1 = a0 * b1, p2 = a1 * b0, p3 = a1 * b1;
    lo = fminf(fminf(p0, p1), fminf(p2, p3));
    hi = fmaxf(fmaxf(p0, p1), fmaxf(p2, p3));
}

// Every cell a lane of this workgroup can name for a beam of cluster c on the fast path or after
// the fp64 correction (which moves X or Y by at most one), clipped to the grid.
__device__ __forceinline__ Box box_of(const PfCluster c, const Bounds& b, int gx, int gy, int gz) {
    // the rotated (lx, ly) at the middle yaw over the cluster's four corners, widened by what half the
    // yaw interval can move a point that far from the origin (|d/dyaw| <= |l|)
    const float am = 0.5f * (b.a0 + b.a1), ah = 0.5f * (b.a1 - b.a0);
    const float cm = __builtin_amdgcn_cosf(am * 0.15915494f), sm = __builtin_amdgcn_sinf(am * 0.15915494f);   // 1e-6 abs: inside the slack
    float l1, h1, l2, h2;
    prod_range(cm, cm, c.lx0, c.lx1, l1, h1);
    prod_range(-sm, -sm, c.ly0, c.ly1, l2, h2);
    const float rad = ah * sqrtf(fmaxf(c.lx0 * c.lx0, c.lx1 * c.lx1) + fmaxf(c.ly0 * c.ly0, c.ly1 * c.ly1));
    int X0 = (int)floorf(l1 + l2 + b.x0 - rad) - 2, X1 = (int)floorf(h1 + h2 + b.x1 + rad) + 2;
    prod_range(sm, sm, c.lx0, c.lx1, l1, h1);
    prod_range(cm, cm, c.ly0, c.ly1, l2, h2);
    int Y0 = (int)floorf(l1 + l2 + b.y0 - rad) - 2, Y1 = (int)floorf(h1 + h2 + b.y1 + rad) + 2;
    int Z0 = c.lz0 + b.z0, Z1 = c.lz1 + b.z1;
    const bool inside = X0 >= 0 && Y0 >= 0 && Z0 >= 0 && X1 < gx && Y1 < gy && Z1 < gz;
    X0 = max(X0, 0); Y0 = max(Y0, 0); Z0 = max(Z0, 0) & ~3;
    X1 = min(X1, gx - 1); Y1 = min(Y1, gy - 1); Z1 = min(Z1, gz - 1);
    Box o;
    o.X0 = X0; o.Y0 = Y0; o.Z0 = Z0;
    o.nx = X1 - X0 + 1; o.ny = Y1 - Y0 + 1; o.nzq = (Z1 - Z0 + 4) >> 2;
    const bool some = o.nx > 0 && o.ny > 0 && Z1 >= Z0;
    const long long dw = some ? (long long)o.nx * o.ny * o.nzq : 0;
    o.lds = dw > 0 && dw <= BOX_CAP_DW;
    o.dwords = o.lds ? (int)dw : 0;
    o.inside = inside;
    return o;
}

template <int KP>
struct Lanes {
    float cf[KP], sf[KP], txh[KP], tyh[KP], thr[KP];   // thr = 0.5 - E; -1 when the lane's z shift is ambiguous
    double c[KP], s[KP], tx[KP], ty[KP];               // for the fp64 correction: a reload from memory would make
                                                       // every wave of the workgroup wait at the next barrier
    int zp[KP], sum[KP], cnt[KP], lane[KP];             // cnt: beams counted beyond the wave's common count
};

// The reference's whole sequence for one beam (S3P:745-747, 880, 85, 882), from global memory.
__device__ __forceinline__ void generic_beam(const double* __restrict__ sorted, int P, int lane, float lx, float ly,
                                             int lz, const PfGridParams& gp, const uint8_t* __restrict__ maps,
                                             int& sum, int& cnt) {
    const double c = sorted[lane], s = sorted[(size_t)P + lane];
    const double tx = sorted[2 * (size_t)P + lane], ty = sorted[3 * (size_t)P + lane], tz = sorted[4 * (size_t)P + lane];
    const double ms = -s, dlx = (double)lx, dly = (double)ly;
    const double Xd = round((c * dlx + ms * dly) + tx);
    const double Yd = round((s * dlx + c * dly) + ty);
    const double Zd = round((double)lz + tz);
    const double fd = fma(Xd, gp.gyz, fma(Yd, gp.gzd, Zd));
    if (fd >= 0.0 && fd < gp.elements) { sum += maps[(unsigned)fd]; cnt += 1; }
}

// fp32 evaluation of one column for one particle: X, Y and how far the fractions are from 0.5
__device__ __forceinline__ float col_xy(float cf, float sf, float txh, float tyh, float lx, float ly, int& X, int& Y) {
    const v2f_b rotA = {cf, sf}, rotB = {-sf, cf}, trh = {txh, tyh};
    const v2f_b lxx = {lx, lx}, lyy = {ly, ly};
    const v2f_b h = __builtin_elementwise_fma(rotA, lxx, __builtin_elementwise_fma(rotB, lyy, trh));
    X = flr_i32(h.x); Y = flr_i32(h.y);
    const float dx = __builtin_amdgcn_fractf(h.x) - 0.5f, dy = __builtin_amdgcn_fractf(h.y) - 0.5f;
    return __builtin_fmaxf(__builtin_fabsf(dx), __builtin_fabsf(dy));
}

// A column some lane cannot settle in fp32, or a column of a cluster whose box is not wholly inside
// the grid or not in LDS: every case, for one particle per lane. Runs in a loop of its own behind
// the fast columns of a cluster, so that the fast loop stays a few dozen instructions.
struct SlowArgs {
    float lx, ly; int n; int lz0, lz1, lz2, lz3;   // (no array: a dynamically indexed one would live in scratch)
    int use_lds, sy, sz, K0;
    lds_u8* lds_b; const uint8_t* maps; const double* sorted;
    int P, gx, gy, gz;
};
__device__ __forceinline__ void slow_column(const SlowArgs& a, const PfGridParams& gp, float cf, float sf,
                                            float txh, float tyh, float thr, int zp, int lane, double c, double s,
                                            double tx, double ty, int& sum, int& cnt) {
    int X, Y;
    const float dev = col_xy(cf, sf, txh, tyh, a.lx, a.ly, X, Y);
    auto lz_of = [&](int j) { return j == 0 ? a.lz0 : j == 1 ? a.lz1 : j == 2 ? a.lz2 : a.lz3; };
    const unsigned zlo = (unsigned)(a.lz0 + zp), zhi = (unsigned)(lz_of(a.n - 1) + zp);
    bool ok = (dev <= thr) & ((unsigned)X < (unsigned)a.gx) & ((unsigned)Y < (unsigned)a.gy) &
              (max(zlo, zhi) < (unsigned)a.gz);
    if (__any(!ok)) {
        // x and y in the reference's fp64 sequence (S3P:745-747, 880) for the lanes that need it
        const double ms = -s, dlx = (double)a.lx, dly = (double)a.ly;
        const double Xd = round((c * dlx + ms * dly) + tx);
        const double Yd = round((s * dlx + c * dly) + ty);
        const bool rep = fabs(Xd) < 1073741824.0 && fabs(Yd) < 1073741824.0;
        const int Xe = rep ? (int)Xd : -1, Ye = rep ? (int)Yd : -1;
        const bool inside = rep & ((unsigned)Xe < (unsigned)a.gx) & ((unsigned)Ye < (unsigned)a.gy) &
                            (max(zlo, zhi) < (unsigned)a.gz) & (thr >= 0.0f);
        if (!ok) {
            if (inside) { X = Xe; Y = Ye; ok = true; }
            else {
                // off an axis (the flat-only bounds test may still accept the index) or an ambiguous
                // z shift: the whole reference sequence, beam by beam
                for (int j = 0; j < a.n; ++j)
                    generic_beam(a.sorted, a.P, lane, a.lx, a.ly, lz_of(j), gp, a.maps, sum, cnt);
            }
        }
    }
    if (ok) {
        unsigned acc = 0;
        if (a.use_lds) {
            const int cb = X * a.sy + Y * a.sz + (a.K0 + zp);
            for (int j = 0; j < a.n; ++j) acc += a.lds_b[cb + lz_of(j)];
        } else {
            const unsigned cb = ((unsigned)X * (unsigned)a.gy + (unsigned)Y) * (unsigned)a.gz + (unsigned)zp;
            for (int j = 0; j < a.n; ++j) acc += a.maps[cb + (unsigned)lz_of(j)];
        }
        sum += (int)acc;                                             // S3P:887-888
        cnt += a.n;
    }
}

// The common case: the cluster's box is in LDS and wholly inside the grid (no lane can leave an
// axis). Returns false, having done nothing, when some lane's fractions are too close to a
// rounding boundary. Kz[k] = K0 + zp[k].
template <int KP, int N>
__device__ __forceinline__ bool fast_column(float lx, float ly, const int (&lz)[4], Lanes<KP>& L, const int (&Kz)[KP],
                                            int sy, int sz, lds_u8* lds_b) {
    int X[KP], Y[KP];
    float over = -1.0f;
#pragma unroll
    for (int k = 0; k < KP; ++k)
        over = __builtin_fmaxf(over, col_xy(L.cf[k], L.sf[k], L.txh[k], L.tyh[k], lx, ly, X[k], Y[k]) - L.thr[k]);
    if (__builtin_expect(__any(over > 0.0f), 0)) return false;
    unsigned v[KP][N];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const int cb = __mul24(X[k], sy) + __mul24(Y[k], sz) + Kz[k];      // box offsets: far below 2^23
#pragma unroll
        for (int j = 0; j < N; ++j) v[k][j] = lds_b[cb + lz[j]];
    }
#pragma unroll
    for (int k = 0; k < KP; ++k)
#pragma unroll
        for (int j = 0; j < N; ++j) L.sum[k] += (int)v[k][j];            // S3P:887-888
    return true;
}

template <int KP>
__global__ __launch_bounds__(BOX_WG) void k_measure_box(
        const double* __restrict__ sorted, int P, int Lp, const uint8_t* __restrict__ maps, int gx, int gy, int gz,
        const PfColG* __restrict__ cg, const PfCluster* __restrict__ cl, const int* __restrict__ group_start,
        PfGridParams gp, double lmax, int2* __restrict__ partial_l, unsigned long long* __restrict__ stats) {
    extern __shared__ unsigned box_lds[];                  // [2][BOX_CAP_DW], then uint4 [2][2 * BOX_COLCAP] column groups
    __shared__ float red_f[16][8];
    __shared__ int red_i[16][2];
    const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
    const unsigned long long t_begin = wall_clock64();
    const int lane0 = blockIdx.x * Lp, lane1 = min(P, lane0 + Lp);
    Lanes<KP> L;
    Bounds b{INFINITY, -INFINITY, INFINITY, -INFINITY, INFINITY, -INFINITY, INT_MAX, INT_MIN};
    // yaw is taken relative to the workgroup's first particle, so that an interval across +-pi stays short
    const float a_ref = atan2f((float)sorted[(size_t)P + lane0], (float)sorted[lane0]);
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const int want = lane0 + tid + BOX_WG * k;
        const int lane = min(want, lane1 - 1);             // surplus threads shadow the last particle
        L.lane[k] = lane;
        const double c = sorted[lane], s = sorted[(size_t)P + lane];
        const double tx = sorted[2 * (size_t)P + lane], ty = sorted[3 * (size_t)P + lane], tz = sorted[4 * (size_t)P + lane];
        L.cf[k] = (float)c; L.sf[k] = (float)s;
        L.c[k] = c; L.s[k] = s; L.tx[k] = tx; L.ty[k] = ty;
        L.txh[k] = (float)(tx + 0.5); L.tyh[k] = (float)(ty + 0.5);
        L.zp[k] = (int)round(tz);
        const double tmax = fmax(fmax(fabs(tx), fabs(ty)), fabs(tz));
        L.thr[k] = (float)(0.5 - (2.5e-7 * (lmax + tmax + 1.0) + 1.0e-7));
        if (z_ambiguous(tz)) L.thr[k] = -1.0f;             // never settled in fp32: this lane goes beam by beam
        L.sum[k] = 0; L.cnt[k] = 0;
        float da = atan2f(L.sf[k], L.cf[k]) - a_ref;
        da -= 6.2831853f * rintf(da * 0.15915494f);
        b.a0 = fminf(b.a0, da); b.a1 = fmaxf(b.a1, da);
        b.x0 = fminf(b.x0, L.txh[k]); b.x1 = fmaxf(b.x1, L.txh[k]);
        b.y0 = fminf(b.y0, L.tyh[k]); b.y1 = fmaxf(b.y1, L.tyh[k]);
        b.z0 = min(b.z0, L.zp[k]); b.z1 = max(b.z1, L.zp[k]);
    }
    // bounds of the workgroup's particles: wave butterflies, then the sixteen wave results through LDS
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        b.a0 = fminf(b.a0, __shfl_xor(b.a0, o, 64)); b.a1 = fmaxf(b.a1, __shfl_xor(b.a1, o, 64));
        b.x0 = fminf(b.x0, __shfl_xor(b.x0, o, 64)); b.x1 = fmaxf(b.x1, __shfl_xor(b.x1, o, 64));
        b.y0 = fminf(b.y0, __shfl_xor(b.y0, o, 64)); b.y1 = fmaxf(b.y1, __shfl_xor(b.y1, o, 64));
        b.z0 = min(b.z0, __shfl_xor(b.z0, o, 64)); b.z1 = max(b.z1, __shfl_xor(b.z1, o, 64));
    }
    if (ln == 0) {
        red_f[wv][0] = b.a0; red_f[wv][1] = b.a1;
        red_f[wv][4] = b.x0; red_f[wv][5] = b.x1; red_f[wv][6] = b.y0; red_f[wv][7] = b.y1;
        red_i[wv][0] = b.z0; red_i[wv][1] = b.z1;
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        b.a0 = fminf(b.a0, red_f[w][0]); b.a1 = fmaxf(b.a1, red_f[w][1]);
        b.x0 = fminf(b.x0, red_f[w][4]); b.x1 = fmaxf(b.x1, red_f[w][5]);
        b.y0 = fminf(b.y0, red_f[w][6]); b.y1 = fmaxf(b.y1, red_f[w][7]);
        b.z0 = min(b.z0, red_i[w][0]); b.z1 = max(b.z1, red_i[w][1]);
    }
    // (atan2f and the fp32 cos/sin are a few ulp off the lanes' own cf, sf: the two cells of slack
    // on every side of the box are there for that and for the fp64 correction)
    b.a0 = uni(b.a0 + a_ref - 1.0e-5f); b.a1 = uni(b.a1 + a_ref + 1.0e-5f);
    b.x0 = uni(b.x0); b.x1 = uni(b.x1); b.y0 = uni(b.y0); b.y1 = uni(b.y1);
    b.z0 = uni(b.z0); b.z1 = uni(b.z1);

    int tl = 8;
    const bool tl_on = stats && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
#define TL_MARK() do { if (tl_on && tl < 120) stats[tl++] = wall_clock64(); } while (0)
    if (tl_on) stats[tl++] = t_begin;
    TL_MARK();
    const int c_begin = group_start[blockIdx.y], c_end = group_start[blockIdx.y + 1];
    int cnt_common = 0;                                     // beams every lane of this wave counted (fast columns)
    typedef __attribute__((address_space(1))) const unsigned g_u32;
    typedef __attribute__((address_space(3))) unsigned l_u32;
    unsigned* col_lds = box_lds + 2 * BOX_CAP_DW;          // [2][8 * BOX_COLCAP] dwords of column groups
    unsigned short* slow_list = (unsigned short*)(col_lds + 2 * 8 * BOX_COLCAP);   // [16 waves][BOX_COLCAP]
    // Starts the copy of cluster c's box (reference order: rows of nzq dwords along z) and of its
    // column groups into buffer `which`: loads that write LDS directly (no registers held while the
    // current cluster is evaluated), lane l of a wave fills dword l of the wave's 64-dword piece.
    auto stage = [&](const Box& bx, const PfCluster& c, int which) {
        // d < 2^13 and the divisors are small: (d + 0.5) * (1/n) truncates to d / n exactly in fp32
        const unsigned ny = (unsigned)bx.ny, nzq = (unsigned)bx.nzq;
        const float r_nzq = 1.0f / (float)bx.nzq, r_ny = 1.0f / (float)bx.ny;
        unsigned* buf = box_lds + which * BOX_CAP_DW;
#pragma unroll
        for (int j = 0; j < BOX_PRE; ++j) {
            const int d0 = uni((tid & ~63) + BOX_WG * j);                  // the wave's first dword of this round
            if (d0 >= bx.dwords) break;
            const unsigned d = (unsigned)d0 + (unsigned)ln;
            if ((int)d < bx.dwords) {
                const unsigned row = (unsigned)(((float)d + 0.5f) * r_nzq), zq = d - row * nzq;
                const unsigned xr = (unsigned)(((float)row + 0.5f) * r_ny), yr = row - xr * ny;
                const size_t a = ((size_t)(bx.X0 + xr) * gy + (bx.Y0 + yr)) * gz + bx.Z0 + 4 * zq;
                __builtin_amdgcn_global_load_lds((g_u32*)(maps + a), (l_u32*)(buf + d0), 4, 0, 0);
            }
        }
        const int ncd = 8 * (c.g1 - c.g0);                                  // dwords of column groups
#pragma unroll
        for (int r = 0; r < 8 * BOX_COLCAP / BOX_WG; ++r) {
            const int d0 = uni((tid & ~63) + BOX_WG * r);
            if (d0 < ncd && d0 + ln < ncd)
                __builtin_amdgcn_global_load_lds((g_u32*)((const unsigned*)(cg + c.g0) + d0 + ln),
                                                 (l_u32*)(col_lds + which * 8 * BOX_COLCAP + d0), 4, 0, 0);
        }
    };
    auto box_uni = [&](const PfCluster& c) {
        Box o = box_of(c, b, gx, gy, gz);
        o.X0 = uni(o.X0); o.Y0 = uni(o.Y0); o.Z0 = uni(o.Z0); o.nx = uni(o.nx); o.ny = uni(o.ny); o.nzq = uni(o.nzq);
        o.dwords = uni(o.dwords); o.lds = uni(o.lds); o.inside = uni(o.inside);
        return o;
    };
    if (c_begin < c_end) {
        PfCluster c = cl[c_begin];
        Box cur = box_uni(c);
        stage(cur, c, 0);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        for (int ci = c_begin; ci < c_end; ++ci) {
            const int which = (ci - c_begin) & 1;
            Box nxt{};
            PfCluster cn{};
            if (ci + 1 < c_end) {
                cn = cl[ci + 1];
                nxt = box_uni(cn);
                stage(nxt, cn, which ^ 1);
            }
            lds_u8* lds_b = (lds_u8*)(box_lds + which * BOX_CAP_DW);
            if (stats && tid == 0) {                     // development aid (SLAM_PF_BOX_STATS)
                atomicAdd(&stats[0], 1ull); atomicAdd(&stats[1], (unsigned long long)cur.lds);
                atomicAdd(&stats[2], (unsigned long long)cur.dwords); atomicAdd(&stats[3], (unsigned long long)(c.g1 - c.g0));
            }
            lds_u4* cols = (lds_u4*)(col_lds + which * 8 * BOX_COLCAP);
            const int ng = c.g1 - c.g0;
            const int sy = cur.ny * cur.nzq * 4, sz = cur.nzq * 4;
            const int K0 = -(cur.X0 * sy + cur.Y0 * sz + cur.Z0);
            int Kz[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) Kz[k] = K0 + L.zp[k];
            const bool fast = cur.lds && cur.inside;
            unsigned short* my_slow = slow_list + wv * BOX_COLCAP;      // this wave's columns for the second loop
            int n_slow = 0;
            if (fast) {
                // every lane reads the same 32 bytes (a broadcast); the group after the one being
                // evaluated is already on its way
                u4v qa = cols[0], qb = cols[1];
                for (int g = 0; g < ng; ++g) {
                    const float lx = __uint_as_float(qa.x), ly = __uint_as_float(qa.y);
                    const int n = uni((int)qa.z);
                    const int lz[4] = {(int)qb.x, (int)qb.y, (int)qb.z, (int)qb.w};
                    const int gn = min(g + 1, ng - 1);
                    qa = cols[2 * gn]; qb = cols[2 * gn + 1];
                    bool done;
                    switch (n) {
                    case 4: done = fast_column<KP, 4>(lx, ly, lz, L, Kz, sy, sz, lds_b); break;
                    case 3: done = fast_column<KP, 3>(lx, ly, lz, L, Kz, sy, sz, lds_b); break;
                    case 2: done = fast_column<KP, 2>(lx, ly, lz, L, Kz, sy, sz, lds_b); break;
                    default: done = fast_column<KP, 1>(lx, ly, lz, L, Kz, sy, sz, lds_b); break;
                    }
                    if (done) cnt_common += n;
                    else { if (ln == 0) my_slow[n_slow] = (unsigned short)g; ++n_slow; }
                }
            }
            const int n_second = fast ? n_slow : ng;
#pragma nounroll
            for (int i = 0; i < n_second; ++i) {
                const int g = fast ? uni((int)my_slow[i]) : i;          // (written by this wave's lane 0 above)
                const u4v qa = cols[2 * g], qb = cols[2 * g + 1];
                const SlowArgs sa{__uint_as_float(qa.x), __uint_as_float(qa.y), uni((int)qa.z),
                                  (int)qb.x, (int)qb.y, (int)qb.z, (int)qb.w, cur.lds, sy, sz, K0, lds_b, maps, sorted,
                                  P, gx, gy, gz};
#pragma unroll
                for (int k = 0; k < KP; ++k)
                    slow_column(sa, gp, L.cf[k], L.sf[k], L.txh[k], L.tyh[k], L.thr[k], L.zp[k], L.lane[k], L.c[k], L.s[k], L.tx[k],
                                L.ty[k], L.sum[k], L.cnt[k]);
            }
            TL_MARK();
            __builtin_amdgcn_s_waitcnt(0);               // the next cluster's box has landed (this lane's part)
            __syncthreads();
            TL_MARK();
            cur = nxt; c = cn;
        }
    }
#pragma unroll
    for (int k = 0; k < KP; ++k)
        if (lane0 + tid + BOX_WG * k < lane1)
            partial_l[(size_t)blockIdx.y * P + L.lane[k]] = make_int2(L.sum[k], L.cnt[k] + cnt_common);
#undef TL_MARK
}

// Sum of the groups' partials per lane, scattered to the particle the lane stands for; the block
// minima K3 starts from (any partition of the particles serves: K3 takes the minimum of all of them).
__global__ __launch_bounds__(256) void k_reduce_scores_l(const int2* __restrict__ partial_l, int n_groups, int P,
                                                         const int32_t* __restrict__ order, int2* __restrict__ scores,
                                                         int32_t* __restrict__ bmin, int32_t* __restrict__ zero_a,
                                                         int32_t* __restrict__ zero_b, int n_zero) {
    __shared__ int2 sh[3][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int l = blockIdx.x * 64 + lane;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < n_zero; k += gridDim.x * 256) { zero_a[k] = 0; zero_b[k] = 0; }
    int sx = 0, sy = 0;
    if (l < P) {
#pragma unroll 8
        for (int k = q; k < n_groups; k += 4) {
            const int2 v = partial_l[(size_t)k * P + l];
            sx += v.x; sy += v.y;
        }
    }
    if (q) sh[q - 1][lane] = make_int2(sx, sy);
    __syncthreads();
    if (q) return;
#pragma unroll
    for (int r = 0; r < 3; ++r) { sx += sh[r][lane].x; sy += sh[r][lane].y; }
    // (a hole of a padded lane order, order[] = -1, stands for no particle: k_measure_tile)
    const int who = l < P ? order[l] : -1;
    if (who >= 0) scores[who] = make_int2(sx, sy);
    int m = who >= 0 ? sx : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o, 64));
    if (lane == 0) bmin[blockIdx.x] = m;
}

}  // namespace

void pfk_box_plan(int P, int n_clusters, int* kp_out, int* parts_out, int* lp_out, int* groups_out) {
    int best_kp = 1; double best_u = -1;
    const int forced = getenv("SLAM_PF_BOX_KP") ? atoi(getenv("SLAM_PF_BOX_KP")) : 0;   // development knob
    for (int kp : {2, 1}) {
        if (forced == 1 || forced == 2) { best_kp = forced; break; }
        const int parts = (P + BOX_WG * kp - 1) / (BOX_WG * kp);
        const double u = (double)P / ((double)parts * BOX_WG * kp);
        if (u >= 0.93) { best_kp = kp; best_u = u; break; }
        if (u > best_u) { best_u = u; best_kp = kp; }
    }
    const int parts = (P + BOX_WG * best_kp - 1) / (BOX_WG * best_kp);
    int groups = 256 / parts;
    if (groups < 1) groups = 1;
    if (groups > n_clusters) groups = n_clusters > 0 ? n_clusters : 1;
    *kp_out = best_kp; *parts_out = parts; *lp_out = (P + parts - 1) / parts; *groups_out = groups;
}

hipError_t pfk_measure_box(hipStream_t st, const double* sorted, int P, const uint8_t* maps, int gx, int gy, int gz,
                           const PfColG* cg, const PfCluster* cl, const int* group_start, int kp, int parts, int lp,
                           int groups, PfGridParams gp, double lmax, int2* partial_l) {
    const dim3 grid(parts, groups);
    const size_t lds = 2 * BOX_CAP_DW * sizeof(unsigned) + 2 * 8 * BOX_COLCAP * sizeof(unsigned) + 16 * BOX_COLCAP * sizeof(unsigned short);
    static const bool want_stats = getenv("SLAM_PF_BOX_STATS") != nullptr;
    unsigned long long* stats = nullptr;
    if (want_stats) {
        if (hipMalloc((void**)&stats, 1024) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemsetAsync(stats, 0, 1024, st);
    }
#define LAUNCH_BOX(K)                                                                                        \
    hipLaunchKernelGGL((k_measure_box<K>), grid, dim3(BOX_WG), lds, st, sorted, P, lp, maps, gx, gy, gz, cg, cl, \
                       group_start, gp, lmax, partial_l, stats)
    if (kp == 2) LAUNCH_BOX(2);
    else LAUNCH_BOX(1);
#undef LAUNCH_BOX
    if (want_stats) {
        unsigned long long hs[128];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(hs, stats, 1024, hipMemcpyDeviceToHost);
        fprintf(stderr, "  workgroup (0,0) timeline, x10 ns from its first instruction:");
        for (int i = 9; i < 128 && hs[i]; ++i) fprintf(stderr, " %llu", hs[i] - hs[8]);
        fprintf(stderr, "\n");
        (void)hipFree(stats);
        fprintf(stderr, "k_measure_box: kp %d parts %d lp %d groups %d | cluster visits %llu, in LDS %llu, avg box %.1f KB, "
                        "columns per visit %.1f; wave-columns %llu, of them corrected in fp64 %llu\n", kp, parts, lp, groups, hs[0], hs[1],
                hs[1] ? 4.0 * hs[2] / hs[1] / 1024 : 0.0, hs[0] ? (double)hs[3] / hs[0] : 0.0, hs[5], hs[4]);
    }
    return hipGetLastError();
}

hipError_t pfk_reduce_scores_l(hipStream_t st, const int2* partial_l, int n_groups, int P, const int32_t* order,
                               int2* scores, int32_t* bmin, int32_t* zero_a, int32_t* zero_b, int n_zero) {
    hipLaunchKernelGGL(k_reduce_scores_l, dim3((P + 63) / 64), dim3(256), 0, st, partial_l, n_groups, P, order, scores,
                       bmin, zero_a, zero_b, n_zero);
    return hipGetLastError();
}
