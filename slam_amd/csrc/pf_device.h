// Device code shared by pf_kernels.hip and pf_resample.hip: the gather-layout address and the K5
// map update (S3P:628-704), whose work can also ride as extra workgroups of the weight kernels.
#pragma once
#include "pf_kernels.h"

__device__ __forceinline__ unsigned mad_u24(unsigned a, unsigned b, unsigned c) {   // a*b + c, a, b < 2^24
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
// ((X>>2)*a + (Y>>2)*b + (Z>>2)*64) | (X&3)<<4 | (Y&3)<<2 | (Z&3); a, b multiples of 64
__device__ __forceinline__ unsigned swz_col_base(unsigned X, unsigned Y, unsigned a, unsigned b) {
    return mad_u24(X >> 2, a, mad_u24(Y >> 2, b, ((X & 3u) << 4) | ((Y & 3u) << 2)));
}
__device__ __forceinline__ unsigned swz_col_addr(unsigned base, unsigned Z, unsigned b) {
    (void)b;
    return (((Z & ~3u) << 4) + base) | (Z & 3u);
}
__device__ __forceinline__ unsigned swz_addr(unsigned X, unsigned Y, unsigned Z, unsigned a, unsigned b) {
    return swz_col_addr(swz_col_base(X, Y, a, b), Z, b);
}
// tile layout (PF_LAYOUT_TILES): Z*a + (Y/8 + 1)*b + (X/8 + 1)*64 + (X&7)*8 + (Y&7); the products stay below 2^32
// (pf_create checks gz*a)
__device__ __forceinline__ unsigned tile_addr(unsigned X, unsigned Y, unsigned Z, unsigned a, unsigned b) {
    return Z * a + ((Y >> 3) + 1u) * b + ((((X >> 3) + 1u) << 6) | ((X & 7u) << 3) | (Y & 7u));
}
__device__ __forceinline__ unsigned layout_addr(unsigned X, unsigned Y, unsigned Z, unsigned a, unsigned b, unsigned kind) {
    return kind == PF_LAYOUT_TILES ? tile_addr(X, Y, Z, a, b) : swz_addr(X, Y, Z, a, b);
}


// ------------------------------------------------------------------------------------------
// K5 map update: S3P:628-704. Every touch of one phase applies the SAME function to a cell
// (hits: v -> min(v+inc, clamp); misses: v -> max(v+inc, clamp)), so the result of a phase is
// independent of the order of its touches and a compare-and-swap on the cell's 32-bit word
// reproduces the reference's sequential loop exactly. Hits complete (kernel boundary) before
// the ray march starts, as in S3P:632-661 then S3P:668-704.
// ------------------------------------------------------------------------------------------
// Returns false when the cell was already at its clamp for this touch (nothing written).
__device__ inline bool cell_touch(void* grid, long long f, int inc, int clampv, bool hit,
                                  bool is_signed) {
    unsigned* wp = (unsigned*)grid + (f >> 2);
    const unsigned shft = (unsigned)(f & 3) * 8u;
    unsigned old = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (true) {
        const unsigned byte = (old >> shft) & 0xffu;
        const int v = is_signed ? (int)(int8_t)byte : (int)byte;
        int nv = v + inc;
        if (hit ? nv > clampv : nv < clampv) nv = clampv;
        if (nv == v) return false;
        const unsigned nw = (old & ~(0xffu << shft)) | (((unsigned)nv & 0xffu) << shft);
        const unsigned seen = atomicCAS(wp, old, nw);
        if (seen == old) return true;
        old = seen;
    }
}

// S3P:641-645: per-axis round, int flat index, flat-only bounds. Coordinates are validated on
// the host to stay far inside int32, so the double arithmetic below is exact.
__device__ inline long long flat_of(double rx, double ry, double rz, const PfUpdateParams& up) {
    const double X = round(rx), Y = round(ry), Z = round(rz);
    if (!(fabs(X) < 2147483648.0 && fabs(Y) < 2147483648.0 && fabs(Z) < 2147483648.0)) return -1;
    const long long f = (long long)X * up.gy * up.gz + (long long)Y * up.gz + (long long)Z;
    return (f < 0 || f >= up.elements) ? -1 : f;
}

// the same touch on the gather-layout copy of the shared grid (both copies see the same
// order-independent function, so they stay identical)
__device__ inline void shadow_touch(const PfUpdateParams& up, long long f, int inc, int clampv,
                                    bool hit) {
    if (!up.shadow) return;
    const unsigned fu = (unsigned)f, gyz = (unsigned)(up.gy * up.gz);
    const unsigned xc = fu / gyz, r = fu - xc * gyz, yc = r / (unsigned)up.gz;
    const unsigned zc = r - yc * (unsigned)up.gz;
    cell_touch(up.shadow, (long long)layout_addr(xc, yc, zc, up.swz_a, up.swz_b, up.swz_kind), inc, clampv, hit, false);
}

// S3P:632-661 for point i.
__device__ inline void d5_hits(int i, const PfUpdateWork& uw) {
    const PfUpdateParams& up = uw.up;
    void* grid = uw.grid;
    const double* __restrict__ points = uw.points;
    double* __restrict__ wpts = uw.wpts;
    if (i >= uw.M) return;
    const double px = points[3 * (size_t)i], py = points[3 * (size_t)i + 1],
                 pz = points[3 * (size_t)i + 2];
    // S3P:547-549 then S3P:555-557
    const double wx = (up.rot[0] * px + up.rot[1] * py + up.rot[2] * pz) + up.pos[0];
    const double wy = (up.rot[3] * px + up.rot[4] * py + up.rot[5] * pz) + up.pos[1];
    const double wz = (up.rot[6] * px + up.rot[7] * py + up.rot[8] * pz) + up.pos[2];
    wpts[3 * (size_t)i] = wx; wpts[3 * (size_t)i + 1] = wy; wpts[3 * (size_t)i + 2] = wz;
    const double dx = wx - up.pos[0], dy = wy - up.pos[1], dz = wz - up.pos[2];
    const double nrm = sqrt(dx * dx + dy * dy + dz * dz);        // S3P:615-617
    if (nrm > up.max_range) return;                              // S3P:637 (NaN falls through)
    const long long f = flat_of(wx * up.conv, wy * up.conv, wz * up.conv, up);
    if (f < 0) return;
    // a touch that found the cell saturated in one copy finds it saturated in the other: the two
    // copies receive the same touches, and within a phase a saturated cell stays saturated
    if (cell_touch(grid, f, up.hit_inc, up.hit_clamp, true, up.is_signed != 0))
        shadow_touch(up, f, up.hit_inc, up.hit_clamp, true);
}
// The same for a whole wave (every lane calls): neighbouring points of a depth image mostly fall into the same
// voxel, so a run of consecutive lanes with one flat index is ONE touch of (run length) * increment by its first
// lane (k touches of v -> min(v + inc, clamp) are one touch of k * inc), a third of the compare-and-swaps on the
// reference's 19 200-point clouds.
__device__ inline void d5_hits_wave(int i, const PfUpdateWork& uw) {
    const PfUpdateParams& up = uw.up;
    long long f = -1;
    if (i < uw.M) {
        const double* __restrict__ points = uw.points;
        const double px = points[3 * (size_t)i], py = points[3 * (size_t)i + 1], pz = points[3 * (size_t)i + 2];
        const double wx = (up.rot[0] * px + up.rot[1] * py + up.rot[2] * pz) + up.pos[0];   // S3P:547-549, 555-557
        const double wy = (up.rot[3] * px + up.rot[4] * py + up.rot[5] * pz) + up.pos[1];
        const double wz = (up.rot[6] * px + up.rot[7] * py + up.rot[8] * pz) + up.pos[2];
        uw.wpts[3 * (size_t)i] = wx; uw.wpts[3 * (size_t)i + 1] = wy; uw.wpts[3 * (size_t)i + 2] = wz;
        const double dx = wx - up.pos[0], dy = wy - up.pos[1], dz = wz - up.pos[2];
        const double nrm = sqrt(dx * dx + dy * dy + dz * dz);        // S3P:615-617
        if (!(nrm > up.max_range)) f = flat_of(wx * up.conv, wy * up.conv, wz * up.conv, up);   // S3P:637 (NaN falls through)
    }
    const int lane = threadIdx.x & 63;
    const int flo = (int)f;                                          // (f < 2^31: the grids' cell counts are below it)
    const int prev = __shfl_up(flo, 1, 64);
    const bool head = f >= 0 && (lane == 0 || prev != flo);
    const unsigned long long heads = __ballot(head || f < 0);        // a lane without a cell ends a run too
    if (!head) return;
    const unsigned long long above = lane == 63 ? 0ull : heads >> (lane + 1);
    const int run = above ? __ffsll((long long)above) : 64 - lane;
    if (cell_touch(uw.grid, f, run * up.hit_inc, up.hit_clamp, true, up.is_signed != 0))
        shadow_touch(up, f, run * up.hit_inc, up.hit_clamp, true);
}

// S3P:664-704 for ray r: PF_RAY_LANES lanes per ray (two rays per wave: a ray of the 3 m range has
// at most 60 steps, and half as many waves fit the chip in one residency instead of two and a
// half); lane l takes steps j = 1+l, 1+l+PF_RAY_LANES, ... (S3P:685: j = 1 .. n_it-1).
#ifndef PF_RAY_LANES
#define PF_RAY_LANES 32
#endif
#define PF_RAYS_PER_WG (256 / PF_RAY_LANES)
__device__ inline void d5_rays(int r, int lane, const PfUpdateWork& uw) {
    const PfUpdateParams& up = uw.up;
    void* grid = uw.grid;
    const double* __restrict__ wpts = uw.wpts;
    const int32_t* __restrict__ ray_idx = uw.ray_idx;
    if (r >= uw.R) return;
    const double* e = wpts + 3 * (size_t)ray_idx[r];
    const double dx = e[0] - up.pos[0], dy = e[1] - up.pos[1], dz = e[2] - up.pos[2];
    const double nrm = sqrt(dx * dx + dy * dy + dz * dz);
    const double nc = nrm * up.conv;
    const double rr = round(nc);                                 // S3P:676
    if (!(rr < 2147483648.0)) return;                            // also drops NaN
    const int n_it = (int)rr;
    const double sx = dx / nc, sy = dy / nc, sz = dz / nc;       // S3P:681-683
    for (int j = 1 + uw.near_steps + lane; j < n_it; j += PF_RAY_LANES) {
        const double jd = (double)j;
        const long long f = flat_of((up.pos[0] + sx * jd) * up.conv, (up.pos[1] + sy * jd) * up.conv,
                                    (up.pos[2] + sz * jd) * up.conv, up);   // S3P:687-690
        if (f >= 0) {
            if (cell_touch(grid, f, up.miss_inc, up.miss_clamp, false, up.is_signed != 0))
                shadow_touch(up, f, up.miss_inc, up.miss_clamp, false);
        }
    }
}


// The first uw.near_steps steps of the rays r0, r0 + stride, ... (S3P:685-700), counted per cell in an LDS table
// (16-bit counters, two to a word: a cell sees at most R < 65536 touches) over the box around the sensor's cell,
// then applied once per cell. A step whose cell is off an axis (the flat index aliases, S3P:691) or outside the
// box is touched directly. Call with every thread of the workgroup; tbl: PF_NEAR_TBL_WORDS words of LDS.
#define PF_NEAR_TBL_WORDS ((PF_NEAR_W * PF_NEAR_W * PF_NEAR_W + 1) / 2)
__device__ inline void d5_rays_near(int r0, int stride, const PfUpdateWork& uw, unsigned* tbl) {
    const PfUpdateParams& up = uw.up;
    for (int k = threadIdx.x; k < PF_NEAR_TBL_WORDS; k += blockDim.x) tbl[k] = 0u;
    __syncthreads();
    const long long gyz = (long long)up.gy * up.gz;
    const int gx = (int)(up.elements / gyz);
    // the box's centre: the sensor's own cell (any cell would do: a step outside the box goes the direct way)
    const double cxd = round(up.pos[0] * up.conv), cyd = round(up.pos[1] * up.conv), czd = round(up.pos[2] * up.conv);
    const bool centre_ok = fabs(cxd) < 1e9 && fabs(cyd) < 1e9 && fabs(czd) < 1e9;
    const int cx = centre_ok ? (int)cxd : 0, cy = centre_ok ? (int)cyd : 0, cz = centre_ok ? (int)czd : 0;
    for (int r = r0 + (int)threadIdx.x * stride; r < uw.R; r += (int)blockDim.x * stride) {
        const double* e = uw.wpts + 3 * (size_t)uw.ray_idx[r];
        const double dx = e[0] - up.pos[0], dy = e[1] - up.pos[1], dz = e[2] - up.pos[2];
        const double nrm = sqrt(dx * dx + dy * dy + dz * dz);
        const double nc = nrm * up.conv;
        const double rr = round(nc);                                 // S3P:676
        if (!(rr < 2147483648.0)) continue;
        const int n_it = (int)rr;
        const double sx = dx / nc, sy = dy / nc, sz = dz / nc;       // S3P:681-683
        const int j1 = min(n_it, 1 + uw.near_steps);
        for (int j = 1; j < j1; ++j) {
            const double jd = (double)j;
            const double X = round((up.pos[0] + sx * jd) * up.conv), Y = round((up.pos[1] + sy * jd) * up.conv),
                         Z = round((up.pos[2] + sz * jd) * up.conv);   // S3P:687-690
            if (!(fabs(X) < 2147483648.0 && fabs(Y) < 2147483648.0 && fabs(Z) < 2147483648.0)) continue;
            const long long f = (long long)X * up.gy * up.gz + (long long)Y * up.gz + (long long)Z;
            if (f < 0 || f >= up.elements) continue;                 // S3P:691
            const int xi = (int)X, yi = (int)Y, zi = (int)Z;
            const unsigned bx = (unsigned)(xi - cx + PF_NEAR_R), by = (unsigned)(yi - cy + PF_NEAR_R), bz = (unsigned)(zi - cz + PF_NEAR_R);
            if (centre_ok && bx < PF_NEAR_W && by < PF_NEAR_W && bz < PF_NEAR_W && (unsigned)xi < (unsigned)gx &&
                (unsigned)yi < (unsigned)up.gy && (unsigned)zi < (unsigned)up.gz) {
                const unsigned slot = (bx * PF_NEAR_W + by) * PF_NEAR_W + bz;
                atomicAdd(&tbl[slot >> 1], 1u << (16u * (slot & 1u)));
            } else if (cell_touch(uw.grid, f, up.miss_inc, up.miss_clamp, false, up.is_signed != 0)) {
                shadow_touch(up, f, up.miss_inc, up.miss_clamp, false);
            }
        }
    }
    __syncthreads();
    for (int slot = threadIdx.x; slot < PF_NEAR_W * PF_NEAR_W * PF_NEAR_W; slot += blockDim.x) {
        const unsigned k = (tbl[slot >> 1] >> (16u * ((unsigned)slot & 1u))) & 0xffffu;
        if (!k) continue;
        const int bz = slot % PF_NEAR_W, by = (slot / PF_NEAR_W) % PF_NEAR_W, bx = slot / (PF_NEAR_W * PF_NEAR_W);
        const long long f = (long long)(bx + cx - PF_NEAR_R) * up.gy * up.gz + (long long)(by + cy - PF_NEAR_R) * up.gz + (bz + cz - PF_NEAR_R);
        if (cell_touch(uw.grid, f, (int)k * up.miss_inc, up.miss_clamp, false, up.is_signed != 0))
            shadow_touch(up, f, (int)k * up.miss_inc, up.miss_clamp, false);
    }
}
