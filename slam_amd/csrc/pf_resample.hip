// gfx950 kernels for the weight update (K3, S3P:920-937) and the resampler (K4, S3P:757-821).
// Built with -ffp-contract=off like pf_kernels.hip.
//
// Two orders of summation:
//  EXACT  one workgroup, the raw-weight sum and the CDF are added by ONE lane in the
//         reference's index order (tiles staged through LDS): every fp64 result is bit-identical
//         to the sequential CPU loop. Used for n_particles_global <= 2048 (PF_ORDER_AUTO) or
//         on request.
//  FAST   multi-workgroup, fixed-shape reductions and scans (the shape depends only on the
//         particle count, never on the grid or the GPU, so every rank that runs this redundantly
//         gets bit-identical results). The sums differ from the sequential ones by rounding
//         only; k_search counts the draws that fall within that error of a CDF boundary and
//         they are searched again on the sequential-order CDF, rebuilt only then (inside
//         k_first_c up to PF_FOLD_CDF_MAX particles; beyond by k_cdf_exact_if, whose workgroup
//         computes the chain's exact bits in parallel, par_seq_cdf).
#include "pf_kernels.h"
#include "pf_device.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>

#define WAVE 64
#define WG 1024
#define TILE 4096
#define NB 256                   // threads per block of the multi-block kernels
#define NEVER_DRAWN 0x7f7f7f7f   // "nobody drew this slot": larger than any slot number
#define PF_MID_MAX 16384         // limit of the one-workgroup forms k3_mid / k4_mid
#define PF_FOLD_CDF_MAX 32768     // up to here k_first_c rebuilds an ambiguous step's CDF itself
#define PF_MULTI_CDF_MIN 262144   // from here on the sequential-order CDF is rebuilt by many workgroups (k_cdfx_*): three launches that
                                  // cost 16 us when no draw was flagged against the one guard launch's 7, so only where most steps
                                  // have flagged draws (at 125 k one step in forty has, and the one-workgroup form takes 0.2 ms there)

// prm[] layout (device doubles)
#define PRM_K99 0      // 0.99 * smallest sum (S3P:928)
#define PRM_N 1        // num_match of the last particle
#define PRM_S 2        // sum of the raw weights (S3P:929)
#define PRM_LSE 3      // log-sum-exp of the log-domain raw weights
#define PRM_LOGN 4

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, WAVE);
    return v;
}
__device__ inline double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, WAVE));
    return v;
}
__device__ inline int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_down(v, o, WAVE));
    return v;
}
// Block-wide reductions; result broadcast to every thread. sh: blockDim/64 doubles. The wave
// totals are added in wave order by every thread: one fixed shape.
__device__ inline double block_sum(double v, double* sh) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    double r = 0;
    for (int k = 0; k < nw; ++k) r += sh[k];
    return r;
}
__device__ inline double block_max(double v, double* sh) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    double r = sh[0];
    for (int k = 1; k < nw; ++k) r = fmax(r, sh[k]);
    return r;
}
// The same for one 256-thread group of a larger workgroup: sh4 = this group's 4 doubles, wv4 =
// the wave's index inside the group. Every thread of the WORKGROUP must call it (barriers).
// Same shape as block_sum/block_max in a 256-thread block, so the results are bit-identical.
__device__ inline double group_sum(double v, double* sh4, int lane, int wv4) {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sh4[wv4] = v;
    __syncthreads();
    double r = 0;
    for (int k = 0; k < NB / WAVE; ++k) r += sh4[k];
    return r;
}
__device__ inline double group_max(double v, double* sh4, int lane, int wv4) {
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) sh4[wv4] = v;
    __syncthreads();
    double r = sh4[0];
    for (int k = 1; k < NB / WAVE; ++k) r = fmax(r, sh4[k]);
    return r;
}
__device__ inline int ld_agent(const int32_t* p) {     // a value other waves change with atomics
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline unsigned long long ld_agent(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Inclusive wave scan (Kogge-Stone over the 64 lanes), fixed shape.
__device__ inline double wave_scan(double v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const double t = __shfl_up(v, o, WAVE);
        if (lane >= o) v += t;
    }
    return v;
}
__device__ inline int wave_scan_i(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const int t = __shfl_up(v, o, WAVE);
        if (lane >= o) v += t;
    }
    return v;
}

// Sequential-order sum of src[0..n) (S3P:929): tiles staged in LDS, one lane adds in order.
__device__ inline double seq_sum(const double* src, int n, double* tile, double* sh) {
    double run = 0;
    for (int base = 0; base < n; base += TILE) {
        const int cntv = min(TILE, n - base);
        __syncthreads();
        for (int k = threadIdx.x; k < cntv; k += blockDim.x) tile[k] = src[base + k];
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int k = 0; k < cntv; ++k) run += tile[k];
        }
    }
    if (threadIdx.x == 0) sh[0] = run;
    __syncthreads();
    run = sh[0];
    __syncthreads();
    return run;
}

// Sequential-order inclusive prefix sums (S3P:763-767): cdf[0] = 0, cdf[i+1] = cdf[i] + w[i].
__device__ inline void seq_cdf(const double* w, int n, double* cdf, double* tile) {
    __shared__ double carry;
    if (threadIdx.x == 0) { carry = 0; cdf[0] = 0; }
    for (int base = 0; base < n; base += TILE) {
        const int cntv = min(TILE, n - base);
        __syncthreads();
        for (int k = threadIdx.x; k < cntv; k += blockDim.x) tile[k] = w[base + k];
        __syncthreads();
        if (threadIdx.x == 0) {
            double run = carry;
            for (int k = 0; k < cntv; ++k) { run += tile[k]; tile[k] = run; }
            carry = run;
        }
        __syncthreads();
        for (int k = threadIdx.x; k < cntv; k += blockDim.x) cdf[base + 1 + k] = tile[k];
    }
    __syncthreads();
}

// S3P:928 for one particle: the raw weight and its log-domain twin for the log-sum-exp.
__device__ inline void raw_weight(double w0, int s, double k99, double n_match, double logn,
                                  double& raw, double& l) {
    const double a = (double)s - k99;
    raw = w0 * a * a / n_match;
    l = (w0 > 0.0 && a != 0.0) ? (log(w0) + 2.0 * log(fabs(a)) - logn) : -INFINITY;
}

// ------------------------------------------------------------------------------------------
// EXACT order: everything in one workgroup of 1024 threads.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void k_weights_exact(const int2* __restrict__ scores, int Pg,
                                                      double* w, float* __restrict__ logw,
                                                      double* cdf, double* scratch,
                                                      int32_t* __restrict__ first,
                                                      unsigned long long* counters) {
    __shared__ double tile[TILE];
    __shared__ double sh[WG / WAVE];
    __shared__ int shi[WG / WAVE];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    // S3P:920-925: smallest sum (ints: the order of the reduction does not matter).
    int m = INT_MAX;
    for (int g = tid; g < Pg; g += WG) m = min(m, scores[g].x);
    m = wave_min_i(m);
    if (lane == 0) shi[wv] = m;
    __syncthreads();
    m = shi[0];
    for (int k = 1; k < WG / WAVE; ++k) m = min(m, shi[k]);
    double smallest = 1000000000000.0;
    if (smallest > (double)m) smallest = (double)m;
    const double n_match = (double)scores[Pg - 1].y;     // S3P:865 + S3P:928: the LAST particle's
    const double k99 = 0.99 * smallest;
    const double logn = log(n_match);

    double lmax = -INFINITY;
    for (int g = tid; g < Pg; g += WG) {
        double raw, l;
        raw_weight(w[g], scores[g].x, k99, n_match, logn, raw, l);
        w[g] = raw;
        scratch[g] = l;
        lmax = fmax(lmax, l);
    }
    lmax = block_max(lmax, sh);
    double se = 0;
    for (int g = tid; g < Pg; g += WG) se += exp(scratch[g] - lmax);
    se = block_sum(se, sh);
    const double lse = lmax + log(se);
    for (int g = tid; g < Pg; g += WG) { logw[g] = (float)(scratch[g] - lse); first[g] = NEVER_DRAWN; }
    if (tid < 3) counters[tid] = 0;                       // K4's counters start at zero

    const double total = seq_sum(w, Pg, tile, sh);        // S3P:929
    __syncthreads();
    for (int g = tid; g < Pg; g += WG) w[g] = w[g] / total;   // S3P:935-937
    __syncthreads();
    seq_cdf(w, Pg, cdf, tile);                            // S3P:763-767
}

// The same prefix sums (S3P:763-767), every value bit-identical to the one-lane chain above, but
// computed by the whole workgroup. While the running sum s stays inside one binade
// [2^e, 2^(e+1)), fl(s + w) for 0 <= w < 2^(e+1) is s plus w rounded to a multiple of
// q = ulp(s) = 2^(e-52); round-to-nearest-even looks at the parity of S = s/q only on an exact
// tie. The chain of adds is therefore an integer prefix sum whose increments depend on one bit of
// the running state: an element is the pair (increment when S is even, increment when S is odd),
// pairs compose associatively, and the workgroup scans XE elements per thread and round. A round
// ends at the first element that takes the sum out of the binade or is not of that form
// (negative, NaN, too large); the lane that holds it does that add in hardware and the next round
// starts behind it. A zero sum is the same scan with every non-zero element ending the round;
// the first XCHAIN elements, and sums that are negative, not finite or below 2^-960, go by the
// chain through LDS, XCHAIN elements at a time.
#define XE 8
#define XCHAIN 1024               // the first elements (a binade every few of them) go by the chain
#define XT (32 * (XE + 1))         // LDS doubles per wave for the transposed stores (padded rows)
#define XLIM 9007199254740992.0   // 2^53: S/q beyond it means the sum left the binade
// Integers below 2^53 kept in doubles: adds and compares are single instructions and exact.
struct IncPair { double e, o; };
__device__ __forceinline__ bool x_odd(double v) { const double h = v * 0.5; return h != floor(h); }
__device__ __forceinline__ IncPair inc_then(IncPair a, IncPair b) {  // a's elements first, then b's
    IncPair r;
    r.e = a.e + (x_odd(a.e) ? b.o : b.e);
    r.o = a.o + (x_odd(a.o) ? b.e : b.o);
    return r;
}
__device__ __forceinline__ bool inc_of(double w, double scale, bool zero_sum, IncPair& p) {
    p.e = p.o = 0.0;
    if (zero_sum) return w == 0.0;
    const double x = w * scale;                                      // w / q, exact where it matters
    if (!(w >= 0.0) || !(x < XLIM)) return false;
    const double W = floor(x), f = x - W;
    p.e = p.o = f > 0.5 ? W + 1.0 : W;
    if (f == 0.5) {                                                  // tie: to the even neighbour
        const bool odd = x_odd(W);
        p.e = odd ? W + 1.0 : W;
        p.o = odd ? W : W + 1.0;
    }
    return true;
}
__device__ __forceinline__ IncPair shfl_up_pair(IncPair v, int o) {
    IncPair t;
    t.e = __shfl_up(v.e, o, WAVE); t.o = __shfl_up(v.o, o, WAVE);
    return t;
}
// One round of the scan in its general form: elements as (even, odd) increment pairs.
// Called by the whole workgroup after a barrier; ends with the round's results in cdf / s_sh /
// pos_sh (not yet behind a barrier).
__device__ __forceinline__ void x_round_pairs(const double (&wk)[XE], double scale, bool zero_sum, double S0,
                                              double q, int base, int end, double* cdf, IncPair* wtot,
                                              double* s_sh, int* pos_sh, int* lim_sh) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    IncPair el[XE];
    bool ok[XE];
    IncPair loc{0.0, 0.0};
#pragma unroll
    for (int k = 0; k < XE; ++k) {
        el[k].e = el[k].o = 0.0; ok[k] = true;
        if (base + k < end) ok[k] = inc_of(wk[k], scale, zero_sum, el[k]);
        loc = inc_then(loc, el[k]);
    }
    IncPair inc = loc;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const IncPair t = shfl_up_pair(inc, o);
        if (lane >= o) inc = inc_then(t, inc);
    }
    const IncPair prev = shfl_up_pair(inc, 1);
    if (lane == WAVE - 1) wtot[wv] = inc;
    __syncthreads();
    // the 16 wave totals, scanned by every wave for itself
    IncPair tot{0.0, 0.0};
    if (lane < WG / WAVE) tot = wtot[lane];
#pragma unroll
    for (int o = 1; o < WG / WAVE; o <<= 1) {
        const IncPair t = shfl_up_pair(tot, o);
        if (lane >= o) tot = inc_then(t, tot);
    }
    IncPair ex{0.0, 0.0};
    if (wv > 0) { ex.e = __shfl(tot.e, wv - 1, WAVE); ex.o = __shfl(tot.o, wv - 1, WAVE); }
    if (lane > 0) ex = inc_then(ex, prev);
    double S = S0 + (x_odd(S0) ? ex.o : ex.e);
    double val[XE];
    double r_stop = 0.0;
    int my_lim = INT_MAX;
    bool live = true;
#pragma unroll
    for (int k = 0; k < XE; ++k) {
        val[k] = 0.0;
        if (live && base + k < end) {
            const double Sn = S + (x_odd(S) ? el[k].o : el[k].e);
            if (!ok[k] || !(Sn < XLIM)) {
                // the add that ends the round, in hardware (only the first such in the
                // workgroup has a valid S; the others are not used)
                my_lim = base + k; live = false; r_stop = S * q + wk[k];
            } else { S = Sn; val[k] = S * q; }
        }
    }
    if (my_lim != INT_MAX) atomicMin(lim_sh, my_lim);
    __syncthreads();
    const int lim = *lim_sh;                                         // pos <= lim <= end
#pragma unroll
    for (int k = 0; k < XE; ++k) {
        const int idx = base + k;
        if (idx < lim) {
            cdf[idx + 1] = val[k];
            if (idx == end - 1) { *s_sh = val[k]; *pos_sh = end; }    // (then lim == end)
        }
    }
    if (my_lim == lim) { cdf[lim + 1] = r_stop; *s_sh = r_stop; *pos_sh = lim + 1; }
}

// tile: XCHAIN doubles of LDS. blockDim.x == WG. The elements [pos0, n) behind a running sum s0 (= cdf[pos0], already
// stored); returns the running sum at n (the same value in every thread).
__device__ __forceinline__ double par_seq_range(const double* w, int pos0, double s0, int n, double* cdf, double* tile) {
    __shared__ IncPair wtot[WG / WAVE];
    __shared__ double wsum[WG / WAVE];
    __shared__ double s_sh;
    __shared__ int pos_sh, lim_sh, tie_sh;
    __shared__ double xpose[(WG / WAVE) * XT];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __syncthreads();
    if (tid == 0) { s_sh = s0; pos_sh = pos0; tie_sh = 0; }
    __syncthreads();
    while (true) {
        const int pos = pos_sh;
        const double s = s_sh;
        if (pos >= n) break;
        const bool zero_sum = s == 0.0;
        if (pos < XCHAIN || (!zero_sum && !(s >= 0x1p-960 && s < INFINITY))) {   // uniform over the workgroup
            const int cnt = min(n - pos, XCHAIN);
            if (tid < cnt) tile[tid] = w[pos + tid];
            __syncthreads();
            if (tid == 0) {
                double r = s;
                for (int k = 0; k < cnt; ++k) { r += tile[k]; tile[k] = r; }
                s_sh = r; pos_sh = pos + cnt;
            }
            __syncthreads();
            if (tid < cnt) cdf[pos + 1 + tid] = tile[tid];
            __syncthreads();
            continue;
        }
        const int sh = zero_sum ? 0 : 52 - ilogb(s);
        const double scale = ldexp(1.0, sh), q = zero_sum ? 0.0 : ldexp(1.0, -sh);
        const double S0 = zero_sum ? 0.0 : s * scale;
        const int end = (int)min((long long)n, (long long)pos + (long long)WG * XE);
        const int base = pos + tid * XE;
        const int nin = min(max(end - base, 0), XE);                 // this thread's elements
        double wk[XE];
#pragma unroll
        for (int k = 0; k < XE; ++k) wk[k] = k < nin ? w[base + k] : 0.0;
        // Without an exact tie among the round's elements every increment is one number, the
        // same for either parity, and an element that cannot be scanned is an increment of 2^54:
        // it stops the round like a change of binade does.
        double d[XE];
        double loc = 0.0;
        bool tie = false;
#pragma unroll
        for (int k = 0; k < XE; ++k) {
            const double x = wk[k] * scale, W = floor(x), f = x - W;
            const bool valid = zero_sum ? wk[k] == 0.0 : (wk[k] >= 0.0 && x < XLIM);
            double dk = f > 0.5 ? W + 1.0 : W;
            dk = valid ? dk : 2.0 * XLIM;
            dk = k < nin ? dk : 0.0;
            tie = tie || (k < nin && valid && f == 0.5);
            d[k] = dk; loc += dk;
        }
        if (__ballot(tie) != 0ull && lane == 0) tie_sh = 1;
        const double inc = wave_scan(loc);
        if (lane == WAVE - 1) wsum[wv] = inc;
        if (tid == 0) lim_sh = end;
        __syncthreads();
        if (tie_sh) {                                                // uniform
            x_round_pairs(wk, scale, zero_sum, S0, q, base, end, cdf, wtot, &s_sh, &pos_sh, &lim_sh);
            if (tid == 0) tie_sh = 0;
            __syncthreads();
            continue;
        }
        double tot = lane < WG / WAVE ? wsum[lane] : 0.0;
#pragma unroll
        for (int o = 1; o < WG / WAVE; o <<= 1) {
            const double t = __shfl_up(tot, o, WAVE);
            if (lane >= o) tot += t;
        }
        // (the lanes before, from their inclusive sum: inc - loc would lose bits when this
        // thread's own elements hold the 2^54 of a stop)
        const double before = __shfl_up(inc, 1, WAVE);
        const double S_start = S0 + ((wv > 0 ? __shfl(tot, wv - 1, WAVE) : 0.0) + (lane > 0 ? before : 0.0));
        double val[XE];
        double S = S_start;
        int c = 0;                                                   // elements settled by this thread
#pragma unroll
        for (int k = 0; k < XE; ++k) {
            S += d[k];
            val[k] = S * q;
            c += (k < nin && S < XLIM) ? 1 : 0;                      // S never decreases: a prefix of the k
        }
        const int my_lim = c < nin ? base + c : INT_MAX;
        if (my_lim != INT_MAX) atomicMin(&lim_sh, my_lim);
        __syncthreads();
        const int lim = lim_sh;                                      // pos <= lim <= end
#pragma unroll
        for (int k = 0; k < XE; ++k)
            if (base + k < lim && base + k == end - 1) { s_sh = val[k]; pos_sh = end; }   // (then lim == end)
        // results out through a wave-private transpose: a lane's XE values are consecutive in
        // memory, stored directly every store instruction would write 64 separate 8-byte pieces;
        // this way it writes one 512-byte run. Half a wave at a time (XT doubles of LDS per wave).
        {
            double* tw = xpose + wv * XT;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if ((lane >> 5) == half) {
#pragma unroll
                    for (int k = 0; k < XE; ++k) tw[(lane & 31) * (XE + 1) + k] = val[k];
                }
                // (LDS operations of one wave complete in order: no barrier between the two loops)
#pragma unroll
                for (int j = 0; j < XE / 2; ++j) {
                    const int m = j * WAVE + lane;                   // element of this half
                    const int idx = pos + wv * (WAVE * XE) + half * (WAVE * XE / 2) + m;
                    const double v = tw[(m >> 3) * (XE + 1) + (m & 7)];
                    if (idx < lim) cdf[idx + 1] = v;
                }
            }
        }
        if (my_lim == lim) {
            // the add that ends the round, in hardware, by the lane that holds the element
            double sp = S_start * q, wc = wk[0];
#pragma unroll
            for (int k = 1; k < XE; ++k) if (k == c) { sp = val[k - 1]; wc = wk[k]; }
            const double r = sp + wc;
            cdf[lim + 1] = r; s_sh = r; pos_sh = lim + 1;
        }
        __syncthreads();
    }
    return s_sh;
}
__device__ __forceinline__ void par_seq_cdf(const double* w, int n, double* cdf, double* tile) {
    if (threadIdx.x == 0) cdf[0] = 0.0;
    (void)par_seq_range(w, 0, 0.0, n, cdf, tile);
}

// Sequential-order CDF of the current weights, only when counters[1] (ambiguous draws) != 0.
__global__ __launch_bounds__(WG) void k_cdf_exact_if(const double* w, int Pg, double* cdf,
                                                     const unsigned long long* counters) {
    __shared__ double tile[XCHAIN];
    if (ld_agent(&counters[1]) == 0) return;
    par_seq_cdf(w, Pg, cdf, tile);
}
// S3P:763-767 on any weight vector, by the one-lane chain (how = 0) or the workgroup scan (how = 1):
// what pf_sequential_cdf exposes so that the two can be compared bit for bit.
__global__ __launch_bounds__(WG) void k_cdf_of(const double* w, int n, double* cdf, int how) {
    __shared__ double tile[TILE];
    if (how == 0) seq_cdf(w, n, cdf, tile);
    else par_seq_cdf(w, n, cdf, tile);
}

// The same CDF by MANY workgroups (a million particles: one workgroup takes a millisecond, and on a sharded job every
// rank pays it in every step). A segment of XSEG elements whose running sum stays inside one binade is, by the
// argument above, a function of the sum's parity when it enters: (total increment on even, on odd), composed from its
// elements' pairs in index order. Which binade a segment lies in is GUESSED from an approximate CDF at its two ends (the
// parallel CDF the weight kernels left, or plain segment sums), with a margin far above the two CDFs' difference:
//   k_cdfx_summ  every segment but the first: the guess, and the segment's pair for the guessed binade;
//   k_cdfx_walk  one workgroup walks the segments in order with the EXACT running sum: a segment whose guess holds for
//                that sum (same binade on entry, still inside it on exit, every element of the scannable form) is
//                settled by one addition and left to the third kernel; any other (the first segment, the dozen or so that
//                cross a binade, sums that are tiny, negative or not finite) is done on the spot by par_seq_range;
//   k_cdfx_fill  the settled segments, each from its exact entry value, in parallel.
// Nothing rests on the guess being right: the walker tests it against the exact sum, and a wrong guess only sends the
// segment down the slow way. Bit-identical to seq_cdf (tests/test_gpu_parity.py).
#define XSEG (WG * XE)
#define XWALK 256                     // segment records the walker stages in LDS at a time
#define XGUESS 1.4901161193847656e-08          // 2^-26: margin of the binade guess (the two CDFs differ by < n * 2^-52 relative)
// The segments: [0, XCHAIN) (the binades change every few elements there: always the one-lane chain), [1024, 2048),
// [2048, 4096), [4096, 8192) (one binade crossing each on an even CDF), then XSEG elements each.
#define XHEAD 4                        // segments inside the first XSEG elements
__host__ __device__ inline int cdfx_nseg(long long n) { return n <= XSEG ? XHEAD : XHEAD - 1 + (int)((n + XSEG - 1) / XSEG); }
__host__ __device__ inline void cdfx_range(int b, long long n, int& start, int& end) {
    long long a, e;
    if (b < XHEAD) { a = b == 0 ? 0 : (long long)XCHAIN << (b - 1); e = (long long)XCHAIN << b; }
    else { a = (long long)(b - XHEAD + 1) * XSEG; e = a + XSEG; }
    start = (int)(a < n ? a : n); end = (int)(e < n ? e : n);
}
// state: 0 not scannable / done by the walker; 1 guessed: one binade; 2 to fill (one binade); 3 guessed: one crossing at
// element p (the elements before it in binade ex, pair (e, o); p itself added in hardware; the elements behind it in
// binade ex + 1, pair (e2, o2)); 4 to fill (crossing)
struct __attribute__((aligned(64))) CdfSeg { double e, o, start, e2, o2, start2, wp; int ex, state, p, pad; };
// the workgroup's elements' pairs composed in index order: the exclusive prefix of this thread (ex) and the total
__device__ __forceinline__ void cdfx_block_scan(IncPair loc, IncPair* wtot, IncPair& ex, IncPair& total) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    IncPair inc = loc;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const IncPair t = shfl_up_pair(inc, o);
        if (lane >= o) inc = inc_then(t, inc);
    }
    const IncPair prev = shfl_up_pair(inc, 1);
    __syncthreads();
    if (lane == WAVE - 1) wtot[wv] = inc;
    __syncthreads();
    IncPair tot{0.0, 0.0};
    if (lane < WG / WAVE) tot = wtot[lane];
#pragma unroll
    for (int o = 1; o < WG / WAVE; o <<= 1) {
        const IncPair t = shfl_up_pair(tot, o);
        if (lane >= o) tot = inc_then(t, tot);
    }
    ex = IncPair{0.0, 0.0};
    if (wv > 0) { ex.e = __shfl(tot.e, wv - 1, WAVE); ex.o = __shfl(tot.o, wv - 1, WAVE); }
    if (lane > 0) ex = inc_then(ex, prev);
    total.e = __shfl(tot.e, WG / WAVE - 1, WAVE); total.o = __shfl(tot.o, WG / WAVE - 1, WAVE);
}
// cdf: an APPROXIMATE CDF of w (the parallel one of the weight kernels, or k_cdfx_approx's), read here, overwritten later
__global__ __launch_bounds__(WG) void k_cdfx_summ(const double* __restrict__ w, int n, const double* __restrict__ cdf,
                                                  CdfSeg* __restrict__ segs, const unsigned long long* counters) {
    __shared__ IncPair wtot[WG / WAVE];
    __shared__ int bad_sh, p_sh;
    if (counters && ld_agent(&counters[1]) == 0) return;
    const int tid = threadIdx.x, b = blockIdx.x + 1;
    int start, end;
    cdfx_range(b, n, start, end);
    const double lo = cdf[start], hi = cdf[end];
    bool ok = lo >= 0x1p-960 && hi < 0x1p1000 && hi >= lo;
    const int ex = ok ? ilogb(lo * (1.0 - XGUESS)) : 0;
    const int exh = ok ? ilogb(hi * (1.0 + XGUESS)) : 0;
    ok = ok && (exh == ex || exh == ex + 1);
    bool cross = ok && exh == ex + 1;
    if (tid == 0) { bad_sh = 0; p_sh = INT_MAX; }
    __syncthreads();
    const int base = start + tid * XE;
    const double edge = ldexp(1.0, ex + 1);                          // where the next binade starts
    if (cross) {                                                     // (workgroup-uniform) the first element that reaches it
        int mine = INT_MAX;
#pragma unroll
        for (int k = XE - 1; k >= 0; --k)
            if (base + k < end && cdf[base + k + 1] >= edge) mine = base + k;
        if (mine != INT_MAX) atomicMin(&p_sh, mine);
        __syncthreads();
    }
    const int p = cross ? p_sh : INT_MAX;
    // (no element of the guess reaches the edge, only the margin did -- the last segment of normalised weights ends just
    // below 1: one binade as far as the guess can tell, and the walker sees whether the exact sum stays inside it)
    if (cross && p == INT_MAX) cross = false;
    if (cross) {
        // the guess must be clear of the edge on both sides of p (else the walker takes the segment the slow way)
        ok = cdf[p] * (1.0 + XGUESS) < edge && cdf[p + 1] * (1.0 - XGUESS) >= edge;
    }
    IncPair locA{0.0, 0.0}, locB{0.0, 0.0};
    if (ok) {                                                        // workgroup-uniform
        const double scaleA = ldexp(1.0, 52 - ex), scaleB = ldexp(1.0, 51 - ex);
        bool fine = true;
#pragma unroll
        for (int k = 0; k < XE; ++k) {
            IncPair el{0.0, 0.0};
            const int idx = base + k;
            if (idx < end && idx != p) fine = inc_of(w[idx], idx < p ? scaleA : scaleB, false, el) && fine;
            if (idx < p) locA = inc_then(locA, el); else locB = inc_then(locB, el);
        }
        if (!fine) bad_sh = 1;                                       // (benign race: every writer stores 1)
    }
    IncPair exq, totalA, totalB{0.0, 0.0};
    cdfx_block_scan(locA, wtot, exq, totalA);
    if (cross) cdfx_block_scan(locB, wtot, exq, totalB);             // (workgroup-uniform)
    if (tid == 0) {
        CdfSeg sg;
        sg.e = totalA.e; sg.o = totalA.o; sg.start = 0.0; sg.e2 = totalB.e; sg.o2 = totalB.o; sg.start2 = 0.0;
        sg.wp = (ok && cross) ? w[p] : 0.0;
        sg.ex = ex; sg.state = (ok && !bad_sh) ? (cross ? 3 : 1) : 0; sg.p = p; sg.pad = 0;
        segs[b] = sg;
    }
}
// 2^k for -1022 <= k <= 1023 and the exponent of a normal positive double, without the library calls (the walker does a
// dozen of them per segment, one thread's worth of latency each)
__device__ __forceinline__ double x_pow2(int k) { return __longlong_as_double((long long)(1023 + k) << 52); }
__device__ __forceinline__ int x_exp(double v) { return (int)((__double_as_longlong(v) >> 52) & 0x7ff) - 1023; }
__global__ __launch_bounds__(WG) void k_cdfx_walk(const double* w, int n, double* cdf, CdfSeg* segs,
                                                  const unsigned long long* counters) {
    __shared__ double tile[XCHAIN];
    __shared__ CdfSeg seg_sh[XWALK];
    __shared__ double s_walk;
    __shared__ int b_walk;
    if (counters && ld_agent(&counters[1]) == 0) return;
    const int nseg = cdfx_nseg(n);
    if (threadIdx.x == 0) cdf[0] = 0.0;
    int start, end;
    cdfx_range(0, n, start, end);
    double s = par_seq_range(w, 0, 0.0, end, cdf, tile);
    int b = 1;
    // The records go through LDS, XWALK at a time (one dependent global load per segment was a microsecond each), and ONE
    // wave walks the segments whose guess holds while the others wait at the barrier (sixteen waves in lockstep share the
    // CU's issue slots: 0.5 us per segment against 0.15); a segment that needs the slow way brings everybody back.
    while (b < nseg) {
        __syncthreads();
        const int b0 = b;
        for (int k = threadIdx.x; k < XWALK && b0 + k < nseg; k += WG) seg_sh[k] = segs[b0 + k];
        __syncthreads();
        if (threadIdx.x < WAVE) {
            int bb = b0;
            double ss = s;
            for (; bb < nseg && bb < b0 + XWALK; ++bb) {
                const CdfSeg sg = seg_sh[bb - b0];                   // the same for every lane
                bool ok = (sg.state == 1 || sg.state == 3) && ss >= 0x1p-960 && ss < INFINITY && x_exp(ss) == sg.ex;
                double s1 = 0.0, sp = 0.0;
                if (ok) {
                    const double S0 = ss * x_pow2(52 - sg.ex);
                    const double S1 = S0 + (x_odd(S0) ? sg.o : sg.e);
                    ok = S1 < XLIM;                                  // still inside the binade behind the (first) part
                    s1 = S1 * x_pow2(sg.ex - 52);
                }
                if (ok && sg.state == 3) {
                    sp = s1 + sg.wp;                                 // the add that crosses, in hardware
                    ok = sp < INFINITY && x_exp(sp) == sg.ex + 1;
                    if (ok) {
                        const double T0 = sp * x_pow2(51 - sg.ex);
                        const double T1 = T0 + (x_odd(T0) ? sg.o2 : sg.e2);
                        ok = T1 < XLIM;
                        s1 = T1 * x_pow2(sg.ex - 51);
                    }
                }
                if (!ok) break;
                if (threadIdx.x == 0) { segs[bb].start = ss; segs[bb].start2 = sp; segs[bb].state = sg.state + 1; }
                ss = s1;
            }
            if (threadIdx.x == 0) { s_walk = ss; b_walk = bb; }
        }
        __syncthreads();
        s = s_walk; b = b_walk;
        if (b < nseg && b < b0 + XWALK) {                            // (workgroup-uniform) this one the slow way
            if (threadIdx.x == 0) segs[b].state = 0;
#ifdef CDFX_DEBUG
            if (threadIdx.x == 0) printf("walker: segment %d of %d the slow way (s %.17g)\n", b, nseg, s);
#endif
            cdfx_range(b, n, start, end);
            s = par_seq_range(w, start, s, end, cdf, tile);
            ++b;
        }
    }
}
__global__ __launch_bounds__(WG) void k_cdfx_fill(const double* __restrict__ w, int n, double* __restrict__ cdf,
                                                  const CdfSeg* __restrict__ segs, const unsigned long long* counters) {
    __shared__ IncPair wtot[WG / WAVE];
    if (counters && ld_agent(&counters[1]) == 0) return;
    const int tid = threadIdx.x, b = blockIdx.x + 1;
    const CdfSeg sg = segs[b];
    if (sg.state != 2 && sg.state != 4) return;                      // workgroup-uniform
    const bool cross = sg.state == 4;
    const int p = cross ? sg.p : INT_MAX;
    int start, end;
    cdfx_range(b, n, start, end);
    const double scaleA = ldexp(1.0, 52 - sg.ex), qA = ldexp(1.0, sg.ex - 52);
    const double scaleB = ldexp(1.0, 51 - sg.ex), qB = ldexp(1.0, sg.ex - 51);
    const int base = start + tid * XE;
    IncPair el[XE];
    IncPair locA{0.0, 0.0}, locB{0.0, 0.0};
#pragma unroll
    for (int k = 0; k < XE; ++k) {
        el[k] = IncPair{0.0, 0.0};
        const int idx = base + k;
        if (idx < end && idx != p) (void)inc_of(w[idx], idx < p ? scaleA : scaleB, false, el[k]);
        if (idx < p) locA = inc_then(locA, el[k]); else locB = inc_then(locB, el[k]);
    }
    IncPair exA, exB{0.0, 0.0}, total;
    cdfx_block_scan(locA, wtot, exA, total);
    if (cross) cdfx_block_scan(locB, wtot, exB, total);              // (workgroup-uniform)
    const double S0 = sg.start * scaleA, T0 = sg.start2 * scaleB;
    double S = S0 + (x_odd(S0) ? exA.o : exA.e);
    double T = T0 + (x_odd(T0) ? exB.o : exB.e);
#pragma unroll
    for (int k = 0; k < XE; ++k) {
        const int idx = base + k;
        if (idx < p) { S += x_odd(S) ? el[k].o : el[k].e; if (idx < end) cdf[idx + 1] = S * qA; }
        else if (idx == p) cdf[idx + 1] = sg.start2;
        else { T += x_odd(T) ? el[k].o : el[k].e; if (idx < end) cdf[idx + 1] = T * qB; }
    }
}
// an approximate CDF for a weight vector that has none yet (pf_sequential_cdf's how = 2): plain sums of the segments,
// their running total, a plain scan inside each segment
__global__ __launch_bounds__(WG) void k_cdfx_segsum(const double* __restrict__ w, int n, double* __restrict__ sums) {
    __shared__ double sh[WG / WAVE];
    const int start = blockIdx.x * XSEG, end = (int)min((long long)n, (long long)start + XSEG);
    double v = 0.0;
    for (int i = start + threadIdx.x; i < end; i += WG) v += w[i];
    v = block_sum(v, sh);
    if (threadIdx.x == 0) sums[blockIdx.x] = v;
}
__global__ void k_cdfx_bounds(const double* __restrict__ sums, int nseg, double* __restrict__ bounds) {
    if (threadIdx.x || blockIdx.x) return;
    double r = 0.0;
    bounds[0] = 0.0;
    for (int b = 0; b < nseg; ++b) { r += sums[b]; bounds[b + 1] = r; }
}
__global__ __launch_bounds__(WG) void k_cdfx_approx(const double* __restrict__ w, int n, const double* __restrict__ bounds,
                                                    double* __restrict__ cdf) {
    __shared__ double wsum[WG / WAVE];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int start = blockIdx.x * XSEG, end = (int)min((long long)n, (long long)start + XSEG);
    const int base = start + tid * XE;
    double v[XE], loc = 0.0;
#pragma unroll
    for (int k = 0; k < XE; ++k) { v[k] = base + k < end ? w[base + k] : 0.0; loc += v[k]; }
    const double inc = wave_scan(loc);
    if (lane == WAVE - 1) wsum[wv] = inc;
    __syncthreads();
    double off = bounds[blockIdx.x] + (inc - loc);
    for (int k = 0; k < wv; ++k) off += wsum[k];
    if (blockIdx.x == 0 && tid == 0) cdf[0] = 0.0;
#pragma unroll
    for (int k = 0; k < XE; ++k) { off += v[k]; if (base + k < end) cdf[base + k + 1] = off; }
}

// ------------------------------------------------------------------------------------------
// FAST order: K3 as four kernels.
// ------------------------------------------------------------------------------------------
__device__ inline void d3_min(const int2* __restrict__ scores, int Pg, double* prm, int* shi) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int m = INT_MAX;
    for (int g = tid; g < Pg; g += WG) m = min(m, scores[g].x);
    m = wave_min_i(m);
    if (lane == 0) shi[wv] = m;
    __syncthreads();
    if (tid == 0) {
        m = shi[0];
        for (int k = 1; k < WG / WAVE; ++k) m = min(m, shi[k]);
        double smallest = 1000000000000.0;                 // S3P:920
        if (smallest > (double)m) smallest = (double)m;
        const double n_match = (double)scores[Pg - 1].y;
        prm[PRM_K99] = 0.99 * smallest;
        prm[PRM_N] = n_match;
        prm[PRM_LOGN] = log(n_match);
    }
}
__global__ __launch_bounds__(WG) void k3_min(const int2* __restrict__ scores, int Pg, double* prm) {
    __shared__ int shi[WG / WAVE];
    d3_min(scores, Pg, prm, shi);
}

// Block minima of sum_match over 1024 particles per workgroup, for scores that did not come out of this
// handle's own score reduction (a caller-side all-gather): k3_raw_m then takes the minimum of the few
// minima instead of one workgroup walking every score (k3_min: 14.8 us at 80 k particles, this 3 us).
__global__ __launch_bounds__(NB) void k3_bmin(const int2* __restrict__ scores, int Pg, int32_t* __restrict__ bmin) {
    __shared__ int sh[NB / WAVE];
    int m = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = (int)blockIdx.x * 4 * NB + k * NB + (int)threadIdx.x;
        if (i < Pg) m = min(m, scores[i].x);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < NB / WAVE; ++q) m = min(m, sh[q]);
        bmin[blockIdx.x] = m;
    }
}
int pfk_block_minima_count(int Pg) { return (Pg + 4 * NB - 1) / (4 * NB); }
hipError_t pfk_block_minima(hipStream_t st, const int2* scores, int Pg, int32_t* bmin) {
    hipLaunchKernelGGL(k3_bmin, dim3(pfk_block_minima_count(Pg)), dim3(NB), 0, st, scores, Pg, bmin);
    return hipGetLastError();
}

// raw weights + per-block partials: bsum[b] = sum of the block's raw weights (wave shuffle tree,
// then the 4 wave totals in order), (bmax[b], bse[b]) = block max of l and sum of exp(l - bmax).
// vb = the 256-particle block, lt = thread inside it, sh4 = 4 doubles of LDS for this block.
// (w0, s0: this thread's previous weight and sum_match, loaded by the caller, which may do so
// before the constants in prm exist)
__device__ inline void d3_raw_v(int vb, int lt, double* sh4, double w0, int s0, int Pg,
                                double* w, double* __restrict__ lbuf, const double* prm, double* bsum,
                                double* bmax, double* bse) {
    const int g = vb * NB + lt, lane = lt & 63, wv4 = lt >> 6;
    double raw = 0, l = -INFINITY;
    if (g < Pg) {
        raw_weight(w0, s0, prm[PRM_K99], prm[PRM_N], prm[PRM_LOGN], raw, l);
        w[g] = raw;
        lbuf[g] = l;
    }
    const double s = group_sum(raw, sh4, lane, wv4);
    const double m = group_max(l, sh4, lane, wv4);
    const double e = group_sum((g < Pg && l > -INFINITY) ? exp(l - m) : 0.0, sh4, lane, wv4);
    if (lt == 0 && g < Pg) { bsum[vb] = s; bmax[vb] = m; bse[vb] = e; }
}
__device__ inline void d3_raw(int vb, int lt, double* sh4, const int2* __restrict__ scores, int Pg,
                              double* w, double* __restrict__ lbuf, const double* prm, double* bsum,
                              double* bmax, double* bse) {
    const int g = vb * NB + lt;
    d3_raw_v(vb, lt, sh4, g < Pg ? w[g] : 0.0, g < Pg ? scores[g].x : 0, Pg, w, lbuf, prm, bsum, bmax, bse);
}
__global__ __launch_bounds__(NB) void k3_raw(const int2* __restrict__ scores, int Pg, double* w,
                                             double* __restrict__ lbuf,
                                             const double* __restrict__ prm, double* bsum,
                                             double* bmax, double* bse) {
    __shared__ double sh[NB / WAVE];
    d3_raw(blockIdx.x, threadIdx.x, sh, scores, Pg, w, lbuf, prm, bsum, bmax, bse);
}

// (sharded handles: no block minima, k3_min ran before; the hit phase of a map update rides here)
__global__ __launch_bounds__(NB) void k3_raw_u(const int2* __restrict__ scores, int Pg, double* w,
                                               double* __restrict__ lbuf,
                                               const double* __restrict__ prm, double* bsum,
                                               double* bmax, double* bse, int nb, PfUpdateWork uw) {
    __shared__ double sh[NB / WAVE];
    if ((int)blockIdx.x >= nb) { d5_hits(((int)blockIdx.x - nb) * NB + (int)threadIdx.x, uw); return; }
    d3_raw(blockIdx.x, threadIdx.x, sh, scores, Pg, w, lbuf, prm, bsum, bmax, bse);
}

// k3_raw with the min stage inside: the smallest sum comes from pfk_reduce_scores' block minima
// (integers: any order), the three derived constants are computed by every workgroup with the
// expressions of d3_min.
__device__ __forceinline__ void d3_raw_m(int vb, const int2* __restrict__ scores, int Pg, double* w,
                                               double* __restrict__ lbuf,
                                               const int32_t* __restrict__ bmin, int n_bmin,
                                               double* bsum, double* bmax, double* bse) {
    __shared__ double sh[NB / WAVE];
    __shared__ int shi[NB / WAVE];
    __shared__ double prm_s[8];
    const int lt = threadIdx.x, lane = lt & 63, wv = lt >> 6;
    // this thread's inputs first: their latency overlaps the reduction of the block minima
    const int g0 = vb * NB + lt;
    const double w0 = g0 < Pg ? w[g0] : 0.0;
    const int s0 = g0 < Pg ? scores[g0].x : 0;
    int m = INT_MAX;
    for (int k = lt; k < n_bmin; k += NB) m = min(m, bmin[k]);
    m = wave_min_i(m);
    if (lane == 0) shi[wv] = m;
    __syncthreads();
    if (lt == 0) {
        m = shi[0];
        for (int k = 1; k < NB / WAVE; ++k) m = min(m, shi[k]);
        double smallest = 1000000000000.0;                 // S3P:920
        if (smallest > (double)m) smallest = (double)m;
        const double n_match = (double)scores[Pg - 1].y;
        prm_s[PRM_K99] = 0.99 * smallest;
        prm_s[PRM_N] = n_match;
        prm_s[PRM_LOGN] = log(n_match);
    }
    __syncthreads();
    d3_raw_v(vb, lt, sh, w0, s0, Pg, w, lbuf, prm_s, bsum, bmax, bse);
}

__global__ __launch_bounds__(NB) void k3_raw_m(const int2* __restrict__ scores, int Pg, double* w,
                                               double* __restrict__ lbuf,
                                               const int32_t* __restrict__ bmin, int n_bmin,
                                               double* bsum, double* bmax, double* bse) {
    d3_raw_m(blockIdx.x, scores, Pg, w, lbuf, bmin, n_bmin, bsum, bmax, bse);
}
// The same launch carrying the hit phase of a map update in workgroups nb .. (S3P:632-661): the
// two have nothing to do with each other, which is the point: both are a few dependent memory
// round trips on a handful of CUs, and side by side they take the time of the longer one.
__global__ __launch_bounds__(NB) void k3_raw_m_u(const int2* __restrict__ scores, int Pg, double* w,
                                                 double* __restrict__ lbuf,
                                                 const int32_t* __restrict__ bmin, int n_bmin,
                                                 double* bsum, double* bmax, double* bse, int nb,
                                                 PfUpdateWork uw) {
    if ((int)blockIdx.x >= nb) { d5_hits(((int)blockIdx.x - nb) * NB + (int)threadIdx.x, uw); return; }
    d3_raw_m(blockIdx.x, scores, Pg, w, lbuf, bmin, n_bmin, bsum, bmax, bse);
}

// One workgroup: S = sum of bsum (thread t adds blocks t, t+1024, ... then the fixed tree),
// exclusive scan of bsum into boff, global max / log-sum-exp.
__device__ inline void d3_combine(int nb, const double* bsum, const double* bmax, const double* bse,
                                  double* boff, double* prm, double* sh) {
    const int tid = threadIdx.x;
    // contiguous segment per thread so that the scan is a scan
    const int seg = (nb + WG - 1) / WG;
    const int b = tid * seg, e = min(nb, b + seg);
    double local = 0, m = -INFINITY;
    for (int k = b; k < e; ++k) { local += bsum[k]; m = fmax(m, bmax[k]); }
    // exclusive scan of the thread totals: wave scan + wave offsets (fixed shape)
    const double inc = wave_scan(local);
    const int lane = tid & 63, wv = tid >> 6;
    __syncthreads();
    if (lane == 63) sh[wv] = inc;
    __syncthreads();
    double woff = 0;
    for (int k = 0; k < wv; ++k) woff += sh[k];
    double total = 0;
    for (int k = 0; k < WG / WAVE; ++k) total += sh[k];
    double run = woff + inc - local;
    for (int k = b; k < e; ++k) { boff[k] = run; run += bsum[k]; }
    const double gmax = block_max(m, sh);
    double se = 0;
    for (int k = b; k < e; ++k) se += bmax[k] > -INFINITY ? bse[k] * exp(bmax[k] - gmax) : 0.0;
    se = block_sum(se, sh);
    if (tid == 0) {
        prm[PRM_S] = total;
        prm[PRM_LSE] = gmax + log(se);
    }
}
__global__ __launch_bounds__(WG) void k3_combine(int nb, const double* __restrict__ bsum,
                                                 const double* __restrict__ bmax,
                                                 const double* __restrict__ bse, double* boff,
                                                 double* prm) {
    __shared__ double sh[WG / WAVE];
    d3_combine(nb, bsum, bmax, bse, boff, prm, sh);
}

// normalise, log-weights, CDF: cdf[i+1] = (boff[block] + inclusive scan of raw inside the block) / S.
// (raw, lb: this thread's raw weight and log-domain twin, loaded by the caller)
__device__ inline void d3_norm_v(int vb, int lt, double* sh4, int Pg, double raw, double lb, double* w,
                                 float* __restrict__ logw, const double* prm, const double boff_vb,
                                 double* cdf, int32_t* __restrict__ first,
                                 unsigned long long* counters) {
    const int g = vb * NB + lt, lane = lt & 63, wv4 = lt >> 6;
    const double S = prm[PRM_S];
    const double inc = wave_scan(raw);
    __syncthreads();
    if (lane == 63) sh4[wv4] = inc;
    __syncthreads();
    double woff = 0;
    for (int k = 0; k < wv4; ++k) woff += sh4[k];
    if (g < Pg) {
        w[g] = raw / S;                                     // S3P:936
        logw[g] = (float)(lb - prm[PRM_LSE]);
        const double cv = (boff_vb + (woff + inc)) / S;
        cdf[g + 1] = cv;
        if (lt == NB - 1 || g == Pg - 1) cdf[Pg + 1 + vb] = cv;   // the block's end once more, contiguous (k_search_t)
        first[g] = NEVER_DRAWN;                             // K4's first-occurrence table
    }
    if (g == 0) cdf[0] = 0;
    if (g < 3) counters[g] = 0;                             // K4's counters start at zero
}
__device__ inline void d3_norm(int vb, int lt, double* sh4, int Pg, double* w, const double* lbuf,
                               float* __restrict__ logw, const double* prm, const double* boff,
                               double* cdf, int32_t* __restrict__ first,
                               unsigned long long* counters) {
    const int g = vb * NB + lt;
    d3_norm_v(vb, lt, sh4, Pg, g < Pg ? w[g] : 0.0, g < Pg ? lbuf[g] : 0.0, w, logw, prm, vb * NB < Pg ? boff[vb] : 0.0, cdf, first,
              counters);
}
__global__ __launch_bounds__(NB) void k3_norm(int Pg, double* w, const double* __restrict__ lbuf,
                                              float* __restrict__ logw,
                                              const double* __restrict__ prm,
                                              const double* __restrict__ boff, double* cdf,
                                              int32_t* __restrict__ first,
                                              unsigned long long* counters) {
    __shared__ double sh[NB / WAVE];
    d3_norm(blockIdx.x, threadIdx.x, sh, Pg, w, lbuf, logw, prm, boff, cdf, first, counters);
}

// k3_norm with the combine stage inside: for up to 1024 blocks (262 144 particles) the combine is
// a scan over at most 1024 block partials, cheaper to redo in every workgroup than to launch
// (a launch boundary costs ~4 us here, the redundant scan well under one). The 256 threads walk
// the 16 wave-shaped pieces of k3_combine's 1024-thread arithmetic in four passes, same
// shuffles, same order of the wave totals: bit-identical to the separate kernel.
__device__ __forceinline__ void d3_norm_c(int vb, int Pg, int nb, double* w, const double* lbuf,
                                                float* __restrict__ logw, const double* bsum,
                                                const double* bmax, const double* bse, double* cdf,
                                                int32_t* __restrict__ first,
                                                unsigned long long* counters) {
    __shared__ double shT[WG / WAVE], shM[WG / WAVE], shS[WG / WAVE];
    __shared__ double sh4[NB / WAVE];
    __shared__ double prm_s[8], s_boff;
    const int lt = threadIdx.x, lane = lt & 63, wv = lt >> 6;
    // this thread's inputs first: their latency overlaps the combine
    const int g0 = vb * NB + lt;
    const double raw0 = g0 < Pg ? w[g0] : 0.0, lb0 = g0 < Pg ? lbuf[g0] : 0.0;
    double loc[4], inc[4];
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int vw = pass * 4 + wv, vt = vw * 64 + lane;
        loc[pass] = vt < nb ? bsum[vt] : 0.0;
        double m = -INFINITY;
        if (vt < nb) m = fmax(m, bmax[vt]);
        inc[pass] = wave_scan(loc[pass]);
        m = wave_max(m);
        if (lane == 63) shT[vw] = inc[pass];
        if (lane == 0) shM[vw] = m;
    }
    __syncthreads();
    double total = 0;
    for (int k = 0; k < WG / WAVE; ++k) total += shT[k];
    double gmax = shM[0];
    for (int k = 1; k < WG / WAVE; ++k) gmax = fmax(gmax, shM[k]);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int vw = pass * 4 + wv, vt = vw * 64 + lane;
        double se = 0;
        if (vt < nb) se += bmax[vt] > -INFINITY ? bse[vt] * exp(bmax[vt] - gmax) : 0.0;
        se = wave_sum(se);
        if (lane == 0) shS[vw] = se;
        if (vt == vb) {                                   // this workgroup's offset in the scan
            double woff = 0;
            for (int k = 0; k < vw; ++k) woff += shT[k];
            s_boff = woff + inc[pass] - loc[pass];
        }
    }
    __syncthreads();
    if (lt == 0) {
        double se = 0;
        for (int k = 0; k < WG / WAVE; ++k) se += shS[k];
        prm_s[PRM_S] = total;
        prm_s[PRM_LSE] = gmax + log(se);
    }
    __syncthreads();
    d3_norm_v(vb, lt, sh4, Pg, raw0, lb0, w, logw, prm_s, s_boff, cdf, first, counters);
}
__global__ __launch_bounds__(NB) void k3_norm_c(int Pg, int nb, double* w, const double* lbuf,
                                                float* __restrict__ logw, const double* bsum,
                                                const double* bmax, const double* bse, double* cdf,
                                                int32_t* __restrict__ first,
                                                unsigned long long* counters) {
    d3_norm_c(blockIdx.x, Pg, nb, w, lbuf, logw, bsum, bmax, bse, cdf, first, counters);
}
// The same launch carrying the ray phase of the map update (S3P:664-704) in workgroups nb ..,
// PF_RAYS_PER_WG rays each; the hits rode with the launch before, so they are complete.
__global__ __launch_bounds__(NB) void k3_norm_c_u(int Pg, int nb, double* w, const double* lbuf,
                                                  float* __restrict__ logw, const double* bsum,
                                                  const double* bmax, const double* bse, double* cdf,
                                                  int32_t* __restrict__ first,
                                                  unsigned long long* counters, PfUpdateWork uw) {
    if ((int)blockIdx.x >= nb) {
        d5_rays(((int)blockIdx.x - nb) * PF_RAYS_PER_WG + (int)threadIdx.x / PF_RAY_LANES, (int)threadIdx.x % PF_RAY_LANES, uw);
        return;
    }
    d3_norm_c(blockIdx.x, Pg, nb, w, lbuf, logw, bsum, bmax, bse, cdf, first, counters);
}

// K3 in ONE workgroup for up to PF_MID_MAX particles: the same four stages on the same 256-
// particle blocks (four at a time), barriers instead of kernel boundaries. Bit-identical to the
// four-kernel chain; one launch instead of four.
__global__ __launch_bounds__(WG) void k3_mid(const int2* __restrict__ scores, int Pg, double* w,
                                             double* lbuf, float* __restrict__ logw, double* cdf,
                                             int32_t* __restrict__ first,
                                             unsigned long long* counters, PfResampleScratch sc) {
    __shared__ double sh[WG / WAVE];
    __shared__ int shi[WG / WAVE];
    const int tid = threadIdx.x, grp = tid >> 8, lt = tid & (NB - 1);
    const int nb = (Pg + NB - 1) / NB;
    d3_min(scores, Pg, sc.prm, shi);
    __syncthreads();
    for (int vb0 = 0; vb0 < nb; vb0 += WG / NB)
        d3_raw(vb0 + grp, lt, sh + 4 * grp, scores, Pg, w, lbuf, sc.prm, sc.bsum, sc.bmax, sc.bse);
    __syncthreads();
    d3_combine(nb, sc.bsum, sc.bmax, sc.bse, sc.boff, sc.prm, sh);
    __syncthreads();
    for (int vb0 = 0; vb0 < nb; vb0 += WG / NB)
        d3_norm(vb0 + grp, lt, sh + 4 * grp, Pg, w, lbuf, logw, sc.prm, sc.boff, cdf, first, counters);
}

// ------------------------------------------------------------------------------------------
// K4 resample: S3P:757-821.
// ------------------------------------------------------------------------------------------
// S3P:772-780 as a binary search. For a non-decreasing CDF the reference's first j with
// cdf[j-1] < u < cdf[j] can only be j* = first j with cdf[j] > u, and it matches iff
// cdf[j*-1] != u. No match (boundary hit, u above the last boundary, u == 0, NaN): the draw
// keeps its own index and is counted.
__device__ inline int search_draw(const double* __restrict__ cdf, double ui, int i, int Pg,
                                  bool& hit, int& pos) {
    int lo = 1, hi = Pg + 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] > ui) hi = mid; else lo = mid + 1;
    }
    hit = lo <= Pg && cdf[lo - 1] < ui;
    pos = lo;
    return hit ? lo - 1 : i;
}
// Is the draw ui, found in the interval (cdf[lo-1], cdf[lo]) of the parallel CDF, too close to one of the
// two boundaries for the parallel and the sequential CDF to be sure to agree on it? `bound` is the unit
// roundoff with its safety factor (pf_api.hip). The sequential sum s_k = fl(s_(k-1) + w_k) differs from the
// exact one by sum_j d_j (s_(j-1) + w_j) with |d_j| <= 2^-53, i.e. by at most 2^-53 times the sum of its own
// partial sums; the parallel one by ~50 roundings of sums <= cdf[k]; the weights' division adds one each.
// First the cheap form of that (every partial sum <= cdf[k]: k * cdf[k]); a draw inside that band, about one
// in a million, is tested again with the partial sums bounded block by block (the 256 of block q by the CDF at
// the block's end): half the band on an even CDF, and half as many steps that pay for the sequential CDF.
// (bound < 0: the cheap form only, SLAM_PF_LOOSE_BAND=1, for comparing how often the two forms flag a step)
// (top: the block ends staged in LDS when the caller has them, top[q] == cdf[(q + 1) * NB]; a lane that walks
// thousands of them in global memory holds its whole workgroup up for hundreds of microseconds)
__device__ inline bool draw_is_ambiguous(double ui, int lo, const double* cdf, int Pg, double bound,
                                         const double* top = nullptr) {
    const bool loose_only = bound < 0.0;
    bound = fabs(bound);
    const double c = cdf[lo], below = ui - cdf[lo - 1], above = c - ui;
    const double extra = 64.0 + (double)(Pg >> 16);
    const double loose = bound * ((double)lo + extra) * c;
    if (!(below <= loose || above <= loose)) return false;
    if (loose_only) return true;
    const int b = (lo - 1) / NB;                            // block of boundary lo: indices b*NB+1 .. (b+1)*NB
    double ends = 0.0;
    if (top) { for (int q = 0; q < b; ++q) ends += top[q]; }
    else { for (int q = 0; q < b; ++q) ends += cdf[(q + 1) * NB]; }
    const double band = bound * ((double)NB * ends + ((double)(lo - b * NB) + extra) * c);
    return below <= band || above <= band;
}
// What follows the bisection for draw i with threshold ui: lo = first j in [1, Pg] with
// cdf[j] > ui, Pg + 1 when there is none.
__device__ inline void d_search_tail(int i, double ui, int lo, const double* cdf, int Pg, double bound,
                                     int32_t* draw, unsigned long long* counters, const double* top = nullptr) {
    const bool hit = lo <= Pg && cdf[lo - 1] < ui;
    const int a = hit ? lo - 1 : i;
    if (!hit) atomicAdd(&counters[0], 1ull);
    // a draw the two orders of summation might settle differently is counted and marked (~a): d_first searches it
    // again on the sequential-order CDF, which is rebuilt in between when the count is not zero
    const bool amb = bound != 0.0 && (!hit || draw_is_ambiguous(ui, lo, cdf, Pg, bound, top));
    if (amb) atomicAdd(&counters[1], 1ull);
    draw[i] = amb ? ~a : a;
}
// The same search in two levels: the CDF at the end of every 256-particle block staged in LDS
// (XTOP entries at most), bisected there, then the eight probes inside the block that holds the
// boundary. Same result as search_draw on a non-decreasing CDF; a draw close enough to an
// inversion of the parallel CDF (a few ulp) for the order of the probes to matter is inside the
// ambiguity band and searched again on the sequential-order CDF anyway. The probes that cost a
// round trip to L2 drop from ~10 of 14 (10 k particles) or 13 of 17 (80 k) to four.
#define XTOP 4096
// (use_tops: the weight kernels left those block ends behind the CDF, cdf[Pg+1+j]: one coalesced read per
// workgroup instead of nb loads 2 KB apart, which at a million particles was 0.6 ms of this kernel)
__global__ __launch_bounds__(NB) void k_search_t(const double* __restrict__ cdf,
                                                 const double* __restrict__ u, int Pg, double bound,
                                                 int32_t* __restrict__ draw,
                                                 unsigned long long* counters, int use_tops) {
    __shared__ double top[XTOP];
    const int nb = (Pg + NB - 1) / NB;                       // <= XTOP (pfk_resample)
    const int i = blockIdx.x * NB + threadIdx.x;
    const double ui = i < Pg ? u[i] : 0.0;
    for (int j = threadIdx.x; j < nb; j += NB) top[j] = use_tops ? cdf[Pg + 1 + j] : cdf[min((j + 1) * NB, Pg)];
    __syncthreads();
    if (i >= Pg) return;
    int lo = 0, hi = nb;                                    // first block whose last boundary is above ui
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (top[mid] > ui) hi = mid; else lo = mid + 1;
    }
    int pos = Pg + 1;
    if (lo < nb) {
        int a = lo * NB + 1, b = min((lo + 1) * NB, Pg);    // cdf[b] > ui
        while (a < b) {
            const int mid = (a + b) >> 1;
            if (cdf[mid] > ui) b = mid; else a = mid + 1;
        }
        pos = a;
    }
    d_search_tail(i, ui, pos, cdf, Pg, bound, draw, counters, top);
}
__device__ inline void d_search(int i, const double* cdf, const double* __restrict__ u, int Pg,
                                double bound, int32_t* draw, unsigned long long* counters) {
    if (i >= Pg) return;
    const double ui = u[i];
    bool hit; int lo;
    const int a = search_draw(cdf, ui, i, Pg, hit, lo);
    if (!hit) atomicAdd(&counters[0], 1ull);
    const bool amb = bound != 0.0 && (!hit || draw_is_ambiguous(ui, lo, cdf, Pg, bound));
    if (amb) atomicAdd(&counters[1], 1ull);
    draw[i] = amb ? ~a : a;
}
__global__ __launch_bounds__(NB) void k_search(const double* __restrict__ cdf,
                                               const double* __restrict__ u, int Pg, double bound,
                                               int32_t* __restrict__ draw,
                                               unsigned long long* counters) {
    d_search(blockIdx.x * NB + threadIdx.x, cdf, u, Pg, bound, draw, counters);
}

// first[a] = position of the first draw of ancestor a. When the FAST-order search flagged
// ambiguous draws, the CDF has been rebuilt in sequential order in between and the flagged draws
// are searched again here (no-hits recounted in counters[2]).
__device__ inline void d_first(int i, const double* cdf, const double* __restrict__ u, int32_t* draw,
                               int Pg, int32_t* first, unsigned long long* counters) {
    if (i >= Pg) return;
    int a = draw[i];
    if (a < 0) {                                   // marked by the search: only these are searched again (every draw
        bool hit; int lo;                          // that found no interval is among them, so counters[2] is the step's
        a = search_draw(cdf, u[i], i, Pg, hit, lo);   // count of those)
        draw[i] = a;
        if (!hit) atomicAdd(&counters[2], 1ull);
    }
    atomicMin(&first[a], i);
}
__global__ __launch_bounds__(NB) void k_first(const double* __restrict__ cdf,
                                              const double* __restrict__ u, int32_t* draw, int Pg,
                                              int32_t* first, unsigned long long* counters) {
    d_first(blockIdx.x * NB + threadIdx.x, cdf, u, draw, Pg, first, counters);
}

// S3P:782-802 in parallel form. A draw is a duplicate when its ancestor was drawn earlier
// (first[a] != i); slot j is free when nobody drew it. The k-th duplicate (draw order) goes to
// the k-th free slot (ascending); drawn slots keep themselves. Three stages: per-block counts,
// scan of the counts, ranked scatter; then the fill.
// (vb, lt, sd4/sf4: the 256-particle block, the thread in it, 4 ints of LDS each for the block)
__device__ inline void d4_count(int vb, int lt, int* sd4, int* sf4, const int32_t* draw, int Pg,
                                const int32_t* first, int2* bcnt) {
    const int i = vb * NB + lt, lane = lt & 63, wv4 = lt >> 6;
    const int isd = i < Pg && ld_agent(&first[draw[i]]) != i;
    const int isf = i < Pg && ld_agent(&first[i]) == NEVER_DRAWN;
    const int nd = __popcll(__ballot(isd)), nf = __popcll(__ballot(isf));
    __syncthreads();
    if (lane == 0) { sd4[wv4] = nd; sf4[wv4] = nf; }
    __syncthreads();
    if (lt == 0 && i < Pg) {
        int a = 0, b = 0;
        for (int k = 0; k < NB / WAVE; ++k) { a += sd4[k]; b += sf4[k]; }
        bcnt[vb] = make_int2(a, b);
    }
}
__global__ __launch_bounds__(NB) void k4_count(const int32_t* __restrict__ draw, int Pg,
                                               const int32_t* __restrict__ first, int2* bcnt) {
    __shared__ int sd[NB / WAVE], sf[NB / WAVE];
    d4_count(blockIdx.x, threadIdx.x, sd, sf, draw, Pg, first, bcnt);
}
__device__ inline void d4_scan(int nb, const int2* bcnt, int2* boff, int32_t* ndup_out, int* sd,
                               int* sf) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int seg = (nb + WG - 1) / WG;
    const int b = tid * seg, e = min(nb, b + seg);
    int ld = 0, lf = 0;
    for (int k = b; k < e; ++k) { ld += bcnt[k].x; lf += bcnt[k].y; }
    const int id = wave_scan_i(ld), jf = wave_scan_i(lf);
    __syncthreads();
    if (lane == 63) { sd[wv] = id; sf[wv] = jf; }
    __syncthreads();
    int od = 0, of = 0;
    for (int k = 0; k < wv; ++k) { od += sd[k]; of += sf[k]; }
    int rd = od + id - ld, rf = of + jf - lf;
    for (int k = b; k < e; ++k) { boff[k] = make_int2(rd, rf); rd += bcnt[k].x; rf += bcnt[k].y; }
    if (tid == WG - 1) *ndup_out = od + id;
}
__global__ __launch_bounds__(WG) void k4_scan(int nb, const int2* __restrict__ bcnt, int2* boff,
                                              int32_t* ndup_out) {
    __shared__ int sd[WG / WAVE], sf[WG / WAVE];
    d4_scan(nb, bcnt, boff, ndup_out, sd, sf);
}
// (a, isd, isf: this thread's draw and its duplicate / free-slot flags, loaded by the caller)
__device__ inline void d4_scatter_v(int vb, int lt, int* sd4, int* sf4, int a, int isd, int isf, int Pg,
                                    const int2 boff_vb, int32_t* dup_anc, int32_t* slot_anc) {
    const int i = vb * NB + lt, lane = lt & 63, wv4 = lt >> 6;
    const unsigned long long md = __ballot(isd), mf = __ballot(isf);
    const unsigned long long below = (1ull << lane) - 1ull;
    __syncthreads();
    if (lane == 0) { sd4[wv4] = __popcll(md); sf4[wv4] = __popcll(mf); }
    __syncthreads();
    if (i >= Pg) return;                                    // after the last barrier
    int od = boff_vb.x, of = boff_vb.y;
    for (int k = 0; k < wv4; ++k) { od += sd4[k]; of += sf4[k]; }
    if (isd) dup_anc[od + __popcll(md & below)] = a;
    // a free slot takes the duplicate of its own rank; the apply stage resolves -1-rank -> dup_anc[rank]
    if (isf) slot_anc[i] = -1 - (of + __popcll(mf & below));
    else slot_anc[i] = i;
}
__device__ inline void d4_scatter(int vb, int lt, int* sd4, int* sf4, const int32_t* draw, int Pg,
                                  const int32_t* first, const int2* boff, int32_t* dup_anc,
                                  int32_t* slot_anc) {
    const int i = vb * NB + lt;
    const int a = i < Pg ? draw[i] : 0;
    const int isd = i < Pg && ld_agent(&first[a]) != i;
    const int isf = i < Pg && ld_agent(&first[i]) == NEVER_DRAWN;
    d4_scatter_v(vb, lt, sd4, sf4, a, isd, isf, Pg, vb * NB < Pg ? boff[vb] : make_int2(0, 0), dup_anc, slot_anc);
}
__global__ __launch_bounds__(NB) void k4_scatter(const int32_t* __restrict__ draw, int Pg,
                                                 const int32_t* __restrict__ first,
                                                 const int2* __restrict__ boff,
                                                 int32_t* __restrict__ dup_anc,
                                                 int32_t* __restrict__ free_slot,
                                                 int32_t* __restrict__ slot_anc) {
    __shared__ int sd[NB / WAVE], sf[NB / WAVE];
    (void)free_slot;
    d4_scatter(blockIdx.x, threadIdx.x, sd, sf, draw, Pg, first, boff, dup_anc, slot_anc);
}
// k4_scatter with the scan of the block counts inside (integers: any order of addition gives the
// same offsets), for up to 1024 blocks.
__global__ __launch_bounds__(NB) void k4_scatter_s(const int32_t* __restrict__ draw, int Pg,
                                                   const int32_t* __restrict__ first,
                                                   const int2* __restrict__ bcnt,
                                                   int32_t* __restrict__ dup_anc,
                                                   int32_t* __restrict__ slot_anc) {
    __shared__ int sd[NB / WAVE], sf[NB / WAVE];
    __shared__ int2 my_off;
    const int lt = threadIdx.x, lane = lt & 63, wv = lt >> 6, vb = blockIdx.x;
    // this thread's draw and flags first: their latency overlaps the scan of the block counts
    const int i0 = vb * NB + lt;
    const int a0 = i0 < Pg ? draw[i0] : 0;
    const int isd0 = i0 < Pg && ld_agent(&first[a0]) != i0;
    const int isf0 = i0 < Pg && ld_agent(&first[i0]) == NEVER_DRAWN;
    int a = 0, b = 0;
    for (int k = lt; k < vb; k += NB) { const int2 c = bcnt[k]; a += c.x; b += c.y; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, WAVE); b += __shfl_xor(b, o, WAVE); }
    if (lane == 0) { sd[wv] = a; sf[wv] = b; }
    __syncthreads();
    if (lt == 0) my_off = make_int2(sd[0] + sd[1] + sd[2] + sd[3], sf[0] + sf[1] + sf[2] + sf[3]);
    __syncthreads();
    d4_scatter_v(vb, lt, sd, sf, a0, isd0, isf0, Pg, my_off, dup_anc, slot_anc);
}
// k_first with the rare sequential-order CDF inside: when draws were ambiguous every workgroup
// rebuilds the CDF itself (all write the same values) before it searches again.
__global__ __launch_bounds__(NB) void k_first_c(double* cdf, const double* __restrict__ u,
                                                int32_t* draw, int Pg, const double* w,
                                                int32_t* first, unsigned long long* counters) {
    __shared__ double tile[TILE];
    if (ld_agent(&counters[1]) != 0) {                     // uniform over the whole grid
        seq_cdf(w, Pg, cdf, tile);
        __syncthreads();
    }
    d_first(blockIdx.x * NB + threadIdx.x, cdf, u, draw, Pg, first, counters);
}
// S3P:805-812 + the first stage of S3P:814-820. Sources are slots that keep themselves, so the
// in-place parallel copy reads nothing that this stage writes. Per block: the first index of
// the block's strict maximum of the NEW weights above 0.
// POSE: the handle holds all particles (one GPU), the pose rows move in the same stage.
template <bool POSE>
__device__ inline void d4_apply(int vb, int lt, double* sv4, int* si4, int32_t* slot_anc,
                                const int32_t* dup_anc, int Pg, double* w, float* logw, double* pose,
                                const double* pose_src,
                                double* bbest_v, int32_t* bbest_i, int32_t* order_hist = nullptr,
                                int32_t* order_keys = nullptr, PfOrderParams op = PfOrderParams{},
                                int* lds_hist = nullptr) {
    const int g = vb * NB + lt, lane = lt & 63, wv4 = lt >> 6;
    double v = 0.0;
    int idx = -1;
    if (g < Pg) {
        int a = slot_anc[g];
        if (a < 0) { a = dup_anc[-1 - a]; slot_anc[g] = a; }   // k-th free slot <- k-th duplicate
        if (POSE && order_hist) {
            // slot g ends this step with its ancestor's pose: count its key for the next K1's
            // lane order now (the first stage of pfk_order)
            const int key = order_key(pose_src, Pg, a, op);
            order_keys[g] = key;
            if (lds_hist) atomicAdd(&lds_hist[key], 1);     // flushed by the caller, one add per key and workgroup
            else atomicAdd(&order_hist[key], 1);
        }
        v = w[a];
        if (a != g) { w[g] = v; logw[g] = logw[a]; }
        // pose_src != pose: the predicted poses still sit in the caller's block (pf_predict_device
        // did not copy them); every slot takes its ancestor's row from there
        if (POSE && (a != g || pose_src != pose)) {
#pragma unroll
            for (int k = 0; k < PF_POSE_ROWS; ++k) pose[(size_t)k * Pg + g] = pose_src[(size_t)k * Pg + a];
        }
        if (v > 0.0) idx = g; else v = 0.0;                 // NaN and <= 0 never win (S3P:816)
    }
    // wave arg-max: larger value wins, ties go to the lower index
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_down(v, o, WAVE);
        const int oi = __shfl_down(idx, o, WAVE);
        if (oi >= 0 && (idx < 0 || ov > v || (ov == v && oi < idx))) { v = ov; idx = oi; }
    }
    __syncthreads();
    if (lane == 0) { sv4[wv4] = v; si4[wv4] = idx; }
    __syncthreads();
    if (lt == 0 && g < Pg) {
        for (int k = 1; k < NB / WAVE; ++k)
            if (si4[k] >= 0 && (idx < 0 || sv4[k] > v || (sv4[k] == v && si4[k] < idx))) { v = sv4[k]; idx = si4[k]; }
        bbest_v[vb] = v;
        bbest_i[vb] = idx;
    }
}
// host_hdr (may be null): the header of a pinned result slot; bbest_v / bbest_i then point into
// the same slot and the host finishes the arg-max (pfk_result_best).
template <bool POSE>
__global__ __launch_bounds__(NB) void k4_apply(int32_t* __restrict__ slot_anc,
                                               const int32_t* __restrict__ dup_anc, int Pg,
                                               double* w, float* logw, double* pose,
                                               const double* pose_src,
                                               double* bbest_v, int32_t* bbest_i,
                                               int32_t* order_hist, int32_t* order_keys,
                                               PfOrderParams op, const unsigned long long* counters,
                                               unsigned long long* host_hdr) {
    if (host_hdr && blockIdx.x == 0 && threadIdx.x == 0) {
        host_hdr[0] = ld_agent(&counters[0]); host_hdr[1] = ld_agent(&counters[1]);
        host_hdr[2] = ld_agent(&counters[2]);
        host_hdr[3] = PF_RESULT_BLOCKS | (unsigned long long)gridDim.x;
    }
    __shared__ double sv[NB / WAVE];
    __shared__ int si[NB / WAVE];
    // the key counts go through a workgroup-private histogram: a resampled cloud puts hundreds of
    // slots into the same few buckets, and that many global adds on one address serialise
    __shared__ int lh[POSE ? 4096 : 1];
    const bool count = POSE && order_hist;                  // 4096-bucket keys only (pf_api.hip)
    if (count) {
#pragma unroll
        for (int k = 0; k < 16; ++k) lh[(threadIdx.x + NB * k) & (POSE ? 4095 : 0)] = 0;
        __syncthreads();
    }
    d4_apply<POSE>(blockIdx.x, threadIdx.x, sv, si, slot_anc, dup_anc, Pg, w, logw, pose, pose_src, bbest_v,
                   bbest_i, order_hist, order_keys, op, count ? lh : nullptr);
    if (count) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int b = (threadIdx.x + NB * k) & (POSE ? 4095 : 0);
            const int c = lh[b];
            if (c) atomicAdd(&order_hist[b], c);
        }
    }
}
// host_out (may be null): pinned host memory the last stage of K4 writes the step's results to
// ({no-hit draws, ambiguous draws, no-hit draws after the re-search, best}); the stream
// synchronisation the caller does anyway makes them visible, no copy is enqueued.
__device__ inline void d4_argmax(int nb, const double* bbest_v, const int32_t* bbest_i,
                                 int32_t* best_inout, double* sv, int* si,
                                 const unsigned long long* counters = nullptr,
                                 unsigned long long* host_out = nullptr) {
    const int tid = threadIdx.x;
    double v = 0.0; int idx = -1;
    for (int k = tid; k < nb; k += WG) {
        const int oi = bbest_i[k]; const double ov = bbest_v[k];
        if (oi >= 0 && (idx < 0 || ov > v || (ov == v && oi < idx))) { v = ov; idx = oi; }
    }
    // larger value wins, ties go to the lower index: associative and commutative, any tree will do
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_down(v, o, WAVE);
        const int oi = __shfl_down(idx, o, WAVE);
        if (oi >= 0 && (idx < 0 || ov > v || (ov == v && oi < idx))) { v = ov; idx = oi; }
    }
    __syncthreads();
    if ((tid & 63) == 0) { sv[tid >> 6] = v; si[tid >> 6] = idx; }
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < WG / WAVE; ++k)
            if (si[k] >= 0 && (idx < 0 || sv[k] > v || (sv[k] == v && si[k] < idx))) { v = sv[k]; idx = si[k]; }
        si[0] = idx;
        int b = *best_inout;
        if (si[0] >= 0) *best_inout = b = si[0];            // otherwise the previous index stays
        if (host_out) {
            host_out[0] = ld_agent(&counters[0]); host_out[1] = ld_agent(&counters[1]);
            host_out[2] = ld_agent(&counters[2]); host_out[3] = (unsigned long long)(unsigned)b;
        }
    }
}
__global__ __launch_bounds__(WG) void k4_argmax(int nb, const double* __restrict__ bbest_v,
                                                const int32_t* __restrict__ bbest_i,
                                                int32_t* best_inout,
                                                const unsigned long long* counters,
                                                unsigned long long* host_out) {
    __shared__ double sv[WG];
    __shared__ int si[WG];
    d4_argmax(nb, bbest_v, bbest_i, best_inout, sv, si, counters, host_out);
}

// K4 in ONE workgroup for up to PF_MID_MAX particles on one GPU: the same stages on the same
// 256-particle blocks, barriers instead of kernel boundaries (one launch instead of eight).
// tile: the LDS the sequential-order CDF needs when draws were ambiguous; it doubles as the
// arg-max scratch.
__global__ __launch_bounds__(WG) void k4_mid(double* cdf, const double* __restrict__ u, int Pg,
                                             double bound, double* w, float* logw, double* pose,
                                             int32_t* draw, int32_t* first, int32_t* dup_anc,
                                             int32_t* slot_anc, int32_t* best,
                                             unsigned long long* counters, PfResampleScratch sc,
                                             unsigned long long* host_out, int32_t* order_hist,
                                             int32_t* order_keys, PfOrderParams op) {
    __shared__ double tile[TILE];
    __shared__ int sd[WG / WAVE], sf[WG / WAVE];
    __shared__ double sv[WG / WAVE];
    const int tid = threadIdx.x, grp = tid >> 8, lt = tid & (NB - 1);
    const int nb = (Pg + NB - 1) / NB;
    for (int i = tid; i < Pg; i += WG) d_search(i, cdf, u, Pg, bound, draw, counters);
    __syncthreads();
    if (bound > 0.0 && ld_agent(&counters[1]) != 0) {       // workgroup-uniform
        seq_cdf(w, Pg, cdf, tile);
        __syncthreads();
    }
    for (int i = tid; i < Pg; i += WG) d_first(i, cdf, u, draw, Pg, first, counters);
    __syncthreads();
    for (int vb0 = 0; vb0 < nb; vb0 += WG / NB)
        d4_count(vb0 + grp, lt, sd + 4 * grp, sf + 4 * grp, draw, Pg, first, sc.bcnt);
    __syncthreads();
    d4_scan(nb, sc.bcnt, sc.boff2, sc.ndup, sd, sf);
    __syncthreads();
    for (int vb0 = 0; vb0 < nb; vb0 += WG / NB)
        d4_scatter(vb0 + grp, lt, sd + 4 * grp, sf + 4 * grp, draw, Pg, first, sc.boff2, dup_anc, slot_anc);
    __syncthreads();
    for (int vb0 = 0; vb0 < nb; vb0 += WG / NB)
        d4_apply<true>(vb0 + grp, lt, sv + 4 * grp, sd + 4 * grp, slot_anc, dup_anc, Pg, w, logw, pose, pose,
                       sc.bbest_v, sc.bbest_i, order_hist, order_keys, op);
    __syncthreads();
    d4_argmax(nb, sc.bbest_v, sc.bbest_i, best, tile, (int*)(tile + WG), counters, host_out);
}

// (vb: the 256-particle block of the shard; lh: 4096 ints of LDS; dup_anc: a slot the fill stage has
// not resolved yet still holds -1 - rank, which names dup_anc[rank])
__device__ __forceinline__ void d_apply_pose(int vb, int* lh, const int32_t* __restrict__ slot_anc,
                                             const int32_t* __restrict__ dup_anc, int P, int offset,
                                             double* pose, const double* __restrict__ gathered,
                                             int copy_all, int32_t* order_hist, int32_t* order_keys,
                                             PfOrderParams op) {
    if (order_hist) {
#pragma unroll
        for (int k = 0; k < 16; ++k) lh[threadIdx.x + NB * k] = 0;
        __syncthreads();
    }
    const int i = vb * NB + threadIdx.x;
    if (i < P) {
        const int g = offset + i;
        int a = __hip_atomic_load(&slot_anc[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a < 0) a = dup_anc[-1 - a];
        const double* src = pose;                           // rows of the ancestor: [9][P] block
        int ai = i;
        // copy_all: this rank's own rows are not in `pose` yet either (pf_predict_device left them
        // in the caller's block, which went into the all-gather): every slot reads the gathered blocks
        if (a != g || copy_all) {
            if (gathered) {
                const int r = a / P;
                ai = a - r * P;
                src = gathered + (size_t)r * PF_POSE_ROWS * P;
            } else {
                ai = a - offset;
            }
        }
        if (order_hist) {
            const int key = order_key(src, P, ai, op);
            order_keys[i] = key;
            atomicAdd(&lh[key], 1);
        }
        if (a != g || copy_all) {
#pragma unroll
            for (int k = 0; k < PF_POSE_ROWS; ++k) pose[(size_t)k * P + i] = src[(size_t)k * P + ai];
        }
    }
    if (order_hist) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int b = threadIdx.x + NB * k, c = lh[b];
            if (c) atomicAdd(&order_hist[b], c);
        }
    }
}
// The fill stage of a shard in one launch: workgroups 0 .. nb-1 permute the weights of ALL slots and
// find their blocks' bests (d4_apply<false>, redundantly on every rank), workgroups nb .. write the
// local shard's pose rows from the gathered blocks and (order_hist != null: 4096-bucket keys)
// count K1's lane-order keys of those poses, as the one-GPU fill stage does. The two parts share
// only slot_anc[], which the second reads whether or not the first has resolved an entry yet.
__global__ __launch_bounds__(NB) void k4_apply_shard(int32_t* slot_anc, const int32_t* __restrict__ dup_anc,
                                                     int Pg, double* w, float* logw, double* bbest_v,
                                                     int32_t* bbest_i, const unsigned long long* counters,
                                                     unsigned long long* host_hdr, int nb, int P,
                                                     int offset, double* pose,
                                                     const double* __restrict__ gathered, int copy_all,
                                                     int32_t* order_hist, int32_t* order_keys,
                                                     PfOrderParams op) {
    __shared__ double sv[NB / WAVE];
    __shared__ int si[NB / WAVE];
    __shared__ int lh[4096];
    if ((int)blockIdx.x >= nb) {
        d_apply_pose((int)blockIdx.x - nb, lh, slot_anc, dup_anc, P, offset, pose, gathered, copy_all, order_hist,
                     order_keys, op);
        return;
    }
    if (host_hdr && blockIdx.x == 0 && threadIdx.x == 0) {
        host_hdr[0] = ld_agent(&counters[0]); host_hdr[1] = ld_agent(&counters[1]);
        host_hdr[2] = ld_agent(&counters[2]);
        host_hdr[3] = PF_RESULT_BLOCKS | (unsigned long long)nb;
    }
    d4_apply<false>(blockIdx.x, threadIdx.x, sv, si, slot_anc, dup_anc, Pg, w, logw, nullptr, nullptr, bbest_v,
                    bbest_i, nullptr, nullptr, op, nullptr);
}
// Per-particle map slabs (S3P:810 copies the 7.2 MB map with the struct): 16 B per lane.
__global__ __launch_bounds__(NB) void k_apply_maps(const int32_t* __restrict__ slot_anc, int P,
                                                   int offset, uint8_t* maps,
                                                   unsigned long long stride,
                                                   unsigned long long bytes, int vshard) {
    const int i = blockIdx.x;
    const int a = slot_anc[offset + i];
    if (a == offset + i) return;
    if (a < offset || a >= offset + P) return;               // another rank's slab: pf_resample receives it
    // (test switch SLAM_PF_VIRTUAL_SHARDS: the shard is treated as blocks of vshard slots that exchange their
    // slabs through the library's send/receive groups like ranks do)
    if (vshard > 0 && (a - offset) / vshard != i / vshard) return;
    const uint4* src = (const uint4*)(maps + stride * (unsigned long long)(a - offset));
    uint4* dst = (uint4*)(maps + stride * (unsigned long long)i);
    const unsigned long long n16 = (bytes + 15) / 16;        // slabs are padded to 256 B
    for (unsigned long long k = (unsigned long long)blockIdx.y * NB + threadIdx.x; k < n16;
         k += (unsigned long long)gridDim.y * NB)
        dst[k] = src[k];
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static inline int nblocks(int n) { return (n + NB - 1) / NB; }
// SLAM_PF_CHAIN=1: every stage its own launch (the form the folded kernels are tested against)
static inline bool chain_only() {
    const char* e = getenv("SLAM_PF_CHAIN");
    return e && atoi(e) != 0;
}
// SLAM_PF_MID=1: K3 and K4 as ONE workgroup each (k3_mid, k4_mid) instead of the multi-kernel
// chains. Off by default: one CU cannot hide the latency of 10k binary searches, measured 67 us
// (K3) and 172 us (K4) against 18 and 29 us for the chains at 10k particles. Kept because the
// two forms are the same arithmetic on the same 256-particle blocks and a test compares them
// bit for bit, which pins the fixed-shape claim the multi-GPU path relies on.
static inline bool mid_ok(int Pg) {
    const char* e = getenv("SLAM_PF_MID");
    return Pg <= PF_MID_MAX && e && atoi(e) != 0;
}

bool pfk_weights_carries(int Pg, bool exact, const int32_t* bmin) {
    (void)bmin;
    return !exact && !mid_ok(Pg) && !chain_only() && nblocks(Pg) <= WG;
}
hipError_t pfk_weights(hipStream_t st, const int2* scores, int Pg, double* w, float* logw,
                       double* cdf, double* scratch, bool exact, int32_t* first,
                       unsigned long long* counters, PfResampleScratch sc, const int32_t* bmin,
                       const PfUpdateWork* carry, int n_bmin) {
    if (n_bmin <= 0) n_bmin = (Pg + 63) / 64;
    if (carry && pfk_weights_carries(Pg, exact, bmin)) {
        const int nb = nblocks(Pg);
        if (bmin) {
            hipLaunchKernelGGL(k3_raw_m_u, dim3(nb + (carry->M + NB - 1) / NB), dim3(NB), 0, st, scores, Pg, w,
                               scratch, bmin, n_bmin, sc.bsum, sc.bmax, sc.bse, nb, *carry);
        } else {
            hipLaunchKernelGGL(k3_min, dim3(1), dim3(WG), 0, st, scores, Pg, sc.prm);
            hipLaunchKernelGGL(k3_raw_u, dim3(nb + (carry->M + NB - 1) / NB), dim3(NB), 0, st, scores, Pg, w,
                               scratch, sc.prm, sc.bsum, sc.bmax, sc.bse, nb, *carry);
        }
        hipLaunchKernelGGL(k3_norm_c_u, dim3(nb + (carry->R + PF_RAYS_PER_WG - 1) / PF_RAYS_PER_WG), dim3(NB), 0, st, Pg, nb, w, scratch, logw,
                           sc.bsum, sc.bmax, sc.bse, cdf, first, counters, *carry);
        return hipGetLastError();
    }
    if (exact) {
        hipLaunchKernelGGL(k_weights_exact, dim3(1), dim3(WG), 0, st, scores, Pg, w, logw, cdf, scratch,
                           first, counters);
        return hipGetLastError();
    }
    const int nb = nblocks(Pg);
    if (mid_ok(Pg)) {
        hipLaunchKernelGGL(k3_mid, dim3(1), dim3(WG), 0, st, scores, Pg, w, scratch, logw, cdf, first,
                           counters, sc);
        return hipGetLastError();
    }
    if (bmin && !chain_only()) {
        hipLaunchKernelGGL(k3_raw_m, dim3(nb), dim3(NB), 0, st, scores, Pg, w, scratch, bmin, n_bmin, sc.bsum,
                           sc.bmax, sc.bse);
    } else {
        hipLaunchKernelGGL(k3_min, dim3(1), dim3(WG), 0, st, scores, Pg, sc.prm);
        hipLaunchKernelGGL(k3_raw, dim3(nb), dim3(NB), 0, st, scores, Pg, w, scratch, sc.prm, sc.bsum,
                           sc.bmax, sc.bse);
    }
    if (nb <= WG && !chain_only()) {
        hipLaunchKernelGGL(k3_norm_c, dim3(nb), dim3(NB), 0, st, Pg, nb, w, scratch, logw, sc.bsum, sc.bmax,
                           sc.bse, cdf, first, counters);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k3_combine, dim3(1), dim3(WG), 0, st, nb, sc.bsum, sc.bmax, sc.bse, sc.boff, sc.prm);
    hipLaunchKernelGGL(k3_norm, dim3(nb), dim3(NB), 0, st, Pg, w, scratch, logw, sc.prm, sc.boff, cdf,
                       first, counters);
    return hipGetLastError();
}

// counters: unsigned long long [3] = {nohit, ambiguous, nohit after the re-search}.
hipError_t pfk_resample(hipStream_t st, const double* cdf, const double* u, int Pg, int P,
                        int offset, double bound, double* w, float* logw, double* pose,
                        const double* pose_src, const double* pose_gathered, uint8_t* maps, unsigned long long map_stride,
                        unsigned long long map_bytes, int32_t* draw, int32_t* first,
                        int32_t* dup_anc, int32_t* free_slot, int32_t* slot_anc, int32_t* best,
                        unsigned long long* counters, PfResampleScratch sc,
                        unsigned long long* host_out, int32_t* order_hist, int32_t* order_keys,
                        PfOrderParams op, bool tops_valid, int map_vshard) {
    const int nb = nblocks(Pg);
    // counters[0..2] and first[] were initialised by the weight kernels of this step
    if (!pose_src) pose_src = pose;
    if (mid_ok(Pg) && P == Pg && !pose_gathered && !(maps && map_stride)) {
        if (pose_src != pose)
            (void)hipMemcpyAsync(pose, pose_src, sizeof(double) * PF_POSE_ROWS * (size_t)P, hipMemcpyDeviceToDevice, st);
        hipLaunchKernelGGL(k4_mid, dim3(1), dim3(WG), 0, st, (double*)cdf, u, Pg, bound, w, logw, pose,
                           draw, first, dup_anc, slot_anc, best, counters, sc, host_out, order_hist,
                           order_keys, op);
        return hipGetLastError();
    }
    const bool folded = nb <= WG && !chain_only();
    static const bool loose_band = getenv("SLAM_PF_LOOSE_BAND") && atoi(getenv("SLAM_PF_LOOSE_BAND")) != 0;
    const double sb = loose_band ? -bound : bound;          // (development knob: the first, cheaper band alone)
    if (nb <= XTOP && !chain_only())
        hipLaunchKernelGGL(k_search_t, dim3(nb), dim3(NB), 0, st, cdf, u, Pg, sb, draw, counters, tops_valid ? 1 : 0);
    else
        hipLaunchKernelGGL(k_search, dim3(nb), dim3(NB), 0, st, cdf, u, Pg, sb, draw, counters);
    // draws that sit within the parallel CDF's rounding error of a boundary are settled on the
    // sequential-order CDF of the same weights (nothing happens when there are none). Such a draw
    // turns up about once in 1/(8e-17 Pg^3) steps: up to PF_FOLD_CDF_MAX particles that is rare
    // enough for every workgroup of k_first_c to rebuild the CDF by the one-lane chain when it
    // happens; beyond, it happens every few steps (every step from a few hundred thousand particles
    // on), and one workgroup rebuilds it with the parallel form in a launch of its own
    if (folded && Pg <= PF_FOLD_CDF_MAX) {
        hipLaunchKernelGGL(k_first_c, dim3(nb), dim3(NB), 0, st, (double*)cdf, u, draw, Pg, w, first, counters);
    } else {
        if (bound > 0.0 && Pg >= PF_MULTI_CDF_MIN && sc.segs) {
            // the sequential-order CDF by many workgroups (each of the three returns at once when no draw was flagged)
            const int nseg = cdfx_nseg(Pg);
            CdfSeg* segs = (CdfSeg*)sc.segs;
            hipLaunchKernelGGL(k_cdfx_summ, dim3(nseg - 1), dim3(WG), 0, st, w, Pg, cdf, segs, counters);
            hipLaunchKernelGGL(k_cdfx_walk, dim3(1), dim3(WG), 0, st, w, Pg, (double*)cdf, segs, counters);
            hipLaunchKernelGGL(k_cdfx_fill, dim3(nseg - 1), dim3(WG), 0, st, w, Pg, (double*)cdf, segs, counters);
        } else if (bound > 0.0)
            hipLaunchKernelGGL(k_cdf_exact_if, dim3(1), dim3(WG), 0, st, w, Pg, (double*)cdf, counters);
        hipLaunchKernelGGL(k_first, dim3(nb), dim3(NB), 0, st, cdf, u, draw, Pg, first, counters);
    }
    hipLaunchKernelGGL(k4_count, dim3(nb), dim3(NB), 0, st, draw, Pg, first, sc.bcnt);
    if (folded) {
        hipLaunchKernelGGL(k4_scatter_s, dim3(nb), dim3(NB), 0, st, draw, Pg, first, sc.bcnt, dup_anc, slot_anc);
    } else {
        hipLaunchKernelGGL(k4_scan, dim3(1), dim3(WG), 0, st, nb, sc.bcnt, sc.boff2, sc.ndup);
        hipLaunchKernelGGL(k4_scatter, dim3(nb), dim3(NB), 0, st, draw, Pg, first, sc.boff2, dup_anc,
                           free_slot, slot_anc);
    }
    // default form: the blocks' bests go straight into the pinned result slot, no arg-max launch
    const bool host_argmax = folded && host_out;
    double* bv = host_argmax ? (double*)(host_out + PF_RESULT_HEADER) : sc.bbest_v;
    int32_t* bi = host_argmax ? (int32_t*)(bv + nb) : sc.bbest_i;
    unsigned long long* hdr = host_argmax ? host_out : nullptr;
    if (P == Pg && !pose_gathered && chain_only()) {
        hipLaunchKernelGGL(k4_apply<true>, dim3(nb), dim3(NB), 0, st, slot_anc, dup_anc, Pg, w, logw,
                           pose, pose_src, bv, bi, order_hist, order_keys, op, counters, hdr);
    } else if (P == Pg && !pose_gathered) {
        // one GPU: the same two-part launch as a shard's (weights and bests in workgroups 0 .. nb-1,
        // pose rows and key count in nb ..): a thread's chain of dependent loads is half as long
        // as with both parts in one thread (7.2 against 8.9 us at 10 k particles)
        const bool lazy = pose_src != pose;                 // rows still in the caller's block
        hipLaunchKernelGGL(k4_apply_shard, dim3(2 * nb), dim3(NB), 0, st, slot_anc, dup_anc, Pg, w, logw,
                           bv, bi, counters, hdr, nb, P, 0, pose, lazy ? pose_src : nullptr, lazy ? 1 : 0,
                           order_hist, order_keys, op);
    } else {
        hipLaunchKernelGGL(k4_apply_shard, dim3(nb + nblocks(P)), dim3(NB), 0, st, slot_anc, dup_anc, Pg, w, logw,
                           bv, bi, counters, hdr, nb, P, offset, pose, pose_gathered,
                           (pose_gathered && pose_src != pose) ? 1 : 0, order_hist, order_keys, op);
    }
    if (maps && map_stride) {
        unsigned long long n16 = (map_bytes + 15) / 16;
        int gy = (int)((n16 + 256ull * 8 - 1) / (256ull * 8));
        if (gy < 1) gy = 1;
        if (gy > 64) gy = 64;
        hipLaunchKernelGGL(k_apply_maps, dim3(P, gy), dim3(NB), 0, st, slot_anc, P, offset, maps,
                           map_stride, map_bytes, map_vshard);
    }
    if (!host_argmax)
        hipLaunchKernelGGL(k4_argmax, dim3(1), dim3(WG), 0, st, nb, sc.bbest_v, sc.bbest_i, best, counters,
                           host_out);
    return hipGetLastError();
}

int32_t pfk_result_best(const unsigned long long* slot, int32_t prev) {
    if (!(slot[3] & PF_RESULT_BLOCKS)) return (int32_t)(slot[3] & 0xffffffffull);   // the arg-max kernel ran
    const size_t nb = (size_t)(slot[3] & 0xffffffffull);
    const double* bv = (const double*)(slot + PF_RESULT_HEADER);
    const int32_t* bi = (const int32_t*)(bv + nb);
    double v = 0.0; int32_t idx = -1;
    for (size_t k = 0; k < nb; ++k)                         // blocks ascend in index: ties keep the earlier one
        if (bi[k] >= 0 && (idx < 0 || bv[k] > v)) { v = bv[k]; idx = bi[k]; }
    return idx >= 0 ? idx : prev;                           // S3P:816: nothing above 0 keeps the previous index
}

// resample() called on its own (S3P:757-821 on whatever weights the particles hold): what the weight
// kernels otherwise leave for K4 -- the CDF in the reference's sequential order (S3P:763-767), the
// first-occurrence table and the counters.
__global__ __launch_bounds__(NB) void k4_init(int Pg, int32_t* __restrict__ first, unsigned long long* counters) {
    const int g = blockIdx.x * NB + threadIdx.x;
    if (g < Pg) first[g] = NEVER_DRAWN;
    if (g < 3) counters[g] = 0;
}
hipError_t pfk_resample_prepare(hipStream_t st, const double* w, int Pg, double* cdf, int32_t* first,
                                unsigned long long* counters) {
    hipLaunchKernelGGL(k_cdf_of, dim3(1), dim3(WG), 0, st, w, Pg, cdf, Pg > 2048 ? 1 : 0);
    hipLaunchKernelGGL(k4_init, dim3(nblocks(Pg)), dim3(NB), 0, st, Pg, first, counters);
    return hipGetLastError();
}

size_t pfk_cdf_scratch_bytes(int n) { const size_t nw = ((size_t)n + XSEG - 1) / XSEG; return ((size_t)cdfx_nseg(n) + 1) * sizeof(CdfSeg) + (2 * nw + 2) * sizeof(double); }
hipError_t pfk_cdf_of(hipStream_t st, const double* w, int n, double* cdf, int how, void* scratch) {
    if (how == 2) {                                             // the many-workgroup form (scratch: pfk_cdf_scratch_bytes)
        const int nseg = cdfx_nseg(n), nw = (n + XSEG - 1) / XSEG;
        CdfSeg* segs = (CdfSeg*)scratch;
        double* sums = (double*)(segs + nseg + 1);
        double* bounds = sums + nw;
        // an approximate CDF first (plain sums), as the weight kernels leave one in the resampler
        hipLaunchKernelGGL(k_cdfx_segsum, dim3(nw), dim3(WG), 0, st, w, n, sums);
        hipLaunchKernelGGL(k_cdfx_bounds, dim3(1), dim3(64), 0, st, sums, nw, bounds);
        hipLaunchKernelGGL(k_cdfx_approx, dim3(nw), dim3(WG), 0, st, w, n, bounds, cdf);
        hipLaunchKernelGGL(k_cdfx_summ, dim3(nseg - 1), dim3(WG), 0, st, w, n, cdf, segs, (const unsigned long long*)nullptr);
        hipLaunchKernelGGL(k_cdfx_walk, dim3(1), dim3(WG), 0, st, w, n, cdf, segs, (const unsigned long long*)nullptr);
        hipLaunchKernelGGL(k_cdfx_fill, dim3(nseg - 1), dim3(WG), 0, st, w, n, cdf, segs, (const unsigned long long*)nullptr);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_cdf_of, dim3(1), dim3(WG), 0, st, w, n, cdf, how);
    return hipGetLastError();
}
