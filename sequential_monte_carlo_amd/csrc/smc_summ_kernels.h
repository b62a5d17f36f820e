// smc_summ_kernels.h -- summaries of the current weights of filters of ANY size (several segments): weighted quantiles of one
// state coordinate and mean / variance of every coordinate (README.md:41,51 quantile(x, ...);
// examples/inflation_example.jl:45-46 quantile(x, weights(w), p) and the weighted variance), five launches on the handle's
// stream, nothing waits for the host - per step inside the multi-step calls (smc_set_summaries) and behind smc_get_quantiles /
// smc_get_moments.  (Single-segment filters: resident_summaries in smc_resident.h, inside the resident kernels.)
//
// Integer definition of a quantile (include/smc_hip.h smc_get_quantiles; the oracle SORTS): W_i = q_i >> sh_b the weight of
// particle i in the units of the combined segment table, T = floor(p * sum W); result = the smallest value v with
// sum{W_i : x_i <= v} > T, values ordered by the IEEE total order of their bits.  Any selection that narrows by a MONOTONE map
// of x finds that same particle, so:
//   k_ms_range    per segment: the range of the values that carry weight (and, for the moments, the segment's partial sums of
//                 w x and w x^2 with the dense weights w = q 2^-48-dk / (Dtot 2^(SH-48)) in a fixed order)
//   k_ms_hist     histogram of the integer weights over MS_BINS equal-width VALUE bins (LDS per workgroup, then 64-bit atomics)
//   k_ms_pick     one workgroup per filter: total weight, every level's target, bin and the weight below it; the moments' sums
//   k_ms_collect  the particles of the chosen bins into a candidate list per level
//   k_ms_select   one workgroup per (level, filter): 8-pass radix select on the order-preserving key among the candidates in LDS -
//                 or, when the range is unusable (a non-finite value carrying weight, all values alike, an overflowing width) or
//                 a bin holds more than MS_CAP particles, the same radix select streaming over ALL particles of the filter
//                 (slow, rare, always right).
// Every cross-workgroup combination is an integer sum or a maximum: the quantiles are bit-identical to the oracle's sort.  The
// moments are sums of doubles in a fixed order (segment by segment), equal to the oracle's to rounding.
#pragma once
#include "smc_kernels.h"

namespace smc {

constexpr int MS_BINS = 4096, MS_CAP = 3072, MS_THREADS = 256, MS_SEL_THREADS = 512;   // (MS_CAP: 48 KB of candidates in LDS)
constexpr int MS_STREAM = 1024;   // threads of the kernels that stream over the particles (k_ms_range, k_ms_hist, k_ms_collect)
constexpr int64_t MS_TWO_LEVEL = (int64_t)1 << 21;   // filters of more particles cut the chosen bin a second time (4096 x 4096 value bins)
constexpr int MS_STASH = 64;      // matches a workgroup of k_ms_collect keeps in LDS per level before it asks for room in the list

// scratch of one handle, in 8-byte words (zeroed once when allocated; the kernels leave hist and cnt zeroed again)
struct MsScratch {
    double* rng;                 // [ntheta][parts][2]     largest value / largest negated value carrying weight (+inf: non-finite), per
                                 //                        workgroup of k_ms_range (parts <= nseg: room for nseg)
    double* mpart;               // [ntheta][d][parts][2]  partial sums of w x, w x^2
    unsigned long long* hist;    // [ntheta][MS_BINS]
    uint64_t* hdr;               // [ntheta][4]            lo, scale (doubles) | usable (0 / 1) | total weight
    uint64_t* st;                // [ntheta][QMAX][6]      bin | weight below | target | state (0 fallback, 1 binned, 2 no weight) | sub-bin | -
    unsigned long long* hist2;   // [ntheta][QMAX][MS_BINS]  second level (filters beyond MS_TWO_LEVEL particles): the chosen bin cut again
    unsigned* cnt;               // [ntheta][QMAX]         candidates collected per level
    uint64_t* cand;              // [ntheta][QMAX][MS_CAP][2]   (key, W)
};
__host__ __device__ inline size_t ms_words(size_t nth, size_t nseg, size_t d) {
    return nth * nseg * 2 + nth * d * nseg * 2 + nth * MS_BINS + nth * 4 + nth * QMAX * 6 + nth * QMAX * MS_BINS + nth * QMAX + nth * QMAX * MS_CAP * 2;
}
__host__ __device__ inline MsScratch ms_carve(uint64_t* base, size_t nth, size_t nseg, size_t d) {
    MsScratch s;
    s.rng = (double*)base; base += nth * nseg * 2;
    s.mpart = (double*)base; base += nth * d * nseg * 2;
    s.hist = (unsigned long long*)base; base += nth * MS_BINS;
    s.hdr = base; base += nth * 4;
    s.st = base; base += nth * QMAX * 6;
    s.hist2 = (unsigned long long*)base; base += nth * QMAX * MS_BINS;
    s.cnt = (unsigned*)base; base += nth * QMAX;
    s.cand = base;
    return s;
}

// integer weight (table units) and dense-weight scale of segment b of filter th
struct MsSeg { int sh; double sc; };
__device__ __forceinline__ MsSeg ms_segment(const FilterView& v, int cur, int th, int b) {
    const double K = v.last_K[th], kb = v.segk[cur][(size_t)th * v.nseg + b], dk = K - kb;
    MsSeg s;
    s.sh = seg_shift(K, kb, v.SH);
    s.sc = (dk >= 0.0 && dk < 900.0) ? pow2i(-48 - (int)dk) : 0.0;
    return s;
}
__device__ __forceinline__ int ms_bin(double x, double lo, double scale) {
    const int b = (int)((x - lo) * scale);   // monotone in x: differences, products and truncation all are
    return b < MS_BINS - 1 ? b : MS_BINS - 1;
}
// ... and the position inside that bin, cut into MS_BINS again (monotone in x among the values of one bin)
__device__ __forceinline__ int ms_bin2(double x, double lo, double scale, int& sub) {
    const double f = (x - lo) * scale;
    int b = (int)f;
    b = b < MS_BINS - 1 ? b : MS_BINS - 1;
    const int s2 = (int)((f - (double)b) * (double)MS_BINS);
    sub = s2 < MS_BINS - 1 ? s2 : MS_BINS - 1;
    return b;
}

// The particles of the segments this workgroup owns (grid.x workgroups share the nseg segments of filter th, consecutive segments
// each), two per thread and load (16-byte loads of C and of state coordinate comp): f(i, q, x, segment constants)
template <class F>
__device__ __forceinline__ void ms_for_each(const FilterView& v, int cur, int th, int comp, F f) {
    const int G = gridDim.x, spw = (v.nseg + G - 1) / G, b0 = blockIdx.x * spw, b1 = b0 + spw < v.nseg ? b0 + spw : v.nseg;
    const uint64_t* C = v.C[cur] + (size_t)th * v.npad;
    const double* x = v.x[cur] + ((size_t)comp * v.ntheta + th) * v.npad;
    for (int b = b0; b < b1; ++b) {
        const MsSeg sg = ms_segment(v, cur, th, b);
        const int64_t base = (int64_t)b * v.seg, left = v.n - base;
        const int m = left < v.seg ? (int)left : v.seg;
        const ulonglong2* C2 = reinterpret_cast<const ulonglong2*>(C + base);
        const double2* X2 = reinterpret_cast<const double2*>(x + base);
        for (int p = threadIdx.x; 2 * p < m; p += blockDim.x) {
            const ulonglong2 c = C2[p];
            const uint64_t prev = p ? C[base + 2 * p - 1] : 0;
            const double2 xx = X2[p];
            f(base + 2 * p, (uint64_t)(c.x - prev), xx.x, sg);
            if (2 * p + 1 < m) f(base + 2 * p + 1, (uint64_t)(c.y - c.x), xx.y, sg);
        }
    }
}

// grid (G, ntheta): per workgroup the range of the weighted values and the partial sums of the moments
__global__ __launch_bounds__(MS_STREAM) void k_ms_range(FilterView v, int cur, int d, MsScratch ms) {
    constexpr int NW = MS_STREAM / WAVE;
    __shared__ double red[8][NW];
    const int g = blockIdx.x, G = gridDim.x, th = blockIdx.y, tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const uint64_t Dtot = v.last_D[th];
    const double Dd = (double)Dtot * pow2i(v.SH - 48);
    double vhi = -inf(), vlo = -inf(), mo[3] = {0.0, 0.0, 0.0}, mo2[3] = {0.0, 0.0, 0.0};
    bool odd = false;
    const bool want_q = v.sum_np != 0, want_m = v.sum_mom != 0;
    ms_for_each(v, cur, th, v.sum_comp, [&](int64_t i, uint64_t q, double xq, const MsSeg& sg) {
        if (want_q) {
            const uint64_t W = sg.sh < 64 ? q >> sg.sh : 0;
            if (W) {
                odd = odd || !(fabs(xq) < inf());
                vhi = xq > vhi ? xq : vhi;
                vlo = -xq > vlo ? -xq : vlo;
            }
        }
        if (want_m) {
            const double w = Dtot ? ((double)q * sg.sc) / Dd : 0.0;
            for (int c = 0; c < d; ++c) {
                const double x = c == v.sum_comp ? xq : v.x[cur][((size_t)c * v.ntheta + th) * v.npad + i];
                mo[c] += w * x;
                mo2[c] += w * x * x;
            }
        }
    });
    if (want_q) {
        odd = __ballot(odd) != 0;
        vhi = wave_max_f64(odd ? 0.0 : vhi);
        vlo = wave_max_f64(odd ? 0.0 : vlo);
        if (lane == 0) { red[0][wave] = odd ? inf() : vhi; red[1][wave] = vlo; }
    }
    if (want_m)
        for (int c = 0; c < d; ++c) {
            const double a = wave_sum_f64(mo[c]), a2 = wave_sum_f64(mo2[c]);
            if (lane == 0) { red[2 + 2 * c][wave] = a; red[3 + 2 * c][wave] = a2; }
        }
    __syncthreads();
    if (tid == 0) {
        if (want_q) {
            double h = red[0][0], l = red[1][0];
            for (int w = 1; w < NW; ++w) { h = red[0][w] > h ? red[0][w] : h; l = red[1][w] > l ? red[1][w] : l; }
            ms.rng[((size_t)th * G + g) * 2] = h;
            ms.rng[((size_t)th * G + g) * 2 + 1] = l;
        }
        if (want_m)
            for (int c = 0; c < d; ++c) {
                double a = 0.0, a2 = 0.0;
                for (int w = 0; w < NW; ++w) { a += red[2 + 2 * c][w]; a2 += red[3 + 2 * c][w]; }
                ms.mpart[(((size_t)th * d + c) * G + g) * 2] = a;
                ms.mpart[(((size_t)th * d + c) * G + g) * 2 + 1] = a2;
            }
    }
}

// the filter's value range from the partial ranges of k_ms_range's `nparts` workgroups (every workgroup that needs it computes the
// same numbers): lo, scale, usable.  red: 2 * (blockDim.x / 64) doubles of LDS; one barrier inside.
__device__ __forceinline__ void ms_range_of(const MsScratch& ms, int th, int nparts, double* red, double& lo, double& scale, bool& usable) {
    const int NW = blockDim.x / WAVE, tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    double vhi = -inf(), vlo = -inf();
    for (int g = tid; g < nparts; g += blockDim.x) {
        const double h = ms.rng[((size_t)th * nparts + g) * 2], l = ms.rng[((size_t)th * nparts + g) * 2 + 1];
        vhi = h > vhi ? h : vhi;
        vlo = l > vlo ? l : vlo;
    }
    vhi = wave_max_f64(vhi);
    vlo = wave_max_f64(vlo);
    if (lane == 0) { red[wave] = vhi; red[NW + wave] = vlo; }
    __syncthreads();
    vhi = red[0]; vlo = red[NW];
    for (int w = 1; w < NW; ++w) { vhi = red[w] > vhi ? red[w] : vhi; vlo = red[NW + w] > vlo ? red[NW + w] : vlo; }
    const double hi = vhi;
    lo = -vlo;
    // (any positive factor gives a monotone map: the hardware's approximate reciprocal will do)
    scale = (double)MS_BINS * __builtin_amdgcn_rcp(hi - lo);
    usable = fabs(lo) < inf() && fabs(hi) < inf() && lo < hi && scale < inf();
}

// grid (G, ntheta): few, large workgroups - each flushes its LDS histogram with one device-scope atomic per occupied bin
__global__ __launch_bounds__(MS_STREAM) void k_ms_hist(FilterView v, int cur, int nparts, MsScratch ms) {
    __shared__ unsigned long long lh[MS_BINS];
    __shared__ double red[2 * (MS_STREAM / WAVE)];
    const int th = blockIdx.y, tid = threadIdx.x;
    double lo, scale;
    bool usable;
    ms_range_of(ms, th, nparts, red, lo, scale, usable);
    if (blockIdx.x == 0 && tid == 0) {
        ms.hdr[(size_t)th * 4] = d2bits(lo);
        ms.hdr[(size_t)th * 4 + 1] = d2bits(scale);
        ms.hdr[(size_t)th * 4 + 2] = usable ? 1 : 0;
    }
    if (!usable) return;   // (workgroup-uniform, the same in every workgroup of the filter)
    for (int i = tid; i < MS_BINS; i += MS_STREAM) lh[i] = 0;
    __syncthreads();
    ms_for_each(v, cur, th, v.sum_comp, [&](int64_t, uint64_t q, double x, const MsSeg& sg) {
        const uint64_t W = sg.sh < 64 ? q >> sg.sh : 0;
        if (W) atomicAdd(&lh[ms_bin(x, lo, scale)], (unsigned long long)W);
    });
    __syncthreads();
    for (int i = tid; i < MS_BINS; i += MS_STREAM)
        if (lh[i]) atomicAdd(&ms.hist[(size_t)th * MS_BINS + i], lh[i]);
}

// grid (ntheta): the filter's histogram -> total, targets, bins; the moments' sums.  mean / var: [d][ntheta] rows of the output
__global__ __launch_bounds__(MS_THREADS) void k_ms_pick(FilterView v, int d, int nparts, MsScratch ms, double* q_out, double* mean, double* var) {
    constexpr int NW = MS_THREADS / WAVE, PER = MS_BINS / MS_THREADS;
    __shared__ uint64_t wt[NW];
    __shared__ double red[2][NW];
    const int th = blockIdx.x, tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    if (v.sum_mom)
        for (int c = 0; c < d; ++c) {   // fixed order: thread t takes the partial sums t, t + 256, ...; lanes, then waves, in order
            double a = 0.0, a2 = 0.0;
            for (int g = tid; g < nparts; g += MS_THREADS) {
                a += ms.mpart[(((size_t)th * d + c) * nparts + g) * 2];
                a2 += ms.mpart[(((size_t)th * d + c) * nparts + g) * 2 + 1];
            }
            a = wave_sum_f64(a);
            a2 = wave_sum_f64(a2);
            __syncthreads();
            if (lane == 0) { red[0][wave] = a; red[1][wave] = a2; }
            __syncthreads();
            if (tid == 0) {
                double s = 0.0, s2 = 0.0;
                for (int w = 0; w < NW; ++w) { s += red[0][w]; s2 += red[1][w]; }
                mean[(size_t)c * v.ntheta + th] = s;
                var[(size_t)c * v.ntheta + th] = s2 - s * s;
            }
        }
    const int nq = v.sum_np;
    if (nq == 0) return;
    uint64_t* st = ms.st + (size_t)th * QMAX * 6;
    if (!ms.hdr[(size_t)th * 4 + 2]) {   // no usable range: the select kernel streams over the whole filter
        if (tid < nq) st[tid * 6 + 3] = 0;
        return;
    }
    unsigned long long* hb = ms.hist + (size_t)th * MS_BINS + (size_t)tid * PER;
    uint64_t h[PER], sum = 0;
#pragma unroll
    for (int t = 0; t < PER; ++t) { h[t] = hb[t]; hb[t] = 0; sum += h[t]; }   // (zeroed for the next use)
    const uint64_t incl_w = wave_incl_scan(sum, lane);
    if (lane == WAVE - 1) wt[wave] = incl_w;
    __syncthreads();
    uint64_t off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { off += w < wave ? wt[w] : 0; tot += wt[w]; }
    const uint64_t excl = off + incl_w - sum;
    if (tid == 0) ms.hdr[(size_t)th * 4 + 3] = tot;
    for (int j = 0; j < nq; ++j) {
        if (!tot) {   // every weight is zero: no quantile
            if (tid == 0) { st[j * 6 + 3] = 2; q_out[(size_t)th * nq + j] = bits2d(0x7ff8000000000000ULL); }
            continue;
        }
        const uint64_t target = __umul64hi(v.sum_p64[j], tot);
        if (sum && excl <= target && target < excl + sum) {   // exactly one thread
            uint64_t run = excl;
#pragma unroll
            for (int t = 0; t < PER; ++t) {
                if (h[t] && run <= target && target < run + h[t]) {
                    st[j * 6] = (uint64_t)(tid * PER + t);
                    st[j * 6 + 1] = run;
                    st[j * 6 + 2] = target;
                    st[j * 6 + 3] = 1;
                }
                run += h[t];
            }
        }
    }
}

// second level (filters beyond MS_TWO_LEVEL particles): the particles of every level's chosen bin, a few thousand, cut into MS_BINS
// sub-bins - few enough for device-scope atomics straight into the level's histogram.  grid (G, ntheta)
__global__ __launch_bounds__(MS_STREAM) void k_ms_hist2(FilterView v, int cur, MsScratch ms) {
    const int th = blockIdx.y, nq = v.sum_np;
    if (!ms.hdr[(size_t)th * 4 + 2] || !ms.hdr[(size_t)th * 4 + 3]) return;
    const double lo = bits2d(ms.hdr[(size_t)th * 4]), scale = bits2d(ms.hdr[(size_t)th * 4 + 1]);
    int sb[QMAX];
#pragma unroll
    for (int j = 0; j < QMAX; ++j) sb[j] = j < nq ? (int)ms.st[((size_t)th * QMAX + j) * 6] : -1;
    ms_for_each(v, cur, th, v.sum_comp, [&](int64_t, uint64_t q, double x, const MsSeg& sg) {
        const uint64_t W = sg.sh < 64 ? q >> sg.sh : 0;
        if (!W) return;
        int sub;
        const int bin = ms_bin2(x, lo, scale, sub);
#pragma unroll
        for (int l = 0; l < QMAX; ++l)
            if (bin == sb[l]) atomicAdd(&ms.hist2[((size_t)th * QMAX + l) * MS_BINS + sub], (unsigned long long)W);
    });
}
// grid (nq, ntheta): the sub-bin the level's target falls in, and the weight below it
__global__ __launch_bounds__(MS_THREADS) void k_ms_pick2(FilterView v, MsScratch ms) {
    constexpr int NW = MS_THREADS / WAVE, PER = MS_BINS / MS_THREADS;
    __shared__ uint64_t wt[NW];
    const int jq = blockIdx.x, th = blockIdx.y, tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    uint64_t* st = ms.st + ((size_t)th * QMAX + jq) * 6;
    if (!ms.hdr[(size_t)th * 4 + 2] || st[3] != 1) return;   // no usable range / no weight: nothing was binned
    unsigned long long* hb = ms.hist2 + ((size_t)th * QMAX + jq) * MS_BINS + (size_t)tid * PER;
    uint64_t h[PER], sum = 0;
#pragma unroll
    for (int t = 0; t < PER; ++t) { h[t] = hb[t]; hb[t] = 0; sum += h[t]; }   // (zeroed for the next use)
    const uint64_t incl_w = wave_incl_scan(sum, lane);
    if (lane == WAVE - 1) wt[wave] = incl_w;
    const uint64_t below = st[1], target = st[2];
    __syncthreads();
    uint64_t off = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) off += w < wave ? wt[w] : 0;
    const uint64_t excl = below + off + incl_w - sum;
    if (sum && excl <= target && target < excl + sum) {   // exactly one thread (the bin's weight holds the target)
        uint64_t run = excl;
#pragma unroll
        for (int t = 0; t < PER; ++t) {
            if (h[t] && run <= target && target < run + h[t]) { st[4] = (uint64_t)(tid * PER + t); st[1] = run; }
            run += h[t];
        }
    }
}

// grid (G, ntheta): a workgroup keeps its matches in LDS (MS_STASH per level) and asks for room in the level's list ONCE - the
// list's counter is one address all workgroups of a filter share; a match beyond the stash asks by itself
template <bool TWO>
__global__ __launch_bounds__(MS_STREAM) void k_ms_collect(FilterView v, int cur, MsScratch ms) {
    __shared__ uint64_t stash[QMAX][MS_STASH][2];
    __shared__ unsigned ln[QMAX], lbase[QMAX];
    const int th = blockIdx.y, tid = threadIdx.x, nq = v.sum_np;
    if (!ms.hdr[(size_t)th * 4 + 2] || !ms.hdr[(size_t)th * 4 + 3]) return;
    const double lo = bits2d(ms.hdr[(size_t)th * 4]), scale = bits2d(ms.hdr[(size_t)th * 4 + 1]);
    int sb[QMAX];
#pragma unroll
    for (int j = 0; j < QMAX; ++j) sb[j] = j < nq ? (int)ms.st[((size_t)th * QMAX + j) * 6] : -1;
    int ssub[QMAX];
#pragma unroll
    for (int j = 0; j < QMAX; ++j) ssub[j] = (TWO && j < nq) ? (int)ms.st[((size_t)th * QMAX + j) * 6 + 4] : 0;
    if (tid < QMAX) ln[tid] = 0;
    __syncthreads();
    ms_for_each(v, cur, th, v.sum_comp, [&](int64_t, uint64_t q, double x, const MsSeg& sg) {
        const uint64_t W = sg.sh < 64 ? q >> sg.sh : 0;
        if (!W) return;
        int sub = 0;
        const int bin = TWO ? ms_bin2(x, lo, scale, sub) : ms_bin(x, lo, scale);
#pragma unroll
        for (int l = 0; l < QMAX; ++l)
            if (bin == sb[l] && (!TWO || sub == ssub[l])) {
                const unsigned k = atomicAdd(&ln[l], 1u);
                if (k < (unsigned)MS_STASH) {
                    stash[l][k][0] = order_key(x);
                    stash[l][k][1] = W;
                } else {
                    const unsigned idx = atomicAdd(&ms.cnt[(size_t)th * QMAX + l], 1u);
                    if (idx < (unsigned)MS_CAP) {
                        uint64_t* cd = ms.cand + (((size_t)th * QMAX + l) * MS_CAP + idx) * 2;
                        cd[0] = order_key(x);
                        cd[1] = W;
                    }
                }
            }
    });
    __syncthreads();
    if (tid < nq) {
        const unsigned k = ln[tid] < (unsigned)MS_STASH ? ln[tid] : (unsigned)MS_STASH;
        lbase[tid] = k ? atomicAdd(&ms.cnt[(size_t)th * QMAX + tid], k) : 0u;
    }
    __syncthreads();
    for (int l = 0; l < nq; ++l) {
        const unsigned k = ln[l] < (unsigned)MS_STASH ? ln[l] : (unsigned)MS_STASH;
        if ((unsigned)tid < k && lbase[l] + tid < (unsigned)MS_CAP) {
            uint64_t* cd = ms.cand + (((size_t)th * QMAX + l) * MS_CAP + lbase[l] + tid) * 2;
            cd[0] = stash[l][tid][0];
            cd[1] = stash[l][tid][1];
        }
    }
}

// grid (nq, ntheta): the quantile of one level of one filter
__global__ __launch_bounds__(MS_SEL_THREADS) void k_ms_select(FilterView v, int cur, MsScratch ms, double* q_out) {
    constexpr int NW = MS_SEL_THREADS / WAVE;
    __shared__ uint64_t ck[MS_CAP], cw[MS_CAP];
    __shared__ unsigned long long lh[256];
    __shared__ uint64_t wt[NW], ck_mx[NW];
    __shared__ uint64_t sel[2];   // prefix, below
    const int jq = blockIdx.x, th = blockIdx.y, tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE, nq = v.sum_np;
    const uint64_t* st = ms.st + ((size_t)th * QMAX + jq) * 6;
    const uint64_t state = st[3];
    const unsigned c = ms.cnt[(size_t)th * QMAX + jq];
    __syncthreads();
    if (tid == 0) ms.cnt[(size_t)th * QMAX + jq] = 0;   // ready for the next use
    if (state == 2) return;                               // no weight: k_ms_pick wrote the NaN
    const bool listed = state == 1 && c <= (unsigned)MS_CAP;   // workgroup-uniform
    uint64_t target, below0;
    const uint64_t* C = v.C[cur] + (size_t)th * v.npad;
    const double* x = v.x[cur] + ((size_t)v.sum_comp * v.ntheta + th) * v.npad;
    if (listed) {
        const uint64_t* cd = ms.cand + ((size_t)th * QMAX + jq) * MS_CAP * 2;
        for (unsigned i = tid; i < c; i += MS_SEL_THREADS) { ck[i] = cd[2 * i]; cw[i] = cd[2 * i + 1]; }
        target = st[2];
        below0 = st[1];
    } else {   // the whole filter: its total weight first (integer sum: any order)
        uint64_t s = 0;
        for (int64_t i = tid; i < v.n; i += MS_SEL_THREADS) {
            const int b = (int)(i / v.seg), j = (int)(i % v.seg);
            const int sh = ms_segment(v, cur, th, b).sh;
            const uint64_t q = C[i] - (j ? C[i - 1] : 0);
            s += sh < 64 ? q >> sh : 0;
        }
        s = wave_sum(s);
        if (lane == 0) wt[wave] = s;
        __syncthreads();
        uint64_t tot = 0;
        for (int w = 0; w < NW; ++w) tot += wt[w];
        __syncthreads();
        if (!tot) {
            if (tid == 0) q_out[(size_t)th * nq + jq] = bits2d(0x7ff8000000000000ULL);
            return;
        }
        target = __umul64hi(v.sum_p64[jq], tot);
        below0 = 0;
    }
    // the candidates of one value bin share the leading bytes of their keys (sign, exponent, the first mantissa bits): those passes
    // would put every weight on ONE histogram word - skipped; all keys alike: that key is the quantile
    int first_pass = 0;
    uint64_t prefix0 = 0;
    if (listed) {
        uint64_t kmn = ~0ULL, kmx = 0;
        for (unsigned i = tid; i < c; i += MS_SEL_THREADS) { kmn = ck[i] < kmn ? ck[i] : kmn; kmx = ck[i] > kmx ? ck[i] : kmx; }
        __syncthreads();   // (ck, cw written above are visible; wt is free)
        for (int dd = WAVE / 2; dd >= 1; dd >>= 1) {
            const uint64_t a = __shfl_xor((unsigned long long)kmn, dd, WAVE), b = __shfl_xor((unsigned long long)kmx, dd, WAVE);
            kmn = a < kmn ? a : kmn;
            kmx = b > kmx ? b : kmx;
        }
        if (lane == 0) { wt[wave] = kmn; ck_mx[wave] = kmx; }
        __syncthreads();
        kmn = wt[0]; kmx = ck_mx[0];
        for (int w = 1; w < NW; ++w) { kmn = wt[w] < kmn ? wt[w] : kmn; kmx = ck_mx[w] > kmx ? ck_mx[w] : kmx; }
        if (kmn == kmx) {
            if (tid == 0) q_out[(size_t)th * nq + jq] = key_value(kmn);
            return;
        }
        first_pass = __builtin_clzll(kmn ^ kmx) / 8;
        prefix0 = first_pass ? kmn >> (64 - 8 * first_pass) : 0;
    }
    if (tid == 0) { sel[0] = prefix0; sel[1] = below0; }
    if (tid < 256) lh[tid] = 0;
    __syncthreads();
    for (int pass = first_pass; pass < 8; ++pass) {
        const int hs = 64 - 8 * pass;   // the prefix is key >> hs (pass > 0)
        const uint64_t pref = sel[0];
        if (listed) {
            for (unsigned i = tid; i < c; i += MS_SEL_THREADS)
                if (pass == 0 || (ck[i] >> hs) == pref) atomicAdd(&lh[(int)((ck[i] >> (hs - 8)) & 255)], (unsigned long long)cw[i]);
        } else {
            for (int64_t i = tid; i < v.n; i += MS_SEL_THREADS) {
                const int b = (int)(i / v.seg), j = (int)(i % v.seg);
                const int sh = ms_segment(v, cur, th, b).sh;
                const uint64_t q = C[i] - (j ? C[i - 1] : 0);
                const uint64_t W = sh < 64 ? q >> sh : 0;
                if (!W) continue;
                const uint64_t key = order_key(x[i]);
                if (pass == 0 || (key >> hs) == pref) atomicAdd(&lh[(int)((key >> (hs - 8)) & 255)], (unsigned long long)W);
            }
        }
        __syncthreads();
        if (wave == 0) {   // lane l owns the digits 4 l .. 4 l + 3
            const uint64_t h0 = lh[4 * lane], h1 = lh[4 * lane + 1], h2 = lh[4 * lane + 2], h3 = lh[4 * lane + 3];
            lh[4 * lane] = lh[4 * lane + 1] = lh[4 * lane + 2] = lh[4 * lane + 3] = 0;
            const uint64_t sum = h0 + h1 + h2 + h3;
            uint64_t run = sel[1] + wave_incl_scan(sum, lane) - sum;
            const uint64_t hh[4] = {h0, h1, h2, h3};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (hh[t] && run <= target && target < run + hh[t]) {   // exactly one digit of one lane
                    const uint64_t np_ = (pref << 8) | (uint64_t)(4 * lane + t);
                    sel[0] = np_;
                    sel[1] = run;
                    if (pass == 7) q_out[(size_t)th * nq + jq] = key_value(np_);
                }
                run += hh[t];
            }
        }
        __syncthreads();
    }
}

}  // namespace smc
