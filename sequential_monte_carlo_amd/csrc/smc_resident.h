// smc_resident.h -- k_resident: log_likelihood(N, y, model) (particles.jl:132-147) for the
// batched callers (smc_samplers.jl:112-121,223-229): ONE workgroup per filter (theta), the whole
// T loop inside the kernel, particle states and weight prefix sums resident in LDS (160 KiB/CU
// on gfx950).  Requires a single segment (n_x <= seg).  Bit-identical to the k_init/k_step path.
#pragma once
#include "smc_kernels.h"

namespace smc {

// per-step record written by the loop, turned into (logmu_t, ess_t) after it
struct StepRec { double kb; uint64_t S, hi, lo; };

// LDS of the per-step summaries (SUMM): weight histograms [nq][256] of the radix select, its state (prefix, below, target per
// level), the partial sums of the moments; and of the usual (value-binned) selection: one histogram of SUMM_BINS bins shared by
// the levels, SUMM_CAND candidates (key, weight) per level, the chosen bin and the candidate count per level, the waves' key range
constexpr int SUMM_BINS = 1024, SUMM_CAND = 64;
__host__ __device__ inline size_t summary_lds_words(int nq) {
    return (size_t)nq * 256 + 3 * QMAX + 2 * 3 * 16 + SUMM_BINS + (size_t)nq * SUMM_CAND * 2 + 2 * QMAX + 2 * 16;
}
template <int MODEL>
__host__ __device__ inline size_t resident_lds_bytes(int seg, int threads, int np, int sum_nq = -1) {
    return (size_t)lds_padded_len(seg) * 8 * (1 + model_dim<MODEL>::value) + scr_words(threads, np) * 8 +
           (sum_nq >= 0 ? summary_lds_words(sum_nq) * 8 : 0);
}

// Per-step filtered summaries of ONE single-segment filter whose weights (inclusive fixed-point sums Cs) and states xs sit in
// LDS (both padded by lds_pad): what the README loop computes on the host after every bootstrap_filter! (README.md:41,51
// quantile(x, ...); examples/inflation_example.jl:45-46 quantile(x, weights(w), p) and the weighted variance).
//   quantiles: the definition of smc_get_quantiles (inverse of the weighted empirical CDF in the integer weights: smallest
//              value v with sum{q_i : x_i <= v} > floor(p S)).  Any selection that narrows by a MONOTONE map of x finds that same
//              particle, so the usual path bins by value instead of by key digits (the top key bytes - sign and exponent - of a
//              filter's cloud are all alike: a thousand LDS atomics on two or three addresses per pass): the range of the keys
//              that carry weight (one wave reduction), ONE histogram of 1024 equal-width value bins of the 64-bit integer weights
//              for all levels (order-free LDS atomics, about one particle per bin), one wave per level picks its bin, the few
//              particles of that bin are collected and ranked inside the wave - four barriers.  Whenever that does not apply
//              (a non-finite value carrying weight, an empty or overflowing range, more than 64 particles in a chosen bin) the
//              step falls back to the 8-pass radix select on the order-preserving key, one wave per level picking the digit;
//   moments:   sum w x and sum w x^2 with the dense weights w = q 2^-48 / (S 2^-48) (smc_get_state's w).
// Called by every thread of the workgroup after the step's last barrier; ends with the histograms cleared for the next step.
template <int THREADS, int NP, int D>
__device__ __forceinline__ void resident_summaries(const FilterView& v, int th, int64_t row, const uint64_t* Cs, const double* xs, int SEGP,
                                                   uint64_t S, uint64_t* sm) {
    constexpr int NW = THREADS / WAVE, NQ = 2 * NP;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int nq = v.sum_np;
    unsigned long long* hist = (unsigned long long*)sm;   // [nq][256]
    uint64_t* st_prefix = sm + (size_t)nq * 256;          // [QMAX]
    uint64_t* st_below = st_prefix + QMAX;                // [QMAX]
    uint64_t* st_target = st_below + QMAX;                // [QMAX]
    double* red = (double*)(st_target + QMAX);            // [2][3][16]
    uint64_t q[NQ];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int i0 = 2 * (tid + k * THREADS), pp = lds_pad(i0);
        const uint64_t c0 = Cs[pp], c1 = Cs[pp + 1], prev = i0 ? Cs[lds_pad(i0 - 1)] : 0;
        q[2 * k] = c0 - prev;
        q[2 * k + 1] = c1 - c0;
    }
    if (v.sum_mom) {
        const double Dd = (double)S * pow2i(-48), sc = pow2i(-48);
#pragma unroll
        for (int c = 0; c < D; ++c) {
            double m = 0.0, m2 = 0.0;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int pp = lds_pad(2 * (tid + k * THREADS));
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const double x = xs[c * SEGP + pp + j];
                    const double w = S ? ((double)q[2 * k + j] * sc) / Dd : 0.0;
                    m += w * x;
                    m2 += w * x * x;
                }
            }
            m = wave_sum_f64(m);
            m2 = wave_sum_f64(m2);
            if (lane == 0) { red[(0 * 3 + c) * 16 + wave] = m; red[(1 * 3 + c) * 16 + wave] = m2; }
        }
    }
    if (nq > 0 && S == 0) {   // collapsed filter: no quantile (workgroup-uniform)
        if (tid < nq) v.sum_q[((size_t)row * v.ntheta + th) * nq + tid] = bits2d(0x7ff8000000000000ULL);
    } else if (nq > 0) {
        uint64_t key[NQ];
        double xv[NQ];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int pp = lds_pad(2 * (tid + k * THREADS));
            xv[2 * k] = xs[v.sum_comp * SEGP + pp];
            xv[2 * k + 1] = xs[v.sum_comp * SEGP + pp + 1];
            key[2 * k] = order_key(xv[2 * k]);
            key[2 * k + 1] = order_key(xv[2 * k + 1]);
        }
        // ---- the usual path: selection through equal-width value bins ----
        unsigned long long* hbin = (unsigned long long*)(red + 2 * 3 * 16);   // [SUMM_BINS]
        uint64_t* cand = (uint64_t*)(hbin + SUMM_BINS);                        // [nq][SUMM_CAND][2]
        uint64_t* st_bin = cand + (size_t)nq * SUMM_CAND * 2;                  // [QMAX]
        unsigned* cnt = (unsigned*)(st_bin + QMAX);                            // [QMAX] (two per word)
        uint64_t* kmm = st_bin + 2 * QMAX;                                     // [2][16]
        double vhi = -inf(), vlo = -inf();   // the largest value and the largest negated value that carry weight
        bool odd = false;                     // ... or a non-finite one
#pragma unroll
        for (int i = 0; i < NQ; ++i)
            if (q[i]) {
                odd = odd || !(fabs(xv[i]) < inf());
                vhi = xv[i] > vhi ? xv[i] : vhi;
                vlo = -xv[i] > vlo ? -xv[i] : vlo;
            }
        odd = __ballot(odd) != 0;
        vhi = wave_max_f64(odd ? 0.0 : vhi);
        vlo = wave_max_f64(odd ? 0.0 : vlo);
        if (lane == 0) { kmm[wave] = d2bits(odd ? inf() : vhi); kmm[16 + wave] = d2bits(vlo); }
        if (tid < nq) cnt[tid] = 0;
        __syncthreads();
        vhi = vlo = -inf();
#pragma unroll
        for (int w = 0; w < NW; ++w) { const double a = bits2d(kmm[w]), b = bits2d(kmm[16 + w]); vhi = a > vhi ? a : vhi; vlo = b > vlo ? b : vlo; }
        // (any positive factor gives a monotone map: the hardware's approximate reciprocal will do)
        const double lo = -vlo, hi = vhi, scale = (double)SUMM_BINS * __builtin_amdgcn_rcp(hi - lo);
        // workgroup-uniform: every value that carries weight is finite, they are not all alike, the bin width is a normal number
        bool binned = fabs(lo) < inf() && fabs(hi) < inf() && lo < hi && scale < inf();
        if (binned) {
            constexpr int PER = SUMM_BINS / WAVE;
            static_assert(PER == 16, "one row of lanes picks the bin");
            int bin[NQ];
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int b = (int)((xv[i] - lo) * scale);   // monotone in x: differences, products and truncation all are
                bin[i] = b < SUMM_BINS - 1 ? b : SUMM_BINS - 1;
                // bin b lives at word (b mod 16) * 64 + b / 16: the lanes of the wave that sums 16 consecutive bins each read
                // consecutive words (16 l + t at t * 64 + l) instead of all hitting the same two banks
                if (q[i]) atomicAdd(&hbin[(bin[i] & (PER - 1)) * WAVE + (bin[i] >> 4)], (unsigned long long)q[i]);
            }
            __syncthreads();
            for (int j = wave; j < nq; j += NW) {   // one wave per level: lane l sums the bins 16 l .. 16 l + 15
                uint64_t sum = 0;
#pragma unroll 4
                for (int t = 0; t < PER; ++t) sum += hbin[t * WAVE + lane];
                const uint64_t excl = wave_incl_scan(sum, lane) - sum;
                const uint64_t target = __umul64hi(v.sum_p64[j], S);
                const unsigned long long own = __ballot(sum && excl <= target && target < excl + sum);   // exactly one lane
                const int L = own ? __builtin_ctzll(own) : 0;
                // ... and the 16 bins of that lane, one per lane of the first row
                const uint64_t h = lane < PER ? hbin[lane * WAVE + L] : 0;
                uint64_t incl = h;
                incl = dpp_add_u64<0x111>(incl);
                incl = dpp_add_u64<0x112>(incl);
                incl = dpp_add_u64<0x114>(incl);
                incl = dpp_add_u64<0x118>(incl);
                const uint64_t run = readlane_u64(excl, L) + incl - h;
                if (lane < PER && h && run <= target && target < run + h) { st_bin[j] = (uint64_t)(PER * L + lane); st_below[j] = run; st_target[j] = target; }
            }
            __syncthreads();
            for (int j = 0; j < nq; ++j) {   // the particles of the chosen bins, in any order
                const int sb = (int)st_bin[j];
#pragma unroll
                for (int i = 0; i < NQ; ++i)
                    if (q[i] && bin[i] == sb) {
                        const unsigned idx = atomicAdd(&cnt[j], 1u);
                        if (idx < (unsigned)SUMM_CAND) { cand[((size_t)j * SUMM_CAND + idx) * 2] = key[i]; cand[((size_t)j * SUMM_CAND + idx) * 2 + 1] = q[i]; }
                    }
            }
            for (int i = tid; i < SUMM_BINS; i += THREADS) hbin[i] = 0;   // ready for the next step
            __syncthreads();
            for (int j = 0; j < nq; ++j) binned = binned && cnt[j] <= (unsigned)SUMM_CAND;   // workgroup-uniform
            if (binned) {
                for (int j = wave; j < nq; j += NW) {   // one wave per level ranks its candidates: lane l holds candidate l
                    const int c = __builtin_amdgcn_readfirstlane((int)cnt[j]);
                    const uint64_t ck = lane < c ? cand[((size_t)j * SUMM_CAND + lane) * 2] : ~0ULL;
                    const uint64_t cw = lane < c ? cand[((size_t)j * SUMM_CAND + lane) * 2 + 1] : 0;
                    uint64_t less = st_below[j], upto = less;   // weight of everything below / not above this candidate
                    for (int m = 0; m < c; ++m) {
                        const uint64_t km = readlane_u64(ck, m), wm = readlane_u64(cw, m);
                        less += km < ck ? wm : 0;
                        upto += km <= ck ? wm : 0;
                    }
                    // the candidates whose value the target falls on (all of them hold the same key): the quantile
                    const uint64_t tg = st_target[j];
                    const unsigned long long hit = __ballot(lane < c && less <= tg && tg < upto);
                    const uint64_t best = readlane_u64(ck, hit ? __builtin_ctzll(hit) : 0);
                    if (lane == 0) v.sum_q[((size_t)row * v.ntheta + th) * nq + j] = key_value(best);
                }
            }
        }
        // ---- otherwise: radix select on the key, eight digits of eight bits ----
        if (!binned)
        for (int pass = 0; pass < 8; ++pass) {
            const int hs = 64 - 8 * pass;   // the prefix is key >> hs (pass > 0)
            for (int j = 0; j < nq; ++j) {
                const uint64_t pref = pass ? st_prefix[j] : 0;
#pragma unroll
                for (int i = 0; i < NQ; ++i)
                    if (q[i] && (pass == 0 || (key[i] >> hs) == pref))
                        atomicAdd(&hist[j * 256 + (int)((key[i] >> (hs - 8)) & 255)], (unsigned long long)q[i]);
            }
            __syncthreads();
            for (int j = wave; j < nq; j += NW) {   // one wave per level: lane l owns the bins 4l .. 4l+3
                unsigned long long* hb = hist + j * 256 + 4 * lane;
                const uint64_t h0 = hb[0], h1 = hb[1], h2 = hb[2], h3 = hb[3];
                hb[0] = hb[1] = hb[2] = hb[3] = 0;   // ready for the next pass / step
                const uint64_t sum = h0 + h1 + h2 + h3;
                const uint64_t excl = wave_incl_scan(sum, lane) - sum;
                const uint64_t target = pass ? st_target[j] : __umul64hi(v.sum_p64[j], S);
                const uint64_t prefix = pass ? st_prefix[j] : 0;
                uint64_t run = (pass ? st_below[j] : 0) + excl;
                const uint64_t hh[4] = {h0, h1, h2, h3};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (hh[t] && run <= target && target < run + hh[t]) {   // exactly one bin of one lane
                        const uint64_t np_ = (prefix << 8) | (uint64_t)(4 * lane + t);
                        st_prefix[j] = np_;
                        st_below[j] = run;
                        st_target[j] = target;
                        if (pass == 7) v.sum_q[((size_t)row * v.ntheta + th) * nq + j] = key_value(np_);
                    }
                    run += hh[t];
                }
            }
            __syncthreads();
        }
    }
    if (v.sum_mom) {
        __syncthreads();
        if (tid < D) {
            double a = 0.0, b2 = 0.0;
            for (int w = 0; w < NW; ++w) { a += red[(0 * 3 + tid) * 16 + w]; b2 += red[(1 * 3 + tid) * 16 + w]; }
            v.sum_m[(((size_t)row * 2 + 0) * D + tid) * v.ntheta + th] = a;
            v.sum_m[(((size_t)row * 2 + 1) * D + tid) * v.ntheta + th] = b2 - a * a;
        }
    }
}

// The summaries of the CURRENT state of single-segment filters in one launch (smc_get_quantiles / smc_get_moments between two
// steps of the README loop, README.md:41,51): state and weights of buffer `cur` into LDS, then the per-step routine above, row 0.
template <int THREADS, int NP, int D>
__global__ __launch_bounds__(THREADS) void k_summ_once(FilterView v, int cur) {
    constexpr int SEG = 2 * NP * THREADS, SEGP = lds_padded_len(SEG);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* Cs = (uint64_t*)smem;
    double* xs = (double*)(smem + (size_t)SEGP * 8);
    uint64_t* sm = (uint64_t*)(smem + (size_t)SEGP * 8 * (1 + D));
    const int tid = threadIdx.x, th = blockIdx.x;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int i0 = 2 * (tid + k * THREADS);
#pragma unroll
        for (int c = 0; c < D; ++c)
            *reinterpret_cast<double2*>(xs + c * SEGP + lds_pad(i0)) = *reinterpret_cast<const double2*>(v.x[cur] + ((size_t)c * v.ntheta + th) * v.npad + i0);
        *reinterpret_cast<ulonglong2*>(Cs + lds_pad(i0)) = *reinterpret_cast<const ulonglong2*>(v.C[cur] + (size_t)th * v.npad + i0);
    }
    for (int i = tid; i < (int)summary_lds_words(v.sum_np); i += THREADS) sm[i] = 0;
    __syncthreads();
    resident_summaries<THREADS, NP, D>(v, th, 0, Cs, xs, SEGP, v.segS[cur][th], sm);
}
template <int D>
__host__ inline size_t summ_once_lds_bytes(int seg, int nq) { return (size_t)lds_padded_len(seg) * 8 * (1 + D) + summary_lds_words(nq) * 8; }

// Window mode (WIN): the same loop over the steps [t0, t0 + T) of filters that already exist (t0 >= 1): the state is
// read from buffer `bin`, the T steps run in LDS, the state after them goes to buffer `bout`, and (logmu_t, ess_t) of
// every step go to `win` ([2][T][ntheta], pinned host memory) - nothing else of the handle changes (logZ and the
// "last emitted" values are advanced by k_commit once the host has decided to keep the steps).  This is the
// speculative multi-step call of the online sampler (smc_step_window): bootstrap_filter! k times in one launch.
// Waves per SIMD the register budget must allow (second launch bound).  Only the 512-thread, one-pair-per-thread variant
// (1024-particle filters: the samplers' shape) is pinned - to 4, i.e. two workgroups per CU: there a 129th vector register
// halves the occupancy (measured: 512 UCSV filters 1.69 -> 2.24 ms).  Pinning the larger variants the same way makes the
// compiler spill and costs more than the occupancy gains (2048 particles, UCSV: 1.07 -> 1.40 ms), so they keep the default.
template <int MODEL, int THREADS, int NP>
constexpr int resident_min_waves() {
    return (THREADS == 512 && NP == 1) ? 4 : 1;
}
// SUMM: per-step summaries (resident_summaries) after every step, for the levels / coordinate the view names
template <int MODEL, int THREADS, int NP, bool SYS = false, bool WIN = false, bool SUMM = false>
__global__ __launch_bounds__(THREADS, (resident_min_waves<MODEL, THREADS, NP>())) void k_resident(FilterView v, int T, StepRec* recs /*[ntheta][T]*/, int t0, int bin, int bout,
                                                      double* win) {
    constexpr int D = model_dim<MODEL>::value;
    constexpr int SEG = 2 * NP * THREADS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SEGP = lds_padded_len(SEG);
    uint64_t* Cs = (uint64_t*)smem;                       // [SEGP] padded against bank conflicts (lds_pad)
    double* xs = (double*)(smem + (size_t)SEGP * 8);      // [D][SEGP] padded the same way
    uint64_t* scr = (uint64_t*)(smem + (size_t)SEGP * 8 + (size_t)SEGP * 8 * D);
    uint64_t* usys = scr + scr_words(THREADS, NP) - 8;   // SYS: the steps' uniforms (tail words nobody else uses)
    uint64_t* sm = scr + scr_words(THREADS, NP);         // SUMM: histograms and state of the per-step summaries
    constexpr int NU = (THREADS / WAVE) < 8 ? (THREADS / WAVE) : 8;
    const int tid = threadIdx.x;
    int th = blockIdx.x;
    if (v.order) {   // active filters first (see FilterView::order); the others are not run: logZ = -inf
        th = v.order[blockIdx.x];
        if ((int)blockIdx.x >= *v.n_active) {
            if (tid == 0) {
                v.logZ[th] = -inf();
                if (v.host_out) v.host_out[th] = -inf();
            }
            return;
        }
    } else if (v.skip && v.skip[th]) {   // a filter that is not run (proposal outside the prior's support): logZ = -inf
        if (tid == 0) {
            v.logZ[th] = -inf();
            if (v.host_out) v.host_out[th] = -inf();
        }
        return;
    }
    const Params prm = v.params[th];
    const uint32_t stream = v.stream[th];
    StepRec* rec = recs + (size_t)th * T;
#ifdef SMC_ABLATE
    if (v.dbg && tid == 0) {   // diagnostic build: where and when this workgroup runs (HW_ID: cu [11:8], sh [12], se [15:13]; XCC_ID)
        v.dbg[(size_t)blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memrealtime();
        v.dbg[(size_t)blockIdx.x * 8 + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) |
                                            ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) << 32);
    }
#endif

    constexpr int NQ = 2 * NP;
    double xn[NP][2][D];
    double lw[NP][2];
    int anc[NQ];
    uint64_t S = 0;
    if (WIN) {   // the filters' current state into LDS
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int i0 = 2 * (tid + k * THREADS);
#pragma unroll
            for (int c = 0; c < D; ++c)
                *reinterpret_cast<double2*>(xs + c * SEGP + lds_pad(i0)) =
                    *reinterpret_cast<const double2*>(v.x[bin] + ((size_t)c * v.ntheta + th) * v.npad + i0);
            *reinterpret_cast<ulonglong2*>(Cs + lds_pad(i0)) = *reinterpret_cast<const ulonglong2*>(v.C[bin] + (size_t)th * v.npad + i0);
        }
        S = v.segS[bin][th];
        if (SYS && tid < NU) {   // the uniforms of the first NU steps (afterwards refreshed inside the loop as usual)
            const u32x4 uw = draw(v.seed, 0u, stream, (uint32_t)(t0 + tid), SLOT_SYS);
            usys[(t0 - 1 + tid) % NU] = ((uint64_t)uw.v[1] << 32) | uw.v[0];
        }
        __syncthreads();
    }

    if (SUMM) {   // the histograms start out empty (every selection leaves them empty again)
        for (int i = tid; i < (int)summary_lds_words(v.sum_np); i += THREADS) sm[i] = 0;
    }
    const bool ragged = (int)v.n != SEG;   // workgroup-uniform
    for (int t = t0; t < t0 + T; ++t) {
        const double y = v.y[t - t0];
        double xp[NQ][D];
        SMC_PRIO(0);   // two workgroups share a CU for the whole series: whoever is in the earlier half of its step wins the
                       // issue slot, so neither runs ahead and leaves the other alone at the end (LG 512 filters: 0.92 -> 0.85 ms)
        if (t > 0) {
            // a = resample(weights); xp = x[a]: the NQ searches of a thread advance level by level
            uint64_t T2[NQ];
            if (SYS) {   // opt-in systematic resampling: child j takes T_j = floor((j S + v0) / n)  (single segment: SH = 0)
                const SysBase sbase = sys_base(S, (uint32_t)v.n, v.inv_n, usys[(t - 1) % NU], 0u);   // published NU steps at a time
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    const int j = 2 * (tid + (i >> 1) * THREADS) + (i & 1);
                    T2[i] = sys_target(sbase, (uint32_t)(j < (int)v.n ? j : (int)v.n - 1));
                }
            } else {
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const u32x4 rw = draw(v.seed, (uint32_t)(tid + k * THREADS), stream, (uint32_t)t, SLOT_RESAMPLE);
                    uint64_t lo;
#if SMC_EXP_PICKF64
                    T2[2 * k] = (uint64_t)(fma((double)rw.v[1], 0x1p-32, (double)(rw.v[0] >> 11) * 0x1p-53) * (double)S);
                    T2[2 * k + 1] = (uint64_t)(fma((double)rw.v[3], 0x1p-32, (double)(rw.v[2] >> 11) * 0x1p-53) * (double)S);
#else
                    mul64wide(((uint64_t)rw.v[1] << 32) | rw.v[0], S, T2[2 * k], lo);
                    mul64wide(((uint64_t)rw.v[3] << 32) | rw.v[2], S, T2[2 * k + 1], lo);
#endif
                }
            }
            // the search carries the padded position as an LDS byte pointer (one ds_read_b64 with an
            // immediate offset per probe); xs is padded like Cs, so the gather uses it as it stands
            lds_byte* pb[NQ];
#pragma unroll
            for (int i = 0; i < NQ; ++i) pb[i] = lds_ptr(Cs);
            if (SMC_ABL(v, 0)) {   // ablation (profiling build): no search, a pseudo-random in-range position instead
#pragma unroll
                for (int i = 0; i < NQ; ++i) pb[i] += 8 * lds_pad((int)(T2[i] >> 7) & (SEG - 1));
            } else {
#pragma unroll
            for (int s = SEG >> 1; s >= 1; s >>= 1) {
                uint64_t val[NQ];
#pragma unroll
                for (int i = 0; i < NQ; ++i) val[i] = lds_load_u64(pb[i] + 8 * lds_probe_off(s));
#pragma unroll
                for (int i = 0; i < NQ; ++i) pb[i] += (val[i] <= T2[i]) ? 8 * lds_step_inc(s) : 0;
            }
            }
            const int last_p = lds_pad((int)v.n - 1);
            int apos[NQ];
#pragma unroll
            for (int i = 0; i < NQ; ++i) apos[i] = (int)(pb[i] - lds_ptr(Cs)) >> 3;   // T2 < S = C[n-1]: a position below n
            if (S == 0 || ragged) {   // workgroup-uniform and rare: a real branch
                asm volatile("; collapsed or ragged");
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    const int own = 2 * (tid + (i >> 1) * THREADS) + (i & 1);
                    const int ap = S ? apos[i] : lds_pad(own);   // collapsed filter: identity
                    apos[i] = ap < last_p ? ap : last_p;         // lds_pad is increasing: the clamp to n-1 commutes with it
                }
            }
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int ap = apos[i];
                anc[i] = ap;
#pragma unroll
                for (int c = 0; c < D; ++c) xp[i][c] = xs[c * SEGP + ap];
            }
        } else {
#pragma unroll
            for (int i = 0; i < NQ; ++i) anc[i] = lds_pad(2 * (tid + (i >> 1) * THREADS) + (i & 1));
        }
        double z[NP][D][2];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const uint32_t pg = (uint32_t)(tid + k * THREADS);
#pragma unroll
            for (int c = 0; c < D; ++c) box_muller(draw(v.seed, pg, stream, (uint32_t)t, SLOT_NORMAL0 + c), z[k][c][0], z[k][c][1]);
        }
        if (t > 0) {
#pragma unroll
            for (int k = 0; k < NP; ++k)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    double zz[D];
#pragma unroll
                    for (int c = 0; c < D; ++c) zz[c] = z[k][c][j];
                    model_transition<MODEL>(prm, xp[2 * k + j], zz, xn[k][j]);
                }
        } else {
            // a real (workgroup-uniform) branch: flattened into selects, every step of the series would also evaluate the
            // initial distribution
            asm volatile("; t = 0");
#pragma unroll
            for (int k = 0; k < NP; ++k)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    double zz[D];
#pragma unroll
                    for (int c = 0; c < D; ++c) zz[c] = z[k][c][j];
                    model_initial<MODEL>(prm, zz, xn[k][j]);
                }
        }
#pragma unroll
        for (int k = 0; k < NP; ++k)
#pragma unroll
            for (int j = 0; j < 2; ++j) lw[k][j] = model_logobs<MODEL>(prm, xn[k][j], y);
        if (ragged) {   // n < SEG: the particles beyond n carry NaN weights and zero states (a real branch: the samplers' filters are full)
            asm volatile("; ragged");
#pragma unroll
            for (int k = 0; k < NP; ++k)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if ((2 * (tid + k * THREADS) + j) >= v.n) {
                        lw[k][j] = nan_mask();
#pragma unroll
                        for (int c = 0; c < D; ++c) xn[k][j][c] = 0.0;
                    }
        }
        __syncthreads();  // every gather from xs / Cs is done
        SMC_PRIO(2);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int i0 = 2 * (tid + k * THREADS);
#pragma unroll
            for (int c = 0; c < D; ++c) {
                double2 o;
                o.x = xn[k][0][c];
                o.y = xn[k][1][c];
                *reinterpret_cast<double2*>(xs + c * SEGP + lds_pad(i0)) = o;
            }
        }
        const SegRec r = segment_normalize<THREADS, NP, true>(lw, scr, Cs, WIN || v.want_s2 != 0 || t == t0 + T - 1);
        S = r.S;
        if (tid == 0) {
            StepRec o;
            o.kb = r.kb; o.S = r.S; o.hi = r.hi; o.lo = r.lo;
            rec[t - t0] = o;
        }
        if (SYS && (t % NU) == 0 && (tid & (WAVE - 1)) == 0 && tid / WAVE < NU) {
            // the uniforms of the next NU steps: wave w draws u(t+1+w) - every wave pays one Philox call per NU
            // steps and no wave runs ahead of the others; the barrier below publishes them
            const int w = tid / WAVE;
            const u32x4 uw = draw(v.seed, 0u, stream, (uint32_t)(t + 1 + w), SLOT_SYS);
            usys[w] = ((uint64_t)uw.v[1] << 32) | uw.v[0];
        }
        __syncthreads();  // Cs, xs complete; scr free
        if (SUMM) resident_summaries<THREADS, NP, D>(v, th, (int64_t)(t - t0), Cs, xs, SEGP, S, sm);
    }

    // state out: same layout as the k_step path leaves it (log_likelihood: buffer 0)
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int i0 = 2 * (tid + k * THREADS);
#pragma unroll
        for (int c = 0; c < D; ++c)
            *reinterpret_cast<double2*>(v.x[bout] + ((size_t)c * v.ntheta + th) * v.npad + i0) =
                *reinterpret_cast<const double2*>(xs + c * SEGP + lds_pad(i0));
        *reinterpret_cast<ulonglong2*>(v.C[bout] + (size_t)th * v.npad + i0) = *reinterpret_cast<const ulonglong2*>(Cs + lds_pad(i0));
        if (v.anc) {
            int2 o;
            o.x = lds_unpad(anc[2 * k]);   // anc holds padded positions
            o.y = lds_unpad(anc[2 * k + 1]);
            *reinterpret_cast<int2*>(v.anc + (size_t)th * v.npad + i0) = o;
        }
    }
    // (logmu_t, ess_t) for every step in parallel, then logZ = sum_t logmu_t in step order
    double* lm = (double*)Cs;  // reuse LDS as [T] staging when it fits, else global trace
    const bool stage_lds = T <= SEG;
    __syncthreads();
    for (int t = tid; t < T; t += THREADS) {
        const StepRec o = rec[t];
        // single segment: K = kb, sh = SH = 0, so the table is (S) itself
        const uint64_t Qb = o.S, Rb = seg_R(o.hi, o.lo, 0, 0);
        double logmu, ess;
        combine_outputs(o.kb, Qb, Rb, 0, v.n, logmu, ess);
        if (WIN) {
            win[(size_t)t * v.ntheta + th] = logmu;
            win[((size_t)T + t) * v.ntheta + th] = ess;
            if (t == T - 1) {   // the record of the state in buffer bout
                v.segk[bout][th] = o.kb;
                v.segS[bout][th] = o.S;
                v.segS2hi[bout][th] = o.hi;
                v.segS2lo[bout][th] = o.lo;
            }
            continue;
        }
        if (v.trace_logmu) v.trace_logmu[(size_t)t * v.ntheta + th] = logmu;
        if (v.trace_ess) v.trace_ess[(size_t)t * v.ntheta + th] = ess;
        if (stage_lds) lm[t] = logmu;
        else rec[t].kb = logmu;
        if (t == T - 1) {
            v.last_logmu[th] = logmu;
            v.last_ess[th] = ess;
            if (v.host_out) {
                v.host_out[(size_t)v.ntheta + th] = logmu;
                v.host_out[2 * (size_t)v.ntheta + th] = ess;
            }
            v.last_K[th] = o.kb;
            v.last_D[th] = Qb;
            v.segk[bout][th] = o.kb;
            v.segS[bout][th] = o.S;
            v.segS2hi[bout][th] = o.hi;
            v.segS2lo[bout][th] = o.lo;
        }
    }
#ifdef SMC_ABLATE
    if (v.dbg && tid == 0) v.dbg[(size_t)blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
    if (WIN) return;
    __syncthreads();
    if (tid == 0) {
        double z = 0.0;
        for (int t = 0; t < T; ++t) {
            const double l = stage_lds ? lm[t] : rec[t].kb;
            z = t == 0 ? l : z + l;
        }
        v.logZ[th] = z;
        if (v.host_out) v.host_out[th] = z;
    }
}

inline bool resident_supported(int model, int seg) {
    const int d = model_dim_rt(model);
    if (d < 0) return false;
    return (size_t)lds_padded_len(seg) * 8 * (1 + d) + 2048 <= 160 * 1024;
}

}  // namespace smc
