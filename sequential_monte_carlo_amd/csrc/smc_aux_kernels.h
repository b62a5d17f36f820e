// smc_aux_kernels.h -- kernels off the hot path (state read-back, outer resample!, stand-alone
// normalize / resample).  Included by smc_capi.hip only.
#pragma once
#include "smc_kernels.h"
#include "smc_resident.h"

namespace smc {

// ---------------------------------------------------------------------------------------------
// dense normalised weights w_i (normalize()'s `w`, particles.jl:11) for smc_get_state
// ---------------------------------------------------------------------------------------------
__global__ void k_dense_weights(FilterView v, int cur, double* w /*[ntheta][n]*/) {
    const int th = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.n) return;
    const double K = v.last_K[th];
    const uint64_t Dtot = v.last_D[th];
    const int b = (int)(i / v.seg), j = (int)(i % v.seg);
    const uint64_t* C = v.C[cur] + (size_t)th * v.npad;
    const uint64_t q = C[i] - (j ? C[i - 1] : 0);
    const double dk = K - v.segk[cur][(size_t)th * v.nseg + b];
    const double sc = (dk >= 0.0 && dk < 900.0) ? pow2i(-48 - (int)dk) : 0.0;
    const double Dd = (double)Dtot * pow2i(v.SH - 48);
    w[(size_t)th * v.n + i] = Dtot ? ((double)q * sc) / Dd : 0.0;
}

// ---------------------------------------------------------------------------------------------
// outer resample!(smc) (smc_samplers.jl:74-84): theta slot m <- slot a[m], value copy of the
// whole filter state (x cloud, C, segment records, logZ).  grid (blocks, ntheta)
// ---------------------------------------------------------------------------------------------
__global__ void k_permute(FilterView v, int cur, int d, const int32_t* a, const double* logZ_src) {
    const int th = blockIdx.y, src = a[th], nxt = cur ^ 1;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < v.npad) {
        for (int c = 0; c < d; ++c)
            v.x[nxt][((size_t)c * v.ntheta + th) * v.npad + i] = v.x[cur][((size_t)c * v.ntheta + src) * v.npad + i];
        v.C[nxt][(size_t)th * v.npad + i] = v.C[cur][(size_t)src * v.npad + i];
    }
    if (i < v.nseg) {
        const size_t o = (size_t)th * v.nseg + i, s = (size_t)src * v.nseg + i;
        v.segk[nxt][o] = v.segk[cur][s];
        v.segS[nxt][o] = v.segS[cur][s];
        v.segS2hi[nxt][o] = v.segS2hi[cur][s];
        v.segS2lo[nxt][o] = v.segS2lo[cur][s];
    }
    if (i == 0) v.logZ[th] = logZ_src[src];
}

// exact scalar Kalman filter, one lane per parameter row   kalman_filter.jl:29-70
__global__ void k_kalman(const double* raw, int64_t ntheta, const double* y, int64_t T, int predict_first, double* out) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= ntheta) return;
    const double A = raw[m * 6 + 0], B = raw[m * 6 + 1], Q = raw[m * 6 + 2], R = raw[m * 6 + 3];
    double x = raw[m * 6 + 4], S = raw[m * 6 + 5], logZ = 0.0;
    for (int64_t t = 0; t < T; ++t) {
        if (predict_first || t > 0) { x = A * x; S = (A * A) * S + Q; }
        const double s = (B * B) * S + R, dy = y[t] - B * x;
        const double K = S * B, inv = 1.0 / s;
        x = x + (K * inv) * dy;
        S = S - (K * K) * inv;
        logZ += -0.5 * (0x1.d67f1c864beb5p+0 + sp_log(s) + (dy / s) * dy);
    }
    out[m * 3 + 0] = x; out[m * 3 + 1] = S; out[m * 3 + 2] = logZ;
}

// ---------------------------------------------------------------------------------------------
// k_breaks: the break points of the resampling steps t0 .. t0+gridDim.x-1 of every filter (smc_spec.h
// "break points"): F[(tt * ntheta + th) * (nseg + 1) + w], 2^-64 fixed point, F[..][0] = 0.  They depend
// on (seed, stream, t, block sizes) only - never on the particles - so this runs ahead of the steps, off
// the critical path.  grid (steps, ntheta); one Gamma variate per block, an integer scan, one long division.
// ---------------------------------------------------------------------------------------------
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_breaks(FilterView v, uint32_t t0, uint64_t* F) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* g = (uint64_t*)smem;               // [nseg + 1]
    uint64_t* wt = g + v.nseg + 1;               // [THREADS / WAVE]
    constexpr int NW = THREADS / WAVE;
    const int tt = blockIdx.x, th = blockIdx.y, tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const uint32_t t = t0 + (uint32_t)tt, stream = v.stream[th];
    const int B = v.nseg;
    for (int w = tid; w < B; w += THREADS) {
        int64_t m = v.n - (int64_t)w * v.seg;
        m = m > v.seg ? v.seg : m;
        g[w] = gamma_fix(v.seed, (uint32_t)w, stream, t, m);
    }
    if (tid == 0) g[B] = exp1_fix(v.seed, (uint32_t)B, stream, t);
    __syncthreads();
    const int E = (B + THREADS - 1) / THREADS;   // consecutive entries per thread
    uint64_t run = 0;
    for (int e = 0; e < E; ++e) {
        const int i = tid * E + e;
        if (i < B) { run += g[i]; g[i] = run; }
    }
    const uint64_t incl = wave_incl_scan(run, lane);
    if (lane == WAVE - 1) wt[wave] = incl;
    __syncthreads();
    uint64_t off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { off += w < wave ? wt[w] : 0; tot += wt[w]; }
    const uint64_t excl = off + incl - run, S = tot + g[B];
    uint64_t* out = F + ((size_t)tt * v.ntheta + th) * ((size_t)B + 1);
    if (tid == 0) out[0] = 0;
    for (int e = 0; e < E; ++e) {
        const int i = tid * E + e;
        if (i < B) out[i + 1] = div_frac64(g[i] + excl, S);
    }
}

// slot <-> packed buffer.  Packed slot layout in 8-byte words:
//   x [d][npad] | C [npad] | kb,S,S2hi,S2lo [4][nseg] | logZ,last_logmu,last_ess,last_K,last_D [5]   (+ pad to even)
__host__ __device__ inline int64_t slot_words(int d, int64_t npad, int nseg) {
    const int64_t w = (int64_t)(d + 1) * npad + 4 * (int64_t)nseg + 5;
    return (w + 1) & ~(int64_t)1;
}
template <bool PACK>
__global__ void k_pack_slots(FilterView v, int cur, int d, const int32_t* idx, uint64_t* buf) {
    const int s = blockIdx.y, th = idx[s];
    const int64_t W = slot_words(d, v.npad, v.nseg);
    uint64_t* b = buf + (size_t)s * W;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto mv = [&](uint64_t* slot_word, uint64_t* dev_word) {
        if (PACK) *slot_word = *dev_word; else *dev_word = *slot_word;
    };
    if (i < v.npad) {
        for (int c = 0; c < d; ++c) mv(b + (size_t)c * v.npad + i, (uint64_t*)(v.x[cur] + ((size_t)c * v.ntheta + th) * v.npad + i));
        mv(b + (size_t)d * v.npad + i, v.C[cur] + (size_t)th * v.npad + i);
    }
    uint64_t* r = b + (size_t)(d + 1) * v.npad;
    if (i < v.nseg) {
        const size_t o = (size_t)th * v.nseg + i;
        mv(r + i, (uint64_t*)(v.segk[cur] + o));
        mv(r + v.nseg + i, v.segS[cur] + o);
        mv(r + 2 * (size_t)v.nseg + i, v.segS2hi[cur] + o);
        mv(r + 3 * (size_t)v.nseg + i, v.segS2lo[cur] + o);
    }
    if (i == 0) {
        uint64_t* t = r + 4 * (size_t)v.nseg;
        mv(t + 0, (uint64_t*)(v.logZ + th));
        mv(t + 1, (uint64_t*)(v.last_logmu + th));
        mv(t + 2, (uint64_t*)(v.last_ess + th));
        mv(t + 3, (uint64_t*)(v.last_K + th));
        mv(t + 4, v.last_D + th);
    }
}

// accept step: slot th of dst <- slot th of src where mask[th]   grid (blocks, ntheta)
__global__ void k_copy_slots(FilterView dst, int dcur, FilterView src, int scur, int d, const unsigned char* mask) {
    const int th = blockIdx.y;
    if (!mask[th]) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < dst.npad) {
        for (int c = 0; c < d; ++c)
            dst.x[dcur][((size_t)c * dst.ntheta + th) * dst.npad + i] = src.x[scur][((size_t)c * src.ntheta + th) * src.npad + i];
        dst.C[dcur][(size_t)th * dst.npad + i] = src.C[scur][(size_t)th * src.npad + i];
    }
    if (i < dst.nseg) {
        const size_t o = (size_t)th * dst.nseg + i;
        dst.segk[dcur][o] = src.segk[scur][o];
        dst.segS[dcur][o] = src.segS[scur][o];
        dst.segS2hi[dcur][o] = src.segS2hi[scur][o];
        dst.segS2lo[dcur][o] = src.segS2lo[scur][o];
    }
    if (i == 0) {
        dst.logZ[th] = src.logZ[th];
        dst.last_logmu[th] = src.last_logmu[th];
        dst.last_ess[th] = src.last_ess[th];
        dst.last_K[th] = src.last_K[th];
        dst.last_D[th] = src.last_D[th];
    }
}

// Keeps the first j steps of a window whose k_resident<WIN> launch ran exactly j steps: logZ += logmu_1 + .. + logmu_j in
// step order (the same additions smc_step makes), and the "last emitted" values become those of step j.  grid over theta.
__global__ void k_commit(FilterView v, int j, const StepRec* recs /*[ntheta][j]*/) {
    const int th = blockIdx.x * blockDim.x + threadIdx.x;
    if (th >= v.ntheta) return;
    double z = v.logZ[th], logmu = 0.0, ess = 0.0;
    StepRec o{};
    for (int t = 0; t < j; ++t) {
        o = recs[(size_t)th * j + t];
        combine_outputs(o.kb, o.S, seg_R(o.hi, o.lo, 0, 0), 0, v.n, logmu, ess);
        z = z + logmu;
    }
    v.logZ[th] = z;
    v.last_logmu[th] = logmu;
    v.last_ess[th] = ess;
    v.last_K[th] = o.kb;
    v.last_D[th] = o.S;
    host_emit(v, th, z, logmu, ess);
}

// ---------------------------------------------------------------------------------------------
// PMMH rejuvenation on the device (rejuvenate!, smc_samplers.jl:103-146): one lane per parameter particle.
//   k_pmmh_propose  theta' ~ MvNormal(theta, scale Sigma) (:114), insupport (:116), prior logpdfs (:123), and the
//                   parameter row smc.model(theta') of the proposal filter (:120)
//   [the proposal filters run: log_likelihood over y for every in-support theta']
//   k_pmmh_accept   the accept test (:123-129) and theta / logZ of the accepted particles (:130-131); the filter
//                   state x, w (:132-133) is then copied by k_copy_slots under the same mask
// ---------------------------------------------------------------------------------------------
struct PmmhDev {
    double* theta;        // [ntheta][MAX_DTHETA] current parameter particles
    double* prop;         // [ntheta][MAX_DTHETA] proposals of this chain position
    double* logZ;         // [ntheta] log-likelihood estimates of the current particles
    double* lp;           // [ntheta][2] log prior of (proposal, current)
    unsigned char* skip;  // [ntheta] proposal outside the support: its filter is not run
    unsigned char* mask;  // [ntheta] accepted at this chain position
    unsigned char* any;   // [ntheta] accepted at least once in this rejuvenation (acc_array, :135)
    unsigned long long* nrun;   // [1] proposal filters executed so far
    double* chol;         // [d][d] lower Cholesky factor of the random-walk covariance (:95-100)
    int32_t* order;       // [ntheta] the filters of this chain position, those to run first (FilterView::order)
    int32_t* counts;      // [2] how many are to run / skipped (filled by k_pmmh_propose, cleared by k_pmmh_accept)
};
__global__ void k_pmmh_propose(FilterView v, PmmhSpec s, PmmhDev p, int model, uint64_t move_seed, uint32_t c, double sq, Params* params) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= v.ntheta) return;
    const double* th = p.theta + (size_t)m * MAX_DTHETA;
    double pr[MAX_DTHETA];
    pmmh_propose(s, move_seed, v.stream[m], c, th, p.chol, sq, pr);
    const bool ok = pmmh_insupport(s, pr);
    p.skip[m] = ok ? 0 : 1;
    for (int i = 0; i < s.d; ++i) p.prop[(size_t)m * MAX_DTHETA + i] = pr[i];
    // which workgroup runs which filter does not matter for the results: any order of the two groups will do
    if (ok) p.order[atomicAdd(&p.counts[0], 1)] = m;
    else p.order[v.ntheta - 1 - atomicAdd(&p.counts[1], 1)] = m;
    if (!ok) return;
    p.lp[2 * (size_t)m] = pmmh_logprior(s, pr);
    p.lp[2 * (size_t)m + 1] = pmmh_logprior(s, th);
    Params P;
    pmmh_raw_row(s, pr, P.raw);
    derive_params(model, P.raw, P.der);
    params[m] = P;
}
__global__ void k_pmmh_accept(FilterView v, PmmhDev p, int d, uint64_t move_seed, uint32_t c, double xi) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= v.ntheta) return;
    if (m == 0) p.counts[0] = p.counts[1] = 0;   // the filters have run: ready for the next chain position
    bool acc = false;
    if (!p.skip[m]) {
        const double logZp = v.logZ[m], lpp = p.lp[2 * (size_t)m], lpc = p.lp[2 * (size_t)m + 1];
        const double likelihood_ratio = xi * (logZp - p.logZ[m]), prior_ratio = lpp - lpc;
        const double acc_ratio = likelihood_ratio + prior_ratio, log_post_prop = logZp + lpp;
        acc = log_post_prop > -inf() && pmmh_log_uniform(move_seed, v.stream[m], c) < acc_ratio;
        atomicAdd(p.nrun, 1ull);
        if (acc) {
            for (int i = 0; i < d; ++i) p.theta[(size_t)m * MAX_DTHETA + i] = p.prop[(size_t)m * MAX_DTHETA + i];
            p.logZ[m] = logZp;
            p.any[m] = 1;
        }
    }
    p.mask[m] = acc ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// stand-alone A1 / A2 (outer theta-level reweight / resample; n <= a few thousand): one
// workgroup, single level, all integer sums.
// ---------------------------------------------------------------------------------------------
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_normalize(const double* logw, int64_t n, int K, double* w, double* out2) {
    constexpr int NW = THREADS / WAVE;
    __shared__ double red[NW];
    __shared__ uint64_t acc[3][NW];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    double kmax = -inf();
    for (int64_t i = tid; i < n; i += THREADS) {
        const double l = logw[i];
        if (lw_alive(l)) { double k; (void)sp_exp_parts(l, k); kmax = k > kmax ? k : kmax; }
    }
    kmax = block_max<THREADS>(kmax, red);
    uint64_t S = 0;
    U128 s2{0, 0};
    for (int64_t i = tid; i < n; i += THREADS) {
        const double l = logw[i];
        uint64_t q = 0;
        if (lw_alive(l)) { double k; const double p = sp_exp_parts(l, k); q = fix_weight(p, k - kmax, K); }
        S += q;
        s2 = add128(s2, sq128(q));
    }
    S = wave_sum(S);
    s2 = wave_sum128(s2);
    if (lane == 0) { acc[0][wave] = S; acc[1][wave] = s2.lo; acc[2][wave] = s2.hi; }
    __syncthreads();
    uint64_t St = 0;
    U128 t2{0, 0};
#pragma unroll
    for (int k = 0; k < NW; ++k) { St += acc[0][k]; t2 = add128(t2, U128{acc[1][k], acc[2][k]}); }
    const double Sd = (double)St;
    for (int64_t i = tid; i < n; i += THREADS) {
        const double l = logw[i];
        uint64_t q = 0;
        if (lw_alive(l)) { double k; const double p = sp_exp_parts(l, k); q = fix_weight(p, k - kmax, K); }
        w[i] = St ? (double)q / Sd : 0.0;
    }
    if (tid == 0) {
        out2[0] = St ? fma(kmax, LN2_HI, fma(kmax, LN2_LO, sp_log(Sd * pow2i(-K)))) - sp_log((double)n) : -inf();
        out2[1] = St ? (Sd * Sd) / u128_to_double(t2.hi, t2.lo) : 0.0;
    }
}

// The same normalize() for long vectors (a whole particle cloud's log-weights): three grid-wide passes.  Every
// cross-workgroup combination is an integer operation (max of the integer exponents, sums of the fixed-point weights, sums
// of the three 32-bit limbs of their squares), so the order the workgroups arrive in cannot change a bit: the results are
// those of the one-workgroup kernel above.  acc: [0] sum q  [1..3] sums of the limbs of q^2; kmax_i: exponent maximum.
constexpr int NORM_DEAD = (int)0x80000000;
__device__ __forceinline__ uint64_t norm_q(double l, int kmax_i, int K) {
    if (!lw_alive(l) || kmax_i == NORM_DEAD) return 0;
    double k;
    const double p = sp_exp_parts(l, k);
    return fix_weight(p, k - (double)kmax_i, K);
}
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_normalize_max(const double* logw, int64_t n, int* kmax_i) {
    __shared__ int red[THREADS / WAVE];
    int km = NORM_DEAD;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS) {
        const double l = logw[i];
        if (lw_alive(l)) { double k; (void)sp_exp_parts(l, k); const int ki = (int)k; km = ki > km ? ki : km; }
    }
    km = block_max_i32<THREADS>(km, red);
    if (threadIdx.x == 0 && km != NORM_DEAD) atomicMax(kmax_i, km);
}
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_normalize_sum(const double* logw, int64_t n, int K, const int* kmax_i,
                                                          unsigned long long* acc) {
    constexpr int NW = THREADS / WAVE;
    __shared__ uint64_t part[4][NW];
    const int km = *kmax_i, lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    uint64_t s[4] = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS) {
        const uint64_t q = norm_q(logw[i], km, K);
        const U128 q2 = sq128(q);
        s[0] += q;
        s[1] += q2.lo & 0xffffffffULL;
        s[2] += q2.lo >> 32;
        s[3] += q2.hi;                     // q < 2^48: q^2 < 2^96, the top limb is below 2^32
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint64_t t = wave_sum(s[j]);
        if (lane == 0) part[j][wave] = t;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        uint64_t t = 0;
        for (int w = 0; w < NW; ++w) t += part[threadIdx.x][w];
        if (t) atomicAdd(&acc[threadIdx.x], (unsigned long long)t);
    }
}
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_normalize_write(const double* logw, int64_t n, int K, const int* kmax_i,
                                                            const unsigned long long* acc, double* w, double* out2) {
    const int km = *kmax_i;
    const uint64_t St = acc[0];
    const double Sd = (double)St;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS) {
        const uint64_t q = norm_q(logw[i], km, K);
        w[i] = St ? (double)q / Sd : 0.0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // sum q^2 = l0 + l1 2^32 + l2 2^64 as a 128-bit integer
        const uint64_t l0 = acc[1], l1 = acc[2], l2 = acc[3];
        const uint64_t lo = l0 + (l1 << 32);
        const uint64_t hi = l2 + (l1 >> 32) + (lo < l0 ? 1u : 0u);
        const double kmax = km == NORM_DEAD ? -inf() : (double)km;
        out2[0] = St ? fma(kmax, LN2_HI, fma(kmax, LN2_LO, sp_log(Sd * pow2i(-K)))) - sp_log((double)n) : -inf();
        out2[1] = St ? (Sd * Sd) / u128_to_double(hi, lo) : 0.0;
    }
}

// q_i = rint(w_i / wmax * 2^K) ; C = inclusive scan (single workgroup, chunked)
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_resample_cdf(const double* w, int64_t n, int K, uint64_t* C, int* status) {
    constexpr int NW = THREADS / WAVE;
    __shared__ double red[NW];
    __shared__ uint64_t wt[NW];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    double m = 0.0;
    for (int64_t i = tid; i < n; i += THREADS) { const double x = w[i]; m = x > m ? x : m; }
    m = block_max<THREADS>(m, red);
    if (!(m > 0.0) || m == inf()) { if (tid == 0) *status = -2; return; }
    const double scale = pow2i(K);
    uint64_t carry = 0;
    for (int64_t base = 0; base < n; base += THREADS) {
        const int64_t i = base + tid;
        uint64_t q = 0;
        if (i < n) { const double r = w[i] / m; q = (r == r && r > 0.0) ? (uint64_t)rne_pos(r * scale) : 0; }
        const uint64_t incl = wave_incl_scan(q, lane);
        if (lane == WAVE - 1) wt[wave] = incl;
        __syncthreads();
        uint64_t off = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < NW; ++k) { off += (k < wave) ? wt[k] : 0; tot += wt[k]; }
        if (i < n) C[i] = carry + off + incl;
        carry += tot;
        __syncthreads();
    }
    if (tid == 0) *status = 0;
}

// The same inclusive sums for a long vector, grid-wide (every cross-workgroup combination an integer sum or a maximum: the same
// bits as the single workgroup): the maximum; the sum of every workgroup's contiguous chunk; their exclusive scan (one workgroup);
// the chunk's inclusive sums on top of its offset.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_rs_max(const double* w, int64_t n, unsigned long long* mbits) {
    __shared__ double red[THREADS / WAVE];
    double m = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS) { const double x = w[i]; m = x > m ? x : m; }
    m = block_max<THREADS>(m, red);
    if (threadIdx.x == 0) atomicMax(mbits, (unsigned long long)d2bits(m));   // (non-negative doubles order like their bits; +inf included)
}
__device__ __forceinline__ uint64_t rs_q(double w, double m, double scale) {
    const double r = w / m;
    return (r == r && r > 0.0) ? (uint64_t)rne_pos(r * scale) : 0;
}
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_rs_chunk_sums(const double* w, int64_t n, int K, const unsigned long long* mbits, int64_t chunk, uint64_t* bs) {
    __shared__ uint64_t wt[THREADS / WAVE];
    const double m = bits2d(*mbits);
    if (!(m > 0.0) || m == inf()) return;
    const double scale = pow2i(K);
    const int64_t i0 = (int64_t)blockIdx.x * chunk, i1 = i0 + chunk < n ? i0 + chunk : n;
    uint64_t s = 0;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += THREADS) s += rs_q(w[i], m, scale);
    s = wave_sum(s);
    if ((threadIdx.x & (WAVE - 1)) == 0) wt[threadIdx.x / WAVE] = s;
    __syncthreads();
    if (threadIdx.x == 0) { uint64_t t = 0; for (int k = 0; k < THREADS / WAVE; ++k) t += wt[k]; bs[blockIdx.x] = t; }
}
// one workgroup: exclusive scan of the nb chunk sums in place; status = -2 for an unusable maximum
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_rs_scan_chunks(const unsigned long long* mbits, int nb, uint64_t* bs, int* status) {
    constexpr int NW = THREADS / WAVE;
    __shared__ uint64_t wt[NW];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const double m = bits2d(*mbits);
    if (!(m > 0.0) || m == inf()) { if (tid == 0) *status = -2; return; }
    uint64_t carry = 0;
    for (int base = 0; base < nb; base += THREADS) {
        const int i = base + tid;
        const uint64_t v = i < nb ? bs[i] : 0;
        const uint64_t incl = wave_incl_scan(v, lane);
        if (lane == WAVE - 1) wt[wave] = incl;
        __syncthreads();
        uint64_t off = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < NW; ++k) { off += (k < wave) ? wt[k] : 0; tot += wt[k]; }
        if (i < nb) bs[i] = carry + off + incl - v;
        carry += tot;
        __syncthreads();
    }
    if (tid == 0) *status = 0;
}
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_rs_write(const double* w, int64_t n, int K, const unsigned long long* mbits, int64_t chunk, const uint64_t* bs,
                                                      uint64_t* C) {
    constexpr int NW = THREADS / WAVE;
    __shared__ uint64_t wt[NW];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const double m = bits2d(*mbits);
    if (!(m > 0.0) || m == inf()) return;
    const double scale = pow2i(K);
    const int64_t i0 = (int64_t)blockIdx.x * chunk, i1 = i0 + chunk < n ? i0 + chunk : n;
    uint64_t carry = bs[blockIdx.x];
    for (int64_t base = i0; base < i1; base += THREADS) {
        const int64_t i = base + tid;
        const uint64_t q = i < i1 ? rs_q(w[i], m, scale) : 0;
        const uint64_t incl = wave_incl_scan(q, lane);
        if (lane == WAVE - 1) wt[wave] = incl;
        __syncthreads();
        uint64_t off = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < NW; ++k) { off += (k < wave) ? wt[k] : 0; tot += wt[k]; }
        if (i < i1) C[i] = carry + off + incl;
        carry += tot;
        __syncthreads();
    }
}

__global__ void k_resample_draw(const uint64_t* C, int64_t n, int64_t ndraw, uint64_t seed, uint32_t stream, uint32_t t,
                                int32_t* a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ndraw) return;
    const u32x4 rw = draw(seed, (uint32_t)(i >> 1), stream, t, SLOT_RESAMPLE);
    const int j = (int)(i & 1);
    const uint64_t r = ((uint64_t)rw.v[2 * j + 1] << 32) | rw.v[2 * j];
    const uint64_t S = C[n - 1];
    uint64_t T, lo;
    mul64wide(r, S, T, lo);
    int64_t l = 0, h = n;
    while (l < h) {
        const int64_t mid = (l + h) >> 1;
        if (C[mid] > T) h = mid; else l = mid + 1;
    }
    a[i] = (int32_t)(l < n ? l : n - 1);
}

}  // namespace smc
