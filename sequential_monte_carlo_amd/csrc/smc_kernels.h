// smc_kernels.h -- hand-written HIP kernels for gfx950 (CDNA4, wave64) of the bootstrap
// particle filter hot path.  One workgroup owns one SEGMENT (seg = 2*NP*THREADS particles)
// of one filter (theta); thread tau owns the NP particle pairs p = tau + k*THREADS, so every
// global store is a 16-byte-per-lane, fully coalesced double2 / ulonglong2.
//
//   k_init      bootstrap_filter        particles.jl:87-105   (sample x_1, weigh, normalise)
//   k_step      bootstrap_filter!       particles.jl:107-129  (resample+gather, propagate,
//                                                              weigh, normalise) - ONE launch
//   k_finalize  the (logmu, ess) return of normalize          particles.jl:10,12
//   k_resident  log_likelihood          particles.jl:132-147  whole T loop, one workgroup per
//                                                              filter, state resident in LDS
//
// Weights never exist as doubles in memory: a segment keeps the inclusive prefix sums C of
// q_i = rint(exp(logw_i - m_seg) * 2^48) (uint64, exact, order independent) plus its record
// (m_seg, S = sum q, S2 = sum q^2 as 128 bit).  The next launch's prologue turns the records
// of all segments of the filter into a second integer table (Dcum) in LDS; a particle then
// draws its ancestor with one 64-bit Philox draw: table search in LDS, segment search in C.
#pragma once
#include "smc_spec.h"

namespace smc {

constexpr int WAVE = 64;

struct FilterView {
    int64_t n;        // particles per filter (Nx)
    int64_t npad;     // nseg * seg
    int seg, nseg, nseg_p2, QK;
    int ntheta;
    uint64_t seed;
    const Params* params;    // [ntheta]
    const uint32_t* stream;  // [ntheta]
    double* x[2];            // [d][ntheta][npad]   ping-pong
    uint64_t* C[2];          // [ntheta][npad]
    double* segm[2];         // [ntheta][nseg]
    uint64_t* segS[2];
    uint64_t* segS2hi[2];
    uint64_t* segS2lo[2];
    int32_t* anc;            // [ntheta][npad] or nullptr
    double* logZ;            // [ntheta]
    double* last_logmu;      // [ntheta]  (logmu, ess, g, D) of the most recently emitted weights
    double* last_ess;        // [ntheta]
    double* last_g;          // [ntheta]
    uint64_t* last_D;        // [ntheta]
    double* trace_logmu;     // [T][ntheta] or nullptr
    double* trace_ess;       // [T][ntheta] or nullptr
    const double* y;         // [T] on device (log_likelihood) or nullptr
};

// ---------------------------------------------------------------------------------------------
// wave / block primitives (wave64)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int d) {
    return (uint64_t)__shfl_up((unsigned long long)v, d, WAVE);
}
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int d) {
    return (uint64_t)__shfl_xor((unsigned long long)v, d, WAVE);
}
__device__ __forceinline__ uint64_t wave_incl_scan(uint64_t v, int lane) {
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint64_t o = shfl_up_u64(v, d);
        if (lane >= d) v += o;
    }
    return v;
}
__device__ __forceinline__ uint64_t wave_sum(uint64_t v) {
#pragma unroll
    for (int d = WAVE / 2; d >= 1; d >>= 1) v += shfl_xor_u64(v, d);
    return v;
}
// 128-bit unsigned accumulator (sum of q^2).  NOTE: do not "optimise" this into 24-bit limb
// products: hipcc 7.2 folds (q & 0xFFFFFF)^2 accumulations into v_mad_u64_u32 on the UNMASKED
// register (wrong sums); the parity tests against the oracle caught it.
struct U128 {
    uint64_t lo, hi;
};
__device__ __forceinline__ U128 add128(U128 a, U128 b) {
    U128 r;
    r.lo = a.lo + b.lo;
    r.hi = a.hi + b.hi + (r.lo < a.lo ? 1u : 0u);
    return r;
}
__device__ __forceinline__ U128 sq128(uint64_t q) {
    U128 r;
    r.hi = __umul64hi(q, q);
    r.lo = q * q;
    return r;
}
__device__ __forceinline__ U128 wave_sum128(U128 v) {
#pragma unroll
    for (int d = WAVE / 2; d >= 1; d >>= 1) {
        U128 o;
        o.lo = shfl_xor_u64(v.lo, d);
        o.hi = shfl_xor_u64(v.hi, d);
        v = add128(v, o);
    }
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int d = WAVE / 2; d >= 1; d >>= 1) {
        const double o = __shfl_xor(v, d, WAVE);
        v = o > v ? o : v;
    }
    return v;
}

// max over the workgroup; every thread gets the result. `red` : NW doubles of LDS.
template <int THREADS>
__device__ __forceinline__ double block_max(double v, double* red) {
    constexpr int NW = THREADS / WAVE;
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    v = wave_max(v);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double r = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) r = red[w] > r ? red[w] : r;
    __syncthreads();
    return r;
}

// count of entries <= T in a non-decreasing array of length N (power of two): first index with
// arr[idx] > T.  Branch-free, log2(N) dependent loads.
template <class PTR>
__device__ __forceinline__ int upper_bound_pow2(PTR arr, int N, uint64_t T) {
    int pos = 0;
    for (int s = N >> 1; s >= 1; s >>= 1) pos += (arr[pos + s - 1] <= T) ? s : 0;
    return pos;
}

// ---------------------------------------------------------------------------------------------
// LDS carve of the step / finalize kernels (dynamic LDS, 16-B aligned, no static LDS)
// ---------------------------------------------------------------------------------------------
struct TableLds {
    uint64_t* Dcum;  // [nseg_p2] inclusive sums of Q_b
    double* fd;      // [nseg_p2] S_b / Q_b
    uint64_t* Sseg;  // [nseg_p2]
    uint64_t* scr;   // [4*NW + 8] scratch
};
__host__ __device__ inline size_t scr_words(int threads, int np) {
    return (size_t)((np + 3 > 4 ? np + 3 : 4) * (threads / WAVE) + 8);
}
__host__ __device__ inline size_t table_lds_bytes(int nseg_p2, int threads, int np) {
    return (size_t)nseg_p2 * 24 + scr_words(threads, np) * 8;
}
__device__ __forceinline__ TableLds carve(char* smem, int nseg_p2) {
    TableLds t;
    t.Dcum = (uint64_t*)smem;
    t.fd = (double*)(smem + (size_t)nseg_p2 * 8);
    t.Sseg = (uint64_t*)(smem + (size_t)nseg_p2 * 16);
    t.scr = (uint64_t*)(smem + (size_t)nseg_p2 * 24);
    return t;
}

// Segment-table prologue: builds Dcum / fd / Sseg in LDS from the records of filter `th` in
// buffer `cur`; returns Dtot.  If `emit`, thread 0 also produces (logmu, ess) of those weights
// - the return value of normalize(), particles.jl:10,12 - into last_* and adds logmu to logZ.
template <int THREADS>
__device__ __forceinline__ uint64_t table_prologue(const FilterView& v, int cur, int th, const TableLds& L, bool emit,
                                                   bool first_emit, uint32_t t_emit) {
    constexpr int NW = THREADS / WAVE;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const size_t base = (size_t)th * v.nseg;
    const double* sm = v.segm[cur] + base;
    const uint64_t* sS = v.segS[cur] + base;
    const uint64_t* sHi = v.segS2hi[cur] + base;
    const uint64_t* sLo = v.segS2lo[cur] + base;
    double* red = (double*)L.scr;

    double g = -inf();
    for (int b = tid; b < v.nseg; b += THREADS) { const double m = sm[b]; g = m > g ? m : g; }
    g = block_max<THREADS>(g, red);

    // blocked layout: thread owns E consecutive table entries
    const int E = v.nseg_p2 >= THREADS ? v.nseg_p2 / THREADS : 1;
    uint64_t run = 0, rsum = 0;
    if (tid * E < v.nseg_p2) {
        for (int e = 0; e < E; ++e) {
            const int b = tid * E + e;
            uint64_t Qb = 0, Rb = 0, S = 0;
            if (b < v.nseg) {
                S = sS[b];
                seg_entry(sm[b], S, sHi[b], sLo[b], g, v.QK, Qb, Rb);
            }
            run += Qb;
            rsum += Rb;
            L.Dcum[b] = run;  // thread-local inclusive, fixed up below
            L.Sseg[b] = S;
            L.fd[b] = Qb ? (double)S / (double)Qb : 0.0;
        }
    }
    const uint64_t incl = wave_incl_scan(run, lane);
    const uint64_t rw = wave_sum(rsum);
    uint64_t* wt = L.scr + NW;       // [NW] wave totals of Q
    uint64_t* wr = L.scr + 2 * NW;   // [NW] wave totals of R
    if (lane == WAVE - 1) wt[wave] = incl;
    if (lane == 0) wr[wave] = rw;
    __syncthreads();
    uint64_t off = 0, Dtot = 0, Rtot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const uint64_t t = wt[w];
        off += (w < wave) ? t : 0;
        Dtot += t;
        Rtot += wr[w];
    }
    const uint64_t excl = off + incl - run;
    if (tid * E < v.nseg_p2)
        for (int e = 0; e < E; ++e) L.Dcum[tid * E + e] += excl;
    if (emit && tid == 0) {
        double logmu, ess;
        combine_outputs(g, Dtot, Rtot, v.QK, v.n, logmu, ess);
        v.last_logmu[th] = logmu;
        v.last_ess[th] = ess;
        v.last_g[th] = g;
        v.last_D[th] = Dtot;
        if (v.trace_logmu) v.trace_logmu[(size_t)t_emit * v.ntheta + th] = logmu;
        if (v.trace_ess) v.trace_ess[(size_t)t_emit * v.ntheta + th] = ess;
        v.logZ[th] = first_emit ? logmu : v.logZ[th] + logmu;
    }
    __syncthreads();
    return Dtot;
}

// ---------------------------------------------------------------------------------------------
// segment epilogue: normalize() of one segment.  lw[k][j] = log-weight of particle 2p+j of pair
// p = tau + k*THREADS (masked particles carry -inf).  Writes C (16 B per lane) and the record.
// ---------------------------------------------------------------------------------------------
struct SegRec {
    double m;
    uint64_t S;       // valid in every thread
    uint64_t hi, lo;  // valid in thread 0 only
};

// Cout: where the segment's inclusive sums go (global buffer or LDS), 16-B aligned.
template <int THREADS, int NP>
__device__ __forceinline__ SegRec segment_normalize(double (&lw)[NP][2], uint64_t* scr, uint64_t* Cout) {
    constexpr int NW = THREADS / WAVE;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    double mloc = -inf();
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        mloc = lw[k][0] > mloc ? lw[k][0] : mloc;   // NaN never wins
        mloc = lw[k][1] > mloc ? lw[k][1] : mloc;
    }
    const double mb = block_max<THREADS>(mloc, (double*)scr);

    uint64_t q[NP][2], ps[NP], incl[NP];
    U128 s2{0, 0};   // sum q^2
#pragma unroll
    for (int k = 0; k < NP; ++k) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const double l = lw[k][j];
            uint64_t qq = 0;
            if (l == l && l > -inf()) qq = to_fix48(sp_exp(l - mb));
            q[k][j] = qq;
            s2 = add128(s2, sq128(qq));
        }
        ps[k] = q[k][0] + q[k][1];
        incl[k] = wave_incl_scan(ps[k], lane);
    }
    s2 = wave_sum128(s2);
    uint64_t* wtot = scr;                 // [NP][NW]
    uint64_t* w2 = scr + NP * NW;         // [2][NW]
    if (lane == WAVE - 1) {
#pragma unroll
        for (int k = 0; k < NP; ++k) wtot[k * NW + wave] = incl[k];
    }
    if (lane == 0) { w2[wave] = s2.lo; w2[NW + wave] = s2.hi; }
    __syncthreads();
    uint64_t basek = 0;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        uint64_t off = 0, ktot = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const uint64_t t = wtot[k * NW + w];
            off += (w < wave) ? t : 0;
            ktot += t;
        }
        const uint64_t excl = basek + off + incl[k] - ps[k];
        ulonglong2 cc;
        cc.x = excl + q[k][0];
        cc.y = cc.x + q[k][1];
        *reinterpret_cast<ulonglong2*>(Cout + 2 * (tid + k * THREADS)) = cc;
        basek += ktot;
    }
    SegRec rec;
    rec.m = mb;
    rec.S = basek;
    rec.hi = rec.lo = 0;
    if (tid == 0) {
        U128 t{0, 0};
#pragma unroll
        for (int w = 0; w < NW; ++w) t = add128(t, U128{w2[w], w2[NW + w]});
        rec.hi = t.hi;
        rec.lo = t.lo;
    }
    return rec;
}

// normalize() of one segment into the global ping-pong buffers
template <int THREADS, int NP>
__device__ __forceinline__ void segment_epilogue(const FilterView& v, int nxt, int th, int sb, double (&lw)[NP][2],
                                                 uint64_t* scr) {
    uint64_t* Cout = v.C[nxt] + (size_t)th * v.npad + (size_t)sb * v.seg;
    const SegRec rec = segment_normalize<THREADS, NP>(lw, scr, Cout);
    if (threadIdx.x == 0) {
        const size_t r = (size_t)th * v.nseg + sb;
        v.segm[nxt][r] = rec.m;
        v.segS[nxt][r] = rec.S;
        v.segS2hi[nxt][r] = rec.hi;
        v.segS2lo[nxt][r] = rec.lo;
    }
}

// ---------------------------------------------------------------------------------------------
// k_init : bootstrap_filter  (particles.jl:87-105)    grid (nseg, ntheta)
// ---------------------------------------------------------------------------------------------
template <int MODEL, int THREADS, int NP>
__global__ __launch_bounds__(THREADS) void k_init(FilterView v, int nxt, double y) {
    constexpr int D = model_dim<MODEL>::value;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* scr = (uint64_t*)smem;
    const int sb = blockIdx.x, th = blockIdx.y, tid = threadIdx.x;
    const Params prm = v.params[th];
    const uint32_t stream = v.stream[th];
    const int64_t seg0 = (int64_t)sb * v.seg;
    double lw[NP][2];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int pl = tid + k * THREADS;            // pair within segment
        const int64_t i0 = seg0 + 2 * pl;            // first particle of the pair
        const uint32_t pg = (uint32_t)(i0 >> 1);     // pair index within the filter
        double z[D][2], xn[2][D];
#pragma unroll
        for (int c = 0; c < D; ++c) box_muller(draw(v.seed, pg, stream, 0u, SLOT_NORMAL0 + c), z[c][0], z[c][1]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            double zz[D];
#pragma unroll
            for (int c = 0; c < D; ++c) zz[c] = z[c][j];
            model_initial<MODEL>(prm, zz, xn[j]);
            const bool valid = (i0 + j) < v.n;
            lw[k][j] = valid ? model_logobs<MODEL>(prm, xn[j], y) : -inf();
        }
#pragma unroll
        for (int c = 0; c < D; ++c) {
            double2 o;
            o.x = (i0 < v.n) ? xn[0][c] : 0.0;
            o.y = (i0 + 1 < v.n) ? xn[1][c] : 0.0;
            *reinterpret_cast<double2*>(v.x[nxt] + ((size_t)c * v.ntheta + th) * v.npad + i0) = o;
        }
        if (v.anc) {
            int2 o;
            o.x = (int)i0;
            o.y = (int)i0 + 1;
            *reinterpret_cast<int2*>(v.anc + (size_t)th * v.npad + i0) = o;
        }
    }
    segment_epilogue<THREADS, NP>(v, nxt, th, sb, lw, scr);
}

// ---------------------------------------------------------------------------------------------
// k_step : bootstrap_filter!  (particles.jl:107-129)   grid (nseg, ntheta), ONE launch per step
//   prologue  segment table of the weights in buffer `cur`  (+ emit logmu/ess of step t-1)
//   a = resample(weights)          -> two-level inverse CDF, one 64-bit draw per particle
//   xp = x[a]                      -> gather from buffer `cur`
//   x[i] = rand(transition(xp[i])); logw[i] = logpdf(observation(x[i]), y)
//   normalize(logw)                -> segment epilogue into buffer `cur^1`
// ---------------------------------------------------------------------------------------------
template <int MODEL, int THREADS, int NP>
__global__ __launch_bounds__(THREADS) void k_step(FilterView v, int cur, uint32_t t, int emit_prev, double yval) {
    constexpr int D = model_dim<MODEL>::value;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int sb = blockIdx.x, th = blockIdx.y, tid = threadIdx.x;
    const int nxt = cur ^ 1;
    const TableLds L = carve(smem, v.nseg_p2);
    const Params prm = v.params[th];
    const uint32_t stream = v.stream[th];
    const double y = v.y ? v.y[t] : yval;
    const int64_t seg0 = (int64_t)sb * v.seg;

    uint64_t Dtot;
    if (v.nseg > 1 || (emit_prev && sb == 0)) {
        Dtot = table_prologue<THREADS>(v, cur, th, L, emit_prev && sb == 0, t == 1u, t - 1u);
    } else {
        Dtot = 1;  // unused on the single-segment path (S0 read below)
    }
    const uint64_t S0 = v.segS[cur][(size_t)th * v.nseg];
    const uint64_t* Cprev = v.C[cur] + (size_t)th * v.npad;
    const double* xprev = v.x[cur];

    double lw[NP][2];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int pl = tid + k * THREADS;
        const int64_t i0 = seg0 + 2 * pl;
        const uint32_t pg = (uint32_t)(i0 >> 1);
        const u32x4 rw = draw(v.seed, pg, stream, t, SLOT_RESAMPLE);
        int64_t anc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint64_t r = ((uint64_t)rw.v[2 * j + 1] << 32) | rw.v[2 * j];
            int64_t a;
            if (v.nseg == 1) {
                uint64_t T2, lo;
                mul64wide(r, S0, T2, lo);
                a = S0 ? upper_bound_pow2(Cprev, v.seg, T2) : (i0 + j);
            } else if (Dtot == 0) {
                a = i0 + j;
            } else {
                uint64_t T1, lo;
                mul64wide(r, Dtot, T1, lo);
                const int b = upper_bound_pow2(L.Dcum, v.nseg_p2, T1);
                const uint64_t rho = T1 - (b ? L.Dcum[b - 1] : 0);
                const double pos = ((double)rho + (double)(lo >> 11) * TWO_M53) * L.fd[b];
                uint64_t T2 = (uint64_t)pos;
                const uint64_t Sb = L.Sseg[b];
                T2 = T2 > Sb - 1 ? Sb - 1 : T2;
                a = (int64_t)b * v.seg + upper_bound_pow2(Cprev + (size_t)b * v.seg, v.seg, T2);
            }
            if (a >= v.n) a = v.n - 1;  // cannot happen (padding has q = 0); keeps the gather in bounds
            anc[j] = a;
        }
        double z[D][2], xn[2][D];
#pragma unroll
        for (int c = 0; c < D; ++c) box_muller(draw(v.seed, pg, stream, t, SLOT_NORMAL0 + c), z[c][0], z[c][1]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            double zz[D], xp[D];
#pragma unroll
            for (int c = 0; c < D; ++c) {
                zz[c] = z[c][j];
                xp[c] = xprev[((size_t)c * v.ntheta + th) * v.npad + anc[j]];
            }
            model_transition<MODEL>(prm, xp, zz, xn[j]);
            const bool valid = (i0 + j) < v.n;
            lw[k][j] = valid ? model_logobs<MODEL>(prm, xn[j], y) : -inf();
        }
#pragma unroll
        for (int c = 0; c < D; ++c) {
            double2 o;
            o.x = (i0 < v.n) ? xn[0][c] : 0.0;
            o.y = (i0 + 1 < v.n) ? xn[1][c] : 0.0;
            *reinterpret_cast<double2*>(v.x[nxt] + ((size_t)c * v.ntheta + th) * v.npad + i0) = o;
        }
        if (v.anc) {
            int2 o;
            o.x = (int)anc[0];
            o.y = (int)anc[1];
            *reinterpret_cast<int2*>(v.anc + (size_t)th * v.npad + i0) = o;
        }
    }
    segment_epilogue<THREADS, NP>(v, nxt, th, sb, lw, L.scr);
}

// ---------------------------------------------------------------------------------------------
// k_finalize : (logmu, ess) of the weights currently in buffer `cur`.   grid (ntheta)
// ---------------------------------------------------------------------------------------------
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_finalize(FilterView v, int cur, int first_emit, uint32_t t_emit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const TableLds L = carve(smem, v.nseg_p2);
    table_prologue<THREADS>(v, cur, blockIdx.x, L, true, first_emit != 0, t_emit);
}

// ---------------------------------------------------------------------------------------------
// dense normalised weights w_i (normalize()'s `w`, particles.jl:11) for smc_get_state
// ---------------------------------------------------------------------------------------------
__global__ void k_dense_weights(FilterView v, int cur, double* w /*[ntheta][n]*/) {
    const int th = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.n) return;
    const double g = v.last_g[th];
    const uint64_t Dtot = v.last_D[th];
    const int b = (int)(i / v.seg), j = (int)(i % v.seg);
    const uint64_t* C = v.C[cur] + (size_t)th * v.npad;
    const uint64_t q = C[i] - (j ? C[i - 1] : 0);
    const double e = (g > -inf()) ? sp_exp(v.segm[cur][(size_t)th * v.nseg + b] - g) : 0.0;
    const double Dd = (double)Dtot * pow2i(-v.QK);
    w[(size_t)th * v.n + i] = Dtot ? ((double)q * TWO_M48) * e / Dd : 0.0;
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
        v.segm[nxt][o] = v.segm[cur][s];
        v.segS[nxt][o] = v.segS[cur][s];
        v.segS2hi[nxt][o] = v.segS2hi[cur][s];
        v.segS2lo[nxt][o] = v.segS2lo[cur][s];
    }
    if (i == 0) v.logZ[th] = logZ_src[src];
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
    double m = -inf();
    for (int64_t i = tid; i < n; i += THREADS) { const double l = logw[i]; m = l > m ? l : m; }
    m = block_max<THREADS>(m, red);
    const double scale = pow2i(K);
    uint64_t S = 0;
    U128 s2{0, 0};
    for (int64_t i = tid; i < n; i += THREADS) {
        const double l = logw[i];
        const double e = (l == l && m > -inf()) ? sp_exp(l - m) : 0.0;
        const uint64_t q = (uint64_t)rne_pos(e * scale);
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
        const double e = (l == l && m > -inf()) ? sp_exp(l - m) : 0.0;
        const uint64_t q = (uint64_t)rne_pos(e * scale);
        w[i] = St ? (double)q / Sd : 0.0;
    }
    if (tid == 0) {
        out2[0] = St ? (m + sp_log(Sd * pow2i(-K))) - sp_log((double)n) : -inf();
        out2[1] = St ? (Sd * Sd) / u128_to_double(t2.hi, t2.lo) : 0.0;
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
