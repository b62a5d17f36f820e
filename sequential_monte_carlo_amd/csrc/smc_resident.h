// smc_resident.h -- k_resident: log_likelihood(N, y, model) (particles.jl:132-147) for the
// batched callers (smc_samplers.jl:112-121,223-229): ONE workgroup per filter (theta), the whole
// T loop inside the kernel, particle states and weight prefix sums resident in LDS (160 KiB/CU
// on gfx950).  Requires a single segment (n_x <= seg).  Bit-identical to the k_init/k_step path.
#pragma once
#include "smc_kernels.h"

namespace smc {

// per-step record written by the loop, turned into (logmu_t, ess_t) after it
struct StepRec { double m; uint64_t S, hi, lo; };

template <int MODEL>
__host__ __device__ inline size_t resident_lds_bytes(int seg, int threads, int np) {
    return (size_t)seg * 8 * (model_dim<MODEL>::value + 1) + scr_words(threads, np) * 8;
}

template <int MODEL, int THREADS, int NP>
__global__ __launch_bounds__(THREADS) void k_resident(FilterView v, int T, StepRec* recs /*[ntheta][T]*/) {
    constexpr int D = model_dim<MODEL>::value;
    constexpr int SEG = 2 * NP * THREADS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* Cs = (uint64_t*)smem;                       // [SEG]
    double* xs = (double*)(smem + (size_t)SEG * 8);       // [D][SEG]
    uint64_t* scr = (uint64_t*)(smem + (size_t)SEG * 8 * (D + 1));
    const int th = blockIdx.x, tid = threadIdx.x;
    const Params prm = v.params[th];
    const uint32_t stream = v.stream[th];
    StepRec* rec = recs + (size_t)th * T;

    double xn[NP][2][D];
    double lw[NP][2];
    int anc[NP][2];
    uint64_t S = 0;

    for (int t = 0; t < T; ++t) {
        const double y = v.y[t];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int i0 = 2 * (tid + k * THREADS);
            const uint32_t pg = (uint32_t)(i0 >> 1);
            double xp[2][D];
            if (t > 0) {
                // a = resample(weights); xp = x[a]
                const u32x4 rw = draw(v.seed, pg, stream, (uint32_t)t, SLOT_RESAMPLE);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const uint64_t r = ((uint64_t)rw.v[2 * j + 1] << 32) | rw.v[2 * j];
                    uint64_t T2, lo;
                    mul64wide(r, S, T2, lo);
                    int a = S ? upper_bound_pow2(Cs, SEG, T2) : (i0 + j);
                    if (a >= v.n) a = (int)v.n - 1;
                    anc[k][j] = a;
#pragma unroll
                    for (int c = 0; c < D; ++c) xp[j][c] = xs[c * SEG + a];
                }
            } else {
                anc[k][0] = i0;
                anc[k][1] = i0 + 1;
            }
            double z[D][2];
#pragma unroll
            for (int c = 0; c < D; ++c) box_muller(draw(v.seed, pg, stream, (uint32_t)t, SLOT_NORMAL0 + c), z[c][0], z[c][1]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                double zz[D];
#pragma unroll
                for (int c = 0; c < D; ++c) zz[c] = z[c][j];
                if (t > 0) model_transition<MODEL>(prm, xp[j], zz, xn[k][j]);
                else model_initial<MODEL>(prm, zz, xn[k][j]);
                lw[k][j] = (i0 + j) < v.n ? model_logobs<MODEL>(prm, xn[k][j], y) : -inf();
            }
        }
        __syncthreads();  // every gather from xs / Cs is done
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int i0 = 2 * (tid + k * THREADS);
#pragma unroll
            for (int c = 0; c < D; ++c) {
                double2 o;
                o.x = i0 < v.n ? xn[k][0][c] : 0.0;
                o.y = (i0 + 1) < v.n ? xn[k][1][c] : 0.0;
                *reinterpret_cast<double2*>(xs + c * SEG + i0) = o;
            }
        }
        const SegRec r = segment_normalize<THREADS, NP>(lw, scr, Cs);
        S = r.S;
        if (tid == 0) {
            StepRec o;
            o.m = r.m; o.S = r.S; o.hi = r.hi; o.lo = r.lo;
            rec[t] = o;
        }
        __syncthreads();  // Cs, xs complete; scr free
    }

    // state out: same layout as the k_step path leaves in buffer 0
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int i0 = 2 * (tid + k * THREADS);
#pragma unroll
        for (int c = 0; c < D; ++c)
            *reinterpret_cast<double2*>(v.x[0] + ((size_t)c * v.ntheta + th) * v.npad + i0) =
                *reinterpret_cast<const double2*>(xs + c * SEG + i0);
        *reinterpret_cast<ulonglong2*>(v.C[0] + (size_t)th * v.npad + i0) = *reinterpret_cast<const ulonglong2*>(Cs + i0);
        if (v.anc) {
            int2 o;
            o.x = anc[k][0];
            o.y = anc[k][1];
            *reinterpret_cast<int2*>(v.anc + (size_t)th * v.npad + i0) = o;
        }
    }
    // (logmu_t, ess_t) for every step in parallel, then logZ = sum_t logmu_t in step order
    double* lm = (double*)Cs;  // reuse LDS as [T] staging when it fits, else global trace
    const bool stage_lds = T <= SEG;
    __syncthreads();
    for (int t = tid; t < T; t += THREADS) {
        const StepRec o = rec[t];
        uint64_t Qb, Rb;
        seg_entry(o.m, o.S, o.hi, o.lo, o.m, v.QK, Qb, Rb);
        double logmu, ess;
        combine_outputs(o.m, Qb, Rb, v.QK, v.n, logmu, ess);
        if (v.trace_logmu) v.trace_logmu[(size_t)t * v.ntheta + th] = logmu;
        if (v.trace_ess) v.trace_ess[(size_t)t * v.ntheta + th] = ess;
        if (stage_lds) lm[t] = logmu;
        else rec[t].m = logmu;
        if (t == T - 1) {
            v.last_logmu[th] = logmu;
            v.last_ess[th] = ess;
            v.last_g[th] = o.m;
            v.last_D[th] = Qb;
            v.segm[0][th] = o.m;
            v.segS[0][th] = o.S;
            v.segS2hi[0][th] = o.hi;
            v.segS2lo[0][th] = o.lo;
        }
    }
    __syncthreads();
    if (tid == 0) {
        double z = 0.0;
        for (int t = 0; t < T; ++t) {
            const double l = stage_lds ? lm[t] : rec[t].m;
            z = t == 0 ? l : z + l;
        }
        v.logZ[th] = z;
    }
}

inline bool resident_supported(int model, int seg) {
    const int d = model_dim_rt(model);
    if (d < 0) return false;
    return (size_t)seg * 8 * (d + 1) + 2048 <= 160 * 1024;
}

template <int MODEL, int THREADS, int NP>
static hipError_t launch_resident_t(const FilterView& v, int T, StepRec* recs, hipStream_t s) {
    const size_t lds = resident_lds_bytes<MODEL>(2 * NP * THREADS, THREADS, NP);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k_resident<MODEL, THREADS, NP>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_resident<MODEL, THREADS, NP>), dim3(v.ntheta), dim3(THREADS), lds, s, v, T, recs);
    return hipGetLastError();
}

template <int MODEL>
static hipError_t launch_resident_m(const FilterView& v, int T, StepRec* recs, hipStream_t s) {
    switch (v.seg) {
    case 256: return launch_resident_t<MODEL, 128, 1>(v, T, recs, s);
    case 512: return launch_resident_t<MODEL, 256, 1>(v, T, recs, s);
    case 1024: return launch_resident_t<MODEL, 256, 2>(v, T, recs, s);
    case 2048: return launch_resident_t<MODEL, 512, 2>(v, T, recs, s);
    case 4096: return launch_resident_t<MODEL, 1024, 2>(v, T, recs, s);
    case 8192:
        if constexpr (model_dim<MODEL>::value == 1) return launch_resident_t<MODEL, 1024, 4>(v, T, recs, s);
        else return hipErrorInvalidValue;
    }
    return hipErrorInvalidValue;
}

static hipError_t launch_resident(int model, const FilterView& v, int T, StepRec* recs, hipStream_t s) {
    switch (model) {
    case MODEL_LG1D: return launch_resident_m<MODEL_LG1D>(v, T, recs, s);
    case MODEL_SV1D: return launch_resident_m<MODEL_SV1D>(v, T, recs, s);
    case MODEL_UCSV3D: return launch_resident_m<MODEL_UCSV3D>(v, T, recs, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace smc
