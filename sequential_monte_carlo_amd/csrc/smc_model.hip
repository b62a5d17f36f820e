// smc_model.hip -- instantiates the kernels of ONE model family (-DSMC_MODEL=1|2|3) for every
// workgroup geometry.  Three objects are built in parallel and linked into libsmchip.so.
#include "smc_launch.h"
#include <cstdlib>

#ifndef SMC_MODEL
#error "compile with -DSMC_MODEL=<model id>"
#endif

namespace smc {

// the attribute is per device: raise it wherever this process has not done so yet (cheap: a table lookup afterwards)
template <class K>
static hipError_t raise_lds_limit(K kernel, size_t lds, bool (&raised)[16]) {
    if (lds <= 64 * 1024) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 16 && raised[dev]) return hipSuccess;
    e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess && dev >= 0 && dev < 16) raised[dev] = true;
    return e;
}


template <int THREADS, int NP>
static hipError_t init_t(const FilterView& v, int nxt, double y, hipStream_t s) {
    const size_t lds = scr_words(THREADS, NP) * 8;
    hipLaunchKernelGGL((k_init<SMC_MODEL, THREADS, NP>), dim3(v.nseg, v.ntheta), dim3(THREADS), lds, s, v, nxt, y);
    return hipGetLastError();
}
template <int THREADS, int NP, bool SYS>
static hipError_t step_sys_t(const FilterView& v, int cur, uint32_t t, int emit_prev, double y, hipStream_t s) {
    if (v.tabD) {   // the table comes from k_table (launched by the caller before this step): no table in LDS, nothing to emit
        const size_t lds = step_lds_bytes(0, THREADS, NP, true);
        static bool raised[16] = {};
        hipError_t e = raise_lds_limit(k_step<SMC_MODEL, THREADS, NP, true, SYS, true>, lds, raised);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_step<SMC_MODEL, THREADS, NP, true, SYS, true>), dim3(v.nseg, v.ntheta), dim3(THREADS), lds, s, v, cur, t, 0, y);
        return hipGetLastError();
    }
    const size_t lds = step_lds_bytes(v.nseg_p2, THREADS, NP, v.nseg > 1);
    if (v.nseg_p2 > THREADS) {   // up to twice as many segments as threads: the window prologue with two records per thread
        static bool raised[16] = {};
        hipError_t e = raise_lds_limit(k_step<SMC_MODEL, THREADS, NP, true, SYS, false, 2>, lds, raised);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_step<SMC_MODEL, THREADS, NP, true, SYS, false, 2>), dim3(v.nseg, v.ntheta), dim3(THREADS), lds, s, v, cur, t, emit_prev, y);
        return hipGetLastError();
    }
    {
        static bool raised[2][16] = {};   // per instantiation, variant and device
        hipError_t e = v.nseg > 1 ? raise_lds_limit(k_step<SMC_MODEL, THREADS, NP, true, SYS>, lds, raised[1])
                                  : raise_lds_limit(k_step<SMC_MODEL, THREADS, NP, false, SYS>, lds, raised[0]);
        if (e != hipSuccess) return e;
    }
    if (v.nseg > 1)
        hipLaunchKernelGGL((k_step<SMC_MODEL, THREADS, NP, true, SYS>), dim3(v.nseg, v.ntheta), dim3(THREADS), lds, s, v, cur, t,
                           emit_prev, y);
    else
        hipLaunchKernelGGL((k_step<SMC_MODEL, THREADS, NP, false, SYS>), dim3(v.nseg, v.ntheta), dim3(THREADS), lds, s, v, cur, t,
                           emit_prev, y);
    return hipGetLastError();
}
template <int THREADS, int NP>
static hipError_t step_t(const FilterView& v, int cur, uint32_t t, int emit_prev, double y, hipStream_t s) {
    return v.systematic ? step_sys_t<THREADS, NP, true>(v, cur, t, emit_prev, y, s)
                        : step_sys_t<THREADS, NP, false>(v, cur, t, emit_prev, y, s);
}

#define SMC_GEO_SWITCH(FN, ...)                                                   \
    switch (g.threads * 8 + g.np) {                                               \
    case 64 * 8 + 2: return FN<64, 2>(__VA_ARGS__);                               \
    case 128 * 8 + 1: return FN<128, 1>(__VA_ARGS__);                             \
    case 128 * 8 + 2: return FN<128, 2>(__VA_ARGS__);                             \
    case 128 * 8 + 4: return FN<128, 4>(__VA_ARGS__);                             \
    case 256 * 8 + 1: return FN<256, 1>(__VA_ARGS__);                             \
    case 256 * 8 + 2: return FN<256, 2>(__VA_ARGS__);                             \
    case 256 * 8 + 4: return FN<256, 4>(__VA_ARGS__);                             \
    case 512 * 8 + 1: return FN<512, 1>(__VA_ARGS__);                             \
    case 512 * 8 + 2: return FN<512, 2>(__VA_ARGS__);                             \
    case 512 * 8 + 4: return FN<512, 4>(__VA_ARGS__);                             \
    case 1024 * 8 + 1: return FN<1024, 1>(__VA_ARGS__);                           \
    case 1024 * 8 + 2: return FN<1024, 2>(__VA_ARGS__);                           \
    case 1024 * 8 + 4: return FN<1024, 4>(__VA_ARGS__);                           \
    }                                                                             \
    return hipErrorInvalidValue;

template <>
hipError_t launch_init<SMC_MODEL>(const FilterView& v, Geo g, int nxt, double y, hipStream_t s) {
    SMC_GEO_SWITCH(init_t, v, nxt, y, s)
}
template <>
hipError_t launch_step<SMC_MODEL>(const FilterView& v, Geo g, int cur, uint32_t t, int emit_prev, double y, hipStream_t s) {
    SMC_GEO_SWITCH(step_t, v, cur, t, emit_prev, y, s)
}

template <int THREADS, int NP, bool SYS, bool WIN, bool SUMM = false>
static hipError_t resident_sys_t(const FilterView& v, int T, StepRec* recs, int t0, int bin, int bout, double* win, hipStream_t s) {
    const size_t lds = resident_lds_bytes<SMC_MODEL>(2 * NP * THREADS, THREADS, NP, SUMM ? v.sum_np : -1);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static bool raised[16] = {};   // per instantiation and device
    hipError_t e = raise_lds_limit(k_resident<SMC_MODEL, THREADS, NP, SYS, WIN, SUMM>, lds, raised);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_resident<SMC_MODEL, THREADS, NP, SYS, WIN, SUMM>), dim3(v.ntheta), dim3(THREADS), lds, s, v, T, recs, t0, bin, bout, win);
    return hipGetLastError();
}
// per-step summaries (smc_set_summaries) select the SUMM kernels; they exist for the multinomial default only (the C ABI sends
// a systematic filter with summaries through the one-launch-per-step path)
template <int THREADS, int NP>
static hipError_t resident_t(const FilterView& v, int T, StepRec* recs, hipStream_t s) {
    if (v.sum_np || v.sum_mom) return resident_sys_t<THREADS, NP, false, false, true>(v, T, recs, 0, 0, 0, nullptr, s);
    return v.systematic ? resident_sys_t<THREADS, NP, true, false>(v, T, recs, 0, 0, 0, nullptr, s)
                        : resident_sys_t<THREADS, NP, false, false>(v, T, recs, 0, 0, 0, nullptr, s);
}
template <int THREADS, int NP>
static hipError_t window_t(const FilterView& v, int T, StepRec* recs, int t0, int bin, int bout, double* win, hipStream_t s) {
    if (v.sum_np || v.sum_mom) return resident_sys_t<THREADS, NP, false, true, true>(v, T, recs, t0, bin, bout, win, s);
    return v.systematic ? resident_sys_t<THREADS, NP, true, true>(v, T, recs, t0, bin, bout, win, s)
                        : resident_sys_t<THREADS, NP, false, true>(v, T, recs, t0, bin, bout, win, s);
}

// threads / particle pairs per thread of the LDS-resident kernels for a segment length.  Measured (scripts/res_tune.py,
// scripts/prof_c2.py, scripts/dbg/res1024.py): with at most two workgroups per CU in flight (n_theta <= 512) every model runs faster
// with one pair per thread (twice the waves: 512 UCSV filters 2.05 -> 1.81 ms); beyond, workgroups of half the size with two
// pairs per thread pack the CUs without a ragged last round (576 .. 768 filters: +30 % LG, +11 % UCSV; 1024: +8 %).  SMC_RES_NP
// (1, 2 or 4) overrides the choice for tuning runs; results do not depend on it (tests/test_gpu_parity.py::test_launch_geometry_knobs).
static int resident_np(const FilterView& v) {
    const char* e = getenv("SMC_RES_NP");
    return e ? atoi(e) : (v.seg == 1024 && v.ntheta <= 512) ? 1 : 0;
}

template <>
hipError_t launch_resident<SMC_MODEL>(const FilterView& v, int T, StepRec* recs, hipStream_t s) {
    const int np = resident_np(v);
    switch (v.seg) {
    case 256: return resident_t<128, 1>(v, T, recs, s);
    case 512: return resident_t<256, 1>(v, T, recs, s);
    case 1024: return np == 1 ? resident_t<512, 1>(v, T, recs, s) : np == 4 ? resident_t<128, 4>(v, T, recs, s) : resident_t<256, 2>(v, T, recs, s);
    case 2048: return np == 1 ? resident_t<1024, 1>(v, T, recs, s) : np == 4 ? resident_t<256, 4>(v, T, recs, s) : resident_t<512, 2>(v, T, recs, s);
    case 4096: return np == 4 ? resident_t<512, 4>(v, T, recs, s) : resident_t<1024, 2>(v, T, recs, s);   // (1024 x 2 spills a little and still wins: +5-10 %, scripts/dbg/res4096.py)
    case 8192:
        if constexpr (model_dim<SMC_MODEL>::value == 1) return resident_t<1024, 4>(v, T, recs, s);
        else return hipErrorInvalidValue;
    }
    return hipErrorInvalidValue;
}

template <int THREADS, int NP>
static hipError_t summ_once_t(const FilterView& v, int cur, hipStream_t s) {
    constexpr int D = model_dim<SMC_MODEL>::value;
    const size_t lds = summ_once_lds_bytes<D>(2 * NP * THREADS, v.sum_np);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static bool raised[16] = {};
    hipError_t e = raise_lds_limit(k_summ_once<THREADS, NP, D>, lds, raised);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_summ_once<THREADS, NP, D>), dim3(v.ntheta), dim3(THREADS), lds, s, v, cur);
    return hipGetLastError();
}
template <>
hipError_t launch_summ_once<SMC_MODEL>(const FilterView& v, int cur, hipStream_t s) {
    if (v.nseg != 1) return hipErrorInvalidValue;
    switch (v.seg) {
    case 256: return summ_once_t<128, 1>(v, cur, s);
    case 512: return summ_once_t<256, 1>(v, cur, s);
    case 1024: return summ_once_t<512, 1>(v, cur, s);
    case 2048: return summ_once_t<512, 2>(v, cur, s);
    case 4096: return summ_once_t<512, 4>(v, cur, s);
    case 8192: return summ_once_t<1024, 4>(v, cur, s);
    }
    return hipErrorInvalidValue;
}

template <>
hipError_t launch_window<SMC_MODEL>(const FilterView& v, int T, StepRec* recs, int t0, int bin, int bout, double* win, hipStream_t s) {
    const int np = resident_np(v);
    switch (v.seg) {
    case 256: return window_t<128, 1>(v, T, recs, t0, bin, bout, win, s);
    case 512: return window_t<256, 1>(v, T, recs, t0, bin, bout, win, s);
    case 1024: return np == 1 ? window_t<512, 1>(v, T, recs, t0, bin, bout, win, s) : window_t<256, 2>(v, T, recs, t0, bin, bout, win, s);
    case 2048: return window_t<512, 2>(v, T, recs, t0, bin, bout, win, s);
    case 4096: return np == 4 ? window_t<512, 4>(v, T, recs, t0, bin, bout, win, s) : window_t<1024, 2>(v, T, recs, t0, bin, bout, win, s);
    case 8192:
        if constexpr (model_dim<SMC_MODEL>::value == 1) return window_t<1024, 4>(v, T, recs, t0, bin, bout, win, s);
        else return hipErrorInvalidValue;
    }
    return hipErrorInvalidValue;
}

}  // namespace smc
