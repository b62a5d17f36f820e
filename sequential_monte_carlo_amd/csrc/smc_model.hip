// smc_model.hip -- instantiates the kernels of ONE model family (-DSMC_MODEL=1|2|3) for every
// workgroup geometry.  Three objects are built in parallel and linked into libsmchip.so.
#include "smc_launch.h"
#include <cstdlib>

#ifndef SMC_MODEL
#error "compile with -DSMC_MODEL=<model id>"
#endif

namespace smc {

template <int THREADS, int NP>
static hipError_t init_t(const FilterView& v, int nxt, double y, hipStream_t s) {
    const size_t lds = scr_words(THREADS, NP) * 8;
    hipLaunchKernelGGL((k_init<SMC_MODEL, THREADS, NP>), dim3(v.nseg, v.ntheta), dim3(THREADS), lds, s, v, nxt, y);
    return hipGetLastError();
}
template <int THREADS, int NP, bool SYS>
static hipError_t step_sys_t(const FilterView& v, int cur, uint32_t t, int emit_prev, double y, hipStream_t s) {
    const size_t lds = step_lds_bytes(v.nseg_p2, THREADS, NP, v.nseg > 1);
    if (lds > 64 * 1024) {
        static bool raised[2] = {false, false};   // per instantiation
        const int m = v.nseg > 1 ? 1 : 0;
        if (!raised[m]) {
            hipError_t e = m ? hipFuncSetAttribute((const void*)k_step<SMC_MODEL, THREADS, NP, true, SYS>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                             : hipFuncSetAttribute((const void*)k_step<SMC_MODEL, THREADS, NP, false, SYS>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            raised[m] = true;
        }
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

template <int THREADS, int NP, bool SYS>
static hipError_t resident_sys_t(const FilterView& v, int T, StepRec* recs, hipStream_t s) {
    const size_t lds = resident_lds_bytes<SMC_MODEL>(2 * NP * THREADS, THREADS, NP);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k_resident<SMC_MODEL, THREADS, NP, SYS>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_resident<SMC_MODEL, THREADS, NP, SYS>), dim3(v.ntheta), dim3(THREADS), lds, s, v, T, recs);
    return hipGetLastError();
}
template <int THREADS, int NP>
static hipError_t resident_t(const FilterView& v, int T, StepRec* recs, hipStream_t s) {
    return v.systematic ? resident_sys_t<THREADS, NP, true>(v, T, recs, s) : resident_sys_t<THREADS, NP, false>(v, T, recs, s);
}

template <>
hipError_t launch_resident<SMC_MODEL>(const FilterView& v, int T, StepRec* recs, hipStream_t s) {
    const char* e = getenv("SMC_RES_NP");   // tuning knob
    // measured (scripts/res_tune.py): with few filters in flight (<= 4 waves per SIMD at two pairs per
    // thread) the cheap LG model runs faster with one pair per thread (twice the waves); SV / UCSV and
    // large batches prefer two pairs per thread
    const int np = e ? atoi(e) : (SMC_MODEL == MODEL_LG1D && v.seg == 1024 && v.ntheta <= 1024) ? 1 : 0;
    switch (v.seg) {
    case 256: return resident_t<128, 1>(v, T, recs, s);
    case 512: return resident_t<256, 1>(v, T, recs, s);
    case 1024: return np == 1 ? resident_t<512, 1>(v, T, recs, s) : np == 4 ? resident_t<128, 4>(v, T, recs, s) : resident_t<256, 2>(v, T, recs, s);
    case 2048: return np == 1 ? resident_t<1024, 1>(v, T, recs, s) : np == 4 ? resident_t<256, 4>(v, T, recs, s) : resident_t<512, 2>(v, T, recs, s);
    case 4096: return np == 2 ? resident_t<1024, 2>(v, T, recs, s) : resident_t<512, 4>(v, T, recs, s);   // 1024 threads spill
    case 8192:
        if constexpr (model_dim<SMC_MODEL>::value == 1) return resident_t<1024, 4>(v, T, recs, s);
        else return hipErrorInvalidValue;
    }
    return hipErrorInvalidValue;
}

}  // namespace smc
