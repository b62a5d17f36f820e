// smc_persist.hip -- the OPT-IN persistent step kernel (k_persist, smc_kernels.h; SMC_PERSIST=1), one translation unit of its
// own for all three model families: it is compiled with SMC_FMAK_PINNED (smc_spec.h), which keeps the polynomial constants of
// exp / log / sincos next to their uses inside the step loop.  Measured slower than one launch per step.
#define SMC_FMAK_PINNED 1
#define SMC_TID_OPAQUE 1
#include "smc_launch.h"
#include <cstdlib>

namespace smc {

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

template <int SMC_MODEL, int THREADS, int NP>
static hipError_t persist_t(const FilterView& v, int cur, uint32_t t0, uint32_t t1, PersistCtl pc, hipStream_t s) {
    const size_t lds = step_lds_bytes(v.nseg_p2, THREADS, NP, true);
    static bool raised[16] = {};
    hipError_t e = raise_lds_limit(k_persist<SMC_MODEL, THREADS, NP>, lds, raised);
    if (e != hipSuccess) return e;
    // every workgroup must be resident at once: the occupancy the runtime reports for this kernel times the CUs of the device
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if ((e = hipGetDevice(&dev)) != hipSuccess || (e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
    if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_persist<SMC_MODEL, THREADS, NP>, THREADS, lds)) != hipSuccess) return e;
    if ((long long)v.nseg * v.ntheta > (long long)per_cu * prop.multiProcessorCount) return hipErrorCooperativeLaunchTooLarge;
    hipLaunchKernelGGL((k_persist<SMC_MODEL, THREADS, NP>), dim3(v.nseg, v.ntheta), dim3(THREADS), lds, s, v, cur, t0, t1, pc);
    return hipGetLastError();
}
template <int SMC_MODEL>
static hipError_t launch_persist_t(const FilterView& v, Geo g, int cur, uint32_t t0, uint32_t t1, PersistCtl pc, hipStream_t s) {
    // the default geometries of segments of 256 .. 2048 particles (three staged segments, one record per thread)
    if (v.nseg < 2 || v.nseg_p2 > g.threads || v.systematic) return hipErrorCooperativeLaunchTooLarge;
    switch (g.threads * 8 + g.np) {
    case 128 * 8 + 1: return persist_t<SMC_MODEL, 128, 1>(v, cur, t0, t1, pc, s);
    case 256 * 8 + 1: return persist_t<SMC_MODEL, 256, 1>(v, cur, t0, t1, pc, s);
    case 512 * 8 + 1: return persist_t<SMC_MODEL, 512, 1>(v, cur, t0, t1, pc, s);
    case 512 * 8 + 2: return persist_t<SMC_MODEL, 512, 2>(v, cur, t0, t1, pc, s);
    }
    return hipErrorCooperativeLaunchTooLarge;
}

template <> hipError_t launch_persist<MODEL_LG1D>(const FilterView& v, Geo g, int cur, uint32_t t0, uint32_t t1, PersistCtl pc, hipStream_t s) {
    return launch_persist_t<MODEL_LG1D>(v, g, cur, t0, t1, pc, s);
}
template <> hipError_t launch_persist<MODEL_SV1D>(const FilterView& v, Geo g, int cur, uint32_t t0, uint32_t t1, PersistCtl pc, hipStream_t s) {
    return launch_persist_t<MODEL_SV1D>(v, g, cur, t0, t1, pc, s);
}
template <> hipError_t launch_persist<MODEL_UCSV3D>(const FilterView& v, Geo g, int cur, uint32_t t0, uint32_t t1, PersistCtl pc, hipStream_t s) {
    return launch_persist_t<MODEL_UCSV3D>(v, g, cur, t0, t1, pc, s);
}

}  // namespace smc
