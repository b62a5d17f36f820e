// smc_launch.h -- launch entry points, one explicit specialisation set per model family
// (smc_model.hip is compiled once per model with -DSMC_MODEL=<id>, in parallel).
#pragma once
#include "smc_kernels.h"
#include "smc_resident.h"

namespace smc {

// geometry of a workgroup: SEG = 2 * np * threads
struct Geo {
    int threads, np;
};
// the default geometry of a segment size, and whether (threads, np) is an instantiated one
bool geo_default(int seg, Geo& g);
bool geo_valid(int seg, int np, Geo& g);

template <int MODEL> hipError_t launch_init(const FilterView& v, Geo g, int nxt, double y, hipStream_t s);
template <int MODEL> hipError_t launch_step(const FilterView& v, Geo g, int cur, uint32_t t, int emit_prev, double y, hipStream_t s);
template <int MODEL> hipError_t launch_resident(const FilterView& v, int T, StepRec* recs, hipStream_t s);
// opt-in persistent step kernel: the steps [t0, t1) of a multi-segment filter in one launch; hipErrorCooperativeLaunchTooLarge when
// the grid cannot be resident all at once (or the geometry has no instantiation): the caller then launches step by step
template <int MODEL> hipError_t launch_persist(const FilterView& v, Geo g, int cur, uint32_t t0, uint32_t t1, PersistCtl pc, hipStream_t s);
// summaries (quantile levels / moments named by the view's sum_* fields, row 0) of the current state of single-segment filters in
// one launch; hipErrorInvalidValue when the segment length has no instantiation or the state does not fit LDS
template <int MODEL> hipError_t launch_summ_once(const FilterView& v, int cur, hipStream_t s);
// window mode: steps [t0, t0 + T) from the state in buffer bin to buffer bout, (logmu, ess) of every step to win
template <int MODEL> hipError_t launch_window(const FilterView& v, int T, StepRec* recs, int t0, int bin, int bout, double* win, hipStream_t s);

}  // namespace smc
