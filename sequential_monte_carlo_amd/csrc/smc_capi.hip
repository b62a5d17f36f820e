// smc_capi.hip -- C ABI (include/smc_hip.h) over the gfx950 kernels in smc_kernels.h.
// Handle-owned device state, one HIP stream per handle, HIP-event timing of every call.
// There is NO CPU fallback: every filter entry point launches HIP kernels or fails.
#include "../../include/smc_hip.h"
#include "smc_launch.h"
#include "smc_aux_kernels.h"
#include "smc_summ_kernels.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

using namespace smc;

static thread_local std::string g_err;
#ifdef SMC_ABLATE
static int h_abl_tmp = 0;
#endif
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
extern "C" int smc_set_error_(int code, const char* msg) { return fail(code, msg ? msg : ""); }   // for smc_comm.hip
#define HIPCHK(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess)                                                                                 \
            return fail(SMC_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e) + " (" __FILE__ ":" +      \
                                      std::to_string(__LINE__) + ")");                                        \
    } while (0)

struct smc_filter_s {
    int model = 0, d = 0, device = 0;
    uint32_t flags = 0;
    FilterView v{};
    Params* d_params = nullptr;
    uint32_t* d_stream = nullptr;
    int32_t* d_perm = nullptr;
    double* d_logZ_tmp = nullptr;
    double* d_y = nullptr;
    int64_t ycap = 0;
    double *d_tr_logmu = nullptr, *d_tr_ess = nullptr;
    int64_t trcap = 0;
    double* d_wdense = nullptr;
    StepRec* d_recs = nullptr;
    double* h_pin = nullptr;                   // pinned host mirror [4][ntheta]: logZ | last_logmu | last_ess | ticket of the step API
    uint32_t seq = 0;                          // last ticket handed to a step-API launch
    size_t slab_bytes = 0, pin_bytes = 0;
    char* d_slab = nullptr;                    // ONE allocation behind x, C, the segment records, the per-filter scalars, params / streams / perm
    uint64_t* d_ms = nullptr;                  // scratch of the summaries of multi-segment filters (smc_summ_kernels.h) + [ntheta][QMAX] results
    uint64_t* d_brk = nullptr;                 // break points of the steps [v.brk_t0, v.brk_t0 + brk_count)
    uint32_t brk_cap = 0, brk_count = 0;
    int64_t reccap = 0;
    unsigned char* d_skip = nullptr;           // smc_set_skip: filters log_likelihood leaves out
    int32_t* d_order = nullptr;                //   and the order the one-workgroup-per-filter kernel takes them in: [ntheta] | n_active
    bool skip_on = false;
    // per-step summaries inside the multi-step calls (smc_set_summaries / smc_get_summaries)
    int sum_np = 0, sum_comp = 0, sum_mom = 0;
    uint64_t sum_p64[QMAX] = {};
    double *d_sum_q = nullptr, *d_sum_m = nullptr;   // [T][ntheta][np] | [T][2][d][ntheta]
    int64_t sum_cap = 0, sum_T = 0;            // steps the traces hold / steps the last call recorded
    unsigned* d_pflags = nullptr;              // opt-in persistent step kernel (SMC_PERSIST=1): completion flags [2][ntheta * nseg]
    int* h_perr = nullptr;                     //   and its pinned "a spin expired" word
    int persist = -1;                          //   -1 not decided yet, 0 off / unavailable, 1 on
    double* h_win = nullptr;                   // pinned [2][WIN_MAX][ntheta]: (logmu, ess) of the steps of a window
    double* h_once = nullptr;                  // pinned [QMAX + 2 d][ntheta]: quantiles / moments of the current state (k_summ_once)
    int win_k = 0;                             // steps of the pending window (smc_step_window), 0 = none
    // PMMH rejuvenation state (smc_pmmh_configure / smc_pmmh_rejuvenate): this handle holds the proposal filters
    PmmhSpec pm_spec{};
    bool pm_cfg = false;
    PmmhDev pm{};
    double* h_pm_out = nullptr;                // pinned mirror: theta [ntheta][d] | logZ [ntheta] | any [ntheta] | nrun
    int32_t* h_perm = nullptr;                 // pinned copy of smc_permute's index vector (the call does not wait for the device)
    Params* h_params = nullptr;                // pinned twin of d_params (smc_set_params does not wait either)
    double* pm_in = nullptr;                   // ONE device block: pm.theta | pm.logZ | pm.chol | pm.nrun | pm.counts | pm.any -
    double* h_pm_in = nullptr;                 //   a rejuvenation call fills its pinned twin and uploads it in one copy
    size_t pm_in_words = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int cur = 0;
    uint32_t t = 0;        // index of the next observation
    bool inited = false;   // weights exist
    bool emitted = false;  // logmu/ess of the current weights already produced
    bool have_params = false;
    bool resident_ok = false;
    Geo geo{0, 0};
    double last_ms = 0.0;
};

// ---- geometry ------------------------------------------------------------------------------
namespace smc {
bool geo_valid(int seg, int np, Geo& g) {
    if (np != 1 && np != 2 && np != 4) return false;
    const int th = seg / (2 * np);
    if (th * 2 * np != seg) return false;
    if (!(th == 128 || th == 256 || th == 512 || th == 1024 || (th == 64 && np == 2))) return false;
    g = {th, np};
    return true;
}
bool geo_default(int seg, Geo& g) {
    switch (seg) {
    case 256: g = {128, 1}; return true;
    case 512: g = {256, 1}; return true;
    case 1024: g = {512, 1}; return true;
    case 2048: g = {512, 2}; return true;
    case 4096: g = {1024, 2}; return true;   // (measured: 2^26 particles 1109 -> 1055 us per step against 512 x 4)
    case 8192: g = {1024, 4}; return true;
    }
    return false;
}
}  // namespace smc

// most segments a filter may have (the break points of a step sit in the LDS of k_breaks: 8 B per segment)
constexpr int MAX_NSEG = 16384;
extern "C" int smc_auto_seg(int model_id, int64_t n) {
    // measured (scripts/nx_sweep.py, scripts/dbg/mid_sizes.py, scripts/dbg/ucsv_seg_sweep.py): one segment for as long as the LDS-
    // resident kernels hold the filter (8192 particles of one state coordinate, 4096 of three: the batched callers' shape);
    // above, the SHORTEST segment whose workgroup still has a thread per segment (nseg <= seg / 2 resp. 512: the one-record-per-
    // thread window prologue) - a filter of 2^14..2^18 particles is a handful of workgroups on a 256-CU chip and bound by the
    // life of ONE of them, which shorter segments (fewer particles per workgroup) cut from 9.4 to 6.0-7.3 us per step; then the
    // fastest geometry for as long as the segment count allows - one state coordinate: 2048 (512 threads x two pairs, 12.5 us per
    // 2^20 particles) from 2^19 particles on; three coordinates: 1024 (512 threads x ONE pair: with two pairs the step kernel
    // needs 183 vector registers and a CU holds one workgroup instead of two - 2^20 UCSV particles 2.9 -> 3.4e10 p-steps/s);
    // beyond 512 segments the table of the segments is built once per step by k_table instead of by every workgroup; longer
    // segments keep the count at MAX_NSEG for still larger filters.  The segment length is part of the numerical spec (the CPU
    // restatement used by the tests follows the same rule).
    const bool d3 = model_dim_rt(model_id) == 3;
    if (n > (int64_t)MAX_NSEG * 4096) return 8192;
    if (n > (int64_t)MAX_NSEG * 2048) return 4096;
    if (n > (int64_t)MAX_NSEG * 1024) return 2048;
    if (n > ((int64_t)1 << 19)) return d3 ? 1024 : 2048;
    if (n > ((int64_t)1 << 17)) return 1024;
    if (n > ((int64_t)1 << 15)) return 512;
    if (n > (d3 ? 4096 : MAX_SEG)) return 256;
    int s = 256;
    while (s < n) s <<= 1;
    return s;
}
extern "C" int smc_model_dim(int id) { return model_dim_rt(id); }
extern "C" int smc_model_nraw(int id) { return model_nraw_rt(id); }
extern "C" const char* smc_last_error(void) { return g_err.c_str(); }
extern "C" const char* smc_version(void) { return "smchip 0.1 (gfx950)"; }
extern "C" int smc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- kernel dispatch -------------------------------------------------------------------------
static hipError_t do_init(smc_filter_s* h, double y) {
    switch (h->model) {
    case MODEL_LG1D: return launch_init<MODEL_LG1D>(h->v, h->geo, h->cur, y, h->stream);
    case MODEL_SV1D: return launch_init<MODEL_SV1D>(h->v, h->geo, h->cur, y, h->stream);
    case MODEL_UCSV3D: return launch_init<MODEL_UCSV3D>(h->v, h->geo, h->cur, y, h->stream);
    }
    return hipErrorInvalidValue;
}
// filters with more segments than threads per workgroup (v.tabD): the segment table of the current weights, built once (k_table;
// emit as k_step's emit_prev: 0 nothing, 1 (logmu, ess) from the full records, 2 from the totals)
static hipError_t do_table(smc_filter_s* h, int emit, int first_emit, uint32_t t_emit) {
    constexpr int TH = 1024;
    hipLaunchKernelGGL((k_table<TH>), dim3(h->v.ntheta), dim3(TH), 0, h->stream, h->v, h->cur, emit, first_emit, t_emit);
    return hipGetLastError();
}
static hipError_t do_step(smc_filter_s* h, uint32_t t, int emit_prev, double y) {
    if (h->v.tabD) {
        hipError_t e = do_table(h, emit_prev, t == 1u ? 1 : 0, t - 1u);
        if (e != hipSuccess) return e;
        emit_prev = 0;
    }
    switch (h->model) {
    case MODEL_LG1D: return launch_step<MODEL_LG1D>(h->v, h->geo, h->cur, t, emit_prev, y, h->stream);
    case MODEL_SV1D: return launch_step<MODEL_SV1D>(h->v, h->geo, h->cur, t, emit_prev, y, h->stream);
    case MODEL_UCSV3D: return launch_step<MODEL_UCSV3D>(h->v, h->geo, h->cur, t, emit_prev, y, h->stream);
    }
    return hipErrorInvalidValue;
}
// Break points of the multinomial resampling steps (multi-segment filters; smc_spec.h): computed by k_breaks
// for a window of steps ahead of time - they depend on (seed, stream, t) only.  Called before every step
// launch; almost always a no-op.
static hipError_t ensure_breaks(smc_filter_s* h, uint32_t t, uint32_t t_end) {
    FilterView& v = h->v;
    if (v.nseg <= 1 || v.systematic) return hipSuccess;
    if (h->brk_count && t >= v.brk_t0 && t < v.brk_t0 + h->brk_count) return hipSuccess;
    const size_t per_step = (size_t)v.ntheta * ((size_t)v.nseg + 1);
    if (!h->d_brk) {
        size_t steps = ((size_t)32 << 20) / (per_step * 8);    // <= 32 MiB of break points at a time
        steps = steps < 1 ? 1 : (steps > 1024 ? 1024 : steps);
        hipError_t e = hipMalloc((void**)&h->d_brk, steps * per_step * 8);
        if (e != hipSuccess) return e;
        h->brk_cap = (uint32_t)steps;
        v.brk = h->d_brk;
    }
    uint32_t cnt = t_end > t ? t_end - t : 1;                  // no further than the caller will go
    cnt = cnt > h->brk_cap ? h->brk_cap : cnt;
    constexpr int TH = 256;
    const size_t lds = ((size_t)v.nseg + 1 + TH / WAVE) * 8;
    if (lds > 64 * 1024) {   // filters of more than 8187 segments: beyond the default limit of dynamic LDS
        static bool raised[16] = {};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= 16 || !raised[dev]) {
            e = hipFuncSetAttribute((const void*)k_breaks<TH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            if (dev >= 0 && dev < 16) raised[dev] = true;
        }
    }
    hipLaunchKernelGGL((k_breaks<TH>), dim3(cnt, v.ntheta), dim3(TH), lds, h->stream, v, t, h->d_brk);
    v.brk_t0 = t;
    h->brk_count = cnt;
    return hipGetLastError();
}
static hipError_t do_persist(smc_filter_s* h, uint32_t t0, uint32_t t1, PersistCtl pc) {
    switch (h->model) {
    case MODEL_LG1D: return launch_persist<MODEL_LG1D>(h->v, h->geo, h->cur, t0, t1, pc, h->stream);
    case MODEL_SV1D: return launch_persist<MODEL_SV1D>(h->v, h->geo, h->cur, t0, t1, pc, h->stream);
    case MODEL_UCSV3D: return launch_persist<MODEL_UCSV3D>(h->v, h->geo, h->cur, t0, t1, pc, h->stream);
    }
    return hipErrorInvalidValue;
}
static hipError_t do_resident(smc_filter_s* h, int T) {
    switch (h->model) {
    case MODEL_LG1D: return launch_resident<MODEL_LG1D>(h->v, T, h->d_recs, h->stream);
    case MODEL_SV1D: return launch_resident<MODEL_SV1D>(h->v, T, h->d_recs, h->stream);
    case MODEL_UCSV3D: return launch_resident<MODEL_UCSV3D>(h->v, T, h->d_recs, h->stream);
    }
    return hipErrorInvalidValue;
}

static hipError_t do_window(smc_filter_s* h, int k, int bout) {
    switch (h->model) {
    case MODEL_LG1D: return launch_window<MODEL_LG1D>(h->v, k, h->d_recs, (int)h->t, h->cur, bout, h->h_win, h->stream);
    case MODEL_SV1D: return launch_window<MODEL_SV1D>(h->v, k, h->d_recs, (int)h->t, h->cur, bout, h->h_win, h->stream);
    case MODEL_UCSV3D: return launch_window<MODEL_UCSV3D>(h->v, k, h->d_recs, (int)h->t, h->cur, bout, h->h_win, h->stream);
    }
    return hipErrorInvalidValue;
}

static hipError_t do_finalize(smc_filter_s* h, int first_emit, uint32_t t_emit) {
    if (h->v.tabD) return do_table(h, 1, first_emit, t_emit);   // (no room in LDS for a table of that many segments)
    constexpr int TH = 256;
    const size_t lds = table_lds_bytes(h->v.nseg_p2, TH, 1);
    hipLaunchKernelGGL((k_finalize<TH>), dim3(h->v.ntheta), dim3(TH), lds, h->stream, h->v, h->cur, first_emit, t_emit);
    return hipGetLastError();
}

// ---- lifetime ----------------------------------------------------------------------------------
template <class T>
static hipError_t dalloc(T** p, size_t count) {
    return hipMalloc((void**)p, count * sizeof(T) > 0 ? count * sizeof(T) : 16);
}

// What every handle needs and the runtime is slow to give back (hipFree, hipStreamDestroy and hipHostFree made smc_destroy cost
// 0.48 ms - twice the hundred steps of a 1024-particle filter): the slab, the stream, the two events and the pinned mirror of a
// destroyed handle are kept (a few, bounded in bytes) and handed to the next smc_create that fits them.
namespace {
struct Bundle {
    int device = -1;
    char* slab = nullptr; size_t slab_bytes = 0;
    double* pin = nullptr; size_t pin_bytes = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double* d_y = nullptr; int64_t ycap = 0;          // the series buffer and the per-step records of the whole-series calls
    void* d_recs = nullptr; size_t rec_bytes = 0;     // (whatever the destroyed handle had grown them to)
    void* h_params = nullptr; size_t params_bytes = 0;   // pinned twin of the parameter rows
};
struct BundleCache {
    std::mutex mu;
    std::vector<Bundle> free_list;
    size_t bytes = 0;
    static constexpr size_t MAX_BYTES = (size_t)512 << 20, MAX_COUNT = 8;
    bool take(int device, size_t slab_bytes, size_t pin_bytes, Bundle& out) {
        std::lock_guard<std::mutex> g(mu);
        for (size_t i = 0; i < free_list.size(); ++i) {
            const Bundle& b = free_list[i];
            if (b.device == device && b.slab_bytes >= slab_bytes && b.slab_bytes <= 2 * slab_bytes + 4096 && b.pin_bytes >= pin_bytes) {
                out = b;
                bytes -= b.slab_bytes;
                free_list.erase(free_list.begin() + (long)i);
                return true;
            }
        }
        return false;
    }
    bool give(const Bundle& b) {
        std::lock_guard<std::mutex> g(mu);
        if (free_list.size() >= MAX_COUNT || bytes + b.slab_bytes > MAX_BYTES) return false;
        free_list.push_back(b);
        bytes += b.slab_bytes;
        return true;
    }
};
BundleCache& bundle_cache() { static BundleCache* c = new BundleCache(); return *c; }   // (never destroyed: no runtime calls at exit)
}  // namespace

extern "C" int smc_create(int model_id, int64_t n_theta, int64_t n_x, int seg, uint64_t seed, int device, uint32_t flags,
                          smc_handle* out) {
    if (!out) return fail(SMC_EINVAL, "smc_create: out is NULL");
    *out = nullptr;
    const int d = model_dim_rt(model_id);
    if (d < 0) return fail(SMC_EINVAL, "smc_create: unknown model_id " + std::to_string(model_id));
    if (n_theta <= 0 || n_x <= 0) return fail(SMC_EINVAL, "smc_create: n_theta and n_x must be positive");
    if (n_theta > 65535) return fail(SMC_EINVAL, "smc_create: n_theta > 65535 (grid.y limit); shard theta");
    if (seg == 0) seg = smc_auto_seg(model_id, n_x);
    Geo g;
    if (!geo_default(seg, g)) return fail(SMC_EINVAL, "smc_create: seg must be a power of two in [256,8192]");
#ifdef SMC_ABLATE
    if (const char* e = getenv("SMC_ABL")) h_abl_tmp = atoi(e);
#endif
    if (const char* e = getenv("SMC_NP")) {   // tuning knob: particle pairs per thread (1, 2 or 4)
        Geo g2;
        if (geo_valid(seg, atoi(e), g2)) g = g2;
    }
    const int64_t nseg = (n_x + seg - 1) / seg;
    if (nseg > MAX_NSEG) return fail(SMC_EINVAL, "smc_create: more than 16384 segments; use a larger seg");
    if (n_x > ((int64_t)1 << 31)) return fail(SMC_EINVAL, "smc_create: n_x > 2^31");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev == 0) return fail(SMC_EHIP, "smc_create: no HIP device; this library has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(SMC_EINVAL, "smc_create: bad device index");
    HIPCHK(hipSetDevice(device));

    smc_filter_s* h = new smc_filter_s();
    h->model = model_id; h->d = d; h->device = device; h->flags = flags;
    h->geo = g;
    FilterView& v = h->v;
    v.n = n_x; v.seg = seg; v.nseg = (int)nseg; v.npad = nseg * seg; v.ntheta = (int)n_theta; v.seed = seed;
    int p2 = 1;
    while (p2 < v.nseg) p2 <<= 1;
    v.nseg_p2 = p2;
    v.SH = table_shift_extra(v.npad);
    v.want_s2 = 1;
#ifdef SMC_ABLATE
    v.abl = h_abl_tmp;
    if (getenv("SMC_DBG")) {
        if (hipMalloc((void**)&v.dbg, (size_t)v.ntheta * v.nseg * 128) != hipSuccess) v.dbg = nullptr;   // second half: k_persist's accumulators
        else (void)hipMemset(v.dbg, 0, (size_t)v.ntheta * v.nseg * 128);
    }
#endif
    h->resident_ok = (v.nseg == 1) && !(flags & SMC_FLAG_NO_RESIDENT);
    v.systematic = (flags & SMC_FLAG_SYSTEMATIC) ? 1 : 0;
    v.inv_n = 1.0 / (double)n_x;

    const size_t np = (size_t)v.ntheta * (size_t)v.npad, ns = (size_t)v.ntheta * (size_t)v.nseg, nt = (size_t)v.ntheta;
#define TRY(expr)                                       \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) {                         \
            std::string m = hipGetErrorString(_e);      \
            smc_destroy(h);                             \
            return fail(SMC_ENOMEM, "smc_create: " #expr ": " + m); \
        }                                               \
    } while (0)
    Bundle bun;   // filled from the cache once the sizes are known (below); fresh resources otherwise
    {   // every array the handle always owns, in ONE device allocation (smc_create + smc_destroy of a 1024-particle filter: 0.63 ->
        // 0.55 ms; thirty-odd hipMalloc / hipFree calls cost more than the filter's hundred steps)
        const bool gtab = v.nseg_p2 > 2 * g.threads;   // more than twice as many segments as a workgroup has threads: the segment table is built once per step (k_table)
        size_t off = 0;
        auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
        const size_t o_params = take(nt * sizeof(*h->d_params)), o_stream = take(nt * sizeof(*h->d_stream)), o_perm = take(nt * sizeof(*h->d_perm)),
                     o_ztmp = take(nt * 8);
        size_t o_x[2], o_C[2], o_k[2], o_S[2], o_hi[2], o_lo[2];
        for (int b = 0; b < 2; ++b) {
            o_x[b] = take(np * (size_t)d * 8); o_C[b] = take(np * 8);
            o_k[b] = take(ns * 8); o_S[b] = take(ns * 8); o_hi[b] = take(ns * 8); o_lo[b] = take(ns * 8);
        }
        const size_t o_anc = (flags & SMC_FLAG_ANCESTORS) ? take(np * 4) : 0;
        const size_t o_logZ = take(nt * 8), o_lm = take(nt * 8), o_es = take(nt * 8), o_K = take(nt * 8), o_D = take(nt * 8);
        const size_t o_tD = gtab ? take(nt * (size_t)v.nseg_p2 * 8) : 0, o_tsh = gtab ? take(nt * (size_t)v.nseg_p2 * 4) : 0;
        h->slab_bytes = off;
        if (bundle_cache().take(device, off, 4 * nt * 8, bun)) {
            h->d_slab = bun.slab; h->slab_bytes = bun.slab_bytes; h->stream = bun.stream; h->ev0 = bun.ev0; h->ev1 = bun.ev1;
            h->h_pin = bun.pin; h->pin_bytes = bun.pin_bytes;
            h->d_y = bun.d_y; h->ycap = bun.ycap;
            h->d_recs = (decltype(h->d_recs))bun.d_recs; h->reccap = (int64_t)(bun.rec_bytes / (nt * sizeof(*h->d_recs)));
            if (bun.params_bytes >= nt * sizeof(Params)) h->h_params = (Params*)bun.h_params;
            else if (bun.h_params) (void)hipHostFree(bun.h_params);
        } else {
            TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
            TRY(hipEventCreate(&h->ev0));
            TRY(hipEventCreate(&h->ev1));
            TRY(hipMalloc((void**)&h->d_slab, off));
        }
        TRY(hipMemsetAsync(h->d_slab, 0, off, h->stream));
        char* base = h->d_slab;
        h->d_params = (decltype(h->d_params))(base + o_params); h->d_stream = (decltype(h->d_stream))(base + o_stream);
        h->d_perm = (decltype(h->d_perm))(base + o_perm); h->d_logZ_tmp = (double*)(base + o_ztmp);
        for (int b = 0; b < 2; ++b) {
            v.x[b] = (double*)(base + o_x[b]); v.C[b] = (uint64_t*)(base + o_C[b]);
            v.segk[b] = (double*)(base + o_k[b]); v.segS[b] = (uint64_t*)(base + o_S[b]);
            v.segS2hi[b] = (uint64_t*)(base + o_hi[b]); v.segS2lo[b] = (uint64_t*)(base + o_lo[b]);
        }
        if (flags & SMC_FLAG_ANCESTORS) v.anc = (decltype(v.anc))(base + o_anc);
        v.logZ = (double*)(base + o_logZ); v.last_logmu = (double*)(base + o_lm); v.last_ess = (double*)(base + o_es);
        v.last_K = (double*)(base + o_K); v.last_D = (uint64_t*)(base + o_D);
        if (gtab) { v.tabD = (uint64_t*)(base + o_tD); v.tabsh = (int*)(base + o_tsh); }
    }
    if (!h->h_pin) {
        TRY(hipHostMalloc((void**)&h->h_pin, 4 * nt * 8, hipHostMallocDefault));   // coherent, device-visible
        h->pin_bytes = 4 * nt * 8;
    }
    memset(h->h_pin, 0, 4 * nt * 8);
    v.host_out = h->h_pin;
    std::vector<uint32_t> st(nt);
    for (size_t m = 0; m < nt; ++m) st[m] = (uint32_t)m;
    TRY(hipMemcpyAsync(h->d_stream, st.data(), nt * 4, hipMemcpyHostToDevice, h->stream));
    TRY(hipStreamSynchronize(h->stream));
#undef TRY
    v.params = h->d_params;
    v.stream = h->d_stream;
    *out = h;
    return SMC_OK;
}

extern "C" int smc_destroy(smc_handle h) {
    if (!h) return SMC_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
#ifdef SMC_ABLATE
    if (h->v.dbg && h->v.nseg == 1) {   // placement and lifetime of the workgroups of the LAST k_resident launch
        const size_t nwg = (size_t)h->v.ntheta;
        std::vector<unsigned long long> st(nwg * 8);
        (void)hipMemcpy(st.data(), h->v.dbg, nwg * 64, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t1 = 0;
        std::map<unsigned long long, std::vector<size_t>> bycu;
        size_t nact = 0;
        for (size_t w = 0; w < nwg; ++w) {
            if (!st[w * 8 + 1]) continue;
            ++nact;
            if (st[w * 8] < t0) t0 = st[w * 8];
            if (st[w * 8 + 1] > t1) t1 = st[w * 8 + 1];
            const unsigned long long hw = st[w * 8 + 2];
            const unsigned long long key = ((hw >> 32) & 0xf) << 16 | ((hw >> 8) & 0xff);   // (xcc, se/sh/cu)
            bycu[key].push_back(w);
        }
        int hist[9] = {0};
        double life[9] = {0};
        for (auto& kv : bycu) {
            const size_t c = kv.second.size() < 8 ? kv.second.size() : 8;
            hist[c]++;
            for (size_t w : kv.second) life[c] += (double)(st[w * 8 + 1] - st[w * 8]) * 0.01;
        }
        fprintf(stderr, "[dbg] k_resident: %zu workgroups ran on %zu CUs, span %.1f us;", nact, bycu.size(), (double)(t1 - t0) * 0.01);
        for (int c = 1; c <= 8; ++c)
            if (hist[c]) fprintf(stderr, "  %d CUs with %d workgroups (mean life %.1f us)", hist[c], c, life[c] / (hist[c] * c));
        fprintf(stderr, "\n");
        {   // start times of the workgroups sharing a CU: simultaneous or one after the other?
            int shown = 0;
            for (auto& kv : bycu) {
                if (shown++ >= 4) break;
                fprintf(stderr, "[dbg]   cu %05llx:", kv.first);
                for (size_t w : kv.second) fprintf(stderr, " wg %zu [%.1f, %.1f]", w, (double)(st[w * 8] - t0) * 0.01, (double)(st[w * 8 + 1] - t0) * 0.01);
                fprintf(stderr, "\n");
            }
        }
        (void)hipFree(h->v.dbg);
        h->v.dbg = nullptr;
    }
    if (h->v.dbg) {   // phase profile of the LAST k_step launch: mean over workgroups, in microseconds
        const size_t nwg = (size_t)h->v.ntheta * h->v.nseg;
        std::vector<unsigned long long> st(nwg * 8);
        (void)hipMemcpy(st.data(), h->v.dbg + nwg * 8, nwg * 64, hipMemcpyDeviceToHost);
        if (st[7] == 0x5045525349535421ull) {   // the persistent step kernel ran: its own accumulators
            double wait = 0, body = 0, pub = 0, mx_wait = 0;
            for (size_t w = 0; w < nwg; ++w) {
                const double steps = (double)st[w * 8];
                wait += st[w * 8 + 1] * 0.01 / steps; body += st[w * 8 + 2] * 0.01 / steps; pub += st[w * 8 + 3] * 0.01 / steps;
                mx_wait = std::max(mx_wait, st[w * 8 + 1] * 0.01 / steps);
            }
            fprintf(stderr, "[dbg] k_persist, mean over %zu workgroups, us per step: wait (poll + acquire + barrier) %.2f (max %.2f), step body %.2f, "
                            "drain + barrier + publish %.2f\n", nwg, wait / nwg, mx_wait, body / nwg, pub / nwg);
        }
    }
    if (h->v.dbg) {
        const size_t nwg = (size_t)h->v.ntheta * h->v.nseg;
        std::vector<unsigned long long> st(nwg * 8);
        (void)hipMemcpy(st.data(), h->v.dbg, nwg * 64, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t7 = 0;
        double ph[8] = {0};
        for (size_t w = 0; w < nwg; ++w) {
            if (st[w * 8] < t0) t0 = st[w * 8];
            if (st[w * 8 + 7] > t7) t7 = st[w * 8 + 7];
            for (int k = 1; k < 8; ++k) ph[k] += (double)(st[w * 8 + k] - st[w * 8 + k - 1]) * 0.01;
        }
        double start_spread = 0, life = 0;
        for (size_t w = 0; w < nwg; ++w) { start_spread += (double)(st[w * 8] - t0) * 0.01; life += (double)(st[w * 8 + 7] - st[w * 8]) * 0.01; }
        fprintf(stderr, "[dbg] k_step span %.2f us; mean start offset %.2f us; mean WG life %.2f us; phases(us):", (double)(t7 - t0) * 0.01,
                start_spread / nwg, life / nwg);
        const char* nm[8] = {"", "table+picks", "targets+range+lookup", "normals", "T2+stage-write+barrier", "search", "gather+model+store", "epilogue"};
        for (int k = 1; k < 8; ++k) fprintf(stderr, " %s=%.2f", nm[k], ph[k] / nwg);
        fprintf(stderr, "\n");
        {   // the slowest workgroups: where do they lose time, and where do they sit (XCD = launch index % 8)
            std::vector<size_t> idx(nwg);
            for (size_t w = 0; w < nwg; ++w) idx[w] = w;
            std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return st[a * 8 + 7] > st[b * 8 + 7]; });
            for (size_t r = 0; r < 6 && r < nwg; ++r) {
                const size_t w = idx[r];
                fprintf(stderr, "[dbg]   late #%zu wg %zu xcd %zu: start %.2f end %.2f phases", r, w, w % 8, (double)(st[w * 8] - t0) * 0.01,
                        (double)(st[w * 8 + 7] - t0) * 0.01);
                for (int k = 1; k < 8; ++k) fprintf(stderr, " %.2f", (double)(st[w * 8 + k] - st[w * 8 + k - 1]) * 0.01);
                fprintf(stderr, "\n");
            }
            double endq[5];
            std::vector<double> ends(nwg);
            for (size_t w = 0; w < nwg; ++w) ends[w] = (double)(st[w * 8 + 7] - t0) * 0.01;
            std::sort(ends.begin(), ends.end());
            const double qs[5] = {0.1, 0.5, 0.9, 0.99, 1.0};
            for (int k = 0; k < 5; ++k) endq[k] = ends[(size_t)((nwg - 1) * qs[k])];
            fprintf(stderr, "[dbg]   end-time quantiles (us) 10%%=%.2f 50%%=%.2f 90%%=%.2f 99%%=%.2f max=%.2f\n", endq[0], endq[1], endq[2], endq[3], endq[4]);
        }
        (void)hipFree(h->v.dbg);
    }
#endif
    Bundle bun;
    bun.device = h->device; bun.slab = h->d_slab; bun.slab_bytes = h->slab_bytes; bun.pin = h->h_pin; bun.pin_bytes = h->pin_bytes;
    bun.stream = h->stream; bun.ev0 = h->ev0; bun.ev1 = h->ev1;
    bun.d_y = h->d_y; bun.ycap = h->ycap; bun.d_recs = h->d_recs; bun.rec_bytes = (size_t)h->reccap * (size_t)h->v.ntheta * sizeof(*h->d_recs);
    bun.h_params = h->h_params; bun.params_bytes = h->h_params ? (size_t)h->v.ntheta * sizeof(Params) : 0;
    const bool kept = h->d_slab && h->h_pin && h->stream && h->ev0 && h->ev1 && bundle_cache().give(bun);   // (the stream is idle: synchronised above)
    if (!kept) {
        (void)hipFree(h->d_slab);   // x, C, the records, the per-filter scalars, the segment table, params / streams / perm
        if (h->h_pin) (void)hipHostFree(h->h_pin);
    }
    if (h->h_pm_out) (void)hipHostFree(h->h_pm_out);
    if (h->h_pm_in) (void)hipHostFree(h->h_pm_in);
    if (h->h_perm) (void)hipHostFree(h->h_perm);
    if (h->h_params && !kept) (void)hipHostFree(h->h_params);
    (void)hipFree(h->d_skip); (void)hipFree(h->d_order);
    if (h->h_win) (void)hipHostFree(h->h_win);
    if (h->h_once) (void)hipHostFree(h->h_once);
    (void)hipFree(h->pm.order);
    (void)hipFree(h->pm_in);   // pm.theta, pm.logZ, pm.chol, pm.nrun, pm.counts, pm.any live in this block
    (void)hipFree(h->pm.prop); (void)hipFree(h->pm.lp); (void)hipFree(h->pm.skip); (void)hipFree(h->pm.mask);
    if (h->d_brk) (void)hipFree(h->d_brk);
    if (h->d_ms) (void)hipFree(h->d_ms);
    if (!kept) (void)hipFree(h->d_y);
    (void)hipFree(h->d_tr_logmu); (void)hipFree(h->d_tr_ess); (void)hipFree(h->d_wdense); if (!kept) (void)hipFree(h->d_recs);
    (void)hipFree(h->d_sum_q); (void)hipFree(h->d_sum_m); (void)hipFree(h->d_pflags);
    if (h->h_perr) (void)hipHostFree(h->h_perr);
    if (!kept) {
        if (h->ev0) (void)hipEventDestroy(h->ev0);
        if (h->ev1) (void)hipEventDestroy(h->ev1);
        if (h->stream) (void)hipStreamDestroy(h->stream);
    }
    delete h;
    return SMC_OK;
}

extern "C" int smc_set_params(smc_handle h, const double* raw) {
    if (!h || !raw) return fail(SMC_EINVAL, "smc_set_params: NULL argument");
    HIPCHK(hipSetDevice(h->device));
    const int nraw = model_nraw_rt(h->model);
    // the rows are derived into a pinned twin and copied from there: the call does not wait for the device (what follows on the
    // handle is ordered behind the copy); a previous copy out of the twin has completed once the stream is idle
    if (!h->h_params) HIPCHK(hipHostMalloc((void**)&h->h_params, (size_t)h->v.ntheta * sizeof(Params), hipHostMallocDefault));
    else HIPCHK(hipStreamSynchronize(h->stream));
    Params* P = h->h_params;
    for (int m = 0; m < h->v.ntheta; ++m) {
        for (int k = 0; k < NPARAM; ++k) P[m].raw[k] = k < nraw ? raw[(size_t)m * nraw + k] : 0.0;
        derive_params(h->model, P[m].raw, P[m].der);
    }
    HIPCHK(hipMemcpyAsync(h->d_params, P, (size_t)h->v.ntheta * sizeof(Params), hipMemcpyHostToDevice, h->stream));
    h->have_params = true;
    return SMC_OK;
}

extern "C" int smc_set_streams(smc_handle h, const uint32_t* s) {
    if (!h || !s) return fail(SMC_EINVAL, "smc_set_streams: NULL argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(h->d_stream, s, (size_t)h->v.ntheta * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->brk_count = 0;    // cached break points belong to the old stream ids
    return SMC_OK;
}

extern "C" int smc_reseed(smc_handle h, uint64_t seed) {
    if (!h) return fail(SMC_EINVAL, "smc_reseed: NULL handle");
    h->v.seed = seed;
    h->brk_count = 0;    // cached break points belong to the old seed
    return SMC_OK;
}

static int ensure_y(smc_handle h, int64_t T) {
    if (T > h->ycap) {
        HIPCHK(hipStreamSynchronize(h->stream));
        (void)hipFree(h->d_y);
        h->d_y = nullptr;
        HIPCHK(dalloc(&h->d_y, (size_t)T));
        h->ycap = T;
    }
    return SMC_OK;
}
static int ensure_trace(smc_handle h, int64_t T) {
    if (T > h->trcap) {
        HIPCHK(hipStreamSynchronize(h->stream));
        (void)hipFree(h->d_tr_logmu); (void)hipFree(h->d_tr_ess);
        h->d_tr_logmu = h->d_tr_ess = nullptr;
        HIPCHK(dalloc(&h->d_tr_logmu, (size_t)T * h->v.ntheta));
        HIPCHK(dalloc(&h->d_tr_ess, (size_t)T * h->v.ntheta));
        h->trcap = T;
    }
    return SMC_OK;
}

static int ensure_recs(smc_handle h, int64_t T) {
    if (T > h->reccap) {
        HIPCHK(hipStreamSynchronize(h->stream));
        (void)hipFree(h->d_recs);
        h->d_recs = nullptr;
        HIPCHK(dalloc(&h->d_recs, (size_t)T * h->v.ntheta));
        h->reccap = T;
    }
    return SMC_OK;
}

// Close the timed region (ev1) and hand the requested per-filter result vectors to the caller from the
// pinned mirror (FilterView::host_out): ONE stream synchronisation, no copy commands.
static int finish_timing(smc_handle h, double* logZ = nullptr, double* logmu = nullptr, double* ess = nullptr) {
    const size_t nt = (size_t)h->v.ntheta;
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));   // the emitting kernel stored the results in the pinned mirror itself
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->last_ms = ms;
    if (logZ) memcpy(logZ, h->h_pin, nt * 8);
    if (logmu) memcpy(logmu, h->h_pin + nt, nt * 8);
    if (ess) memcpy(ess, h->h_pin + 2 * nt, nt * 8);
    return SMC_OK;
}

// Step API (smc_init / smc_step): the emitting kernel stores a ticket behind its three values (host_emit); the host spins on the
// pinned words - no event records, no stream synchronisation (32 -> 12 us per call for a filter of 1024 particles).  A ticket
// that does not show up within 50 ms is looked for once more after a real synchronisation (a faulted launch reports there).
static int wait_ticket(smc_handle h, uint32_t seq, double* logmu, double* ess) {
    const size_t nt = (size_t)h->v.ntheta;
    const volatile double* tk = h->h_pin + 3 * nt;
    const double want = (double)seq;
    const auto t0 = std::chrono::steady_clock::now();
    bool synced = false;
    unsigned spins = 0;
    for (size_t th = 0; th < nt;) {
        if (tk[th] == want) { ++th; continue; }
        if ((++spins & 0x3ffu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) {
            if (synced) return fail(SMC_EHIP, "step API: the launch completed without its result ticket");
            HIPCHK(hipStreamSynchronize(h->stream));
            synced = true;
        }
        __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (logmu) memcpy(logmu, h->h_pin + nt, nt * 8);
    if (ess) memcpy(ess, h->h_pin + 2 * nt, nt * 8);
    h->last_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();   // host clock of the wait
    return SMC_OK;
}

// the next ticket of the step API - or 0: batches of more than 64 filters synchronise the stream instead (every filter's emitting
// thread releases its values to the host by itself: 4096 of them take 360 us per step against 74 for events + synchronisation;
// measured equal at 512 filters, scripts/dbg/step_latency_batch.py)
static uint32_t step_ticket(smc_handle h) {
    if (h->v.ntheta > 64) return 0;
    return ++h->seq ? h->seq : ++h->seq;
}
static int emit_if_needed(smc_handle h) {
    if (!h->emitted) {
        HIPCHK(do_finalize(h, h->t == 1 ? 1 : 0, h->t - 1));
        h->emitted = true;
    }
    return SMC_OK;
}

// bootstrap_filter(N, y, model)   particles.jl:87-105
extern "C" int smc_init(smc_handle h, double y1, double* logmu) {
    if (!h) return fail(SMC_EINVAL, "smc_init: NULL handle");
    if (!h->have_params) return fail(SMC_ESTATE, "smc_init: smc_set_params has not been called");
    h->win_k = 0;   // an uncommitted window is dropped
    HIPCHK(hipSetDevice(h->device));
    h->v.y = nullptr; h->v.trace_logmu = nullptr; h->v.trace_ess = nullptr;
    h->cur = 0;
    const bool own = h->v.nseg == 1;   // one workgroup owns the filter: it emits (logmu, ess) itself
    const uint32_t seq = step_ticket(h);
    if (!seq) HIPCHK(hipEventRecord(h->ev0, h->stream));
    h->v.emit_now = own ? 1 : 0;
    h->v.host_seq = seq;
    hipError_t le = do_init(h, y1);
    h->v.emit_now = 0;
    if (le == hipSuccess) { h->t = 1; h->inited = true; h->emitted = own; }
    const int rc = le == hipSuccess ? emit_if_needed(h) : SMC_OK;
    h->v.host_seq = 0;
    HIPCHK(le);
    if (rc) return rc;
    return seq ? wait_ticket(h, seq, logmu, nullptr) : finish_timing(h, nullptr, logmu, nullptr);
}

// bootstrap_filter!(x, w, y, model)   particles.jl:107-129
extern "C" int smc_step(smc_handle h, double y_t, double* logmu, double* ess) {
    if (!h) return fail(SMC_EINVAL, "smc_step: NULL handle");
    if (!h->inited) return fail(SMC_ESTATE, "smc_step: call smc_init (bootstrap_filter) first");
    h->win_k = 0;   // an uncommitted window is dropped
    HIPCHK(hipSetDevice(h->device));
    h->v.y = nullptr; h->v.trace_logmu = nullptr; h->v.trace_ess = nullptr;
    int rc = emit_if_needed(h);   // (a pending emission of the previous step goes first, without a ticket)
    if (rc) return rc;
    HIPCHK(ensure_breaks(h, h->t, h->t + 64));   // step API: 64 steps of break points at a time
    const bool own = h->v.nseg == 1;
    const uint32_t seq = step_ticket(h);
    if (!seq) HIPCHK(hipEventRecord(h->ev0, h->stream));
    h->v.emit_now = own ? 1 : 0;
    h->v.host_seq = seq;
    hipError_t le = do_step(h, h->t, 0, y_t);
    h->v.emit_now = 0;
    if (le == hipSuccess) { h->cur ^= 1; h->t += 1; h->emitted = own; rc = emit_if_needed(h); }
    h->v.host_seq = 0;
    HIPCHK(le);
    if (rc) return rc;
    return seq ? wait_ticket(h, seq, logmu, ess) : finish_timing(h, nullptr, logmu, ess);
}

// Restores the fields of the view that a whole-series call sets for its launches, on every exit path (an early HIPCHK return
// must not leave the handle accumulating no sum of squares, or pointing at the series)
struct SeriesScope {
    FilterView& v;
    explicit SeriesScope(FilterView& view) : v(view) {}
    ~SeriesScope() { v.want_s2 = 1; v.y = nullptr; v.trace_logmu = nullptr; v.trace_ess = nullptr; v.sum_np = v.sum_mom = 0; v.sum_q = v.sum_m = nullptr; }
};

// ---- per-step summaries inside the multi-step calls ------------------------------------------------------------------
static bool summaries_on(const smc_filter_s* h) { return h->sum_np > 0 || h->sum_mom != 0; }
// whether the LDS-resident kernels have room for the summaries' histograms next to the filter's state (160 KiB per workgroup)
static bool summaries_fit_lds(const smc_filter_s* h) {
    return (size_t)lds_padded_len(h->v.seg) * 8 * (size_t)(1 + h->d) + scr_words(1024, 4) * 8 + summary_lds_words(h->sum_np) * 8 <= (size_t)160 * 1024;
}
static int ensure_summaries(smc_handle h, int64_t T) {
    if (T > h->sum_cap) {
        HIPCHK(hipStreamSynchronize(h->stream));
        (void)hipFree(h->d_sum_q); (void)hipFree(h->d_sum_m);
        h->d_sum_q = h->d_sum_m = nullptr;
        h->sum_cap = 0;
        HIPCHK(dalloc(&h->d_sum_q, (size_t)T * h->v.ntheta * QMAX));
        HIPCHK(dalloc(&h->d_sum_m, (size_t)T * 2 * h->d * h->v.ntheta));
        h->sum_cap = T;
    }
    return SMC_OK;
}
// what the launches of a multi-step call need to know about the summaries (the scope of the call resets it)
static void view_summaries(smc_handle h) {
    FilterView& v = h->v;
    v.sum_np = h->sum_np; v.sum_comp = h->sum_comp; v.sum_mom = h->sum_mom;
    for (int j = 0; j < QMAX; ++j) v.sum_p64[j] = h->sum_p64[j];
    v.sum_q = h->d_sum_q; v.sum_m = h->d_sum_m;
}
// The summaries of the CURRENT weights of filters of any size, enqueued on the handle's stream behind the launch that produced
// them (no host synchronisation): smc_summ_kernels.h.  q_out [ntheta][np], mean / var [d][ntheta] are device pointers.  The
// weights must have been emitted (last_K, last_D describe them).
static int ensure_ms(smc_handle h) {
    const size_t nth = (size_t)h->v.ntheta, words = ms_words(nth, (size_t)h->v.nseg, (size_t)h->d);
    if (!h->d_ms) {
        HIPCHK(hipMalloc((void**)&h->d_ms, (words + nth * QMAX) * 8));
        HIPCHK(hipMemsetAsync(h->d_ms, 0, (words + nth * QMAX) * 8, h->stream));
    }
    return SMC_OK;
}
static int enqueue_ms(smc_handle h, int component, int np, const uint64_t* p64, bool mom, double* q_out, double* mean, double* var) {
    const size_t nth = (size_t)h->v.ntheta;
    int rc = ensure_ms(h);
    if (rc) return rc;
    const MsScratch ms = ms_carve(h->d_ms, nth, (size_t)h->v.nseg, (size_t)h->d);
    FilterView v = h->v;
    v.sum_np = np; v.sum_comp = np > 0 ? component : 0; v.sum_mom = mom ? 1 : 0;
    for (int j = 0; j < QMAX; ++j) v.sum_p64[j] = j < np ? p64[j] : 0;
    // streaming kernels: workgroups of 1024 threads over consecutive segments - about 256 workgroups in all for the passes that
    // only read, about 64 for the histogram (every workgroup flushes its occupied bins with device-scope atomics)
    auto groups = [&](int want) {
        int g = want / h->v.ntheta;
        g = g < 1 ? 1 : g;
        return g > h->v.nseg ? h->v.nseg : g;
    };
    const int g_read = groups(256), g_hist = groups(64);
    hipLaunchKernelGGL(k_ms_range, dim3(g_read, h->v.ntheta), dim3(MS_STREAM), 0, h->stream, v, h->cur, h->d, ms);
    if (np > 0) hipLaunchKernelGGL(k_ms_hist, dim3(g_hist, h->v.ntheta), dim3(MS_STREAM), 0, h->stream, v, h->cur, g_read, ms);
    hipLaunchKernelGGL(k_ms_pick, dim3(h->v.ntheta), dim3(MS_THREADS), 0, h->stream, v, h->d, g_read, ms, q_out, mean, var);
    int64_t two_level = MS_TWO_LEVEL;
    if (const char* e = getenv("SMC_MS_TWO_LEVEL")) two_level = atoll(e);   // tuning / test knob: results do not depend on it
    if (np > 0 && h->v.n > two_level) {   // big filters: the chosen bins cut a second time before the candidates are collected
        hipLaunchKernelGGL(k_ms_hist2, dim3(g_read, h->v.ntheta), dim3(MS_STREAM), 0, h->stream, v, h->cur, ms);
        hipLaunchKernelGGL(k_ms_pick2, dim3(np, h->v.ntheta), dim3(MS_THREADS), 0, h->stream, v, ms);
        hipLaunchKernelGGL(k_ms_collect<true>, dim3(g_read, h->v.ntheta), dim3(MS_STREAM), 0, h->stream, v, h->cur, ms);
    } else if (np > 0) {
        hipLaunchKernelGGL(k_ms_collect<false>, dim3(g_read, h->v.ntheta), dim3(MS_STREAM), 0, h->stream, v, h->cur, ms);
    }
    if (np > 0) {
        hipLaunchKernelGGL(k_ms_select, dim3(np, h->v.ntheta), dim3(MS_SEL_THREADS), 0, h->stream, v, h->cur, ms, q_out);
    }
    HIPCHK(hipGetLastError());
    return SMC_OK;
}
// ... into row `row` of the traces of a multi-step call
static int enqueue_step_summaries(smc_handle h, int64_t row) {
    const size_t nth = (size_t)h->v.ntheta, nout = (size_t)h->d * nth;
    double* mbase = h->d_sum_m + (size_t)row * 2 * nout;
    return enqueue_ms(h, h->sum_comp, h->sum_np, h->sum_p64, h->sum_mom != 0, h->d_sum_q + (size_t)row * nth * h->sum_np, mbase, mbase + nout);
}

extern "C" int smc_set_summaries(smc_handle h, int component, const double* p, int np, int moments) {
    if (!h) return fail(SMC_EINVAL, "smc_set_summaries: NULL handle");
    if (np < 0 || np > QMAX || (np > 0 && !p)) return fail(SMC_EINVAL, "smc_set_summaries: 0 <= np <= 8");
    if (np > 0 && (component < 0 || component >= h->d)) return fail(SMC_EINVAL, "smc_set_summaries: component out of range");
    h->sum_np = np; h->sum_comp = np > 0 ? component : 0; h->sum_mom = moments ? 1 : 0;
    for (int j = 0; j < QMAX; ++j) h->sum_p64[j] = j < np ? prob_to_u64(p[j]) : 0;
    h->sum_T = 0;
    return SMC_OK;
}

extern "C" int smc_get_summaries(smc_handle h, int64_t T, double* q, double* mean, double* var) {
    if (!h) return fail(SMC_EINVAL, "smc_get_summaries: NULL handle");
    if (T < 1 || T > h->sum_T) return fail(SMC_ESTATE, "smc_get_summaries: more steps than the last multi-step call recorded");
    if ((q && h->sum_np == 0) || ((mean || var) && !h->sum_mom)) return fail(SMC_ESTATE, "smc_get_summaries: not recorded (smc_set_summaries)");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t nth = (size_t)h->v.ntheta, nout = (size_t)h->d * nth;
    if (q) HIPCHK(hipMemcpy(q, h->d_sum_q, (size_t)T * nth * h->sum_np * 8, hipMemcpyDeviceToHost));
    if (mean) HIPCHK(hipMemcpy2D(mean, nout * 8, h->d_sum_m, 2 * nout * 8, nout * 8, (size_t)T, hipMemcpyDeviceToHost));
    if (var) HIPCHK(hipMemcpy2D(var, nout * 8, h->d_sum_m + nout, 2 * nout * 8, nout * 8, (size_t)T, hipMemcpyDeviceToHost));
    return SMC_OK;
}

// The launches of log_likelihood(N, y, model) (particles.jl:132-147) for every filter of the handle, enqueued on
// its stream: nothing here waits for the device.  y must already be in h->d_y (ensure_y + copy by the caller).
static int enqueue_log_likelihood(smc_handle h, double y0, int64_t T, bool want_trace, bool summ = false) {
    // (systematic resampling with per-step summaries: the LDS-resident summary kernels exist for the default law only)
    const bool resident = h->resident_ok && resident_supported(h->model, h->v.seg) && !(summ && (h->v.systematic || !summaries_fit_lds(h)));
    SeriesScope scope(h->v);
    h->v.y = h->d_y;
    h->v.trace_logmu = want_trace ? h->d_tr_logmu : nullptr;
    h->v.trace_ess = want_trace ? h->d_tr_ess : nullptr;
    h->cur = 0;
    h->v.want_s2 = want_trace ? 1 : 0;   // ess_t is read only through the traces; the last step always has it
    int rc = SMC_OK;
    if (summ && resident) view_summaries(h);
    if (summ && !resident) {
        // one launch per step, every step followed by the emission of its (logmu, ess) and by the summary kernels - all on the
        // handle's stream, nothing waits for the device
        h->v.want_s2 = 1;
        HIPCHK(do_init(h, y0));
        h->t = 1; h->inited = true; h->emitted = false;
        if ((rc = emit_if_needed(h))) return rc;
        if ((rc = enqueue_step_summaries(h, 0))) return rc;
        for (int64_t t = 1; t < T; ++t) {
            HIPCHK(ensure_breaks(h, (uint32_t)t, (uint32_t)T));
            HIPCHK(do_step(h, (uint32_t)t, 0, 0.0));
            h->cur ^= 1; h->t += 1; h->emitted = false;
            if ((rc = emit_if_needed(h))) return rc;
            if ((rc = enqueue_step_summaries(h, t))) return rc;
        }
        return SMC_OK;
    }
    if (resident) {
        HIPCHK(do_resident(h, (int)T));
        h->cur = 0; h->t = (uint32_t)T; h->inited = true; h->emitted = true;
    } else {
        if (T == 1) h->v.want_s2 = 1;
        HIPCHK(do_init(h, y0));
        h->t = 1; h->inited = true; h->emitted = false;
        int64_t t_first = 1;
        // OPT-IN (SMC_PERSIST=1; measured slower than one launch per step): the steps 1 .. T-2 in persistent
        // launches, one per window of prepared break points; the last step (which carries the sum of squares) by its own launch
        if (h->persist < 0) { const char* e = getenv("SMC_PERSIST"); h->persist = (e && atoi(e) == 1) ? 1 : 0; }
        if (h->persist == 1 && !want_trace && h->v.nseg > 1 && !h->v.skip && !h->v.anc && T > 3) {
            const size_t nfl = (size_t)h->v.ntheta * h->v.nseg;
            if (!h->d_pflags) {
                HIPCHK(dalloc(&h->d_pflags, 2 * nfl));
                HIPCHK(hipHostMalloc((void**)&h->h_perr, 16, hipHostMallocDefault));
            }
            *h->h_perr = 0;
            while (t_first < T - 1) {
                HIPCHK(ensure_breaks(h, (uint32_t)t_first, (uint32_t)T));
                int64_t t_end = (int64_t)h->v.brk_t0 + h->brk_count;
                t_end = t_end > T - 1 ? T - 1 : t_end;
                HIPCHK(hipMemsetAsync(h->d_pflags, 0, 2 * nfl * 4, h->stream));
                PersistCtl pc{{h->d_pflags, h->d_pflags + nfl}, h->h_perr};
                const hipError_t pe = do_persist(h, (uint32_t)t_first, (uint32_t)t_end, pc);
                if (pe == hipErrorCooperativeLaunchTooLarge) { h->persist = 0; break; }   // not available for this filter: step by step
                HIPCHK(pe);
                if ((t_end - t_first) & 1) h->cur ^= 1;
                h->t += (uint32_t)(t_end - t_first);
                t_first = t_end;
            }
        }
        for (int64_t t = t_first; t < T; ++t) {
            const int emit = h->v.want_s2 ? 1 : 2;   // 2: the records of step t-1 carry no sum of squares - (logmu, 0) from the totals alone
            if (t == T - 1) h->v.want_s2 = 1;
            HIPCHK(ensure_breaks(h, (uint32_t)t, (uint32_t)T));
            HIPCHK(do_step(h, (uint32_t)t, emit, 0.0));
            h->cur ^= 1; h->t += 1;
        }
        h->v.want_s2 = 1;
        rc = emit_if_needed(h);
    }
    return rc;
}

// log_likelihood(N, y, model)   particles.jl:132-147
extern "C" int smc_log_likelihood(smc_handle h, const double* y, int64_t T, double* logZ, double* logmu_trace,
                                  double* ess_trace) {
    if (!h || !y) return fail(SMC_EINVAL, "smc_log_likelihood: NULL argument");
    if (T <= 0) return fail(SMC_EINVAL, "smc_log_likelihood: T must be positive");
    if (!h->have_params) return fail(SMC_ESTATE, "smc_log_likelihood: smc_set_params has not been called");
    h->win_k = 0;   // an uncommitted window is dropped
    HIPCHK(hipSetDevice(h->device));
    int rc = ensure_y(h, T);
    if (rc) return rc;
    const bool want_trace = logmu_trace || ess_trace;
    if (want_trace && (rc = ensure_trace(h, T))) return rc;
    const bool resident = h->resident_ok && resident_supported(h->model, h->v.seg);
    if (resident && (rc = ensure_recs(h, T))) return rc;
    const bool summ = summaries_on(h);
    if (summ && (rc = ensure_summaries(h, T))) return rc;
    h->sum_T = 0;
    HIPCHK(hipMemcpyAsync(h->d_y, y, (size_t)T * 8, hipMemcpyHostToDevice, h->stream));
    if (h->skip_on) { h->v.skip = h->d_skip; h->v.order = h->d_order; h->v.n_active = h->d_order + h->v.ntheta; }
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    rc = enqueue_log_likelihood(h, y[0], T, want_trace, summ);
    h->v.skip = nullptr; h->v.order = nullptr; h->v.n_active = nullptr;
    if (rc) return rc;
    if (summ) h->sum_T = T;
    rc = finish_timing(h, logZ);
    if (rc) return rc;
    if (h->h_perr && *h->h_perr) {
        *h->h_perr = 0;
        h->persist = 0;
        h->inited = false;
        return fail(SMC_EHIP, "smc_log_likelihood: the persistent step kernel gave up (a workgroup waited 100 ms for the previous step: "
                              "not every workgroup resident?); the handle falls back to one launch per step - call again");
    }
    if (logmu_trace) HIPCHK(hipMemcpy(logmu_trace, h->d_tr_logmu, (size_t)T * h->v.ntheta * 8, hipMemcpyDeviceToHost));
    if (ess_trace) HIPCHK(hipMemcpy(ess_trace, h->d_tr_ess, (size_t)T * h->v.ntheta * 8, hipMemcpyDeviceToHost));
    return SMC_OK;
}

// ---- k steps in one launch (the online sampler's window) ---------------------------------------------------------
constexpr int WIN_MAX = 64;
static int launch_window_steps(smc_handle h, int k) {
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    HIPCHK(do_window(h, k, h->cur ^ 1));
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    return SMC_OK;
}
extern "C" int smc_step_window(smc_handle h, const double* y, int k, double* logmu, double* ess) {
    if (!h || !y) return fail(SMC_EINVAL, "smc_step_window: NULL argument");
    if (!h->inited) return fail(SMC_ESTATE, "smc_step_window: call smc_init (bootstrap_filter) first");
    if (k < 1 || k > WIN_MAX) return fail(SMC_EINVAL, "smc_step_window: 1 <= k <= 64");
    if (h->v.nseg != 1 || !resident_supported(h->model, h->v.seg))
        return fail(SMC_EINVAL, "smc_step_window: needs filters that fit the LDS-resident kernel (one segment); use smc_step");
    HIPCHK(hipSetDevice(h->device));
    int rc = emit_if_needed(h);
    if (rc) return rc;
    if ((rc = ensure_y(h, WIN_MAX))) return rc;
    if ((rc = ensure_recs(h, WIN_MAX))) return rc;
    const size_t nt = (size_t)h->v.ntheta;
    if (!h->h_win) HIPCHK(hipHostMalloc((void**)&h->h_win, 2 * (size_t)WIN_MAX * nt * 8, hipHostMallocDefault));
    const bool summ = summaries_on(h);
    if (summ && h->v.systematic) return fail(SMC_EINVAL, "smc_step_window: per-step summaries need the default (multinomial) resampler");
    if (summ && !summaries_fit_lds(h)) return fail(SMC_EINVAL, "smc_step_window: no LDS left for the summaries of filters this long; fewer levels, or smc_step");
    if (summ && (rc = ensure_summaries(h, WIN_MAX))) return rc;
    h->sum_T = 0;
    HIPCHK(hipMemcpyAsync(h->d_y, y, (size_t)k * 8, hipMemcpyHostToDevice, h->stream));
    h->v.y = h->d_y;
    if (summ) view_summaries(h);
    rc = launch_window_steps(h, k);
    h->v.y = nullptr;
    h->v.sum_np = h->v.sum_mom = 0; h->v.sum_q = h->v.sum_m = nullptr;
    if (rc) return rc;
    if (summ) h->sum_T = k;
    HIPCHK(hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->last_ms = ms;
    if (logmu) memcpy(logmu, h->h_win, (size_t)k * nt * 8);
    if (ess) memcpy(ess, h->h_win + (size_t)k * nt, (size_t)k * nt * 8);
    h->win_k = k;
    return SMC_OK;
}
extern "C" int smc_step_commit(smc_handle h, int j) {
    if (!h) return fail(SMC_EINVAL, "smc_step_commit: NULL handle");
    if (h->win_k == 0) return fail(SMC_ESTATE, "smc_step_commit: no window pending (smc_step_window)");
    if (j < 0 || j > h->win_k) return fail(SMC_EINVAL, "smc_step_commit: 0 <= j <= steps of the window");
    HIPCHK(hipSetDevice(h->device));
    const int k = h->win_k;
    h->win_k = 0;
    if (j == 0) return SMC_OK;            // nothing kept: the filters stand where they stood before the window
    if (j < k) {                          // keep a prefix: the same j steps again (counter-based random numbers: the same bits)
        h->v.y = h->d_y;
        int rc = launch_window_steps(h, j);
        h->v.y = nullptr;
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_commit, dim3((unsigned)((h->v.ntheta + 127) / 128)), dim3(128), 0, h->stream, h->v, j, h->d_recs);
    HIPCHK(hipGetLastError());
    h->cur ^= 1; h->t += (uint32_t)j; h->emitted = true;
    return SMC_OK;      // no wait: whatever the caller does next with this handle is ordered behind on its stream
}

// Filters smc_log_likelihood leaves out (logZ = -inf): the proposals outside the prior's support, for which the
// reference never calls log_likelihood (smc_samplers.jl:116).  NULL: run every filter again.
extern "C" int smc_set_skip(smc_handle h, const uint8_t* skip) {
    if (!h) return fail(SMC_EINVAL, "smc_set_skip: NULL handle");
    HIPCHK(hipSetDevice(h->device));
    if (!skip) { h->skip_on = false; return SMC_OK; }
    const int nt = h->v.ntheta;
    if (!h->d_skip) HIPCHK(dalloc(&h->d_skip, (size_t)nt));
    if (!h->d_order) HIPCHK(dalloc(&h->d_order, (size_t)nt + 1));
    std::vector<int32_t> ord((size_t)nt + 1);
    int na = 0, ns = 0;
    for (int m = 0; m < nt; ++m) {
        if (skip[m]) ord[(size_t)nt - 1 - ns++] = m;
        else ord[(size_t)na++] = m;
    }
    ord[(size_t)nt] = na;
    HIPCHK(hipMemcpyAsync(h->d_skip, skip, (size_t)nt, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_order, ord.data(), ord.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->skip_on = true;
    return SMC_OK;
}

// ---- PMMH rejuvenation on the device (rejuvenate!, smc_samplers.jl:103-146) --------------------------------
extern "C" int smc_pmmh_configure(smc_handle h, int d_theta, const int32_t* prior_family, const double* prior_par,
                                  const int32_t* raw_from, const double* raw_const) {
    if (!h || !prior_family || !prior_par || !raw_from || !raw_const) return fail(SMC_EINVAL, "smc_pmmh_configure: NULL argument");
    if (d_theta < 1 || d_theta > MAX_DTHETA) return fail(SMC_EINVAL, "smc_pmmh_configure: 1 <= d_theta <= 8");
    PmmhSpec sp{};
    sp.d = d_theta;
    for (int i = 0; i < d_theta; ++i) {
        if (prior_family[i] < PRIOR_UNIFORM || prior_family[i] > PRIOR_LOGNORMAL)
            return fail(SMC_EINVAL, "smc_pmmh_configure: unknown prior family " + std::to_string(prior_family[i]));
        sp.family[i] = prior_family[i];
        for (int k = 0; k < PRIOR_NPAR; ++k) sp.par[i][k] = prior_par[(size_t)i * PRIOR_NPAR + k];
    }
    sp.nraw = model_nraw_rt(h->model);
    for (int k = 0; k < sp.nraw; ++k) {
        if (raw_from[k] >= d_theta) return fail(SMC_EINVAL, "smc_pmmh_configure: raw_from index out of range");
        sp.raw_from[k] = raw_from[k];
        sp.raw_const[k] = raw_const[k];
    }
    HIPCHK(hipSetDevice(h->device));
    const size_t nt = (size_t)h->v.ntheta;
    if (!h->pm_in) {
        // what a rejuvenation call uploads or clears sits in ONE block (8-byte words): theta | logZ | chol | nrun | counts | any
        const size_t w_theta = nt * MAX_DTHETA, w_chol = (size_t)MAX_DTHETA * MAX_DTHETA, w_any = (nt + 7) / 8;
        h->pm_in_words = w_theta + nt + w_chol + 2 + w_any;
        HIPCHK(dalloc(&h->pm_in, h->pm_in_words));
        HIPCHK(hipHostMalloc((void**)&h->h_pm_in, h->pm_in_words * 8, hipHostMallocDefault));
        h->pm.theta = h->pm_in;
        h->pm.logZ = h->pm.theta + w_theta;
        h->pm.chol = h->pm.logZ + nt;
        h->pm.nrun = (unsigned long long*)(h->pm.chol + w_chol);
        h->pm.counts = (int32_t*)(h->pm.nrun + 1);
        h->pm.any = (unsigned char*)(h->pm.nrun + 2);
        HIPCHK(dalloc(&h->pm.prop, nt * MAX_DTHETA));
        HIPCHK(dalloc(&h->pm.lp, nt * 2));
        HIPCHK(dalloc(&h->pm.skip, nt));
        HIPCHK(dalloc(&h->pm.mask, nt));
        HIPCHK(dalloc(&h->pm.order, nt));
        HIPCHK(hipHostMalloc((void**)&h->h_pm_out, (nt * (MAX_DTHETA + 2) + 1) * 8, hipHostMallocDefault));
    }
    h->pm_spec = sp;
    h->pm_cfg = true;
    return SMC_OK;
}

__global__ void k_pmmh_export(int ntheta, int d, PmmhDev p, double* out) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= ntheta) return;
    for (int i = 0; i < d; ++i) out[(size_t)m * d + i] = p.theta[(size_t)m * MAX_DTHETA + i];
    out[(size_t)ntheta * d + m] = p.logZ[m];
    out[(size_t)ntheta * (d + 1) + m] = p.any[m] ? 1.0 : 0.0;
    if (m == 0) out[(size_t)ntheta * (d + 2)] = (double)*p.nrun;
}

extern "C" int smc_pmmh_rejuvenate(smc_handle h, smc_handle main, const double* y, int64_t T, double xi, const double* chol,
                                   const double* scales, int chain, const uint64_t* filter_seeds, uint64_t move_seed,
                                   double* theta, double* logZ, uint8_t* accepted, int64_t* filters_run) {
    if (!h || !y || !chol || !scales || !filter_seeds || !theta || !logZ) return fail(SMC_EINVAL, "smc_pmmh_rejuvenate: NULL argument");
    if (!h->pm_cfg) return fail(SMC_ESTATE, "smc_pmmh_rejuvenate: smc_pmmh_configure has not been called");
    if (T <= 0 || chain < 0) return fail(SMC_EINVAL, "smc_pmmh_rejuvenate: bad T or chain");
    if (main) {
        if (main == h) return fail(SMC_EINVAL, "smc_pmmh_rejuvenate: main and proposal handles are the same");
        const FilterView &a = main->v, &b = h->v;
        if (main->model != h->model || a.n != b.n || a.seg != b.seg || a.ntheta != b.ntheta || main->device != h->device)
            return fail(SMC_EINVAL, "smc_pmmh_rejuvenate: handles differ in model, geometry or device");
        if (!main->inited) return fail(SMC_ESTATE, "smc_pmmh_rejuvenate: main filters not initialised");
        main->win_k = 0;
    }
    h->win_k = 0;
    HIPCHK(hipSetDevice(h->device));
    const PmmhSpec& sp = h->pm_spec;
    const int nt = h->v.ntheta, d = sp.d;
    int rc = ensure_y(h, T);
    if (rc) return rc;
    const bool resident = h->resident_ok && resident_supported(h->model, h->v.seg);
    if (resident && (rc = ensure_recs(h, T))) return rc;
    if (main) {   // its pending emission, then an idle stream: the accept copies below run on the proposal handle's stream
        if ((rc = emit_if_needed(main))) return rc;
        HIPCHK(hipStreamSynchronize(main->stream));
    }
    {   // theta (rows padded to MAX_DTHETA), logZ, the Cholesky factor and the zeros of nrun / counts / any: one pinned block, one copy
        // (the previous call's copy has completed: every call ends with a stream synchronisation)
        double* in = h->h_pm_in;
        memset(in, 0, h->pm_in_words * 8);
        for (int m = 0; m < nt; ++m)
            for (int i = 0; i < d; ++i) in[(size_t)m * MAX_DTHETA + i] = theta[(size_t)m * d + i];
        double* in_logZ = in + (size_t)nt * MAX_DTHETA;
        memcpy(in_logZ, logZ, (size_t)nt * 8);
        double* in_chol = in_logZ + nt;
        for (int i = 0; i < d * d; ++i) in_chol[i] = chol[i];
    }
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_y, y, (size_t)T * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->pm_in, h->h_pm_in, h->pm_in_words * 8, hipMemcpyHostToDevice, h->stream));
    const dim3 grid((unsigned)((nt + 127) / 128)), block(128);
    for (int c = 0; c < chain; ++c) {
        hipLaunchKernelGGL(k_pmmh_propose, grid, block, 0, h->stream, h->v, sp, h->pm, h->model, move_seed, (uint32_t)c,
                           sqrt(scales[c]), h->d_params);
        HIPCHK(hipGetLastError());
        h->have_params = true;
        h->v.seed = filter_seeds[c];
        h->brk_count = 0;                      // cached break points belong to the previous seed
        h->v.skip = h->pm.skip; h->v.order = h->pm.order; h->v.n_active = h->pm.counts;
        rc = enqueue_log_likelihood(h, y[0], T, false);
        h->v.skip = nullptr; h->v.order = nullptr; h->v.n_active = nullptr;
        if (rc) return rc;
        hipLaunchKernelGGL(k_pmmh_accept, grid, block, 0, h->stream, h->v, h->pm, d, move_seed, (uint32_t)c, xi);
        HIPCHK(hipGetLastError());
        if (main) {   // smc.x[m], smc.w[m] <- x_prop, w_prop of the accepted particles (smc_samplers.jl:132-133)
            const FilterView& a = main->v;
            hipLaunchKernelGGL(k_copy_slots, dim3((unsigned)((a.npad + 255) / 256), a.ntheta), dim3(256), 0, h->stream, a, main->cur,
                               h->v, h->cur, main->d, h->pm.mask);
            HIPCHK(hipGetLastError());
        }
    }
    hipLaunchKernelGGL(k_pmmh_export, grid, block, 0, h->stream, nt, d, h->pm, h->h_pm_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->last_ms = ms;
    if (main && chain > 0) main->t = h->t;
    memcpy(theta, h->h_pm_out, (size_t)nt * d * 8);
    memcpy(logZ, h->h_pm_out + (size_t)nt * d, (size_t)nt * 8);
    if (accepted)
        for (int m = 0; m < nt; ++m) accepted[m] = h->h_pm_out[(size_t)nt * (d + 1) + m] != 0.0 ? 1 : 0;
    if (filters_run) *filters_run = (int64_t)h->h_pm_out[(size_t)nt * (d + 2)];
    return SMC_OK;
}

extern "C" int smc_time_step_kernel(smc_handle h, const double* y, int64_t T, int nsample, double* avg_ms,
                                    double* min_ms) {
    if (!h || !y || T < 2 || nsample < 1) return fail(SMC_EINVAL, "smc_time_step_kernel: bad argument");
    if (!h->have_params) return fail(SMC_ESTATE, "smc_time_step_kernel: smc_set_params has not been called");
    HIPCHK(hipSetDevice(h->device));
    int rc = ensure_y(h, T);
    if (rc) return rc;
    // a bracket spans G consecutive k_step launches (a step IS one launch): the ~5 us an event pair costs on
    // this stack is amortised over the G launches instead of being charged to one
    const int G = T - 1 >= 512 ? 32 : (T - 1 >= 64 ? 8 : 1);
    if ((int64_t)nsample * G > T - 1) nsample = (int)((T - 1) / G);
    std::vector<hipEvent_t> e0((size_t)nsample), e1((size_t)nsample);
    for (int i = 0; i < nsample; ++i) { HIPCHK(hipEventCreate(&e0[i])); HIPCHK(hipEventCreate(&e1[i])); }
    HIPCHK(hipMemcpyAsync(h->d_y, y, (size_t)T * 8, hipMemcpyHostToDevice, h->stream));
    SeriesScope scope(h->v);
    h->v.y = h->d_y; h->v.trace_logmu = nullptr; h->v.trace_ess = nullptr;
    h->cur = 0;
    h->v.want_s2 = 0;   // exactly the launches of log_likelihood without traces (enqueue_log_likelihood)
    HIPCHK(do_init(h, y[0]));
    h->t = 1; h->inited = true; h->emitted = false;
    const int64_t stride = (T - 1) / nsample;   // >= G
    int k = 0, open_left = 0;
    for (int64_t t = 1; t < T; ++t) {
        const int emit = h->v.want_s2 ? 1 : 2;
        if (t == T - 1) h->v.want_s2 = 1;
        HIPCHK(ensure_breaks(h, (uint32_t)t, (uint32_t)T));
        if (!open_left && k < nsample && ((t - 1) % stride) == (stride - G) / 2) {
            HIPCHK(hipEventRecord(e0[k], h->stream));
            open_left = G;
        }
        HIPCHK(do_step(h, (uint32_t)t, emit, 0.0));
        if (open_left && --open_left == 0) { HIPCHK(hipEventRecord(e1[k], h->stream)); ++k; }
        h->cur ^= 1; h->t += 1;
    }
    if (open_left) { HIPCHK(hipEventRecord(e1[k], h->stream)); }   // (cannot happen: every bracket fits its stride)
    h->v.want_s2 = 1;
    rc = emit_if_needed(h);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    double sum = 0.0, mn = 1e30;
    for (int i = 0; i < k; ++i) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e0[i], e1[i]));
        sum += ms / G;
        mn = ms / G < mn ? ms / G : mn;
    }
    for (int i = 0; i < nsample; ++i) { (void)hipEventDestroy(e0[i]); (void)hipEventDestroy(e1[i]); }
    if (avg_ms) *avg_ms = k ? sum / k : 0.0;
    if (min_ms) *min_ms = k ? mn : 0.0;
    return SMC_OK;
}

__global__ void k_nop() {}

extern "C" int smc_event_overhead_ms(smc_handle h, int nsample, double* avg_ms) {
    if (!h || !avg_ms || nsample < 1) return fail(SMC_EINVAL, "smc_event_overhead_ms: bad argument");
    HIPCHK(hipSetDevice(h->device));
    std::vector<hipEvent_t> e0((size_t)nsample), e1((size_t)nsample);
    for (int i = 0; i < nsample; ++i) { HIPCHK(hipEventCreate(&e0[i])); HIPCHK(hipEventCreate(&e1[i])); }
    for (int i = 0; i < nsample; ++i) {
        hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, h->stream);   // keeps the stream busy like the real loop does
        HIPCHK(hipEventRecord(e0[i], h->stream));
        HIPCHK(hipEventRecord(e1[i], h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    double sum = 0.0;
    for (int i = 0; i < nsample; ++i) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e0[i], e1[i]));
        sum += ms;
        (void)hipEventDestroy(e0[i]); (void)hipEventDestroy(e1[i]);
    }
    *avg_ms = sum / nsample;
    return SMC_OK;
}

extern "C" int smc_get_state(smc_handle h, double* x, double* w, int32_t* anc) {
    if (!h) return fail(SMC_EINVAL, "smc_get_state: NULL handle");
    if (!h->inited) return fail(SMC_ESTATE, "smc_get_state: filter not initialised");
    HIPCHK(hipSetDevice(h->device));
    const FilterView& v = h->v;
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t rows = (size_t)h->d * v.ntheta;
    if (x)
        HIPCHK(hipMemcpy2D(x, (size_t)v.n * 8, v.x[h->cur], (size_t)v.npad * 8, (size_t)v.n * 8, rows,
                           hipMemcpyDeviceToHost));
    if (anc) {
        if (!v.anc) return fail(SMC_ESTATE, "smc_get_state: handle created without SMC_FLAG_ANCESTORS");
        HIPCHK(hipMemcpy2D(anc, (size_t)v.n * 4, v.anc, (size_t)v.npad * 4, (size_t)v.n * 4, (size_t)v.ntheta,
                           hipMemcpyDeviceToHost));
    }
    if (w) {
        int rc = emit_if_needed(h);
        if (rc) return rc;
        if (!h->d_wdense) HIPCHK(dalloc(&h->d_wdense, (size_t)v.ntheta * v.n + 2 * (size_t)h->d * v.ntheta));
        hipLaunchKernelGGL(k_dense_weights, dim3((unsigned)((v.n + 255) / 256), v.ntheta), dim3(256), 0, h->stream, v,
                           h->cur, h->d_wdense);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(w, h->d_wdense, (size_t)v.ntheta * v.n * 8, hipMemcpyDeviceToHost));
    }
    return SMC_OK;
}

extern "C" int smc_get_logZ(smc_handle h, double* logZ, double* ess) {
    if (!h) return fail(SMC_EINVAL, "smc_get_logZ: NULL handle");
    if (!h->inited) return fail(SMC_ESTATE, "smc_get_logZ: filter not initialised");
    HIPCHK(hipSetDevice(h->device));
    int rc = emit_if_needed(h);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (logZ) HIPCHK(hipMemcpy(logZ, h->v.logZ, (size_t)h->v.ntheta * 8, hipMemcpyDeviceToHost));
    if (ess) HIPCHK(hipMemcpy(ess, h->v.last_ess, (size_t)h->v.ntheta * 8, hipMemcpyDeviceToHost));
    return SMC_OK;
}

extern "C" int smc_permute(smc_handle h, const int32_t* a) {
    if (!h || !a) return fail(SMC_EINVAL, "smc_permute: NULL argument");
    if (!h->inited) return fail(SMC_ESTATE, "smc_permute: filter not initialised");
    h->win_k = 0;
    for (int m = 0; m < h->v.ntheta; ++m)
        if (a[m] < 0 || a[m] >= h->v.ntheta) return fail(SMC_EINVAL, "smc_permute: index out of range");
    HIPCHK(hipSetDevice(h->device));
    int rc = emit_if_needed(h);
    if (rc) return rc;
    const FilterView& v = h->v;
    // The call returns without waiting for the device (whatever follows on this handle is ordered behind on its stream; the host
    // goes on with the random-walk covariance meanwhile): the indices travel from a pinned copy the handle owns.
    if (!h->h_perm) HIPCHK(hipHostMalloc((void**)&h->h_perm, (size_t)v.ntheta * 4, hipHostMallocDefault));
    else HIPCHK(hipStreamSynchronize(h->stream));   // (a previous permutation's copy out of the same buffer has completed)
    memcpy(h->h_perm, a, (size_t)v.ntheta * 4);
    HIPCHK(hipMemcpyAsync(h->d_perm, h->h_perm, (size_t)v.ntheta * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_logZ_tmp, v.logZ, (size_t)v.ntheta * 8, hipMemcpyDeviceToDevice, h->stream));
    hipLaunchKernelGGL(k_permute, dim3((unsigned)((v.npad + 255) / 256), v.ntheta), dim3(256), 0, h->stream, v, h->cur, h->d,
                       h->d_perm, h->d_logZ_tmp);
    HIPCHK(hipGetLastError());
    h->cur ^= 1;
    // last_* (g, D, logmu, ess) describe slot-local weights: recompute them for the new layout
    HIPCHK(hipMemcpyAsync(h->d_logZ_tmp, v.logZ, (size_t)v.ntheta * 8, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(do_finalize(h, 0, h->t - 1));
    HIPCHK(hipMemcpyAsync(v.logZ, h->d_logZ_tmp, (size_t)v.ntheta * 8, hipMemcpyDeviceToDevice, h->stream));
    return SMC_OK;
}

extern "C" int smc_copy_from(smc_handle dst, smc_handle src, const uint8_t* mask) {
    if (!dst || !src || !mask) return fail(SMC_EINVAL, "smc_copy_from: NULL argument");
    if (dst == src) return fail(SMC_EINVAL, "smc_copy_from: dst and src are the same handle");
    if (!dst->inited || !src->inited) return fail(SMC_ESTATE, "smc_copy_from: filter not initialised");
    dst->win_k = 0;
    const FilterView &a = dst->v, &b = src->v;
    if (dst->model != src->model || a.n != b.n || a.seg != b.seg || a.ntheta != b.ntheta || dst->device != src->device)
        return fail(SMC_EINVAL, "smc_copy_from: handles differ in model, geometry or device");
    HIPCHK(hipSetDevice(dst->device));
    int rc = emit_if_needed(dst);
    if (rc) return rc;
    rc = emit_if_needed(src);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(src->stream));
    unsigned char* d_mask = reinterpret_cast<unsigned char*>(dst->d_perm);   // n_theta bytes fit in n_theta int32
    HIPCHK(hipMemcpyAsync(d_mask, mask, (size_t)a.ntheta, hipMemcpyHostToDevice, dst->stream));
    hipLaunchKernelGGL(k_copy_slots, dim3((unsigned)((a.npad + 255) / 256), a.ntheta), dim3(256), 0, dst->stream, a, dst->cur,
                       b, src->cur, dst->d, d_mask);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(dst->stream));
    dst->t = src->t;   // the accepted filters have seen the same observations
    return SMC_OK;
}

extern "C" int smc_slot_bytes(smc_handle h, int64_t* bytes) {
    if (!h || !bytes) return fail(SMC_EINVAL, "smc_slot_bytes: NULL argument");
    *bytes = slot_words(h->d, h->v.npad, h->v.nseg) * 8;
    return SMC_OK;
}

static int pack_unpack(smc_handle h, const int32_t* idx, int64_t k, void* buf, bool pack) {
    if (!h || k < 0 || (k > 0 && (!idx || !buf))) return fail(SMC_EINVAL, "smc_pack/unpack_slots: bad argument");
    if (!h->inited) return fail(SMC_ESTATE, "smc_pack/unpack_slots: filter not initialised");
    if (!pack) h->win_k = 0;
    if (k == 0) return SMC_OK;
    for (int64_t i = 0; i < k; ++i)
        if (idx[i] < 0 || idx[i] >= h->v.ntheta) return fail(SMC_EINVAL, "smc_pack/unpack_slots: index out of range");
    HIPCHK(hipSetDevice(h->device));
    int rc = emit_if_needed(h);
    if (rc) return rc;
    const FilterView& v = h->v;
    const int64_t W = slot_words(h->d, v.npad, v.nseg);
    const int64_t span = v.npad > v.nseg ? v.npad : v.nseg;
    // a slot may be packed several times (a heavy theta-particle copied to many ranks): k can exceed
    // n_theta, so go in chunks of the index buffer's capacity
    for (int64_t k0 = 0; k0 < k; k0 += v.ntheta) {
        const int64_t kc = k - k0 < v.ntheta ? k - k0 : v.ntheta;
        HIPCHK(hipMemcpyAsync(h->d_perm, idx + k0, (size_t)kc * 4, hipMemcpyHostToDevice, h->stream));
        const dim3 grid((unsigned)((span + 255) / 256), (unsigned)kc);
        uint64_t* b = (uint64_t*)buf + (size_t)k0 * W;
        if (pack)
            hipLaunchKernelGGL((k_pack_slots<true>), grid, dim3(256), 0, h->stream, v, h->cur, h->d, h->d_perm, b);
        else
            hipLaunchKernelGGL((k_pack_slots<false>), grid, dim3(256), 0, h->stream, v, h->cur, h->d, h->d_perm, b);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return SMC_OK;
}
extern "C" int smc_pack_slots(smc_handle h, const int32_t* idx, int64_t k, void* device_buf) {
    return pack_unpack(h, idx, k, device_buf, true);
}
extern "C" int smc_unpack_slots(smc_handle h, const int32_t* idx, int64_t k, const void* device_buf) {
    return pack_unpack(h, idx, k, const_cast<void*>(device_buf), false);
}

extern "C" int smc_get_weights_raw(smc_handle h, uint64_t* C, double* m, uint64_t* S, uint64_t* S2hi, uint64_t* S2lo) {
    if (!h) return fail(SMC_EINVAL, "smc_get_weights_raw: NULL handle");
    if (!h->inited) return fail(SMC_ESTATE, "smc_get_weights_raw: filter not initialised");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const FilterView& v = h->v;
    const size_t np = (size_t)v.ntheta * v.npad * 8, ns = (size_t)v.ntheta * v.nseg * 8;
    const int c = h->cur;
    if (C) HIPCHK(hipMemcpy(C, v.C[c], np, hipMemcpyDeviceToHost));
    if (m) HIPCHK(hipMemcpy(m, v.segk[c], ns, hipMemcpyDeviceToHost));
    if (S) HIPCHK(hipMemcpy(S, v.segS[c], ns, hipMemcpyDeviceToHost));
    if (S2hi) HIPCHK(hipMemcpy(S2hi, v.segS2hi[c], ns, hipMemcpyDeviceToHost));
    if (S2lo) HIPCHK(hipMemcpy(S2lo, v.segS2lo[c], ns, hipMemcpyDeviceToHost));
    return SMC_OK;
}

extern "C" int smc_get_geometry(smc_handle h, int* seg, int* nseg, int* d, int* resident) {
    if (!h) return fail(SMC_EINVAL, "smc_get_geometry: NULL handle");
    if (seg) *seg = h->v.seg;
    if (nseg) *nseg = h->v.nseg;
    if (d) *d = h->d;
    if (resident) *resident = (h->resident_ok && resident_supported(h->model, h->v.seg)) ? 1 : 0;
    return SMC_OK;
}

extern "C" int smc_last_elapsed_ms(smc_handle h, double* ms) {
    if (!h || !ms) return fail(SMC_EINVAL, "smc_last_elapsed_ms: NULL argument");
    *ms = h->last_ms;
    return SMC_OK;
}

extern "C" int smc_synchronize(smc_handle h) {
    if (!h) return fail(SMC_EINVAL, "smc_synchronize: NULL handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SMC_OK;
}

// ---- stand-alone normalize / resample ------------------------------------------------------------
static int fix_bits_for(int64_t n) {
    const int k = 61 - ceil_log2_i64(n);
    return k > FIX_BITS ? FIX_BITS : k;
}

// Scratch of the stand-alone entry points: device buffers kept per (host thread, slot) and grown on demand - hipMalloc / hipFree of a
// whole cloud's worth on every call cost 30 ms per normalize at 2^20 entries, ten times the work - released when the thread
// ends; and a private non-blocking stream per (host thread, device) so that these calls never serialise against the null stream.
namespace {
struct ScratchSlot {
    void* p = nullptr;
    size_t cap = 0;
    int device = -1;
    ~ScratchSlot() { if (p) (void)hipFree(p); }
};
struct DevBuf {   // a view of one cached slot: alloc() may be called once per call and slot
    void* p = nullptr;
    int slot;
    explicit DevBuf(int s) : slot(s) {}
    hipError_t alloc(size_t bytes) {
        static thread_local ScratchSlot slots[8];
        ScratchSlot& c = slots[slot];
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        bytes = bytes ? bytes : 16;
        if (c.cap < bytes || c.device != dev) {
            if (c.p) { (void)hipFree(c.p); c.p = nullptr; c.cap = 0; }
            e = hipMalloc(&c.p, bytes);
            if (e != hipSuccess) return e;
            c.cap = bytes;
            c.device = dev;
        }
        p = c.p;
        return hipSuccess;
    }
    template <class T> T* as() const { return (T*)p; }
};
hipError_t util_stream(int device, hipStream_t* out) {
    static thread_local hipStream_t streams[16] = {};
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return e;
    if (device < 0 || device >= 16) { *out = nullptr; return hipSuccess; }
    if (!streams[device]) {
        e = hipStreamCreateWithFlags(&streams[device], hipStreamNonBlocking);
        if (e != hipSuccess) return e;
    }
    *out = streams[device];
    return hipSuccess;
}
}  // namespace

extern "C" int smc_normalize(const double* logw, int64_t n, double* w, double* logmu, double* ess, int device) {
    if (!logw || !w || n <= 0) return fail(SMC_EINVAL, "smc_normalize: bad argument");
    hipStream_t st = nullptr;
    HIPCHK(util_stream(device, &st));
    DevBuf d_in(0), d_w(1), d_o(2);
    HIPCHK(d_in.alloc((size_t)n * 8));
    HIPCHK(d_w.alloc((size_t)n * 8));
    HIPCHK(d_o.alloc(16));
    HIPCHK(hipMemcpyAsync(d_in.p, logw, (size_t)n * 8, hipMemcpyHostToDevice, st));
    if (n <= 16384) {
        // one workgroup: the outer reweight works on n_theta-vectors (a few thousand entries)
        hipLaunchKernelGGL((k_normalize<1024>), dim3(1), dim3(1024), 0, st, d_in.as<double>(), n, fix_bits_for(n), d_w.as<double>(), d_o.as<double>());
        HIPCHK(hipGetLastError());
    } else {
        // a whole cloud's log-weights: three grid-wide passes, every cross-workgroup combination an integer one (same bits)
        DevBuf d_acc(3);
        HIPCHK(d_acc.alloc(5 * 8));
        unsigned long long* acc = d_acc.as<unsigned long long>();
        int* kmax_i = reinterpret_cast<int*>(acc + 4);
        HIPCHK(hipMemsetAsync(acc, 0, 32, st));
        HIPCHK(hipMemsetD32Async(kmax_i, NORM_DEAD, 1, st));
        int64_t nb = (n + 4 * 256 - 1) / (4 * 256);
        nb = nb > 2048 ? 2048 : nb;
        const int K = fix_bits_for(n);
        hipLaunchKernelGGL((k_normalize_max<256>), dim3((unsigned)nb), dim3(256), 0, st, d_in.as<double>(), n, kmax_i);
        hipLaunchKernelGGL((k_normalize_sum<256>), dim3((unsigned)nb), dim3(256), 0, st, d_in.as<double>(), n, K, kmax_i, acc);
        hipLaunchKernelGGL((k_normalize_write<256>), dim3((unsigned)nb), dim3(256), 0, st, d_in.as<double>(), n, K, kmax_i, acc,
                           d_w.as<double>(), d_o.as<double>());
        HIPCHK(hipGetLastError());
        double o[2];
        HIPCHK(hipMemcpyAsync(w, d_w.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(o, d_o.p, 16, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));       // (d_acc lives until here)
        if (logmu) *logmu = o[0];
        if (ess) *ess = o[1];
        return SMC_OK;
    }
    double o[2];
    HIPCHK(hipMemcpyAsync(w, d_w.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(o, d_o.p, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (logmu) *logmu = o[0];
    if (ess) *ess = o[1];
    return SMC_OK;
}

extern "C" int smc_resample(const double* w, int64_t n, int64_t ndraw, uint64_t seed, uint32_t stream, uint32_t t,
                            int32_t* a, int device) {
    if (!w || !a || n <= 0 || ndraw < 0) return fail(SMC_EINVAL, "smc_resample: bad argument");
    if (n > ((int64_t)1 << 31)) return fail(SMC_EINVAL, "smc_resample: n > 2^31");
    if (ndraw == 0) return SMC_OK;
    hipStream_t st = nullptr;
    HIPCHK(util_stream(device, &st));
    DevBuf d_w(0), d_C(1), d_a(2), d_st(3);
    HIPCHK(d_w.alloc((size_t)n * 8));
    HIPCHK(d_C.alloc((size_t)n * 8));
    HIPCHK(d_a.alloc((size_t)ndraw * 4));
    HIPCHK(d_st.alloc(4));
    HIPCHK(hipMemcpyAsync(d_w.p, w, (size_t)n * 8, hipMemcpyHostToDevice, st));
    if (n <= 65536) {
        hipLaunchKernelGGL((k_resample_cdf<1024>), dim3(1), dim3(1024), 0, st, d_w.as<double>(), n, fix_bits_for(n), d_C.as<uint64_t>(), d_st.as<int>());
    } else {   // a whole cloud's weights: the inclusive sums grid-wide (the same integers)
        constexpr int TH = 256;
        int nb = (int)((n + 4 * TH - 1) / (4 * TH));
        nb = nb > 2048 ? 2048 : nb;
        const int64_t chunk = ((n + nb - 1) / nb + TH - 1) / TH * TH;   // contiguous, a multiple of the workgroup
        nb = (int)((n + chunk - 1) / chunk);
        DevBuf d_bs(4);
        HIPCHK(d_bs.alloc(((size_t)nb + 1) * 8));
        unsigned long long* mbits = d_bs.as<unsigned long long>() + nb;
        HIPCHK(hipMemsetAsync(mbits, 0, 8, st));
        hipLaunchKernelGGL((k_rs_max<TH>), dim3((unsigned)nb), dim3(TH), 0, st, d_w.as<double>(), n, mbits);
        hipLaunchKernelGGL((k_rs_chunk_sums<TH>), dim3((unsigned)nb), dim3(TH), 0, st, d_w.as<double>(), n, fix_bits_for(n), mbits, chunk, d_bs.as<uint64_t>());
        hipLaunchKernelGGL((k_rs_scan_chunks<1024>), dim3(1), dim3(1024), 0, st, mbits, nb, d_bs.as<uint64_t>(), d_st.as<int>());
        hipLaunchKernelGGL((k_rs_write<TH>), dim3((unsigned)nb), dim3(TH), 0, st, d_w.as<double>(), n, fix_bits_for(n), mbits, chunk, d_bs.as<uint64_t>(), d_C.as<uint64_t>());
    }
    HIPCHK(hipGetLastError());
    int status = 0;
    HIPCHK(hipMemcpyAsync(&status, d_st.p, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (status != 0) return fail(SMC_EINVAL, "smc_resample: weights must be finite with a positive maximum");
    hipLaunchKernelGGL(k_resample_draw, dim3((unsigned)((ndraw + 255) / 256)), dim3(256), 0, st, d_C.as<uint64_t>(), n, ndraw, seed,
                       stream, t, d_a.as<int32_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(a, d_a.p, (size_t)ndraw * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return SMC_OK;
}

extern "C" int smc_kalman_log_likelihood(const double* raw, int64_t n_theta, const double* y, int64_t T, int predict_first,
                                         double* out, int device) {
    if (!raw || !y || !out || n_theta <= 0 || T <= 0) return fail(SMC_EINVAL, "smc_kalman_log_likelihood: bad argument");
    hipStream_t st = nullptr;
    HIPCHK(util_stream(device, &st));
    DevBuf d_raw(0), d_y(1), d_out(2);
    HIPCHK(d_raw.alloc((size_t)n_theta * 48));
    HIPCHK(d_y.alloc((size_t)T * 8));
    HIPCHK(d_out.alloc((size_t)n_theta * 24));
    HIPCHK(hipMemcpyAsync(d_raw.p, raw, (size_t)n_theta * 48, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_y.p, y, (size_t)T * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_kalman, dim3((unsigned)((n_theta + 63) / 64)), dim3(64), 0, st, d_raw.as<double>(), n_theta, d_y.as<double>(), T,
                       predict_first, d_out.as<double>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_out.p, (size_t)n_theta * 24, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return SMC_OK;
}

// Quantiles and / or moments of the CURRENT state of single-segment filters in ONE launch that writes to pinned host memory
// (the README loop asks for them after every bootstrap_filter!: one launch and one synchronisation per request instead of
// sixteen launches and a copy).  done = false: no such kernel for this handle (several segments, or the state does not fit LDS).
static int summaries_once(smc_handle h, int component, const double* p, int np, bool mom, double* q_out, double* mean, double* var, bool& done) {
    done = false;
    if (h->v.nseg != 1) return SMC_OK;
    const size_t nth = (size_t)h->v.ntheta, nout = (size_t)h->d * nth;
    if (!h->h_once) HIPCHK(hipHostMalloc((void**)&h->h_once, ((size_t)QMAX * nth + 2 * nout) * 8, hipHostMallocDefault));
    FilterView v = h->v;
    v.sum_np = np; v.sum_comp = component; v.sum_mom = mom ? 1 : 0;
    for (int j = 0; j < QMAX; ++j) v.sum_p64[j] = j < np ? prob_to_u64(p[j]) : 0;
    v.sum_q = h->h_once; v.sum_m = h->h_once + (size_t)QMAX * nth;
    hipError_t e = hipErrorInvalidValue;
    switch (h->model) {
    case MODEL_LG1D: e = launch_summ_once<MODEL_LG1D>(v, h->cur, h->stream); break;
    case MODEL_SV1D: e = launch_summ_once<MODEL_SV1D>(v, h->cur, h->stream); break;
    case MODEL_UCSV3D: e = launch_summ_once<MODEL_UCSV3D>(v, h->cur, h->stream); break;
    }
    if (e == hipErrorInvalidValue) return SMC_OK;
    HIPCHK(e);
    HIPCHK(hipStreamSynchronize(h->stream));
    if (q_out) memcpy(q_out, h->h_once, nth * np * 8);
    if (mean) memcpy(mean, h->h_once + (size_t)QMAX * nth, nout * 8);
    if (var) memcpy(var, h->h_once + (size_t)QMAX * nth + nout, nout * 8);
    done = true;
    return SMC_OK;
}

extern "C" int smc_get_moments(smc_handle h, double* mean, double* var) {
    if (!h || !mean || !var) return fail(SMC_EINVAL, "smc_get_moments: NULL argument");
    if (!h->inited) return fail(SMC_ESTATE, "smc_get_moments: filter not initialised");
    HIPCHK(hipSetDevice(h->device));
    int rc = emit_if_needed(h);
    if (rc) return rc;
    bool done = false;
    if ((rc = summaries_once(h, 0, nullptr, 0, true, nullptr, mean, var, done)) || done) return rc;
    const size_t nout = (size_t)h->d * h->v.ntheta;
    if (!h->d_wdense) HIPCHK(dalloc(&h->d_wdense, (size_t)h->v.ntheta * h->v.n + 2 * nout));
    double *d_mean = h->d_wdense, *d_var = h->d_wdense + nout;
    if ((rc = enqueue_ms(h, 0, 0, nullptr, true, nullptr, d_mean, d_var))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(mean, d_mean, nout * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(var, d_var, nout * 8, hipMemcpyDeviceToHost));
    return SMC_OK;
}

extern "C" int smc_get_quantiles(smc_handle h, int component, const double* p, int np, double* out) {
    if (!h || !p || !out) return fail(SMC_EINVAL, "smc_get_quantiles: NULL argument");
    if (!h->inited) return fail(SMC_ESTATE, "smc_get_quantiles: filter not initialised");
    if (component < 0 || component >= h->d) return fail(SMC_EINVAL, "smc_get_quantiles: component out of range");
    if (np < 1 || np > QMAX) return fail(SMC_EINVAL, "smc_get_quantiles: 1 <= np <= 8");
    HIPCHK(hipSetDevice(h->device));
    int rc = emit_if_needed(h);
    if (rc) return rc;
    bool done = false;
    if ((rc = summaries_once(h, component, p, np, false, out, nullptr, nullptr, done)) || done) return rc;
    const size_t nth = (size_t)h->v.ntheta, nst = nth * np;
    uint64_t hp[QMAX];
    for (int j = 0; j < QMAX; ++j) hp[j] = j < np ? prob_to_u64(p[j]) : 0;
    if ((rc = ensure_ms(h))) return rc;
    double* d_out = (double*)(h->d_ms + ms_words(nth, (size_t)h->v.nseg, (size_t)h->d));
    if ((rc = enqueue_ms(h, component, np, hp, false, d_out, nullptr, nullptr))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, d_out, nst * 8, hipMemcpyDeviceToHost));
    return SMC_OK;
}

// ---- host-side helpers ---------------------------------------------------------------------------
template <int MODEL>
static void simulate_t(const Params& p, int64_t T, uint64_t seed, double* x, double* y) {
    constexpr int D = model_dim<MODEL>::value;
    double xc[D], xn[D], z[D], z1, mean, sd;
    for (int64_t t = 0; t < T; ++t) {
        for (int c = 0; c < D; ++c) box_muller(draw(seed, 0u, SIM_STREAM, (uint32_t)t, SLOT_NORMAL0 + c), z[c], z1);
        if (t == 0) model_initial<MODEL>(p, z, xn); else model_transition<MODEL>(p, xc, z, xn);
        model_obs_moments<MODEL>(p, xn, mean, sd);
        double e;
        box_muller(draw(seed, 0u, SIM_STREAM, (uint32_t)t, SLOT_OBS), e, z1);
        y[t] = fma(sd, e, mean);
        for (int c = 0; c < D; ++c) { xc[c] = xn[c]; if (x) x[(size_t)c * T + t] = xn[c]; }
    }
}

extern "C" int smc_simulate(int model_id, const double* raw, int64_t T, uint64_t seed, double* x, double* y) {
    const int nraw = model_nraw_rt(model_id);
    if (nraw < 0 || !raw || !y || T <= 0) return fail(SMC_EINVAL, "smc_simulate: bad argument");
    Params p;
    for (int k = 0; k < NPARAM; ++k) p.raw[k] = k < nraw ? raw[k] : 0.0;
    derive_params(model_id, p.raw, p.der);
    switch (model_id) {
    case MODEL_LG1D: simulate_t<MODEL_LG1D>(p, T, seed, x, y); break;
    case MODEL_SV1D: simulate_t<MODEL_SV1D>(p, T, seed, x, y); break;
    default: simulate_t<MODEL_UCSV3D>(p, T, seed, x, y); break;
    }
    return SMC_OK;
}

extern "C" double smc_host_exp(double x) { return sp_exp(x); }
extern "C" double smc_host_log(double x) { return sp_log(x); }
extern "C" void smc_host_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    const u32x4 r = philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    for (int i = 0; i < 4; ++i) out[i] = r.v[i];
}
extern "C" void smc_host_box_muller(const uint32_t w[4], double* z0, double* z1) {
    box_muller(u32x4{{w[0], w[1], w[2], w[3]}}, *z0, *z1);
}

// (the outer level of the samplers - reweight, the window walk, the tempering bisection, resample!, the random-walk factor -
// lives in smc_outer.hip)

extern "C" int smc_host_pmmh_propose(int d_theta, uint64_t move_seed, uint32_t stream, uint32_t c, const double* theta,
                                     const double* chol, double scale, double* prop) {
    if (d_theta < 1 || d_theta > MAX_DTHETA || !theta || !chol || !prop) return fail(SMC_EINVAL, "smc_host_pmmh_propose: bad argument");
    PmmhSpec sp{};
    sp.d = d_theta;
    pmmh_propose(sp, move_seed, stream, c, theta, chol, sqrt(scale), prop);
    return SMC_OK;
}
extern "C" double smc_host_pmmh_log_uniform(uint64_t move_seed, uint32_t stream, uint32_t c) { return pmmh_log_uniform(move_seed, stream, c); }
extern "C" double smc_host_prior_logpdf(int family, const double* par, double x) {
    return prior_insupport(family, par, x) ? prior_logpdf(family, par, x) : -inf();
}

__global__ void k_device_math(int which, const double* a, const double* b, int64_t n, double* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = a[i];
    double r = 0.0;
    if (which == 0) r = sp_exp(x);
    else if (which == 1) r = sp_log(x);
    else if (which == 2) r = sqrt(x);
    else if (which == 3 || which == 4) {
        const uint64_t ua = d2bits(x), ub = d2bits(b[i]);
        double z0, z1;
        box_muller(u32x4{{(uint32_t)ua, (uint32_t)(ua >> 32), (uint32_t)ub, (uint32_t)(ub >> 32)}}, z0, z1);
        r = which == 3 ? z0 : z1;
    } else if (which == 5) r = x / b[i];
    out[i] = r;
}

__global__ void k_sys_targets(uint64_t Dtot, uint32_t n, uint64_t u, uint64_t j0, int nk, uint64_t* out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nk) out[k] = sys_target(sys_base(Dtot, n, 1.0 / (double)n, u, j0), (uint32_t)k);
}
// T_{j0+k} = floor(((j0+k) Dtot + mulhi64(u, Dtot)) / n), k < nk <= 8192: the division-free evaluation the
// systematic kernels use, on the host (device < 0) or on a device - tests compare both with exact integers
extern "C" int smc_sys_targets(uint64_t Dtot, uint32_t n, uint64_t u, uint64_t j0, int nk, uint64_t* out, int device) {
    if (!out || nk < 1 || nk > 8192 || n < 1 || n >= (1u << 31) || Dtot >= (1ull << 63) || j0 + (uint64_t)nk > n)
        return fail(SMC_EINVAL, "smc_sys_targets: bad argument");
    if (device < 0) {
        const SysBase sb = sys_base(Dtot, n, 1.0 / (double)n, u, j0);
        for (int k = 0; k < nk; ++k) out[k] = sys_target(sb, (uint32_t)k);
        return SMC_OK;
    }
    hipStream_t st = nullptr;
    HIPCHK(util_stream(device, &st));
    DevBuf d(0);
    HIPCHK(d.alloc((size_t)nk * 8));
    hipLaunchKernelGGL(k_sys_targets, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, st, Dtot, n, u, j0, nk, d.as<uint64_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d.p, (size_t)nk * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return SMC_OK;
}

extern "C" int smc_device_math(int which, const double* a, const double* b, int64_t n, double* out, int device) {
    if (!a || !out || n <= 0 || which < 0 || which > 5) return fail(SMC_EINVAL, "smc_device_math: bad argument");
    if (which >= 3 && !b) return fail(SMC_EINVAL, "smc_device_math: b required");
    hipStream_t st = nullptr;
    HIPCHK(util_stream(device, &st));
    DevBuf da(0), db(1), dout(2);
    HIPCHK(da.alloc((size_t)n * 8));
    HIPCHK(db.alloc((size_t)n * 8));
    HIPCHK(dout.alloc((size_t)n * 8));
    HIPCHK(hipMemcpyAsync(da.p, a, (size_t)n * 8, hipMemcpyHostToDevice, st));
    if (b) HIPCHK(hipMemcpyAsync(db.p, b, (size_t)n * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_device_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, which, da.as<double>(), db.as<double>(), n, dout.as<double>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dout.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return SMC_OK;
}
