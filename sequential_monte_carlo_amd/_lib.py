"""ctypes binding of libsmchip.so (C ABI: include/smc_hip.h).

The HIP extension is the product: if it is missing or cannot be loaded this module raises.
There is no CPU fallback anywhere in this package.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMC_LIB") or os.path.join(_HERE, "lib", "libsmchip.so")   # SMC_LIB: profiling builds

MODEL_LG1D, MODEL_SV1D, MODEL_UCSV3D = 1, 2, 3
FLAG_ANCESTORS, FLAG_NO_RESIDENT, FLAG_SYSTEMATIC = 1, 2, 4

# every symbol include/smc_hip.h declares
EXPORTS = [
    "smc_create", "smc_destroy", "smc_set_params", "smc_set_streams", "smc_reseed", "smc_init", "smc_step",
    "smc_log_likelihood", "smc_get_state", "smc_get_logZ", "smc_permute", "smc_copy_from", "smc_slot_bytes", "smc_pack_slots", "smc_unpack_slots", "smc_get_weights_raw", "smc_get_geometry",
    "smc_last_elapsed_ms", "smc_synchronize", "smc_time_step_kernel", "smc_event_overhead_ms", "smc_normalize", "smc_resample", "smc_kalman_log_likelihood", "smc_get_moments", "smc_get_quantiles", "smc_simulate", "smc_model_dim",
    "smc_model_nraw", "smc_auto_seg", "smc_device_count", "smc_host_exp", "smc_host_log", "smc_host_philox4x32_10",
    "smc_host_box_muller", "smc_sys_targets", "smc_device_math", "smc_last_error", "smc_version",
    "smc_set_skip", "smc_pmmh_configure", "smc_pmmh_rejuvenate", "smc_host_pmmh_propose", "smc_host_pmmh_log_uniform",
    "smc_host_prior_logpdf", "smc_step_window", "smc_step_commit",
    "smc_comm_unique_id", "smc_comm_create", "smc_comm_destroy", "smc_comm_rank", "smc_comm_all_gather", "smc_outer_reweight",
    "smc_comm_exchange_slots", "smc_host_reweight", "smc_comm_plan_exchange",
    "smc_outer_seg", "smc_host_outer_records", "smc_host_outer_combine", "smc_host_outer_window", "smc_host_outer_walk",
    "smc_host_outer_advance", "smc_host_outer_temper", "smc_host_outer_resample", "smc_host_rw_factor",
    "smc_set_summaries", "smc_get_summaries",
]
COMM_ID_BYTES = 128
PRIOR_UNIFORM, PRIOR_NORMAL, PRIOR_TRUNCNORMAL, PRIOR_LOGNORMAL, PRIOR_NPAR, MAX_DTHETA = 1, 2, 3, 4, 5, 8

_dp = C.POINTER(C.c_double)
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_ip = C.POINTER(C.c_int)


class SmcError(RuntimeError):
    pass


_lib = None


def _share_hip_runtime_with_torch():
    """One HIP/HSA runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (same soname
    as /opt/rocm's); whichever copy is loaded first serves both users, but if OUR library pulls in the system
    copy first and torch initialises later, torch's bundled HSA runtime finds "No HIP GPUs".  So when a
    torch wheel is installed we pre-load ITS runtime (a dlopen of one file, torch itself is not imported);
    libsmchip.so's DT_NEEDED libamdhip64.so.7 then binds to it.  SMC_HIP_RUNTIME=system opts out."""
    if os.environ.get("SMC_HIP_RUNTIME", "") == "system":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:   # noqa: BLE001  (fall back to the system runtime)
        pass


def lib():
    """Load libsmchip.so (built by `__graft_entry__.build()` / csrc/Makefile). Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SmcError("HIP extension %s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    h = C.c_void_p
    L.smc_create.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_uint64, C.c_int, C.c_uint32, C.POINTER(h)]
    L.smc_destroy.argtypes = [h]
    L.smc_set_params.argtypes = [h, _dp]
    L.smc_set_streams.argtypes = [h, _u32p]
    L.smc_reseed.argtypes = [h, C.c_uint64]
    L.smc_init.argtypes = [h, C.c_double, _dp]
    L.smc_step.argtypes = [h, C.c_double, _dp, _dp]
    L.smc_log_likelihood.argtypes = [h, _dp, C.c_int64, _dp, _dp, _dp]
    L.smc_get_state.argtypes = [h, _dp, _dp, _i32p]
    L.smc_get_logZ.argtypes = [h, _dp, _dp]
    L.smc_permute.argtypes = [h, _i32p]
    L.smc_copy_from.argtypes = [h, h, C.POINTER(C.c_uint8)]
    L.smc_slot_bytes.argtypes = [h, C.POINTER(C.c_int64)]
    L.smc_pack_slots.argtypes = [h, _i32p, C.c_int64, C.c_void_p]
    L.smc_unpack_slots.argtypes = [h, _i32p, C.c_int64, C.c_void_p]
    L.smc_get_weights_raw.argtypes = [h, _u64p, _dp, _u64p, _u64p, _u64p]
    L.smc_get_geometry.argtypes = [h, _ip, _ip, _ip, _ip]
    L.smc_last_elapsed_ms.argtypes = [h, _dp]
    L.smc_synchronize.argtypes = [h]
    L.smc_time_step_kernel.argtypes = [h, _dp, C.c_int64, C.c_int, _dp, _dp]
    L.smc_event_overhead_ms.argtypes = [h, C.c_int, _dp]
    L.smc_normalize.argtypes = [_dp, C.c_int64, _dp, _dp, _dp, C.c_int]
    L.smc_resample.argtypes = [_dp, C.c_int64, C.c_int64, C.c_uint64, C.c_uint32, C.c_uint32, _i32p, C.c_int]
    L.smc_kalman_log_likelihood.argtypes = [_dp, C.c_int64, _dp, C.c_int64, C.c_int, _dp, C.c_int]
    L.smc_get_moments.argtypes = [h, _dp, _dp]
    L.smc_get_quantiles.argtypes = [h, C.c_int, _dp, C.c_int, _dp]
    L.smc_set_summaries.argtypes = [h, C.c_int, _dp, C.c_int, C.c_int]
    L.smc_get_summaries.argtypes = [h, C.c_int64, _dp, _dp, _dp]
    L.smc_sys_targets.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_uint64), C.c_int]
    L.smc_simulate.argtypes = [C.c_int, _dp, C.c_int64, C.c_uint64, _dp, _dp]
    L.smc_model_dim.argtypes = [C.c_int]
    L.smc_model_nraw.argtypes = [C.c_int]
    L.smc_auto_seg.argtypes = [C.c_int, C.c_int64]
    L.smc_host_exp.restype = C.c_double
    L.smc_host_exp.argtypes = [C.c_double]
    L.smc_host_log.restype = C.c_double
    L.smc_host_log.argtypes = [C.c_double]
    L.smc_host_philox4x32_10.restype = None
    L.smc_host_philox4x32_10.argtypes = [_u32p, _u32p, _u32p]
    L.smc_host_box_muller.restype = None
    L.smc_host_box_muller.argtypes = [_u32p, _dp, _dp]
    L.smc_device_math.argtypes = [C.c_int, _dp, _dp, C.c_int64, _dp, C.c_int]
    L.smc_step_window.argtypes = [h, _dp, C.c_int, _dp, _dp]
    L.smc_step_commit.argtypes = [h, C.c_int]
    L.smc_set_skip.argtypes = [h, C.POINTER(C.c_uint8)]
    L.smc_pmmh_configure.argtypes = [h, C.c_int, _i32p, _dp, _i32p, _dp]
    L.smc_pmmh_rejuvenate.argtypes = [h, h, _dp, C.c_int64, C.c_double, _dp, _dp, C.c_int, _u64p, C.c_uint64, _dp, _dp,
                                      C.POINTER(C.c_uint8), C.POINTER(C.c_int64)]
    L.smc_host_pmmh_propose.argtypes = [C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, _dp, _dp, C.c_double, _dp]
    L.smc_host_pmmh_log_uniform.restype = C.c_double
    L.smc_host_pmmh_log_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
    L.smc_host_prior_logpdf.restype = C.c_double
    L.smc_host_prior_logpdf.argtypes = [C.c_int, _dp, C.c_double]
    L.smc_host_reweight.argtypes = [_dp, C.c_int64, _dp, _dp, _dp]
    L.smc_host_outer_records.argtypes = [_dp, C.c_int64, _u64p]
    L.smc_host_outer_combine.argtypes = [_u64p, C.c_int64, C.c_int64, _dp, _dp]
    L.smc_host_outer_window.argtypes = [_dp, _dp, C.c_int, C.c_int64, _u64p]
    L.smc_host_outer_walk.argtypes = [_u64p, C.c_int, C.c_int64, C.c_int64, C.c_double, _dp, _ip]
    L.smc_host_outer_advance.argtypes = [_dp, _dp, _dp, C.c_int, C.c_int64]
    L.smc_host_outer_temper.argtypes = [_dp, C.c_int64, C.c_double, C.c_double, _dp, _dp, _ip, _dp]
    L.smc_host_outer_resample.argtypes = [_dp, C.c_int64, C.c_int64, C.c_uint64, _i32p]
    L.smc_host_rw_factor.argtypes = [_dp, C.c_int64, C.c_int, _dp, _ip]
    L.smc_comm_unique_id.argtypes = [C.c_void_p]
    L.smc_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(h)]
    L.smc_comm_destroy.argtypes = [h]
    L.smc_comm_rank.argtypes = [h, _ip, _ip]
    L.smc_comm_all_gather.argtypes = [h, _dp, C.c_int64, _dp]
    L.smc_outer_reweight.argtypes = [h, _dp, C.c_int64, _dp, _dp, _dp, _dp]
    L.smc_comm_exchange_slots.argtypes = [h, h, _i32p, C.c_int64]
    L.smc_comm_plan_exchange.argtypes = [_i32p, C.c_int64, C.c_int, C.c_int, _i32p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _i32p,
                                         C.POINTER(C.c_int64)]
    L.smc_last_error.restype = C.c_char_p
    L.smc_version.restype = C.c_char_p
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise SmcError("libsmchip error %d: %s" % (rc, lib().smc_last_error().decode()))


def _d(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def device_count():
    return lib().smc_device_count()


def sys_targets(Dtot, n, u, j0, nk, device=-1):
    """T_{j0+k}, k < nk, of systematic resampling (smc_sys_targets); device < 0: host evaluation."""
    out = np.zeros(nk, dtype=np.uint64)
    check(lib().smc_sys_targets(int(Dtot), int(n), int(u), int(j0), int(nk), out.ctypes.data_as(C.POINTER(C.c_uint64)), device))
    return out


def simulate(model_id, raw, T, seed):
    """simulate(rng, model, T) -> (x [d][T], y [T])   src/state_space_models.jl:11-26 (host code)."""
    raw = np.ascontiguousarray(raw, dtype=np.float64)
    d = lib().smc_model_dim(model_id)
    x = np.zeros((d, T))
    y = np.zeros(T)
    check(lib().smc_simulate(model_id, _d(raw), T, seed, _d(x), _d(y)))
    return x, y


def normalize(logw, device=0):
    logw = np.ascontiguousarray(logw, dtype=np.float64)
    w = np.zeros_like(logw)
    lm, ess = C.c_double(), C.c_double()
    check(lib().smc_normalize(_d(logw), logw.size, _d(w), C.byref(lm), C.byref(ess), device))
    return lm.value, w, ess.value


def resample(w, ndraw=None, seed=0, stream=0, t=0, device=0):
    w = np.ascontiguousarray(w, dtype=np.float64)
    ndraw = w.size if ndraw is None else int(ndraw)
    a = np.zeros(ndraw, dtype=np.int32)
    check(lib().smc_resample(_d(w), w.size, ndraw, seed, stream, t, a.ctypes.data_as(_i32p), device))
    return a


def kalman_log_likelihood(raw, y, predict_first=False, device=0):
    """Batched exact scalar Kalman filter (src/kalman_filter.jl:29-70): rows of (x_T, Sigma_T, logZ)."""
    raw = np.ascontiguousarray(raw, dtype=np.float64).reshape(-1, 6)
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = np.zeros((raw.shape[0], 3))
    check(lib().smc_kalman_log_likelihood(_d(raw), raw.shape[0], _d(y), y.size, int(predict_first), _d(out), device))
    return out


def device_math(which, a, b=None, device=0):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64) if b is not None else None
    out = np.zeros_like(a)
    check(lib().smc_device_math(which, _d(a), _d(b), a.size, _d(out), device))
    return out


OUTER_SEG = 8     # SMC_OUTER_SEG: entries per segment of the outer level's integer normalisation


def host_reweight(logw, want_w=True):
    """reweight(logw) -> (logmu, w, ess): the outer level's integer normalize (smc_host_reweight; no GPU needed)"""
    logw = np.ascontiguousarray(logw, dtype=np.float64)
    w = np.empty_like(logw) if want_w else None
    lm, ess = C.c_double(), C.c_double()
    check(lib().smc_host_reweight(_d(logw), logw.size, _d(w), C.byref(lm), C.byref(ess)))
    return lm.value, w, ess.value


def host_outer_records(logw_local):
    """segment records [nseg_local][4] (uint64 words) of the whole segments a rank holds (smc_host_outer_records)"""
    lw = np.ascontiguousarray(logw_local, dtype=np.float64)
    rec = np.zeros(((lw.size + OUTER_SEG - 1) // OUTER_SEG, 4), dtype=np.uint64)
    check(lib().smc_host_outer_records(_d(lw), lw.size, rec.ctypes.data_as(_u64p)))
    return rec


def host_outer_combine(rec, n_total):
    """(logmu, ess) from the records of ALL segments (smc_host_outer_combine)"""
    rec = np.ascontiguousarray(rec, dtype=np.uint64).reshape(-1, 4)
    lm, ess = C.c_double(), C.c_double()
    check(lib().smc_host_outer_combine(rec.ctypes.data_as(_u64p), rec.shape[0], int(n_total), C.byref(lm), C.byref(ess)))
    return lm.value, ess.value


def host_outer_window(logw_local, lik):
    """records [k][nseg_local][4] of the k steps of a window over this rank's entries (smc_host_outer_window)"""
    lw = np.ascontiguousarray(logw_local, dtype=np.float64)
    lik = np.ascontiguousarray(lik, dtype=np.float64)
    k, n = lik.shape
    assert n == lw.size
    rec = np.zeros((k, (n + OUTER_SEG - 1) // OUTER_SEG, 4), dtype=np.uint64)
    check(lib().smc_host_outer_window(_d(lw), _d(lik), k, n, rec.ctypes.data_as(_u64p)))
    return rec


def host_outer_walk(rec, n_total, ess_min):
    """(ess [j], j): walk through the steps of a window from the records of ALL segments [k][nseg][4] (smc_host_outer_walk)"""
    rec = np.ascontiguousarray(rec, dtype=np.uint64)
    k, nseg = rec.shape[0], rec.shape[1]
    ess = np.zeros(k)
    j = C.c_int()
    check(lib().smc_host_outer_walk(rec.ctypes.data_as(_u64p), k, nseg, int(n_total), float(ess_min), _d(ess), C.byref(j)))
    return ess[:j.value], j.value


def host_outer_advance(logw, logZ, lik, j):
    """keep the first j steps of a window: (logw, logZ) advanced (new arrays)   (smc_host_outer_advance)"""
    lik = np.ascontiguousarray(lik, dtype=np.float64)
    logw = np.array(logw, dtype=np.float64, order="C")
    logZ = np.array(logZ, dtype=np.float64, order="C")
    assert lik.shape[1] == logw.size == logZ.size and 0 <= j <= lik.shape[0]
    check(lib().smc_host_outer_advance(_d(logw), _d(logZ), _d(lik), int(j), logw.size))
    return logw, logZ


def host_outer_temper(logZ, xi, ess_min):
    """the tempering bisection (smc_samplers.jl:240-266): -> (xi_new, ess, resample_flag, logw)   (smc_host_outer_temper)"""
    logZ = np.ascontiguousarray(logZ, dtype=np.float64)
    lw = np.empty_like(logZ)
    nx, e, flag = C.c_double(), C.c_double(), C.c_int()
    check(lib().smc_host_outer_temper(_d(logZ), logZ.size, float(xi), float(ess_min), C.byref(nx), C.byref(e), C.byref(flag), _d(lw)))
    return nx.value, e.value, bool(flag.value), lw


def host_outer_resample(logw, m, seed):
    """ancestors (ascending, 0-based) of m iid draws from the weights exp(logw)   (smc_host_outer_resample)"""
    logw = np.ascontiguousarray(logw, dtype=np.float64)
    a = np.empty(int(m), dtype=np.int32)
    check(lib().smc_host_outer_resample(_d(logw), logw.size, int(m), int(seed), a.ctypes.data_as(_i32p)))
    return a


def host_rw_factor(theta):
    """(L [d][d], univariate) of random_walk_kernel(theta)   (smc_host_rw_factor)"""
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    n, d = theta.shape
    L = np.zeros((d, d))
    uni = C.c_int()
    check(lib().smc_host_rw_factor(_d(theta), n, d, _d(L), C.byref(uni)))
    return L, bool(uni.value)


def comm_unique_id():
    """the bytes rank 0 hands to the other ranks (smc_comm_unique_id)"""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    check(lib().smc_comm_unique_id(buf))
    return buf.raw


class Comm:
    """The samplers' collectives inside libsmchip.so over RCCL (smc_comm_*): what a host without torch.distributed
    binds to shard theta over the GPUs of a node.  Same interface as distributed.ThetaComm."""

    def __init__(self, unique_id, rank, world, device=0):
        self._c = C.c_void_p()
        self.rank, self.world = int(rank), int(world)
        check(lib().smc_comm_create(C.c_char_p(bytes(unique_id)), self.rank, self.world, int(device), C.byref(self._c)))

    def close(self):
        if getattr(self, "_c", None):
            lib().smc_comm_destroy(self._c)
            self._c = None

    def slice(self, M):
        if M % self.world:
            raise ValueError("n_theta (%d) must be a multiple of the number of ranks (%d)" % (M, self.world))
        per = M // self.world
        return self.rank * per, (self.rank + 1) * per

    def all_gather(self, local):
        local = np.ascontiguousarray(local, dtype=np.float64).ravel()
        out = np.zeros(local.size * self.world)
        check(lib().smc_comm_all_gather(self._c, _d(local), local.size, _d(out)))
        return out

    def outer_reweight(self, logw_local, want_w=True):
        """(logmu, w [n_local*world], ess, logw_all) of the sharded log-weights: smc_host_reweight of the concatenated vector,
        bit for bit; want_w=False: (logmu, None, ess, None) - whole-segment slices then exchange segment records only"""
        lw = np.ascontiguousarray(logw_local, dtype=np.float64).ravel()
        allw, w = (np.zeros(lw.size * self.world), np.zeros(lw.size * self.world)) if want_w else (None, None)
        lm, ess = C.c_double(), C.c_double()
        check(lib().smc_outer_reweight(self._c, _d(lw), lw.size, _d(allw), _d(w), C.byref(lm), C.byref(ess)))
        return lm.value, w, ess.value, allw

    def exchange_slots(self, h, a, M):
        a = np.ascontiguousarray(a, dtype=np.int32)
        check(lib().smc_comm_exchange_slots(self._c, h._h, a.ctypes.data_as(_i32p), int(M)))


class Handle:
    """n_theta bootstrap filters of n_x particles on one GPU (opaque smc_handle)."""

    def __init__(self, model_id, n_theta, n_x, seg=0, seed=1, device=0, flags=0):
        self._h = C.c_void_p()
        self.model_id, self.n_theta, self.n_x = model_id, int(n_theta), int(n_x)
        check(lib().smc_create(model_id, self.n_theta, self.n_x, seg, seed, device, flags, C.byref(self._h)))
        seg_, nseg, d, res = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(lib().smc_get_geometry(self._h, C.byref(seg_), C.byref(nseg), C.byref(d), C.byref(res)))
        self.seg, self.nseg, self.d, self.resident = seg_.value, nseg.value, d.value, bool(res.value)
        self.flags = flags

    def close(self):
        if getattr(self, "_h", None):
            lib().smc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, raw):
        raw = np.ascontiguousarray(raw, dtype=np.float64).reshape(self.n_theta, -1)
        assert raw.shape[1] == lib().smc_model_nraw(self.model_id)
        check(lib().smc_set_params(self._h, _d(raw)))

    def set_streams(self, streams):
        s = np.ascontiguousarray(streams, dtype=np.uint32)
        assert s.size == self.n_theta
        if getattr(self, "_streams_set", None) is not None and np.array_equal(self._streams_set, s):
            return                     # unchanged (the samplers hand over the same global indices at every round): no upload, no wait
        check(lib().smc_set_streams(self._h, s.ctypes.data_as(_u32p)))
        self._streams_set = s.copy()

    def reseed(self, seed):
        check(lib().smc_reseed(self._h, seed))

    def init(self, y1):
        lm = np.zeros(self.n_theta)
        check(lib().smc_init(self._h, float(y1), _d(lm)))
        return lm

    def step(self, y):
        lm = np.zeros(self.n_theta)
        ess = np.zeros(self.n_theta)
        check(lib().smc_step(self._h, float(y), _d(lm), _d(ess)))
        return lm, ess

    def step_window(self, y):
        """len(y) bootstrap_filter! steps in one launch, not yet kept: ([k][n_theta] logmu, [k][n_theta] ess);
        follow with step_commit(j)."""
        y = np.ascontiguousarray(y, dtype=np.float64)
        lm = np.zeros((y.size, self.n_theta))
        ess = np.zeros((y.size, self.n_theta))
        check(lib().smc_step_window(self._h, _d(y), y.size, _d(lm), _d(ess)))
        return lm, ess

    def step_commit(self, j):
        check(lib().smc_step_commit(self._h, int(j)))

    @property
    def can_window(self):
        return self.nseg == 1 and self.resident

    def log_likelihood(self, y, trace=False):
        y = np.ascontiguousarray(y, dtype=np.float64)
        logZ = np.zeros(self.n_theta)
        if trace:
            lm = np.zeros((y.size, self.n_theta))
            es = np.zeros((y.size, self.n_theta))
            check(lib().smc_log_likelihood(self._h, _d(y), y.size, _d(logZ), _d(lm), _d(es)))
            return logZ, lm, es
        check(lib().smc_log_likelihood(self._h, _d(y), y.size, _d(logZ), None, None))
        return logZ

    def state(self, want_w=True, want_anc=None):
        x = np.zeros((self.d, self.n_theta, self.n_x))
        w = np.zeros((self.n_theta, self.n_x)) if want_w else None
        if want_anc is None:
            want_anc = bool(self.flags & FLAG_ANCESTORS)
        a = np.zeros((self.n_theta, self.n_x), dtype=np.int32) if want_anc else None
        check(lib().smc_get_state(self._h, _d(x), _d(w), a.ctypes.data_as(_i32p) if a is not None else None))
        return x, w, a

    def logZ(self):
        z = np.zeros(self.n_theta)
        e = np.zeros(self.n_theta)
        check(lib().smc_get_logZ(self._h, _d(z), _d(e)))
        return z, e

    def permute(self, a):
        a = np.ascontiguousarray(a, dtype=np.int32)
        assert a.size == self.n_theta
        check(lib().smc_permute(self._h, a.ctypes.data_as(_i32p)))

    def moments(self):
        """filtered (mean, variance) of every state coordinate, [d][n_theta] each, computed on the device."""
        m = np.zeros((self.d, self.n_theta))
        v = np.zeros((self.d, self.n_theta))
        check(lib().smc_get_moments(self._h, _d(m), _d(v)))
        return m, v

    def quantiles(self, p, component=0):
        """weighted quantiles of one state coordinate under the current weights, [n_theta][len(p)], on the device."""
        p = np.ascontiguousarray(p, dtype=np.float64).ravel()
        out = np.zeros((self.n_theta, p.size))
        check(lib().smc_get_quantiles(self._h, int(component), _d(p), p.size, _d(out)))
        return out

    def set_summaries(self, p=None, component=0, moments=False):
        """per-step summaries inside the following log_likelihood / step_window calls (smc_set_summaries): weighted quantiles of
        one state coordinate at the levels p (<= 8) and / or mean and variance of every coordinate; set_summaries() switches it off"""
        p = np.ascontiguousarray([] if p is None else p, dtype=np.float64).ravel()
        check(lib().smc_set_summaries(self._h, int(component), _d(p) if p.size else None, p.size, int(bool(moments))))
        self._sum_np, self._sum_mom = int(p.size), bool(moments)

    def get_summaries(self, T):
        """(q [T][n_theta][np] or None, mean [T][d][n_theta] or None, var or None) of the first T steps of the last such call"""
        nq, mom = getattr(self, "_sum_np", 0), getattr(self, "_sum_mom", False)
        q = np.zeros((T, self.n_theta, nq)) if nq else None
        mean = np.zeros((T, self.d, self.n_theta)) if mom else None
        var = np.zeros((T, self.d, self.n_theta)) if mom else None
        check(lib().smc_get_summaries(self._h, int(T), _d(q), _d(mean), _d(var)))
        return q, mean, var

    def copy_from(self, src, mask):
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        assert m.size == self.n_theta
        check(lib().smc_copy_from(self._h, src._h, m.ctypes.data_as(C.POINTER(C.c_uint8))))

    def set_skip(self, skip):
        """filters the following log_likelihood calls leave out (logZ = -inf); None: run all again"""
        if skip is None:
            check(lib().smc_set_skip(self._h, None))
            return
        m = np.ascontiguousarray(skip, dtype=np.uint8)
        assert m.size == self.n_theta
        check(lib().smc_set_skip(self._h, m.ctypes.data_as(C.POINTER(C.c_uint8))))

    def pmmh_configure(self, families, pars, raw_from, raw_const):
        """prior components and the theta -> parameter-row map of smc_pmmh_rejuvenate (see include/smc_hip.h)"""
        fam = np.ascontiguousarray(families, dtype=np.int32)
        par = np.ascontiguousarray(pars, dtype=np.float64).reshape(fam.size, PRIOR_NPAR)
        rf = np.ascontiguousarray(raw_from, dtype=np.int32)
        rc = np.ascontiguousarray(raw_const, dtype=np.float64)
        assert rf.size == rc.size == lib().smc_model_nraw(self.model_id)
        check(lib().smc_pmmh_configure(self._h, fam.size, fam.ctypes.data_as(_i32p), _d(par), rf.ctypes.data_as(_i32p), _d(rc)))
        self._pmmh_d = int(fam.size)

    def pmmh_rejuvenate(self, main, y, xi, chol, scales, filter_seeds, move_seed, theta, logZ):
        """rejuvenate!(smc, y, xi) for this handle's parameter particles, on the device.
        Returns (theta, logZ, accepted, filters_run); theta / logZ are new arrays."""
        y = np.ascontiguousarray(y, dtype=np.float64)
        d = self._pmmh_d
        theta = np.array(theta, dtype=np.float64, order="C").reshape(self.n_theta, d)
        logZ = np.array(logZ, dtype=np.float64, order="C").reshape(self.n_theta)
        chol = np.ascontiguousarray(chol, dtype=np.float64).reshape(d, d)
        scales = np.ascontiguousarray(scales, dtype=np.float64)
        seeds = np.ascontiguousarray(filter_seeds, dtype=np.uint64)
        assert seeds.size == scales.size
        acc = np.zeros(self.n_theta, dtype=np.uint8)
        nrun = C.c_int64()
        check(lib().smc_pmmh_rejuvenate(self._h, main._h if main is not None else None, _d(y), y.size, float(xi), _d(chol),
                                        _d(scales), scales.size, seeds.ctypes.data_as(_u64p), int(move_seed), _d(theta), _d(logZ),
                                        acc.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(nrun)))
        return theta, logZ, acc.astype(bool), int(nrun.value)

    def slot_bytes(self):
        b = C.c_int64()
        check(lib().smc_slot_bytes(self._h, C.byref(b)))
        return b.value

    def pack_slots(self, idx, device_ptr):
        """idx: local slot indices; device_ptr: integer address of a device buffer of len(idx)*slot_bytes() bytes."""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        check(lib().smc_pack_slots(self._h, idx.ctypes.data_as(_i32p), idx.size, C.c_void_p(int(device_ptr))))

    def unpack_slots(self, idx, device_ptr):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        check(lib().smc_unpack_slots(self._h, idx.ctypes.data_as(_i32p), idx.size, C.c_void_p(int(device_ptr))))

    def weights_raw(self):
        npad = self.nseg * self.seg
        Cc = np.zeros((self.n_theta, npad), dtype=np.uint64)
        m = np.zeros((self.n_theta, self.nseg))
        S = np.zeros((self.n_theta, self.nseg), dtype=np.uint64)
        hi = np.zeros_like(S)
        lo = np.zeros_like(S)
        check(lib().smc_get_weights_raw(self._h, Cc.ctypes.data_as(_u64p), _d(m), S.ctypes.data_as(_u64p),
                                        hi.ctypes.data_as(_u64p), lo.ctypes.data_as(_u64p)))
        return Cc, m, S, hi, lo

    def elapsed_ms(self):
        ms = C.c_double()
        check(lib().smc_last_elapsed_ms(self._h, C.byref(ms)))
        return ms.value

    def time_step_kernel(self, y, nsample=64):
        """(avg_ms, min_ms) of one k_step launch, HIP events on the kernel's own stream."""
        y = np.ascontiguousarray(y, dtype=np.float64)
        a, m = C.c_double(), C.c_double()
        check(lib().smc_time_step_kernel(self._h, _d(y), y.size, nsample, C.byref(a), C.byref(m)))
        return a.value, m.value

    def event_overhead_ms(self, nsample=64):
        a = C.c_double()
        check(lib().smc_event_overhead_ms(self._h, nsample, C.byref(a)))
        return a.value

    def synchronize(self):
        check(lib().smc_synchronize(self._h))
