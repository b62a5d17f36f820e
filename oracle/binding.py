"""ctypes binding of the CPU oracle (oracle/smc_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under sequential_monte_carlo_amd/ may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

LG1D, SV1D, UCSV3D = 1, 2, 3

_dp = C.POINTER(C.c_double)
_u64p = C.POINTER(C.c_uint64)
_i64p = C.POINTER(C.c_int64)
_u32p = C.POINTER(C.c_uint32)


def build(force=False):
    src = os.path.join(_HERE, "smc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_exp.restype = C.c_double
        L.orc_exp.argtypes = [C.c_double]
        L.orc_log.restype = C.c_double
        L.orc_log.argtypes = [C.c_double]
        L.orc_sincos2pi.argtypes = [C.c_double, _dp, _dp]
        L.orc_box_muller.argtypes = [_u32p, _dp, _dp]
        L.orc_philox4x32_10.argtypes = [_u32p, _u32p, _u32p]
        L.orc_simulate.argtypes = [C.c_int, _dp, C.c_int, C.c_uint64, _dp, _dp]
        L.orc_normalize.argtypes = [_dp, C.c_int64, _dp, _dp, _dp]
        L.orc_resample.argtypes = [_dp, C.c_int64, C.c_int64, C.c_uint64, C.c_uint32, C.c_uint32, _i64p]
        L.orc_auto_seg.argtypes = [C.c_int, C.c_int64]
        L.orc_filter_create.restype = C.c_void_p
        L.orc_filter_create.argtypes = [C.c_int, _dp, C.c_int64, C.c_int, C.c_uint64, C.c_uint32]
        L.orc_filter_destroy.argtypes = [C.c_void_p]
        L.orc_filter_reseed.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
        L.orc_filter_set_rng.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
        L.orc_filter_set_rng.restype = None
        L.orc_filter_set_params.argtypes = [C.c_void_p, _dp]
        L.orc_filter_copy_state.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_filter_state_words.restype = C.c_int64
        L.orc_filter_state_words.argtypes = [C.c_void_p]
        L.orc_filter_export.argtypes = [C.c_void_p, _u64p]
        L.orc_filter_export.restype = None
        L.orc_filter_import.argtypes = [C.c_void_p, _u64p]
        L.orc_filter_import.restype = None
        L.orc_bootstrap_filter.restype = C.c_double
        L.orc_bootstrap_filter.argtypes = [C.c_void_p, C.c_double]
        L.orc_bootstrap_filter_step.restype = C.c_double
        L.orc_bootstrap_filter_step.argtypes = [C.c_void_p, C.c_double, _dp]
        L.orc_log_likelihood.restype = C.c_double
        L.orc_log_likelihood.argtypes = [C.c_void_p, _dp, C.c_int, _dp, _dp]
        L.orc_filter_get_state.argtypes = [C.c_void_p, _dp, _dp, _i64p, _dp]
        L.orc_filter_ess.restype = C.c_double
        L.orc_filter_ess.argtypes = [C.c_void_p]
        L.orc_filter_seg.argtypes = [C.c_void_p]
        L.orc_filter_get_weights_raw.argtypes = [C.c_void_p, _u64p, _dp, _u64p, _u64p, _u64p]
        L.orc_kalman_log_likelihood.argtypes = [_dp, _dp, C.c_int64, C.c_int, _dp]
        L.orc_kalman_log_likelihood.restype = None
        L.orc_filter_set_systematic.argtypes = [C.c_void_p, C.c_int]
        L.orc_filter_set_systematic.restype = None
        L.orc_filter_moments.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_filter_moments.restype = None
        L.orc_filter_quantiles.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int, _dp]
        L.orc_filter_quantiles.restype = C.c_int
        L.orc_log_likelihood_batch.argtypes = [C.c_int, _dp, C.c_int, C.c_int64, C.c_int, C.c_uint64,
                                               C.c_uint32, _dp, C.c_int, _dp]
        L.orc_prior_insupport.argtypes = [C.c_int, _dp, C.c_double]
        L.orc_prior_logpdf.restype = C.c_double
        L.orc_prior_logpdf.argtypes = [C.c_int, _dp, C.c_double]
        L.orc_pmmh_propose.restype = None
        L.orc_pmmh_propose.argtypes = [C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, _dp, _dp, C.c_double, _dp]
        L.orc_pmmh_log_uniform.restype = C.c_double
        L.orc_pmmh_log_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.orc_outer_reweight.argtypes = [_dp, C.c_int64, _dp, _dp, _dp]
        L.orc_outer_steps.argtypes = [_dp, _dp, _dp, C.c_int, C.c_int64, C.c_double, _dp]
        L.orc_outer_temper.argtypes = [_dp, C.c_int64, C.c_double, C.c_double, _dp, _dp, _dp]
        L.orc_outer_resample.argtypes = [_dp, C.c_int64, C.c_int64, C.c_uint64, _i64p]
        L.orc_rw_factor.argtypes = [_dp, C.c_int64, C.c_int, _dp]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


MODEL_DIM = {LG1D: 1, SV1D: 1, UCSV3D: 3}
MODEL_NRAW = {LG1D: 6, SV1D: 3, UCSV3D: 5}


def exp(x):
    L = lib()
    return np.array([L.orc_exp(float(v)) for v in np.atleast_1d(x)])


def log(x):
    L = lib()
    return np.array([L.orc_log(float(v)) for v in np.atleast_1d(x)])


def sincos2pi(u):
    L = lib()
    c, s = C.c_double(), C.c_double()
    L.orc_sincos2pi(float(u), C.byref(c), C.byref(s))
    return c.value, s.value


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return list(o)


def box_muller(words):
    w = (C.c_uint32 * 4)(*words)
    z0, z1 = C.c_double(), C.c_double()
    lib().orc_box_muller(w, C.byref(z0), C.byref(z1))
    return z0.value, z1.value


def simulate(model, raw, T, seed):
    raw = np.ascontiguousarray(raw, dtype=np.float64)
    d = MODEL_DIM[model]
    x = np.zeros((d, T))
    y = np.zeros(T)
    rc = lib().orc_simulate(model, _d(raw), T, seed, _d(x), _d(y))
    assert rc == 0
    return x, y


def normalize(logw):
    logw = np.ascontiguousarray(logw, dtype=np.float64)
    w = np.zeros_like(logw)
    lm, ess = C.c_double(), C.c_double()
    rc = lib().orc_normalize(_d(logw), logw.size, _d(w), C.byref(lm), C.byref(ess))
    assert rc == 0
    return lm.value, w, ess.value


def resample(w, ndraw=None, seed=0, stream=0, t=0):
    w = np.ascontiguousarray(w, dtype=np.float64)
    ndraw = w.size if ndraw is None else ndraw
    a = np.zeros(ndraw, dtype=np.int64)
    rc = lib().orc_resample(_d(w), w.size, ndraw, seed, stream, t, a.ctypes.data_as(_i64p))
    if rc != 0:
        raise ValueError("orc_resample rc=%d" % rc)
    return a


class Filter:
    """One bootstrap particle filter of the oracle (particles.jl:87-147)."""

    def __init__(self, model, raw, n, seg=0, seed=1, stream=0, systematic=False):
        self.model, self.n, self.d = model, int(n), MODEL_DIM[model]
        raw = np.ascontiguousarray(raw, dtype=np.float64)
        assert raw.size == MODEL_NRAW[model]
        self._h = lib().orc_filter_create(model, _d(raw), self.n, seg, seed, stream)
        if not self._h:
            raise ValueError("orc_filter_create failed")
        self.seg = lib().orc_filter_seg(self._h)
        self.nseg = (self.n + self.seg - 1) // self.seg
        if systematic:
            lib().orc_filter_set_systematic(self._h, 1)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_filter_destroy(self._h)
            self._h = None

    def reseed(self, seed, stream=0):
        lib().orc_filter_reseed(self._h, seed, stream)

    def set_rng(self, seed, stream):
        """new Philox key / stream id without restarting the series (smc_reseed / smc_set_streams mid-run)"""
        lib().orc_filter_set_rng(self._h, seed, stream)

    def set_params(self, raw):
        raw = np.ascontiguousarray(raw, dtype=np.float64)
        assert lib().orc_filter_set_params(self._h, _d(raw)) == 0

    def export_state(self):
        buf = np.zeros(lib().orc_filter_state_words(self._h), dtype=np.uint64)
        lib().orc_filter_export(self._h, buf.ctypes.data_as(_u64p))
        return buf

    def import_state(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint64)
        assert buf.size == lib().orc_filter_state_words(self._h)
        lib().orc_filter_import(self._h, buf.ctypes.data_as(_u64p))

    def copy_state_from(self, src):
        assert lib().orc_filter_copy_state(self._h, src._h) == 0

    def bootstrap_filter(self, y):
        return lib().orc_bootstrap_filter(self._h, float(y))

    def step(self, y):
        ess = C.c_double()
        lm = lib().orc_bootstrap_filter_step(self._h, float(y), C.byref(ess))
        return lm, ess.value

    def ess(self):
        return lib().orc_filter_ess(self._h)

    def log_likelihood(self, y, trace=False):
        y = np.ascontiguousarray(y, dtype=np.float64)
        if trace:
            lm = np.zeros(y.size)
            es = np.zeros(y.size)
            z = lib().orc_log_likelihood(self._h, _d(y), y.size, _d(lm), _d(es))
            return z, lm, es
        return lib().orc_log_likelihood(self._h, _d(y), y.size, None, None)

    def state(self):
        x = np.zeros((self.d, self.n))
        w = np.zeros(self.n)
        a = np.zeros(self.n, dtype=np.int64)
        logw = np.zeros(self.n)
        lib().orc_filter_get_state(self._h, _d(x), _d(w), a.ctypes.data_as(_i64p), _d(logw))
        return x, w, a, logw

    def moments(self):
        m, v = np.zeros(self.d), np.zeros(self.d)
        lib().orc_filter_moments(self._h, _d(m), _d(v))
        return m, v

    def quantiles(self, p, component=0):
        p = np.ascontiguousarray(p, dtype=np.float64)
        out = np.zeros(p.size)
        if lib().orc_filter_quantiles(self._h, int(component), _d(p), p.size, _d(out)) != 0:
            raise ValueError("bad component")
        return out

    def weights_raw(self):
        ns = self.nseg
        Cc = np.zeros(ns * self.seg, dtype=np.uint64)
        m = np.zeros(ns)
        S = np.zeros(ns, dtype=np.uint64)
        hi = np.zeros(ns, dtype=np.uint64)
        lo = np.zeros(ns, dtype=np.uint64)
        lib().orc_filter_get_weights_raw(self._h, Cc.ctypes.data_as(_u64p), _d(m), S.ctypes.data_as(_u64p),
                                         hi.ctypes.data_as(_u64p), lo.ctypes.data_as(_u64p))
        return Cc, m, S, hi, lo


def log_likelihood_batch(model, raw, n, y, seg=0, seed=1, stream0=0):
    raw = np.ascontiguousarray(raw, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    nth = raw.shape[0]
    out = np.zeros(nth)
    rc = lib().orc_log_likelihood_batch(model, _d(raw), nth, n, seg, seed, stream0, _d(y), y.size, _d(out))
    assert rc == 0
    return out


def prior_insupport(fam, par, x):
    par = np.ascontiguousarray(par, dtype=np.float64)
    return bool(lib().orc_prior_insupport(int(fam), _d(par), float(x)))


def prior_logpdf(fam, par, x):
    par = np.ascontiguousarray(par, dtype=np.float64)
    return lib().orc_prior_logpdf(int(fam), _d(par), float(x))


def pmmh_propose(seed, stream, c, theta, chol, scale):
    """theta' ~ MvNormal(theta, scale * L L') with the spec's Philox normals (smc_samplers.jl:114)"""
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    chol = np.ascontiguousarray(chol, dtype=np.float64)
    out = np.zeros(theta.size)
    lib().orc_pmmh_propose(theta.size, int(seed), int(stream), int(c), _d(theta), _d(chol), float(scale), _d(out))
    return out


def pmmh_log_uniform(seed, stream, c):
    return lib().orc_pmmh_log_uniform(int(seed), int(stream), int(c))


def kalman_log_likelihood(raw, y, predict_first=False):
    """(x_T, Sigma_T, logZ) of the scalar Kalman filter, src/kalman_filter.jl:29-70."""
    raw = np.ascontiguousarray(raw, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = np.zeros(3)
    lib().orc_kalman_log_likelihood(_d(raw), _d(y), y.size, int(predict_first), _d(out))
    return out[0], out[1], out[2]


# ---- the outer level of the samplers (orc_outer_*: reweight, window walk, tempering bisection, resample!, random-walk factor) ----
def outer_reweight(logw, want_w=True):
    """reweight(logw) -> (logmu, w, ess)   smc_samplers.jl:232,249,265,298,338"""
    logw = np.ascontiguousarray(logw, dtype=np.float64)
    w = np.zeros_like(logw) if want_w else None
    lm, ess = C.c_double(), C.c_double()
    assert lib().orc_outer_reweight(_d(logw), logw.size, _d(w) if want_w else None, C.byref(lm), C.byref(ess)) == 0
    return lm.value, w, ess.value


def outer_steps(logw, logZ, lik, ess_min):
    """up to k smc²! host halves (smc_samplers.jl:323-338): -> (logw, logZ, ess [j], j), stopping below ess_min"""
    lik = np.ascontiguousarray(lik, dtype=np.float64)
    k, n = lik.shape
    logw = np.array(logw, dtype=np.float64, order="C")
    logZ = np.array(logZ, dtype=np.float64, order="C")
    ess = np.zeros(k)
    j = lib().orc_outer_steps(_d(logw), _d(logZ), _d(lik), k, n, float(ess_min), _d(ess))
    return logw, logZ, ess[:j], j


def outer_temper(logZ, xi, ess_min):
    """the bisection of density_tempered (smc_samplers.jl:240-266): -> (xi_new, ess, resample_flag, logw)"""
    logZ = np.ascontiguousarray(logZ, dtype=np.float64)
    lw = np.zeros_like(logZ)
    nx, e = C.c_double(), C.c_double()
    flag = lib().orc_outer_temper(_d(logZ), logZ.size, float(xi), float(ess_min), C.byref(nx), C.byref(e), _d(lw))
    return nx.value, e.value, bool(flag), lw


def outer_resample(logw, m, seed):
    """a = resample(omega) of resample!(smc) (smc_samplers.jl:74-84), ascending, 0-based"""
    logw = np.ascontiguousarray(logw, dtype=np.float64)
    a = np.zeros(int(m), dtype=np.int64)
    assert lib().orc_outer_resample(_d(logw), logw.size, int(m), int(seed), a.ctypes.data_as(_i64p)) == 0
    return a


def rw_factor(theta):
    """random_walk_kernel(theta) (smc_samplers.jl:87-101): -> (L [d][d], univariate)"""
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    n, d = theta.shape
    L = np.zeros((d, d))
    uni = lib().orc_rw_factor(_d(theta), n, d, _d(L))
    return L, bool(uni)
