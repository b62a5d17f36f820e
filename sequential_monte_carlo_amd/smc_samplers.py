"""Host mirror of src/smc_samplers.jl: SMC, density_tempered, smc2 ("smc²"), smc2_step ("smc²!"), smc2_run (the
`for t in 2:T smc²!(smc,y,t) end` loop, several steps per device call), rejuvenate_, resample_, exchange!, and the
summaries of plotting_utils.jl / the example script (estimated_trend, filtered_summaries).

The O(n_theta) outer logic (bisection for the tempering exponent, resample!, the random-walk kernel) stays on the host as in
the reference, but its arithmetic is ONE specification shared by every host (csrc/smc_outer.hip, include/smc_hip.h "the OUTER
level"): `reweight` is the inner filter's integer normalize over segments of 8 parameter particles, so its sums do not depend
on the order of evaluation or on how theta is dealt out to ranks; the sampler carries the un-normalised outer log-weights
`logw` (`omega` = their normalisation, on demand); resample! draws its indices from Philox through the integer weight CDF.
Every particle filter the reference runs inside `Threads.@threads for m in 1:M` (smc_samplers.jl:112-121,174-180,223-229) or
serially (:289-295,:325-335) becomes ONE batched call on the GPU (theta axis = workgroups).

PMMH rejuvenation (rejuvenate!, smc_samplers.jl:103-146) runs entirely on the device when the sampler is given
what a GPU can evaluate instead of closures: `theta_map=ThetaMap(...)` for smc.model(theta) and a prior made of the
enumerated families (distributions.py spec()).  Then one smc_pmmh_rejuvenate call does the whole chain loop for
the rank's parameter particles - proposals, insupport, prior ratio, the proposal filters, the accept test, and the
overwrite of theta / logZ / x / w - with Philox draws keyed by the GLOBAL theta index, and no collective inside.
With arbitrary Python closures / priors the same loop stays on the host (numpy random numbers) and only the filters
are batched on the GPU.  Either way a proposal outside the prior's support is never filtered (:116).

theta sharding (multi-GPU): pass `comm=ThetaComm(...)` (distributed.py).  A rank *filters* its own contiguous slice of theta
and does the outer arithmetic of that slice; what every rank must agree on (the ESS of a step, the ancestors of a resample!,
the random-walk factor) is computed from exchanged integers or from the same counter-based random numbers, so all ranks take
identical decisions.  Collectives: one all-gather of the logZ slices per batched evaluation, one all-gather of the
moved (theta, logZ, accepted) slices per rejuvenation, one all-gather of the SEGMENT RECORDS (32 bytes per 8 parameter
particles and step) per window of online steps, one all-gather of the (logw, logZ) slices and one all-to-all of filter slots
per resample! of the online sampler.  Results do not depend on the number of ranks: filter m always uses Philox stream id m,
and the outer sums are integer sums over fixed segments.
"""
import math
import sys

import numpy as np

from . import _lib
from .models import params_matrix


class RawModels:
    """A batch of models of one family given directly as parameter rows (see SMC(raw_fn=...))."""

    def __init__(self, model_id, raw):
        self.model_id, self.rows = int(model_id), np.ascontiguousarray(raw, dtype=np.float64)


def _rows(models):
    return (models.model_id, models.rows) if isinstance(models, RawModels) else params_matrix(models)


class ThetaMap:
    """smc.model(theta) in a form the device can evaluate: parameter row k of the model family `model_id` is
    theta[raw_from[k]] when raw_from[k] >= 0, else the constant raw_const[k].
    README.md:75-79 (LinearGaussian(theta1, 1.0, theta2, theta3, 0.0)):  ThetaMap(LG, [0,-1,1,2,-1,-1], [0,1,0,0,0,1]);
    examples/inflation_example.jl:227-230 (UCSV(theta1, theta2, (theta3, theta4))):  ThetaMap(UCSV, [0,0,1,2,3], [0]*5)."""

    def __init__(self, model_id, raw_from, raw_const):
        self.model_id = int(model_id)
        self.raw_from = np.ascontiguousarray(raw_from, dtype=np.int32)
        self.raw_const = np.ascontiguousarray(raw_const, dtype=np.float64)
        if self.raw_from.size != self.raw_const.size:
            raise ValueError("raw_from and raw_const must have one entry per parameter of the model row")

    def rows(self, thetas):
        """[m, n_raw] parameter rows of the m parameter vectors in `thetas`"""
        thetas = np.atleast_2d(np.asarray(thetas, dtype=np.float64))
        out = np.tile(self.raw_const, (thetas.shape[0], 1))
        use = self.raw_from >= 0
        out[:, use] = thetas[:, self.raw_from[use]]
        return out


# ---- filter backends ---------------------------------------------------------------------------
class LibOuter:
    """The outer level (reweight, window walk, tempering bisection, resample!, random-walk factor) by the library's host
    routines (csrc/smc_outer.hip).  `window_walk` is where theta sharding shows: a rank whose slice consists of whole
    segments computes the segment records of its own parameter particles and the ranks exchange records; otherwise the
    log-likelihood increments are exchanged and every rank walks the whole vector - the same bits either way."""

    reweight = staticmethod(_lib.host_reweight)
    temper = staticmethod(_lib.host_outer_temper)
    resample = staticmethod(_lib.host_outer_resample)
    rw_factor = staticmethod(_lib.host_rw_factor)

    def window_walk(self, smc, lik_local, ess_min):
        """the host half of k = len(lik_local) smc²! steps (smc_samplers.jl:323-338): advances smc.logw / smc.logZ by the
        steps walked and returns (ess [j], j)"""
        lik_local = np.ascontiguousarray(lik_local, dtype=np.float64)
        k, per = lik_local.shape
        lo, hi = smc.lo, smc.hi
        if smc.comm is None or (per % _lib.OUTER_SEG == 0):
            rec = _lib.host_outer_window(smc.logw[lo:hi], lik_local)                  # [k][nseg_local][4]
            if smc.comm is not None:
                nsl = rec.shape[1]
                allr = smc.comm.all_gather(rec.view(np.float64).ravel()).view(np.uint64)   # 8-byte words, no arithmetic on the way
                rec = np.ascontiguousarray(allr.reshape(-1, k, nsl, 4).transpose(1, 0, 2, 3).reshape(k, -1, 4))
            ess, j = _lib.host_outer_walk(rec, smc.M, ess_min)
            smc.logw[lo:hi], smc.logZ[lo:hi] = _lib.host_outer_advance(smc.logw[lo:hi], smc.logZ[lo:hi], lik_local, j)
            smc._outer_local = smc.comm is not None          # the other ranks' slices are stale until _sync_outer
        else:
            smc._sync_outer()
            lik = smc._gather(lik_local.ravel()).reshape(-1, k, per).transpose(1, 0, 2).reshape(k, smc.M)
            rec = _lib.host_outer_window(smc.logw, lik)
            ess, j = _lib.host_outer_walk(rec, smc.M, ess_min)
            smc.logw, smc.logZ = _lib.host_outer_advance(smc.logw, smc.logZ, lik, j)
        return ess, j


class HipBackend:
    """Runs the batched inner filters on one GPU through the C ABI (no CPU fallback)."""

    outer = LibOuter()

    def __init__(self, device=0, seg=0, resampler="multinomial"):
        self.device, self.seg = device, seg
        self.flags = _lib.FLAG_SYSTEMATIC if resampler == "systematic" else 0   # opt-in; default = the reference's law
        self._handles = {}

    def _handle(self, key, model_id, n_theta, N, seed):
        k = (key, model_id, n_theta, N)
        h = self._handles.get(k)
        if h is None:
            h = _lib.Handle(model_id, n_theta, N, seg=self.seg, seed=seed, device=self.device, flags=self.flags)
            self._handles[k] = h
        h.reseed(seed)
        return h

    def init(self, models, N, y1, seed, streams, key="main"):
        mid, raw = _rows(models)
        h = self._handle(key, mid, raw.shape[0], N, seed)
        h.set_params(raw)
        h.set_streams(streams)
        return h.init(y1), h

    def log_likelihood(self, models, N, y, seed, streams, key="prop", skip=None):
        """skip[m] = True: filter m is not run (proposal outside the prior's support, smc_samplers.jl:116); logZ = -inf"""
        mid, raw = _rows(models)
        h = self._handle(key, mid, raw.shape[0], N, seed)
        h.set_params(raw)
        h.set_streams(streams)
        h.set_skip(skip if skip is not None and np.any(skip) else None)
        try:
            out = h.log_likelihood(y)
        finally:
            h.set_skip(None)     # whatever happened: the mask must not outlive this evaluation on the cached handle
        return out, h

    def rejuvenate(self, tmap, prior_spec, N, y, xi, chol, scales, filter_seeds, move_seed, streams, theta, logZ, main):
        """rejuvenate! for this rank's parameter particles on the device (smc_pmmh_rejuvenate).
        -> (theta, logZ, accepted, filters_run)"""
        h = self._handle("prop", tmap.model_id, theta.shape[0], N, int(filter_seeds[0]))
        h.set_streams(streams)
        cfg = (tmap.raw_from.tobytes(), tmap.raw_const.tobytes(), np.asarray(prior_spec[0]).tobytes(), np.asarray(prior_spec[1]).tobytes())   # by content
        if getattr(h, "_pmmh_cfg", None) != cfg:
            h.pmmh_configure(prior_spec[0], prior_spec[1], tmap.raw_from, tmap.raw_const)
            h._pmmh_cfg = cfg
        return h.pmmh_rejuvenate(main, y, xi, chol, scales, filter_seeds, move_seed, theta, logZ)

    def release(self, h):
        """close one handle (a superseded set of online filters, exchange!)"""
        for k, v in list(self._handles.items()):
            if v is h:
                v.close()
                del self._handles[k]

    def close(self):
        for h in self._handles.values():
            h.close()
        self._handles = {}


class SMC:
    """SMC(N, M, model, prior, chain, ess_threshold, min_ar=-1.0)   smc_samplers.jl:5-59
    N state particles per filter, M parameter particles.  `model` maps a parameter vector to a
    StateSpaceModel (the reference's closure smc.model(theta))."""

    def __init__(self, N, M, model, prior, chain, ess_threshold, min_ar=-1.0, seed=1, backend=None, comm=None,
                 raw_fn=None, theta_map=None):
        """theta_map (optional): ThetaMap, the device-evaluable form of `model`; with a prior of enumerated families it
        moves rejuvenate! onto the device.  raw_fn (optional): vectorised shortcut for `model`: maps an [m, d_theta]
        array to (model_id, [m, n_raw] parameter rows) without building m model objects per evaluation."""
        self.N, self.M, self.model, self.prior, self.chain = int(N), int(M), model, prior, int(chain)
        self.theta_map = theta_map
        if raw_fn is None and theta_map is not None:
            def raw_fn(th, _t=theta_map):
                return _t.model_id, _t.rows(th)
        self.raw_fn = raw_fn
        self.rng = np.random.default_rng(seed)
        self.seed = int(seed)
        if hasattr(prior, "rand_many"):
            self.theta = np.ascontiguousarray(prior.rand_many(self.rng, self.M), dtype=np.float64)
        else:
            self.theta = np.array([np.atleast_1d(prior.rand(self.rng)) for _ in range(self.M)], dtype=np.float64)
        self.logw = np.zeros(self.M)        # un-normalised outer log-weights; omega = their normalisation (property)
        self.logZ = np.zeros(self.M)
        self.ess = float(self.M)
        self.ess_min = self.M * float(ess_threshold)
        self.acc_threshold, self.acc_ratio = float(min_ar), 0.0
        self.backend = backend if backend is not None else HipBackend()
        self.outer = getattr(self.backend, "outer", None) or LibOuter()
        self._outer_local = False           # True: only this rank's slice of logw / logZ is current (sharded online steps)
        self.prior_spec = prior.spec() if hasattr(prior, "spec") else None
        self.device_pmmh = (theta_map is not None and self.prior_spec is not None and hasattr(self.backend, "rejuvenate")
                            and self.theta.shape[1] == len(self.prior_spec[0]))
        self.comm = comm
        self.lo, self.hi = (0, self.M) if comm is None else comm.slice(self.M)
        self._calls = 0          # evaluation counter -> fresh Philox seed per batched evaluation
        self.psteps = 0          # executed inner particle-steps (all ranks), SURVEY 8(d)
        self.psteps_skipped = 0  # particle-steps of proposals outside the prior's support: never run (smc_samplers.jl:116)
        self.psteps_speculated = 0   # particle-steps the windows of smc2_run ran beyond the kept ones (dropped steps + the re-run prefix)
        self._main = None        # device handle of the online filters (smc2)
        self._theta_dev = None   # (handle, theta slice) whose parameter rows are on the device
        self.t = 0

    # -- batched inner filters --------------------------------------------------------------------
    def _next_seed(self):
        self._calls += 1
        return (self.seed << 20) + self._calls

    def _streams(self):
        return np.arange(self.lo, self.hi, dtype=np.uint32)

    def _gather(self, local):
        return local if self.comm is None else self.comm.all_gather(local)

    def _sync_outer(self):
        """after sharded online steps every rank has advanced only its own slice of logw / logZ: ONE all-gather brings the
        full vectors up to date everywhere (before resample!, summaries, and when a public call returns)"""
        if self._outer_local:
            per = self.hi - self.lo
            both = self._gather(np.concatenate([self.logw[self.lo:self.hi], self.logZ[self.lo:self.hi]])).reshape(-1, 2, per)
            self.logw = np.ascontiguousarray(both[:, 0, :].ravel())
            self.logZ = np.ascontiguousarray(both[:, 1, :].ravel())
            self._outer_local = False

    def _set_logw(self, logw):
        """reweight(logw) (smc_samplers.jl:232,298): the sampler's outer weights become exp(logw), normalised"""
        self.logw = np.array(logw, dtype=np.float64)
        self._outer_local = False
        _, _, self.ess = self.outer.reweight(self.logw, want_w=False)

    @property
    def omega(self):
        """the normalised outer weights (the reference's smc.ω after reweight): reweight(logw)[1]"""
        self._sync_outer()
        return self.outer.reweight(self.logw)[1]

    def _models(self, thetas):
        """smc.model(theta[m]) for a block of parameter particles (objects, or parameter rows via raw_fn)."""
        if self.raw_fn is not None:
            return RawModels(*self.raw_fn(np.asarray(thetas, dtype=np.float64)))
        return [self.model(th) for th in thetas]

    def _filter_all(self, thetas, y, key="prop", skip=None):
        """logZ[m] = log_likelihood(N, y, model(theta[m])) for every m: ONE batched GPU call per rank.
        skip[m]: that filter is not run (its logZ reads -inf) and not counted."""
        models = self._models(thetas[self.lo:self.hi])
        sk = None if skip is None else np.asarray(skip[self.lo:self.hi], dtype=bool)
        local, h = self.backend.log_likelihood(models, self.N, np.asarray(y, dtype=np.float64), self._next_seed(),
                                               self._streams(), key=key, skip=sk)
        nskip = 0 if skip is None else int(np.sum(skip))
        self.psteps += (self.M - nskip) * self.N * len(y)
        self.psteps_skipped += nskip * self.N * len(y)
        return self._gather(np.asarray(local)), h

    def __repr__(self):
        return "ess     = %.3f\nmean(theta) = %s" % (self.ess, np.array2string(expected_parameters(self)))


def expected_parameters(smc):
    """sum_m theta[m] * omega[m]   (smc_samplers.jl:61-65; omega normalised)"""
    w = smc.omega
    return (smc.theta * w[:, None]).sum(axis=0)


def _per_theta(smc, local):
    """[M_local, k] rows of per-filter summaries of this rank -> [M, k] on every rank"""
    local = np.ascontiguousarray(local, dtype=np.float64)
    if smc.comm is None:
        return local
    k = local.shape[1]
    return smc._gather(local.ravel()).reshape(smc.M, k)


def filtered_summaries(smc, p=(0.25, 0.5, 0.75), component=0):
    """(quantiles [len(p)], variance) of the filtered state of the online sampler, integrated over the parameter particles:
    per theta-particle the weighted quantiles and variance of its x cloud, then their omega-weighted means - what
    get_quantiles_uc computes every period in examples/inflation_example.jl:39-55.
    The per-filter summaries are computed on the device (smc_get_quantiles / smc_get_moments): no cloud leaves the GPU.
    DEVIATIONS from the example, on purpose (parity unpinned: no reference fixture covers these numbers): (1) the example's
    UCSV variant (:244-248) takes UNWEIGHTED quantile(x_cloud) / var(x_cloud) of the clouds and multiplies by the raw smc.ω;
    here both variants use the w-weighted summaries of the UC variant and the normalised omega (right after rejuvenate! the
    reference's ω is all ones: its sum is then M times the mean).  (2) The quantile is the inverse of the weighted empirical
    CDF in the filter's integer weights (smc_get_quantiles), not StatsBase's interpolating definition (not vendored)."""
    if smc._main is None:
        raise ValueError("filtered_summaries needs the online sampler's filters (call smc2 first)")
    q = np.asarray(smc._main.quantiles(list(p), component))            # [M_local][len(p)]
    _, var = smc._main.moments()                                       # [d][M_local]
    return _integrate(smc.omega, _per_theta(smc, np.column_stack([q, np.asarray(var)[component]])))


def estimated_trend(smc):
    """estimated_trend(smc)   src/plotting_utils.jl:116-124:  sum_m omega[m] * mean(observation(model(theta[m]), w[m]' x[m])),
    with the NORMALISED omega (the reference uses smc.ω as it stands - all ones right after rejuvenate!; stated deviation).
    The filtered means w[m]' x[m] come from the device; mean(observation(.)) is B x for the linear model (ssm.jl:96-103),
    x[1] for UCSV (:244-247) and 0 for the stochastic-volatility model."""
    if smc._main is None:
        raise ValueError("estimated_trend needs the online sampler's filters (call smc2 first)")
    mean, _ = smc._main.moments()                                      # [d][M_local]
    mid, rows = _rows(smc._models(smc.theta[smc.lo:smc.hi]))
    if mid == _lib.MODEL_LG1D:
        obs = rows[:, 1] * np.asarray(mean)[0]
    elif mid == _lib.MODEL_UCSV3D:
        obs = np.asarray(mean)[0]
    else:
        obs = np.zeros(rows.shape[0])
    allobs = _per_theta(smc, obs[:, None])[:, 0]
    w = smc.omega
    return float(w @ allobs)


def resample_(smc):
    """resample!(smc)   smc_samplers.jl:74-84 -- value-copy semantics."""
    smc._sync_outer()
    # iid multinomial (sample(1:M, Weights(w), M)) through the integer CDF of the outer weights, pick numbers from Philox keyed
    # by the evaluation counter (the same on every rank).  The order of the resampled population carries no information (the
    # reference's `sample` returns it unsorted); taken in ascending order slot m inherits from an ancestor close to m, so with
    # theta sharded over GPUs most of the filter copies of the online sampler stay on their rank and only the drift of the
    # offspring counts crosses the links.
    a = np.asarray(smc.outer.resample(smc.logw, smc.M, smc._next_seed()), dtype=np.int64)
    smc.theta = smc.theta[a].copy()
    smc.logw = smc.logw[a].copy()
    smc.logZ = smc.logZ[a].copy()
    if smc._main is not None:
        if smc.comm is not None:
            smc.comm.exchange_slots(smc._main, a, smc.M)      # filters move between GPUs: one all-to-all
        else:
            smc._main.permute(a.astype(np.int32))
    return a


def random_walk_factor(theta, scales, outer=None):
    """(L, s) such that the proposal of chain position c is  theta' = theta + sqrt(s[c]) * L z,  z ~ N(0, I):
    multivariate theta (smc_samplers.jl:95-100): MvNormal(x, scale * Sigma), Sigma = 2.83^2/d * cov(theta) + 1e-10 I
      (1e-2 I when the cloud has collapsed, norm(cov) < 1e-8)  ->  L = chol(Sigma), s = scales;
    univariate theta (:87-92): Normal(x, scale * sigma) with sigma = 2.83^2 * var(theta) + 1e-10 (1e-2 when collapsed)
      handed over as the STANDARD DEVIATION  ->  L = [[sigma]], s = scales^2  (sqrt(s) = scale).
    Covariance and factorisation in the library's fixed order of operations (smc_host_rw_factor): the same bits on every rank."""
    scales = np.asarray(scales, dtype=np.float64)
    L, univariate = (outer or LibOuter).rw_factor(np.ascontiguousarray(theta, dtype=np.float64))
    return L, (scales * scales if univariate else scales)


def random_walk_kernel(theta, outer=None):
    """random_walk_kernel(theta)   smc_samplers.jl:87-101
    returns f(x, scale, rng) -> a draw of the reference's Normal(x, scale*sigma) / MvNormal(x, scale*Sigma)."""
    d = theta.shape[1]
    L, _ = random_walk_factor(theta, [1.0], outer)     # fixed now: theta is updated in place during the chain

    def eff(scale):
        return scale * scale if d == 1 else scale

    def kernel(x, scale, rng):
        return x + math.sqrt(eff(scale)) * (L @ rng.standard_normal(d))

    def many(th, scale, rng):
        """one proposal per row of th: the same normals, in the same order, as M calls of kernel()"""
        z = rng.standard_normal((th.shape[0], d))
        return th + math.sqrt(eff(scale)) * (z @ L.T)

    kernel.many = many
    return kernel


def rejuvenate_(smc, y, xi=1.0, verbose=False, out=sys.stdout):
    """rejuvenate!(smc, y, xi)   smc_samplers.jl:103-146 -- PMMH moves, `chain` per parameter particle.
    On the device in one call per rank when the sampler has a ThetaMap and an enumerated prior; otherwise the
    loop below with all M proposals of one chain position filtered in one batched call."""
    y = np.asarray(y, dtype=np.float64)
    if verbose:
        out.write("\t[rejuvenating]")
    if smc.device_pmmh:
        accepted = _rejuvenate_device(smc, y, xi)
    else:
        accepted = _rejuvenate_host(smc, y, xi)
    smc.logw = np.zeros(smc.M)          # smc.ω = ones(M): the moved particles are equally weighted
    smc._outer_local = False
    smc.acc_ratio = float(accepted.sum()) / smc.M
    if verbose:
        out.write("\tacc_rate: %1.5f" % smc.acc_ratio)
    return smc


def _rejuvenate_device(smc, y, xi):
    """The whole `for m ... for c in 1:chain` loop (smc_samplers.jl:112-138) in one smc_pmmh_rejuvenate call per rank;
    the host contributes the random-walk factor (:95-100, from the full theta cloud every rank holds) and, with
    sharded theta, ONE all-gather of the moved (theta, logZ, accepted) slices afterwards."""
    L, s = random_walk_factor(smc.theta, 0.5 * np.arange(smc.chain, 0, -1), smc.outer)     # 0.5*reverse(1:chain)
    seeds = np.array([smc._next_seed() for _ in range(smc.chain)], dtype=np.uint64)
    move_seed = smc._next_seed()
    lo, hi = smc.lo, smc.hi
    th, lz, acc, nrun = smc.backend.rejuvenate(smc.theta_map, smc.prior_spec, smc.N, y, float(xi), L, s, seeds, move_seed,
                                               smc._streams(), smc.theta[lo:hi], smc.logZ[lo:hi], smc._main)
    d = smc.theta.shape[1]
    packed = np.concatenate([np.asarray(th, dtype=np.float64).ravel(), lz, acc.astype(np.float64), [float(nrun)]])
    allp = smc._gather(packed).reshape(-1, packed.size)
    per = hi - lo
    smc.theta = np.ascontiguousarray(allp[:, :per * d].reshape(-1, d))
    smc.logZ = np.ascontiguousarray(allp[:, per * d:per * (d + 1)].ravel())
    accepted = allp[:, per * (d + 1):per * (d + 2)].ravel() != 0.0
    nrun_all = int(round(allp[:, -1].sum()))
    smc.psteps += nrun_all * smc.N * len(y)
    smc.psteps_skipped += (smc.M * smc.chain - nrun_all) * smc.N * len(y)
    return accepted


def _rejuvenate_host(smc, y, xi):
    kernel = random_walk_kernel(smc.theta, smc.outer)
    scales = 0.5 * np.arange(smc.chain, 0, -1)          # 0.5*reverse(1:chain)
    accepted = np.zeros(smc.M, dtype=bool)
    many = hasattr(smc.prior, "logpdf_many")
    for c in range(smc.chain):
        prop = kernel.many(smc.theta, scales[c], smc.rng)      # all M proposals of this chain position
        u = smc.rng.random(smc.M)
        if many:
            ok = smc.prior.insupport_many(prop)
            lp_prop, lp_cur = smc.prior.logpdf_many(np.where(ok[:, None], prop, smc.theta)), smc.prior.logpdf_many(smc.theta)
        else:
            ok = np.array([smc.prior.insupport(p) for p in prop])
            lp_prop = np.array([smc.prior.logpdf(p) if o else -math.inf for p, o in zip(prop, ok)])
            lp_cur = np.array([smc.prior.logpdf(q) for q in smc.theta])
        # a proposal outside the support is never filtered (smc_samplers.jl:116): its slot is skipped on the device
        # (the parameter row handed over for it is the current, valid theta; it is not read)
        safe = np.where(ok[:, None], prop, smc.theta)
        logZ_prop, hprop = smc._filter_all(safe, y, skip=~ok)
        with np.errstate(invalid="ignore"):
            acc_ratio = xi * (logZ_prop - smc.logZ) + (lp_prop - lp_cur)
        with np.errstate(divide="ignore"):
            acc = ok & (logZ_prop + lp_prop > -math.inf) & (np.log(u) < acc_ratio)
        smc.theta[acc] = prop[acc]
        smc.logZ[acc] = logZ_prop[acc]
        if smc._main is not None and acc[smc.lo:smc.hi].any():
            smc._main.copy_from(hprop, acc[smc.lo:smc.hi])      # x[m], w[m] <- x_prop, w_prop on the device
        accepted |= acc
    return accepted


def density_tempered(smc, y, verbose=True, out=sys.stdout):
    """density_tempered(smc, y)   smc_samplers.jl:222-281 (Duan & Fulop)."""
    y = np.asarray(y, dtype=np.float64)
    smc.logZ, _ = smc._filter_all(smc.theta, y)
    smc._set_logw(smc.logZ)                               # :232
    xi = 0.0
    stages = []
    while xi < 1.0:
        # the bisection for the next exponent (:240-258) and the corner solution (:261-266), one library call:
        # ess == ess_min ends it, ess < ess_min lowers the upper end, else the lower end rises, until they are 1e-6 apart
        xi, smc.ess, resample_flag, smc.logw = smc.outer.temper(smc.logZ, xi, smc.ess_min)
        if verbose:
            out.write("ξ = %1.5f\tess = %4.3f" % (xi, smc.ess))
        if resample_flag:
            resample_(smc)
            rejuvenate_(smc, y, xi, verbose, out)
        stages.append((xi, smc.ess, smc.acc_ratio if resample_flag else None))
        if verbose:
            out.write("\n")
    return stages


def smc2(smc, y):
    """smc²(smc, y): initialisation at t = 1   smc_samplers.jl:288-301"""
    y = np.asarray(y, dtype=np.float64)
    models = smc._models(smc.theta[smc.lo:smc.hi])
    logmu, smc._main = smc.backend.init(models, smc.N, float(y[0]), smc._next_seed(), smc._streams(), key="main")
    smc.psteps += smc.M * smc.N
    smc.logZ = smc._gather(np.asarray(logmu, dtype=np.float64)).copy()
    smc._set_logw(smc.logZ)                               # :297-298
    smc.t = 1
    return smc


def smc2_step(smc, y, t, verbose=True, out=sys.stdout):
    """smc²!(smc, y, t): online step for observation y[t] (1-based t as in the reference, t >= 2)
    smc_samplers.jl:308-340"""
    y = np.asarray(y, dtype=np.float64)
    if verbose:
        out.write("t = %4d\tess = %4.3f" % (t - 1, smc.ess))
    if smc.ess < smc.ess_min:
        resample_(smc)
        rejuvenate_(smc, y[: t - 1], 1.0, verbose, out)
        _exchange(smc, y[: t - 1], verbose, out)
    _step_only(smc, y, t, verbose, out)
    smc._sync_outer()
    return smc


def _integrate(w, rows):
    """[M][np + 1] per-filter (quantiles | variance) rows of every rank -> their omega-weighted means (quantiles [np], variance);
    the products summed over the parameter particles in index order (one definition for every caller: no BLAS in between)"""
    tot = np.add.reduce(w[:, None] * np.ascontiguousarray(rows, dtype=np.float64), axis=0)
    return tot[:-1], float(tot[-1])


def _window_summaries(smc, t, lik_local, j, logw_local):
    """the integrated filtered summaries (filtered_summaries) after each of the j kept steps of a window, from the per-step
    per-filter summaries the window launch recorded on the device: [(t + i, quantiles [np], variance)] - what the loop
    `smc²!(smc, y, t); push!(..., get_quantiles_uc(smc))` of examples/inflation_example.jl:39-55,78-86 collects per period.
    logw_local: this rank's outer log-weights BEFORE the window (the weights of step i follow by the window's own additions)."""
    comp = smc._summ["component"]
    q, _, var = smc._main.get_summaries(j)                      # [j][M_local][np], [j][d][M_local]
    nq, per = q.shape[2], q.shape[1]
    lw = np.array(logw_local, dtype=np.float64)
    blocks = []
    for i in range(j):
        lw = lw + lik_local[i]                                   # == smc_host_outer_advance, step by step
        blocks.append(np.concatenate([lw, q[i].ravel(), var[i][comp]]))
    flat = np.concatenate(blocks)
    allf = smc._gather(flat).reshape(-1, j, per * (nq + 2))      # [ranks][j][...]
    rows = []
    for i in range(j):
        logw = np.concatenate([allf[r, i, :per] for r in range(allf.shape[0])])
        qq = np.concatenate([allf[r, i, per:per * (nq + 1)].reshape(per, nq) for r in range(allf.shape[0])])
        vv = np.concatenate([allf[r, i, per * (nq + 1):] for r in range(allf.shape[0])])
        qi, vi = _integrate(smc.outer.reweight(logw)[1], np.column_stack([qq, vv]))
        rows.append((t + i, qi, vi))
    return rows


def smc2_run(smc, y, t_from, t_to, window=16, verbose=True, out=sys.stdout, summaries=None, component=0):
    """for t in t_from:t_to  smc²!(smc, y, t)  end   (the online loop of smc_samplers.jl:308-340 / README.md:93-101),
    with the same results bit for bit, but up to `window` propagation steps per device call: between two
    resample-move decisions the inner filters only need y[t], so a window of steps runs in ONE launch with the particle
    clouds resident in LDS (smc_step_window), the host then walks through the window's outer ESS values exactly as
    smc²! would, and keeps the steps up to (and including) the first one whose ESS falls below the threshold
    (smc_step_commit; the speculated steps behind it are dropped and redone after the resample-move).
    With sharded theta a window costs ONE all-gather of segment records (LibOuter.window_walk) instead of one exchange per step.
    summaries=[p...] (optional): additionally collect, after every step, what the example's loop collects per period
    (examples/inflation_example.jl:78-86: get_quantiles_uc(smc) after each smc²!) - filtered_summaries(smc, p, component) -
    from per-step summaries recorded on the device inside the window launches; returned as smc.summary_trace =
    [(t, quantiles [len(p)], variance)], bit-identical to calling filtered_summaries after every smc2_step."""
    y = np.asarray(y, dtype=np.float64)
    t = int(t_from)
    smc._summ = None if summaries is None else {"p": [float(v) for v in summaries], "component": int(component)}
    smc.summary_trace = []
    while t <= t_to:
        k = min(int(window), t_to - t + 1)
        if k <= 1 or not getattr(smc._main, "can_window", False):
            smc2_step(smc, y, t, verbose, out)
            if smc._summ:
                qq, vv = filtered_summaries(smc, smc._summ["p"], smc._summ["component"])
                smc.summary_trace.append((t, qq, vv))
            t += 1
            continue
        if verbose:
            out.write("t = %4d\tess = %4.3f" % (t - 1, smc.ess))
        if smc.ess < smc.ess_min:
            resample_(smc)
            rejuvenate_(smc, y[: t - 1], 1.0, verbose, out)
            _exchange(smc, y[: t - 1], verbose, out)
            if not getattr(smc._main, "can_window", False):     # exchange! may have outgrown the resident kernel
                _step_only(smc, y, t, verbose, out)
                if smc._summ:
                    smc._sync_outer()
                    qq, vv = filtered_summaries(smc, smc._summ["p"], smc._summ["component"])
                    smc.summary_trace.append((t, qq, vv))
                t += 1
                continue
        _sync_params(smc)
        if smc._summ:
            smc._main.set_summaries(smc._summ["p"], smc._summ["component"], moments=True)
        try:
            lik, _ = smc._main.step_window(y[t - 1: t - 1 + k])          # [k][M_local]
            lik = np.asarray(lik, dtype=np.float64)
            logw_before = smc.logw[smc.lo:smc.hi].copy()
            ess, j = smc.outer.window_walk(smc, lik, smc.ess_min)
            if smc._summ:
                smc.summary_trace.extend(_window_summaries(smc, t, lik, j, logw_before))
        finally:
            if smc._summ:
                smc._main.set_summaries()                                # (the traces are read: recording off again)
        smc.ess = float(ess[-1])
        smc.t = t + j - 1
        if verbose:
            out.write("\n" + "".join("t = %4d\tess = %4.3f\n" % (t + i, ess[i]) for i in range(j - 1)))
        smc._main.step_commit(j)
        smc.psteps += smc.M * smc.N * j
        smc.psteps_speculated += smc.M * smc.N * ((k - j) + (j if j < k else 0))   # dropped steps + the prefix smc_step_commit re-runs
        t += j
    smc._sync_outer()
    return smc


def _sync_params(smc):
    """the parameter rows on the device follow theta (resample!/rejuvenate! change it); unchanged -> no upload"""
    th_loc = smc.theta[smc.lo:smc.hi]
    if smc._theta_dev is None or smc._theta_dev[0] is not smc._main or not np.array_equal(smc._theta_dev[1], th_loc):
        smc._main.set_params(_rows(smc._models(th_loc))[1])
        smc._theta_dev = (smc._main, th_loc.copy())


def _step_only(smc, y, t, verbose, out):
    """the propagation half of smc²! (smc_samplers.jl:323-338), without the degeneracy check"""
    _sync_params(smc)
    lik, _ = smc._main.step(float(y[t - 1]))
    smc.psteps += smc.M * smc.N
    ess, _ = smc.outer.window_walk(smc, np.asarray(lik, dtype=np.float64)[None, :], 0.0)
    smc.ess = float(ess[0])
    smc.t = t
    if verbose:
        out.write("\n")
    return smc


def _exchange(smc, y, verbose, out):
    """exchange!(smc, y)   smc_samplers.jl:163-189 (off unless min_ar > acceptance ratio; default -1)."""
    if smc.acc_ratio < smc.acc_threshold:
        if smc.N <= 4096:
            smc.N *= 2
            if verbose:
                out.write("\t%d particles added" % smc.N)
            models = smc._models(smc.theta[smc.lo:smc.hi])
            new_logZ, h = smc.backend.log_likelihood(models, smc.N, y, smc._next_seed(), smc._streams(), key="main")
            new_logZ = smc._gather(np.asarray(new_logZ, dtype=np.float64))
            smc.psteps += smc.M * smc.N * len(y)
            old, smc._main = smc._main, h
            smc._theta_dev = None
            if old is not None and old is not h and hasattr(smc.backend, "release"):
                smc.backend.release(old)          # the superseded N-particle filters
            smc._sync_outer()
            smc._set_logw(np.asarray(new_logZ) - smc.logZ)          # :183
            smc.logZ = np.asarray(new_logZ, dtype=np.float64).copy()
        else:
            out.write("\n\t[cannot exceed max state particles]")
