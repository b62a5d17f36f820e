"""Host mirror of src/particles.jl over the HIP library: same names, argument meaning, return tuples.

    normalize(logw)                      -> (logmu, w, ess)     particles.jl:5-15
    resample(w, N=len(w))                -> indices (0-based)   particles.jl:17-19
    bootstrap_filter(N, y1, model)       -> (x, w, logmu)       particles.jl:87-105
    bootstrap_filter_(x, w, y, model)    -> (logmu, w, ess)     particles.jl:107-129   ("bootstrap_filter!")
    log_likelihood(N, y, model)          -> (x, w, logZ)        particles.jl:132-147

`model` may be one StateSpaceModel or a list of them (the batched callers smc_samplers.jl:112-121,
223-229,289-295,325-335): then logmu / logZ / ess are arrays over the list.
x and w are views of device-resident state (`Particles`, `Weights`): they convert to numpy arrays
on demand (np.asarray) and are updated in place by bootstrap_filter_, like the reference's x.
Everything runs on the GPU; there is no CPU path.
"""
import itertools

import numpy as np

from . import _lib
from .models import params_matrix

_seed_counter = itertools.count(1)


def normalize(logw, device=0):
    """(logmu, w, ess) with logmu = log(mean(exp(logw)))   particles.jl:5-15"""
    return _lib.normalize(np.asarray(logw, dtype=np.float64), device)


reweight = normalize   # the samplers' name for it (smc_samplers.jl:232,249,265,298,338)


def resample(w, N=None, seed=None, stream=0, t=0, device=0):
    """N iid draws from Categorical(w), unsorted (sample(1:n, Weights(w), N), particles.jl:17-19).
    0-based indices."""
    if seed is None:
        seed = next(_seed_counter) + (1 << 40)
    return _lib.resample(np.asarray(w, dtype=np.float64), N, seed, stream, t, device)


class _Filters:
    """Device state shared by the Particles / Weights views of one bootstrap_filter call."""

    def __init__(self, N, models, seed, seg, device, streams, ancestors, resampler="multinomial"):
        mid, raw = params_matrix(models)
        self.single = not isinstance(models, (list, tuple))
        if resampler not in ("multinomial", "systematic"):
            raise ValueError("resampler must be 'multinomial' (the reference's law) or 'systematic' (opt-in)")
        flags = (_lib.FLAG_ANCESTORS if ancestors else 0) | (_lib.FLAG_SYSTEMATIC if resampler == "systematic" else 0)
        self.h = _lib.Handle(mid, raw.shape[0], N, seg=seg, seed=seed, device=device, flags=flags)
        self.h.set_params(raw)
        if streams is not None:
            self.h.set_streams(streams)
        self.model_id, self.raw = mid, raw

    def check_model(self, models):
        mid, raw = params_matrix(models)
        if mid != self.model_id or raw.shape != self.raw.shape:
            raise ValueError("model family / batch size differs from the one the particles were created with")
        if not np.array_equal(raw, self.raw):
            self.h.set_params(raw)
            self.raw = raw

    def out(self, a):
        return float(a[0]) if self.single else a


class Particles:
    """x: the particle states, resident on the GPU.  np.asarray(x) -> [N] (scalar state), [N, d], or with
    a leading batch axis for a list of models."""

    def __init__(self, f):
        self._f = f

    def __array__(self, dtype=None, copy=None):
        x, _, _ = self._f.h.state(want_w=False, want_anc=False)      # [d][n_theta][N]
        x = np.moveaxis(x, 0, -1)                                   # [n_theta][N][d]
        if x.shape[-1] == 1:
            x = x[..., 0]
        return x[0] if self._f.single else x

    def __len__(self):
        return self._f.h.n_x

    def moments(self):
        """(mean, variance) of the filtered state under the current weights, computed on the device."""
        m, v = self._f.h.moments()           # [d][n_theta]
        m, v = m.T, v.T
        if m.shape[-1] == 1:
            m, v = m[..., 0], v[..., 0]
        return (m[0], v[0]) if self._f.single else (m, v)

    def quantile(self, p, component=0):
        """Weighted quantiles of the filtered state (`quantile(x, weights(w), p)`,
        examples/inflation_example.jl:45), computed on the device: the inverse of the weighted
        empirical CDF (no interpolation between particles).  [len(p)] or [n_theta][len(p)]."""
        q = self._f.h.quantiles(p, component)
        return q[0] if self._f.single else q

    def ancestors(self):
        _, _, a = self._f.h.state(want_w=False, want_anc=True)
        return a[0] if self._f.single else a


class Weights:
    """w: the normalised weights of normalize() (particles.jl:11), resident on the GPU."""

    def __init__(self, f):
        self._f = f

    def __array__(self, dtype=None, copy=None):
        _, w, _ = self._f.h.state(want_w=True, want_anc=False)
        return w[0] if self._f.single else w

    def __len__(self):
        return self._f.h.n_x


def bootstrap_filter(N, y, model, seed=None, seg=0, device=0, streams=None, ancestors=False, resampler="multinomial"):
    """x, w, logmu = bootstrap_filter(N, y[1], model)   particles.jl:87-105
    resampler="systematic" (opt-in, not the reference's law) applies to the following bootstrap_filter_ steps."""
    if seed is None:
        seed = next(_seed_counter)
    f = _Filters(int(N), model, seed, seg, device, streams, ancestors, resampler)
    logmu = f.h.init(float(y))
    return Particles(f), Weights(f), f.out(logmu)


def bootstrap_filter_(states, weights, y, model):
    """logmu, w, ess = bootstrap_filter!(x, w, y[t], model)   particles.jl:107-129
    `states` is updated in place (device resident); the returned w is the new weight view."""
    f = states._f
    if weights._f is not f:
        raise ValueError("states and weights belong to different filters")
    f.check_model(model)
    logmu, ess = f.h.step(float(y))
    return f.out(logmu), Weights(f), f.out(ess)


def _summaries_out(f, T):
    """per-step summaries of the call just made, in the shapes of the README loop: quantiles [T][len(p)], mean / var [T] (scalar
    state) or [T][d]; with a list of models a batch axis follows T"""
    q, mean, var = f.h.get_summaries(T)
    out = {}
    if q is not None:
        out["quantiles"] = q[:, 0, :] if f.single else q
    if mean is not None:
        mean, var = np.moveaxis(mean, 1, -1), np.moveaxis(var, 1, -1)      # [T][n_theta][d]
        if mean.shape[-1] == 1:
            mean, var = mean[..., 0], var[..., 0]
        out["mean"], out["var"] = (mean[:, 0], var[:, 0]) if f.single else (mean, var)
    return out


def log_likelihood(N, y, model, seed=None, seg=0, device=0, streams=None, ancestors=False, trace=False,
                   resampler="multinomial", quantiles=None, component=0, moments=False):
    """x, w, logZ = log_likelihood(N, y, model)   particles.jl:132-147
    trace=True additionally returns the per-step (logmu_t, ess_t).  resampler="systematic": opt-in systematic
    resampling (same expectation, lower variance, one launch per step for big filters; not the reference's law).
    quantiles=[...] / moments=True: the README loop (README.md:33-61: bootstrap_filter!, then quantile(x, ...) at every
    observation) as ONE call - the per-step weighted quantiles of state coordinate `component` and / or mean and variance
    are computed on the device inside the filter loop and returned as a dict behind the usual results."""
    if seed is None:
        seed = next(_seed_counter)
    f = _Filters(int(N), model, seed, seg, device, streams, ancestors, resampler)
    y = np.ascontiguousarray(y, dtype=np.float64)
    summ = quantiles is not None or moments
    if summ:
        f.h.set_summaries(quantiles, component, moments)
        try:
            res = f.h.log_likelihood(y, trace=trace)
            extra = _summaries_out(f, y.size)
        finally:
            f.h.set_summaries()
        if trace:
            logZ, lm, es = res
            if f.single:
                lm, es = lm[:, 0], es[:, 0]
            return Particles(f), Weights(f), f.out(logZ), lm, es, extra
        return Particles(f), Weights(f), f.out(res), extra
    if trace:
        logZ, lm, es = f.h.log_likelihood(y, trace=True)
        if f.single:
            lm, es = lm[:, 0], es[:, 0]
        return Particles(f), Weights(f), f.out(logZ), lm, es
    logZ = f.h.log_likelihood(y)
    return Particles(f), Weights(f), f.out(logZ)
