"""TEST INFRASTRUCTURE: a filter backend for the samplers that runs every inner filter on the CPU
oracle instead of the GPU.  Same interface as sequential_monte_carlo_amd.smc_samplers.HipBackend, so
the sampler's host logic can be (a) tested on CPU, (b) sharded over gloo ranks, and (c) compared
bit for bit with the HIP backend on the GPU box.  Never imported by the product."""
import numpy as np

from oracle import binding as ob
from sequential_monte_carlo_amd.smc_samplers import _rows as params_matrix


class OracleHandle:
    def __init__(self, model_id, raws, N, seg, seed, streams, systematic=False):
        self.model_id, self.N, self.seg, self.systematic = model_id, N, seg, systematic
        self.n_theta = raws.shape[0]
        self.f = [ob.Filter(model_id, raws[m], N, seg=seg, seed=seed, stream=int(streams[m]), systematic=systematic)
                  for m in range(self.n_theta)]

    def set_params(self, raws):
        for m, f in enumerate(self.f):
            f.set_params(raws[m])

    def init(self, y1):
        return np.array([f.bootstrap_filter(y1) for f in self.f])

    def step(self, y):
        r = [f.step(y) for f in self.f]
        return np.array([a for a, _ in r]), np.array([b for _, b in r])

    def log_likelihood(self, y):
        return np.array([f.log_likelihood(y) for f in self.f])

    # the window of the online sampler (smc_step_window / smc_step_commit): k steps, then keep the first j
    can_window = True

    def _blank(self):
        raw = [0.5, 1, 1, 1, 0, 1] if self.model_id == 1 else ([0, 0.5, 1] if self.model_id == 2 else [1, 1, 0, 0, 0])
        return ob.Filter(self.model_id, raw, self.N, seg=self.seg)

    # per-step summaries inside the window (smc_set_summaries / smc_get_summaries)
    _summ = None

    def set_summaries(self, p=None, component=0, moments=False):
        self._summ = None if (p is None and not moments) else (list(p or []), int(component), bool(moments))
        self._summ_rows = []

    def get_summaries(self, T):
        q = np.stack([r[0] for r in self._summ_rows[:T]]) if self._summ[0] else None
        mean = np.stack([r[1] for r in self._summ_rows[:T]]) if self._summ[2] else None
        var = np.stack([r[2] for r in self._summ_rows[:T]]) if self._summ[2] else None
        return q, mean, var

    def step_window(self, y):
        self._snap = [self._blank() for _ in self.f]
        for s, f in zip(self._snap, self.f):
            s.copy_state_from(f)
        self._win_y = np.array(y, dtype=np.float64)
        r = []
        self._summ_rows = []
        for v in self._win_y:
            r.append(self.step(float(v)))
            if self._summ:
                m, vv = self.moments()
                self._summ_rows.append((self.quantiles(self._summ[0], self._summ[1]) if self._summ[0] else None, m, vv))
        return np.array([a for a, _ in r]), np.array([b for _, b in r])

    def step_commit(self, j):
        if j < len(self._win_y):
            for s, f in zip(self._snap, self.f):
                f.copy_state_from(s)
            for v in self._win_y[:j]:
                self.step(float(v))
        self._snap = None

    def permute(self, a):
        # value copy: snapshot the sources first
        snap = [self._blank() for _ in self.f]
        for s, f in zip(snap, self.f):
            s.copy_state_from(f)
        for m, f in enumerate(self.f):
            f.copy_state_from(snap[int(a[m])])

    def copy_from(self, src, mask):
        for m, f in enumerate(self.f):
            if mask[m]:
                f.copy_state_from(src.f[m])

    # movement of whole filters between ranks (same interface distributed.py uses for the HIP handle)
    def export_slots(self, idx):
        import torch
        rows = [self.f[int(i)].export_state().view(np.int64) for i in idx]
        w = int(ob.lib().orc_filter_state_words(self.f[0]._h))
        return torch.from_numpy(np.stack(rows)) if rows else torch.empty((0, w), dtype=torch.int64)

    def import_slots(self, idx, buf):
        arr = buf.numpy().view(np.uint64)
        for k, i in enumerate(idx):
            self.f[int(i)].import_state(arr[k])

    def moments(self):
        r = [f.moments() for f in self.f]
        return np.stack([a for a, _ in r], axis=1), np.stack([b for _, b in r], axis=1)          # [d][n_theta]

    def quantiles(self, p, component=0):
        return np.stack([f.quantiles(p, component) for f in self.f])                             # [n_theta][len(p)]

    def state(self):
        xs = [f.state() for f in self.f]
        return np.stack([x[0] for x in xs], axis=1), np.stack([x[1] for x in xs]), None


class OracleOuter:
    """The outer level of the samplers (reweight, the window walk of smc²!, the tempering bisection, resample!, the
    random-walk factor) by the oracle's whole-vector restatements (oracle/smc_oracle.c orc_outer_*, orc_rw_factor): the
    independent twin of sequential_monte_carlo_amd.smc_samplers.LibOuter (csrc/smc_outer.hip, which works rank by rank on
    segment records).  With theta sharded every rank here walks the WHOLE vector: the log-likelihood increments travel."""

    @staticmethod
    def reweight(logw, want_w=True):
        return ob.outer_reweight(logw, want_w)

    temper = staticmethod(ob.outer_temper)
    resample = staticmethod(ob.outer_resample)
    rw_factor = staticmethod(ob.rw_factor)

    def window_walk(self, smc, lik_local, ess_min):
        lik_local = np.ascontiguousarray(lik_local, dtype=np.float64)
        k, per = lik_local.shape
        smc._sync_outer()
        lik = lik_local if smc.comm is None else smc._gather(lik_local.ravel()).reshape(-1, k, per).transpose(1, 0, 2).reshape(k, smc.M)
        smc.logw, smc.logZ, ess, j = ob.outer_steps(smc.logw, smc.logZ, lik, ess_min)     # smc_samplers.jl:323-338
        return ess, j


class OracleBackend:
    outer = OracleOuter()

    def __init__(self, seg=0, resampler="multinomial"):
        self.seg, self.systematic = seg, resampler == "systematic"
        self.filters_run = 0

    def log_likelihood(self, models, N, y, seed, streams, key="prop", skip=None):
        mid, raw = params_matrix(models)
        h = OracleHandle(mid, raw, N, self.seg, seed, streams, self.systematic)
        y = np.asarray(y, dtype=np.float64)
        if skip is None or not np.any(skip):
            self.filters_run += h.n_theta
            return h.log_likelihood(y), h
        # skipped filters are never run (smc_samplers.jl:116): logZ = -inf
        self.filters_run += int(h.n_theta - np.sum(skip))
        return np.array([-np.inf if skip[m] else f.log_likelihood(y) for m, f in enumerate(h.f)]), h

    def rejuvenate(self, tmap, prior_spec, N, y, xi, chol, scales, filter_seeds, move_seed, streams, theta, logZ, main):
        """rejuvenate!(smc, y, xi) (smc_samplers.jl:103-146) for the parameter particles of one rank, statement by
        statement with the oracle's pieces; the twin of smc_pmmh_rejuvenate.  -> (theta, logZ, accepted, filters_run)"""
        fam, par = prior_spec
        theta = np.array(theta, dtype=np.float64)
        logZ = np.array(logZ, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        M, d = theta.shape
        accepted = np.zeros(M, dtype=bool)
        nrun = 0
        for c in range(len(scales)):                                    # for c in 1:smc.chain        :113
            for m in range(M):                                          # (Threads.@threads for m     :112)
                st = int(streams[m])
                prop = ob.pmmh_propose(move_seed, st, c, theta[m], chol, scales[c])               # :114
                if not all(ob.prior_insupport(fam[i], par[i], prop[i]) for i in range(d)):       # :116
                    continue
                raw = tmap.rows(prop[None, :])[0]                                                  # smc.model(theta_prop)
                f = ob.Filter(tmap.model_id, raw, N, seg=self.seg, seed=int(filter_seeds[c]), stream=st, systematic=self.systematic)
                logZ_prop = f.log_likelihood(y)                                                    # :117-121
                nrun += 1
                lp_prop = lp_cur = 0.0
                for i in range(d):
                    lp_prop = lp_prop + ob.prior_logpdf(fam[i], par[i], prop[i])
                    lp_cur = lp_cur + ob.prior_logpdf(fam[i], par[i], theta[m, i])
                prior_ratio = lp_prop - lp_cur                                                     # :123
                likelihood_ratio = xi * (logZ_prop - logZ[m])                                      # :124
                log_post_prop = logZ_prop + lp_prop                                                # :126
                acc_ratio = likelihood_ratio + prior_ratio                                         # :127
                if log_post_prop > -np.inf and ob.pmmh_log_uniform(move_seed, st, c) < acc_ratio:  # :129
                    logZ[m] = logZ_prop                                                            # :130
                    theta[m] = prop                                                                # :131
                    if main is not None:
                        main.f[m].copy_state_from(f)                                               # :132-133
                    accepted[m] = True                                                             # :135
        self.filters_run += nrun
        return theta, logZ, accepted, nrun

    def release(self, h):
        pass

    def init(self, models, N, y1, seed, streams, key="main"):
        mid, raw = params_matrix(models)
        h = OracleHandle(mid, raw, N, self.seg, seed, streams, self.systematic)
        return h.init(float(y1)), h

    def close(self):
        pass
