"""CPU tests of the host mirror of src/smc_samplers.jl (outer SMC logic) with the oracle as the
filter backend, including the world_size-2 gloo test of the theta sharding."""
import io
import os
import subprocess
import sys

import numpy as np
import pytest

import sequential_monte_carlo_amd as smc
from conftest import ROOT
from oracle_backend import OracleBackend

LG = dict(A=0.5, B=1.0, Q=0.9, R=0.8)


def lg_mod(theta):
    return smc.UnivariateLinearGaussian(A=theta[0], B=1.0, Q=theta[1], R=theta[2])


def lg_prior():
    return smc.product_distribution([smc.TruncatedNormal(0, 1, -1, 1), smc.LogNormal(), smc.LogNormal()])   # README.md:81-85


def test_priors():
    rng = np.random.default_rng(0)
    p = lg_prior()
    th = np.array([p.rand(rng) for _ in range(4000)])
    assert np.all(np.abs(th[:, 0]) <= 1) and np.all(th[:, 1:] > 0)
    assert abs(np.log(th[:, 1]).mean()) < 0.06 and abs(th[:, 0].mean()) < 0.05
    assert p.insupport([0.5, 1.0, 2.0]) and not p.insupport([1.5, 1.0, 2.0]) and not p.insupport([0.5, -1.0, 2.0])
    from scipy.stats import lognorm, truncnorm
    x = [0.3, 0.7, 1.9]
    ref = truncnorm(-1, 1).logpdf(0.3) + lognorm(1).logpdf(0.7) + lognorm(1).logpdf(1.9)
    assert p.logpdf(x) == pytest.approx(ref, rel=1e-12)
    assert p.logpdf([2.0, 1, 1]) == -np.inf
    u = smc.product_distribution([smc.Uniform(0, 1), smc.Normal(3, 2), smc.Uniform(0, 2), smc.Uniform(0, 2)])
    assert u.insupport([0.2, 3, 1, 1]) and not u.insupport([1.2, 3, 1, 1])


def test_models_and_simulate():
    m = smc.UnivariateLinearGaussian(**LG)
    assert m.raw() == [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
    x, y = smc.simulate(m, 50, seed=1998)
    assert x.shape == (50,) and y.shape == (50,)
    uc = smc.unobserved_components(sigma_eps=0.3, sigma_eta=0.5, x0=1.0)
    assert uc.raw() == [1.0, 1.0, 0.3, 0.5, 1.0, 0.3]
    u = smc.unobserved_components_stochastic_volatility(x0=3.0, gamma_eps=0.2, gamma_eta=0.2, log_sigma_eps=0.0, log_sigma_eta=0.0)
    xs, ys = smc.simulate(u, 20)
    assert xs.shape == (20, 3)
    with pytest.raises(ValueError):
        smc.UnivariateLinearGaussian(A=0.5, B=1, Q=-1, R=1)
    with pytest.raises(ValueError):
        smc.StochasticVolatility(0, 1.5, 1)


LG_TMAP = smc.ThetaMap(1, [0, -1, 1, 2, -1, -1], [0.0, 1.0, 0.0, 0.0, 0.0, 1.0])    # lg_mod in device-evaluable form


def _backend(lib_outer=False):
    """the oracle's filters; lib_outer: the OUTER level by the library's host routines (LibOuter: segment records, what the
    product runs) instead of the oracle's whole-vector twin (OracleOuter) - the two must agree bit for bit"""
    b = OracleBackend()
    if lib_outer:
        b.outer = smc.smc_samplers.LibOuter()
    return b


def run_dt(M=32, N=128, T=25, seed=3, comm=None, device=False, lib_outer=False, backend=None):
    """device=True: the sampler gets a ThetaMap, i.e. rejuvenate! takes the one-call-per-rank path (on the GPU:
    smc_pmmh_rejuvenate; here its oracle twin); False: the host loop with numpy random numbers."""
    _, y = smc.simulate(smc.UnivariateLinearGaussian(**LG), T, seed=1998)
    s = smc.SMC(N, M, lg_mod, lg_prior(), 2, 0.5, seed=seed, backend=backend or _backend(lib_outer), comm=comm,
                theta_map=LG_TMAP if device else None)
    assert s.device_pmmh == device
    buf = io.StringIO()
    stages = smc.density_tempered(s, y, verbose=True, out=buf)
    return s, stages, buf.getvalue()


def test_density_tempered_ladder():
    s, stages, text = run_dt()
    xis = [st[0] for st in stages]
    assert xis[-1] == 1.0 and all(b > a for a, b in zip(xis, xis[1:])) and len(xis) >= 2
    for xi, ess, acc in stages[:-1]:
        assert abs(ess - s.ess_min) < 0.5          # bisection lands on ess_min (smc_samplers.jl:245-258)
        assert 0.0 <= acc <= 1.0
    assert stages[-1][2] is None and stages[-1][1] >= s.ess_min - 0.5
    assert text.startswith("ξ = ") and "[rejuvenating]" in text and "acc_rate: " in text    # reference's log format
    th = smc.expected_parameters(s)
    assert -1 < th[0] < 1 and th[1] > 0 and th[2] > 0
    # every executed filter is counted, every proposal outside the prior's support is skipped (smc_samplers.jl:116)
    assert s.psteps + s.psteps_skipped == (1 + 2 * (len(stages) - 1)) * 32 * 128 * 25
    assert s.psteps == s.backend.filters_run * 128 * 25 and s.psteps_skipped > 0
    s2, stages2, _ = run_dt()
    assert np.array_equal(s.theta, s2.theta) and np.array_equal(s.logZ, s2.logZ)      # deterministic


def test_density_tempered_device_style_rejuvenation():
    """The ThetaMap path (one rejuvenate call per rank, Philox-keyed proposals and accept uniforms): same ladder
    properties, executed filters counted exactly, deterministic."""
    s, stages, text = run_dt(device=True)
    xis = [st[0] for st in stages]
    assert xis[-1] == 1.0 and all(b > a for a, b in zip(xis, xis[1:])) and len(xis) >= 2
    for xi, ess, acc in stages[:-1]:
        assert abs(ess - s.ess_min) < 0.5 and 0.0 < acc <= 1.0
    assert "[rejuvenating]" in text and "acc_rate: " in text
    assert s.psteps + s.psteps_skipped == (1 + 2 * (len(stages) - 1)) * 32 * 128 * 25
    assert s.psteps == s.backend.filters_run * 128 * 25 and s.psteps_skipped > 0
    th = smc.expected_parameters(s)
    assert -1 < th[0] < 1 and th[1] > 0 and th[2] > 0 and lg_prior().insupport_many(s.theta).all()
    s2, _, _ = run_dt(device=True)
    assert np.array_equal(s.theta, s2.theta) and np.array_equal(s.logZ, s2.logZ)


def test_pmmh_pieces_host_library_equals_oracle(L, ob):
    """The spec's PMMH pieces as compiled into libsmchip.so (host build of csrc/smc_spec.h) against the oracle's
    independent C restatement: proposals, log-uniforms, prior log densities - bit for bit."""
    import ctypes as C
    rng = np.random.default_rng(5)
    lib = L.lib()
    dp = C.POINTER(C.c_double)
    for d in (1, 2, 3, 4, 7, 8):
        A = rng.normal(size=(d, d))
        chol = np.linalg.cholesky(A @ A.T + 0.1 * np.eye(d))
        th = rng.normal(size=d)
        for c in (0, 1, 5):
            out = np.zeros(d)
            assert lib.smc_host_pmmh_propose(d, 12345 + d, 77, c, th.ctypes.data_as(dp), chol.ctypes.data_as(dp), 1.5, out.ctypes.data_as(dp)) == 0
            assert np.array_equal(out.view(np.uint64), ob.pmmh_propose(12345 + d, 77, c, th, chol, 1.5).view(np.uint64))
            assert lib.smc_host_pmmh_log_uniform(99, d, c) == ob.pmmh_log_uniform(99, d, c) <= 0.0
    fams = lg_prior().spec()
    u = smc.product_distribution([smc.Uniform(0, 1), smc.Normal(3, 2), smc.Uniform(0, 2), smc.Uniform(0, 2)]).spec()
    for fam, par in list(zip(*fams)) + list(zip(*u)):
        for x in np.concatenate([rng.normal(size=200) * 2, [0.0, 1.0, -1.0, 2.0, np.inf, -np.inf]]):
            a = lib.smc_host_prior_logpdf(int(fam), par.ctypes.data_as(dp), float(x))
            b = ob.prior_logpdf(fam, par, x)
            assert (a == b) or (np.isnan(a) and np.isnan(b)), (fam, x, a, b)
    # and the numbers are the densities they claim to be
    from scipy.stats import lognorm, norm, truncnorm, uniform
    assert ob.prior_logpdf(*[f[0] for f in fams], 0.3) == pytest.approx(truncnorm(-1, 1).logpdf(0.3), rel=1e-13)
    assert ob.prior_logpdf(fams[0][1], fams[1][1], 1.7) == pytest.approx(lognorm(1).logpdf(1.7), rel=1e-13)
    assert ob.prior_logpdf(u[0][1], u[1][1], 2.2) == pytest.approx(norm(3, 2).logpdf(2.2), rel=1e-13)
    assert ob.prior_logpdf(u[0][2], u[1][2], 1.2) == pytest.approx(uniform(0, 2).logpdf(1.2), rel=1e-13)
    assert ob.prior_logpdf(u[0][2], u[1][2], 2.2) == -np.inf


def test_random_walk_factor_follows_both_reference_kernels():
    """smc_samplers.jl:87-92 (univariate: Normal(x, scale*sigma), a standard deviation) and :95-100 (MvNormal(x, scale*Sigma))"""
    from sequential_monte_carlo_amd.smc_samplers import random_walk_factor
    rng = np.random.default_rng(1)
    th1 = rng.normal(size=(200, 1)) * 0.3
    L, s = random_walk_factor(th1, [1.5, 1.0, 0.5])
    sigma = 2.83 ** 2 * np.var(th1[:, 0], ddof=1) + 1e-10
    assert L.shape == (1, 1) and L[0, 0] == pytest.approx(sigma) and np.allclose(np.sqrt(s) * L[0, 0], np.array([1.5, 1.0, 0.5]) * sigma)
    th3 = rng.normal(size=(300, 3))
    L, s = random_walk_factor(th3, [1.5, 1.0])
    assert np.allclose(L @ L.T, 2.83 ** 2 / 3 * np.cov(th3.T) + 1e-10 * np.eye(3)) and np.array_equal(s, [1.5, 1.0])
    L, _ = random_walk_factor(np.ones((50, 2)), [1.0])
    assert np.allclose(L @ L.T, 1e-2 * np.eye(2))


def test_exchange_doubles_state_particles():
    """exchange! (smc_samplers.jl:163-189): with min_ar above the acceptance ratio the number of state particles
    doubles after a rejuvenation, the online filters are re-run over y[1:t-1] with 2N particles and the outer
    weights become the ratio of the new and old likelihood estimates."""
    _, y = smc.simulate(smc.UnivariateLinearGaussian(**LG), 30, seed=1998)
    for device in (False, True):
        s = smc.SMC(64, 24, lg_mod, lg_prior(), 2, 0.5, min_ar=2.0, seed=5, backend=OracleBackend(),
                    theta_map=LG_TMAP if device else None)
        smc.smc2(s, y)
        buf = io.StringIO()
        grown = []
        for t in range(2, 31):
            n0, logZ0 = s.N, s.logZ.copy()
            smc.smc2_step(s, y, t, verbose=True, out=buf)
            if s.N != n0:
                grown.append((t, n0, s.N))
                assert s._main.N == s.N and s._main.f[0].n == s.N
        assert grown and all(b == 2 * a for _, a, b in grown) and s.N == 64 * 2 ** len(grown)
        assert "%d particles added" % grown[0][2] in buf.getvalue()
        assert abs(s.omega.sum() - 1) < 1e-12 and np.all(np.isfinite(s.logZ))
        assert np.array_equal(s.omega, smc._lib.host_reweight(s.logw)[1])
        x, w, _ = s._main.state()
        assert x.shape == (1, 24, s.N) and np.allclose(w.sum(axis=1), 1.0, atol=1e-12)


def test_smc2_online_runs_and_tracks():
    _, y = smc.simulate(smc.UnivariateLinearGaussian(**LG), 30, seed=1998)
    s = smc.SMC(64, 24, lg_mod, lg_prior(), 2, 0.5, seed=5, backend=OracleBackend())
    smc.smc2(s, y)
    assert s.t == 1 and np.all(np.isfinite(s.logZ))
    rej = 0
    for t in range(2, 31):
        before = s._calls
        smc.smc2_step(s, y, t, verbose=False)
        rej += s._calls > before
        assert abs(s.omega.sum() - 1) < 1e-12 and 1 <= s.ess <= 24 + 1e-9
    assert rej >= 1          # the ESS threshold triggered at least one resample-move
    # logZ of every theta-particle equals a fresh filter run with its current parameters? (not after moves;
    # but it must be finite and the posterior mean sane)
    assert np.all(np.isfinite(s.logZ))
    th = smc.expected_parameters(s)
    assert -1 < th[0] < 1 and th[1] > 0 and th[2] > 0


def run_online(M=24, N=64, T=30, seed=5, comm=None, device=False, window=0, text=None, lib_outer=False, backend=None):
    """window = 0: the reference's loop, one smc²! per observation; > 0: smc2_run with that many steps per call"""
    _, y = smc.simulate(smc.UnivariateLinearGaussian(**LG), T, seed=1998)
    s = smc.SMC(N, M, lg_mod, lg_prior(), 2, 0.5, seed=seed, backend=backend or _backend(lib_outer), comm=comm,
                theta_map=LG_TMAP if device else None)
    smc.smc2(s, y)
    moves = 0
    if window:
        before = s._calls
        smc.smc2_run(s, y, 2, T, window=window, verbose=text is not None, out=text)
        moves = s._calls - before
    else:
        for t in range(2, T + 1):
            before = s._calls
            smc.smc2_step(s, y, t, verbose=text is not None, out=text)
            moves += s._calls > before
    x, w, _ = s._main.state()
    return s, moves, x, w


def test_windowed_online_run_equals_step_loop():
    """smc2_run (several propagation steps per device call, speculated and rolled back around resample-moves) gives
    the same sampler as the reference's `for t in 2:T smc²!(smc,y,t)` loop: bits, log text, counted particle-steps."""
    for device in (False, True):
        ta = io.StringIO()
        a, moves, xa, wa = run_online(device=device, text=ta)
        assert moves >= 1
        for window in (1, 3, 8, 64):
            tb = io.StringIO()
            b, _, xb, wb = run_online(device=device, window=window, text=tb)
            assert ta.getvalue() == tb.getvalue(), (device, window)
            assert np.array_equal(a.theta, b.theta) and np.array_equal(a.logZ, b.logZ) and np.array_equal(a.omega, b.omega)
            assert np.array_equal(xa, xb) and np.array_equal(wa, wb) and a.psteps == b.psteps and a.t == b.t


def test_outer_level_library_equals_oracle_twin():
    """A9, the outer level: whole sampler runs with the library's outer routines (csrc/smc_outer.hip: segment
    records, vector code) against the same runs with the oracle's independent whole-vector restatements (orc_outer_*): theta,
    logZ, omega, ladders, log text - bit for bit.  (The inner filters are the oracle's in both: only the outer level differs.)"""
    for device in (False, True):
        a, st_a, txt_a = run_dt(device=device)
        b, st_b, txt_b = run_dt(device=device, lib_outer=True)
        assert txt_a == txt_b and st_a == st_b
        assert np.array_equal(a.theta, b.theta) and np.array_equal(a.logZ, b.logZ) and np.array_equal(a.omega, b.omega)
        for window in (0, 5):
            ta, tb = io.StringIO(), io.StringIO()
            a, ma, xa, wa = run_online(M=32, device=device, window=window, text=ta)
            b, mb, xb, wb = run_online(M=32, device=device, window=window, text=tb, lib_outer=True)
            assert ma == mb >= 1 and ta.getvalue() == tb.getvalue()
            assert np.array_equal(a.theta, b.theta) and np.array_equal(a.logZ, b.logZ) and np.array_equal(a.logw, b.logw)
            assert np.array_equal(a.omega, b.omega) and np.array_equal(xa, xb) and np.array_equal(wa, wb) and a.ess == b.ess


WORKER = r'''
import os, sys, io, json
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch.distributed as dist
import sequential_monte_carlo_amd as smc
from sequential_monte_carlo_amd.distributed import ThetaComm
from test_samplers_cpu import run_dt, run_online
WS = int(sys.argv[5])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=WS)
for device in (False, True):      # host-loop rejuvenation / one-call-per-rank rejuvenation (ThetaMap)
    # the OUTER level: the oracle's whole-vector twin with device=False, the library's segment records (what the product runs;
    # whole-segment slices exchange records, others the likelihood increments) with device=True
    tag = sys.argv[4] + (".dev" if device else "")
    s, stages, text = run_dt(comm=ThetaComm(dist), device=device, lib_outer=device)
    assert (s.lo, s.hi) == (dist.get_rank() * (32 // WS), (dist.get_rank() + 1) * (32 // WS))
    np.save(tag + ".%d.npy" % dist.get_rank(), np.concatenate([s.theta.ravel(), s.logZ, [st[0] for st in stages], [s.psteps, s.psteps_skipped]]))
    # online SMC^2 with theta sharded: resample! moves whole filters between the ranks (all-to-all)
    so, moves, x, w = run_online(M=32, comm=ThetaComm(dist), device=device, window=5 if device else 0, lib_outer=device)
    assert moves >= 1
    np.save(tag + ".online.%d.npy" % dist.get_rank(), np.concatenate([so.theta.ravel(), so.logZ, so.omega, so.logw, x.ravel(), w.ravel()]))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 4, 8])
def test_theta_sharding_world_size_2_gloo(tmp_path, world):
    """N > 1 path: `world` gloo ranks each filter their share of theta and all-gather logZ; the result is
    identical on every rank and identical to the single-process run (stream id = global theta index; outer sums over fixed
    segments).  With four and eight ranks resample! of the online sampler moves filters between ranks that are not neighbours;
    with 32 parameter particles the slices of 2 and 4 ranks are whole segments of 8 (the ranks exchange segment records), those
    of 8 ranks are not (the likelihood increments travel): the same bits every way."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = str(29500 + (os.getpid() % 2000) + world)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r), str(tmp_path / "out"), str(world)], env=env)
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    for device in (False, True):
        tag = str(tmp_path / "out") + (".dev" if device else "")
        s, stages, _ = run_dt(device=device)
        ref = np.concatenate([s.theta.ravel(), s.logZ, [st[0] for st in stages], [s.psteps, s.psteps_skipped]])
        for r in range(world):
            got = np.load(tag + ".%d.npy" % r)
            assert np.array_equal(got, ref), device
        # online SMC^2: the ranks' shares of the filter states, concatenated, equal the single-process run
        so, moves, x, w = run_online(M=32, device=device)
        assert moves >= 1
        head = np.concatenate([so.theta.ravel(), so.logZ, so.omega, so.logw])
        parts = [np.load(tag + ".online.%d.npy" % r) for r in range(world)]
        for p_ in parts:
            assert np.array_equal(p_[:head.size], head), device
        d, M, N = x.shape[0], x.shape[1], x.shape[2]
        per = M // world
        xs = np.concatenate([p_[head.size:head.size + d * per * N].reshape(d, per, N) for p_ in parts], axis=1)
        ws = np.concatenate([p_[head.size + d * per * N:].reshape(per, N) for p_ in parts], axis=0)
        assert np.array_equal(xs, x) and np.array_equal(ws, w), device


def test_filtered_summaries_and_trend():
    """get_quantiles_* of examples/inflation_example.jl:39-55 and estimated_trend of plotting_utils.jl:116-124 over the
    online sampler: per-filter weighted quantiles / variance / mean, integrated over the parameter particles."""
    s, _, x, w = run_online(T=12)
    q, var = smc.filtered_summaries(s, [0.25, 0.5, 0.75])
    om = s.omega / s.omega.sum()
    means = np.einsum("mn,mn->m", w, x[0])
    per_var = np.einsum("mn,mn->m", w, x[0] ** 2) - means ** 2
    assert var == pytest.approx(float(om @ per_var), rel=1e-9) and q.shape == (3,) and q[0] <= q[1] <= q[2]
    # the per-filter quantile is the inverse of the weighted empirical CDF: check against a sort-based evaluation
    ref = []
    for m in range(s.M):
        o = np.argsort(x[0, m], kind="stable")
        cw = np.cumsum(w[m][o])
        ref.append([x[0, m][o][min(np.searchsorted(cw, pp, side="right"), x.shape[2] - 1)] for pp in (0.25, 0.5, 0.75)])
    assert np.allclose(q, om @ np.array(ref), atol=0.05)
    assert smc.estimated_trend(s) == pytest.approx(float(om @ (1.0 * means)), rel=1e-9)      # B = 1 in lg_mod


def test_windowed_run_collects_per_period_summaries():
    """smc2_run(..., summaries=p): the per-period filtered summaries the example's loop collects (examples/inflation_example.jl:
    78-86: get_quantiles_uc(smc) after every smc²!) from summaries recorded per step inside the window launches - the same
    numbers, bit for bit, as filtered_summaries(smc) called after every smc2_step; the sampler itself is not changed by them."""
    _, y = smc.simulate(smc.UnivariateLinearGaussian(**LG), 30, seed=1998)
    p = [0.25, 0.5, 0.75]
    for device in (False, True):
        ref = smc.SMC(64, 32, lg_mod, lg_prior(), 2, 0.5, seed=5, backend=OracleBackend(), theta_map=LG_TMAP if device else None)
        smc.smc2(ref, y)
        rows = []
        for t in range(2, 31):
            smc.smc2_step(ref, y, t, verbose=False)
            rows.append((t,) + smc.filtered_summaries(ref, p))
        for window in (1, 5, 16):
            for lib_outer in (False, True):
                s = smc.SMC(64, 32, lg_mod, lg_prior(), 2, 0.5, seed=5, backend=_backend(lib_outer), theta_map=LG_TMAP if device else None)
                smc.smc2(s, y)
                smc.smc2_run(s, y, 2, 30, window=window, verbose=False, summaries=p)
                assert np.array_equal(s.theta, ref.theta) and np.array_equal(s.logZ, ref.logZ) and np.array_equal(s.logw, ref.logw)
                assert len(s.summary_trace) == 29
                for (t, q, v), (t0, q0, v0) in zip(s.summary_trace, rows):
                    assert t == t0 and np.array_equal(q, q0) and v == v0, (device, window, lib_outer, t)


def _golden_sampler_runs(backend_factory):
    """re-run the two committed sampler cases (tests/golden/sampler_vectors.json) on a backend -> dict like the fixture"""
    import json
    from conftest import GOLDEN
    g = json.load(open(os.path.join(GOLDEN, "sampler_vectors.json")))
    hx = lambda a: [float(v).hex() for v in np.asarray(a, dtype=np.float64).ravel()]
    out = {}
    c = g["density_tempered_lg"]
    _, y = smc.simulate(smc.UnivariateLinearGaussian(**LG), c["T"], seed=1998)
    s = smc.SMC(c["N"], c["M"], lg_mod, lg_prior(), c["chain"], 0.5, seed=c["seed"], backend=backend_factory(), theta_map=LG_TMAP)
    buf = io.StringIO()
    stages = smc.density_tempered(s, y, verbose=True, out=buf)
    out["density_tempered_lg"] = dict(text=buf.getvalue(), theta=hx(s.theta), logZ=hx(s.logZ), xi=hx([st[0] for st in stages]),
                                      psteps=int(s.psteps), psteps_skipped=int(s.psteps_skipped))
    c = g["smc2_lg"]
    s = smc.SMC(c["N"], c["M"], lg_mod, lg_prior(), c["chain"], c["ess_threshold"], seed=c["seed"], backend=backend_factory(), theta_map=LG_TMAP)
    buf = io.StringIO()
    smc.smc2(s, y)
    smc.smc2_run(s, y, 2, c["T"], window=c["window"], verbose=True, out=buf)
    x, w, _ = s._main.state()
    out["smc2_lg"] = dict(text=buf.getvalue(), theta=hx(s.theta), logZ=hx(s.logZ), omega=hx(s.omega), x_sum=float(np.sum(x)).hex(),
                          w_head=hx(w[:, :4]), psteps=int(s.psteps))
    return g, out


def test_golden_sampler_vectors_oracle_backend(ob):
    """The committed sampler fixtures (PMMH pieces and whole runs, generated by tests/golden/make_golden.py from the
    oracle): the oracle still produces them (drift detector); the GPU suite checks the HIP backend against the same file."""
    g, out = _golden_sampler_runs(OracleBackend)
    for case in ("density_tempered_lg", "smc2_lg"):
        for k, v in out[case].items():
            assert g[case][k] == v, (case, k)
    p = g["pmmh"]
    chol = np.array([float.fromhex(v) for v in p["chol"]]).reshape(4, 4)
    theta = np.array([float.fromhex(v) for v in p["theta"]])
    for c in ("0", "1", "2"):
        assert [float(v).hex() for v in ob.pmmh_propose(p["seed"], p["stream"], int(c), theta, chol, p["scale"])] == p["proposals"][c]
        assert float(ob.pmmh_log_uniform(p["seed"], p["stream"], int(c))).hex() == p["log_uniform"][c]
    for k, q in g["prior_logpdf"].items():
        assert [float(ob.prior_logpdf(q["family"], q["par"], x)).hex() for x in q["x"]] == q["logpdf"], k
