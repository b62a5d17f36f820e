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


def run_dt(M=32, N=128, T=25, seed=3, comm=None):
    _, y = smc.simulate(smc.UnivariateLinearGaussian(**LG), T, seed=1998)
    s = smc.SMC(N, M, lg_mod, lg_prior(), 2, 0.5, seed=seed, backend=OracleBackend(), comm=comm)
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
    assert s.psteps == (1 + 2 * (len(stages) - 1)) * 32 * 128 * 25
    s2, stages2, _ = run_dt()
    assert np.array_equal(s.theta, s2.theta) and np.array_equal(s.logZ, s2.logZ)      # deterministic


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


def run_online(M=24, N=64, T=30, seed=5, comm=None):
    _, y = smc.simulate(smc.UnivariateLinearGaussian(**LG), T, seed=1998)
    s = smc.SMC(N, M, lg_mod, lg_prior(), 2, 0.5, seed=seed, backend=OracleBackend(), comm=comm)
    smc.smc2(s, y)
    moves = 0
    for t in range(2, T + 1):
        before = s._calls
        smc.smc2_step(s, y, t, verbose=False)
        moves += s._calls > before
    x, w, _ = s._main.state()
    return s, moves, x, w


WORKER = r'''
import os, sys, io, json
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch.distributed as dist
import sequential_monte_carlo_amd as smc
from sequential_monte_carlo_amd.distributed import ThetaComm
from test_samplers_cpu import run_dt, run_online
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=2)
s, stages, text = run_dt(comm=ThetaComm(dist))
assert (s.lo, s.hi) == ((0, 16) if dist.get_rank() == 0 else (16, 32))
np.save(sys.argv[4] + ".%d.npy" % dist.get_rank(), np.concatenate([s.theta.ravel(), s.logZ, [st[0] for st in stages]]))
# online SMC^2 with theta sharded: resample! moves whole filters between the two ranks (all-to-all)
so, moves, x, w = run_online(comm=ThetaComm(dist))
assert moves >= 1
np.save(sys.argv[4] + ".online.%d.npy" % dist.get_rank(), np.concatenate([so.theta.ravel(), so.logZ, so.omega, x.ravel(), w.ravel()]))
dist.destroy_process_group()
'''


def test_theta_sharding_world_size_2_gloo(tmp_path):
    """N > 1 path: two gloo ranks each filter half of theta and all-gather logZ; the result is
    identical on both ranks and identical to the single-process run (stream id = global theta index)."""
    s, stages, _ = run_dt()
    ref = np.concatenate([s.theta.ravel(), s.logZ, [st[0] for st in stages]])
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = str(29500 + (os.getpid() % 2000))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r), str(tmp_path / "out")], env=env)
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    for r in range(2):
        got = np.load(str(tmp_path / "out") + ".%d.npy" % r)
        assert np.array_equal(got, ref)
    # online SMC^2: the two halves of the filter states, concatenated, equal the single-process run
    so, moves, x, w = run_online()
    assert moves >= 1
    head = np.concatenate([so.theta.ravel(), so.logZ, so.omega])
    parts = [np.load(str(tmp_path / "out") + ".online.%d.npy" % r) for r in range(2)]
    for p_ in parts:
        assert np.array_equal(p_[:head.size], head)
    d, M, N = x.shape[0], x.shape[1], x.shape[2]
    xs = np.concatenate([p_[head.size:head.size + d * (M // 2) * N].reshape(d, M // 2, N) for p_ in parts], axis=1)
    ws = np.concatenate([p_[head.size + d * (M // 2) * N:].reshape(M // 2, N) for p_ in parts], axis=0)
    assert np.array_equal(xs, x) and np.array_equal(ws, w)
