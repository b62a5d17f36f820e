"""CPU tests of the oracle (oracle/smc_oracle.c): pinned against every known answer available
for this path -- Philox KATs of the published algorithm, libm, scipy's logsumexp, the
reference's exact Kalman likelihood (src/kalman_filter.jl:29-70) and the committed golden
vectors (which detect drift of the oracle itself)."""
import json
import os

import numpy as np
import pytest
from scipy.special import logsumexp
from scipy.stats import chi2

from conftest import GOLDEN

LG = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


# Random123 kat_vectors, philox4x32-10
PHILOX_KAT = [
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


def test_philox_known_answers(ob):
    for ctr, key, out in PHILOX_KAT:
        assert ob.philox(ctr, key) == out


def test_exp_log_within_one_ulp_of_libm(ob):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-708, 5, 20000), rng.uniform(-1, 1, 20000), [0.0, -0.0, -708.0, -745.0, -1e9]])
    e, ref = ob.exp(x), np.exp(x)
    ok = ref > 1e-300
    assert np.max(np.abs(e[ok] - ref[ok]) / np.spacing(ref[ok])) <= 1.0
    assert ob.exp([0.0])[0] == 1.0 and ob.exp([-1e9])[0] == 0.0 and np.isnan(ob.exp([np.nan])[0])
    x = np.concatenate([rng.uniform(0, 1, 20000), np.exp(rng.uniform(-700, 700, 20000)), 1 - rng.uniform(0, 1e-6, 2000)])
    l, ref = ob.log(x), np.log(x)
    nz = ref != 0
    assert np.max(np.abs(l[nz] - ref[nz]) / np.spacing(np.abs(ref[nz]))) <= 1.0
    assert ob.log([1.0])[0] == 0.0 and ob.log([0.0])[0] == -np.inf


def test_sincos_and_box_muller_moments(ob):
    rng = np.random.default_rng(1)
    for u in rng.integers(0, 2**53, 3000) * 2.0**-53:
        c, s = ob.sincos2pi(u)
        assert abs(c - np.cos(2 * np.pi * u)) < 2e-15 and abs(s - np.sin(2 * np.pi * u)) < 2e-15
    for u, (c, s) in [(0.0, (1.0, 0.0)), (0.25, (0.0, 1.0)), (0.5, (-1.0, 0.0)), (0.75, (0.0, -1.0))]:
        cc, ss = ob.sincos2pi(u)
        assert abs(cc - c) < 1e-16 and abs(ss - s) < 1e-16
    z = np.array([ob.box_muller(list(rng.integers(0, 2**32, 4))) for _ in range(20000)]).ravel()
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1) < 0.03 and abs((z**4).mean() - 3) < 0.2


def test_normalize_matches_logsumexp(ob):
    rng = np.random.default_rng(2)
    for n in (1, 2, 7, 1024, 5000):
        logw = rng.normal(size=n) * 4 - 30
        lmu, w, ess = ob.normalize(logw)
        assert lmu == pytest.approx(logsumexp(logw) - np.log(n), rel=1e-12, abs=1e-12)
        assert w.sum() == pytest.approx(1.0, abs=1e-12)
        ref = np.exp(logw - logsumexp(logw))
        assert np.allclose(w, ref, rtol=1e-12, atol=2.0**-48)   # 2^-48 fixed-point resolution
        assert ess == pytest.approx(1.0 / np.sum(ref**2), rel=1e-9)
        assert 1.0 - 1e-9 <= ess <= n + 1e-9
    lmu, w, ess = ob.normalize(np.full(100, -3.25))     # constant weights: ess = N
    assert ess == 100.0 and lmu == pytest.approx(-3.25, abs=1e-15) and np.all(w == 0.01)
    lmu, w, ess = ob.normalize(np.array([-np.inf, 0.0, np.nan]))
    assert w[0] == 0 and w[2] == 0 and w[1] == 1.0 and ess == 1.0


def test_resample_is_multinomial(ob):
    rng = np.random.default_rng(3)
    w = rng.dirichlet(np.ones(50))
    N = 200000
    a = ob.resample(w, N, seed=99, stream=1, t=4)
    assert a.min() >= 0 and a.max() < 50
    cnt = np.bincount(a, minlength=50)
    stat = np.sum((cnt - N * w) ** 2 / (N * w))
    assert stat < chi2.ppf(0.9999, 49)
    # unsorted (iid) like StatsBase.sample, and reproducible
    assert np.any(np.diff(a) < 0)
    assert np.array_equal(a, ob.resample(w, N, seed=99, stream=1, t=4))
    assert not np.array_equal(a, ob.resample(w, N, seed=99, stream=1, t=5))
    # zero-weight entries are never drawn
    w2 = w.copy(); w2[::2] = 0
    assert np.all(ob.resample(w2, 5000, seed=1) % 2 == 1)
    with pytest.raises(ValueError):
        ob.resample(np.zeros(4), 4)


def test_golden_vectors_reproduced(ob):
    meta = json.load(open(os.path.join(GOLDEN, "filter_vectors.json")))
    g = np.load(os.path.join(GOLDEN, "filter_vectors.npz"))
    for name, m in meta.items():
        y = g[name + "/y"]
        f = ob.Filter(m["model"], m["raw"], m["n"], seg=m["seg"], seed=m["seed"], stream=m["stream"])
        logZ, lm, es = f.log_likelihood(y, trace=True)
        x, w, a, _ = f.state()
        C, mm, S, hi, lo = f.weights_raw()
        assert float(logZ).hex() == m["logZ"], name
        for key, val in (("logmu", lm), ("ess", es), ("x", x), ("w", w), ("m", mm)):
            assert np.array_equal(bits(g[name + "/" + key]), bits(val)), (name, key)
        assert np.array_equal(g[name + "/anc"], a) and np.array_equal(g[name + "/C"], C)
        assert np.array_equal(g[name + "/S"], S) and np.array_equal(g[name + "/S2hi"], hi)
    lmu, w, ess = ob.normalize(g["normalize/logw"])
    assert np.array_equal(bits(w), bits(g["normalize/w"])) and np.array_equal(bits([lmu, ess]), bits(g["normalize/out"]))
    assert np.array_equal(ob.resample(w, 2000, seed=17, stream=2, t=9), g["resample/a"])


def test_simulated_data_matches_fixture(ob):
    k = json.load(open(os.path.join(GOLDEN, "kalman_lg.json")))
    for T in (100, 200, 1000):
        _, y = ob.simulate(ob.LG1D, k["params"], T, k["sim_seed"])
        c = k["cases"]["T%d" % T]
        assert [float(v).hex() for v in y[:8]] == c["y_head"] and float(np.sum(y)).hex() == c["y_sum"]


def test_kalman_restatement_identities():
    """oracle/kalman.py against a dense multivariate-normal evaluation of the same model."""
    from oracle import kalman
    from scipy.stats import multivariate_normal
    A, B, Q, R, x0, s0 = 0.5, 1.0, 0.9, 0.8, 0.0, 1.0
    rng = np.random.default_rng(5)
    T = 6
    y = rng.normal(size=T)
    var = np.zeros(T); var[0] = s0
    for t in range(1, T):
        var[t] = A * A * var[t - 1] + Q
    cov = np.zeros((T, T))
    for i in range(T):
        for j in range(T):
            lo, d = min(i, j), abs(i - j)
            cov[i, j] = B * B * (A ** d) * var[lo] + (R if i == j else 0.0)
    mean = np.array([B * (A ** t) * x0 for t in range(T)])
    ref = multivariate_normal(mean, cov).logpdf(y)
    assert kalman.log_likelihood(y, A, B, Q, R, x0, s0, predict_first=False)[2] == pytest.approx(ref, rel=1e-12)


def test_particle_filter_pinned_by_kalman(ob):
    """E[exp(logZ_PF)] = exp(logZ_KF) (unbiasedness) and mean(logZ_PF) - logZ_KF = -Var/2 + O(se):
    the reference's exact likelihood src/kalman_filter.jl:29-70 pins the oracle (config C1)."""
    k = json.load(open(os.path.join(GOLDEN, "kalman_lg.json")))
    _, y = ob.simulate(ob.LG1D, LG, 100, 1998)
    kf = k["cases"]["T100"]["logZ_kf"]
    K = 96
    for seg in (0, 256):     # single segment and 4 segments obey the same law
        z = np.array([ob.Filter(ob.LG1D, LG, 1024, seg=seg, seed=1000 + s).log_likelihood(y) for s in range(K)])
        se = z.std(ddof=1) / np.sqrt(K)
        assert abs(z.mean() + 0.5 * z.var(ddof=1) - kf) < 4.5 * se
        r = np.exp(z - kf)
        assert abs(r.mean() - 1.0) < 4.5 * r.std(ddof=1) / np.sqrt(K)
    # variance of logZ falls like 1/Nx
    z4 = np.array([ob.Filter(ob.LG1D, LG, 4096, seed=2000 + s).log_likelihood(y) for s in range(48)])
    assert z4.var(ddof=1) < 0.6 * z.var(ddof=1)


def test_block_sorted_resampling_is_multinomial(ob):
    """Multi-segment resampling (uniforms generated sorted by block between Gamma break points) has the
    multinomial law of `sample(1:n, Weights(w), n)`: chi-square of the children counts per segment over
    many steps, and N p / N p (1-p) means and variances of the counts of arbitrary index ranges over
    many independent draws from the SAME weights (export/import + reseed)."""
    _, y = ob.simulate(ob.LG1D, LG, 60, 1998)
    n, seg = 4096, 256
    f = ob.Filter(ob.LG1D, LG, n, seg=seg, seed=5)
    f.bootstrap_filter(y[0])
    tot, dof = 0.0, 0
    for t in range(1, 60):
        _, w, _, _ = f.state()
        f.step(y[t])
        a = f.state()[2]
        # ordered by block of the CDF: no ancestor of block k+1 precedes one of block k
        blocks = [a[k:k + seg] for k in range(0, n, seg)]
        assert all(blocks[k].max() <= blocks[k + 1].min() for k in range(len(blocks) - 1))
        cnt = np.bincount(a // seg, minlength=n // seg)
        p = w.reshape(-1, seg).sum(axis=1)
        tot += float(np.sum((cnt - n * p) ** 2 / (n * p)))
        dof += n // seg - 1
    assert chi2.cdf(tot, dof) < 0.9995 and chi2.cdf(tot, dof) > 0.0005
    n, seg, K = 2000, 256, 1500
    f = ob.Filter(ob.LG1D, LG, n, seg=seg, seed=7)
    f.bootstrap_filter(y[0])
    snap = f.export_state()
    _, w, _, _ = f.state()
    cuts = [0, 100, 300, 777, 1000, 1290, 1600, 2000]
    cnts = np.zeros((K, len(cuts) - 1))
    for k in range(K):
        f.import_state(snap)
        f.reseed(1000 + k, 0)
        f.step(y[1])
        cnts[k] = np.histogram(f.state()[2], bins=cuts)[0]
    p = np.array([w[cuts[i]:cuts[i + 1]].sum() for i in range(len(cuts) - 1)])
    se_mean = np.sqrt(n * p * (1 - p) / K)
    assert np.all(np.abs(cnts.mean(axis=0) - n * p) < 4.5 * se_mean)
    assert np.all(np.abs(cnts.var(axis=0, ddof=1) / (n * p * (1 - p)) - 1.0) < 5 * np.sqrt(2.0 / K))


def test_systematic_resampling_option(ob):
    """The opt-in systematic resampler of the oracle: children sorted by ancestor, every particle gets
    floor(N w) or ceil(N w) children (the defining property), and the likelihood estimate stays unbiased
    against the reference's exact Kalman likelihood with no more variance than multinomial resampling."""
    _, y = ob.simulate(ob.LG1D, LG, 100, 1998)
    for n, seg in ((1000, 256), (1024, 0), (5000, 1024), (777, 256), (2049, 2048)):
        f = ob.Filter(ob.LG1D, LG, n, seg=seg, seed=3, systematic=True)
        f.bootstrap_filter(y[0])
        for t in range(1, 4):
            _, w, _, _ = f.state()
            f.step(y[t])
            a = f.state()[2]
            cnt = np.bincount(a, minlength=n)
            assert np.all(np.diff(a) >= 0) and cnt.sum() == n
            assert np.all(cnt >= np.floor(n * w - 1e-6)) and np.all(cnt <= np.ceil(n * w + 1e-6))
    kf = json.load(open(os.path.join(GOLDEN, "kalman_lg.json")))["cases"]["T100"]["logZ_kf"]
    K = 96
    zs = np.array([ob.Filter(ob.LG1D, LG, 1024, seg=256, seed=1000 + s, systematic=True).log_likelihood(y) for s in range(K)])
    zm = np.array([ob.Filter(ob.LG1D, LG, 1024, seg=256, seed=1000 + s).log_likelihood(y) for s in range(K)])
    r = np.exp(zs - kf)
    assert abs(r.mean() - 1.0) < 4.5 * r.std(ddof=1) / np.sqrt(K)
    assert abs(zs.mean() + 0.5 * zs.var(ddof=1) - kf) < 4.5 * zs.std(ddof=1) / np.sqrt(K)
    assert zs.var(ddof=1) < 1.3 * zm.var(ddof=1)


def test_edge_cases(ob):
    # Nx = 1, odd Nx, Nx not a multiple of seg, T = 1
    _, y = ob.simulate(ob.LG1D, LG, 5, 1998)
    for n, seg in ((1, 0), (3, 0), (257, 256), (513, 256)):
        f = ob.Filter(ob.LG1D, LG, n, seg=seg, seed=4)
        z, lm, es = f.log_likelihood(y, trace=True)
        x, w, a, _ = f.state()
        assert np.isfinite(z) and w.sum() == pytest.approx(1.0, abs=1e-12) and a.min() >= 0 and a.max() < n
        assert np.all(es >= 1 - 1e-9) and np.all(es <= n + 1e-9)
    f = ob.Filter(ob.LG1D, LG, 64, seed=4)
    assert f.log_likelihood(y[:1]) == f.bootstrap_filter(y[0])
    # an observation 1e3 sigma away collapses every weight: logmu = -inf, identity ancestors after
    f = ob.Filter(ob.SV1D, [-1.0, 0.95, 0.25], 64, seed=4)
    f.bootstrap_filter(0.1)
    lm, ess = f.step(1e200)
    assert lm == -np.inf and ess == 0.0
    lm2, _ = f.step(0.1)
    assert np.array_equal(f.state()[2], np.arange(64)) and np.isfinite(lm2)


def test_oracle_kalman_matches_numpy_restatement(ob):
    from oracle import kalman
    rng = np.random.default_rng(9)
    for _ in range(20):
        raw = [rng.uniform(-0.95, 0.95), rng.uniform(0.5, 1.5), rng.lognormal(), rng.lognormal(), rng.normal(), rng.lognormal()]
        y = rng.normal(size=60)
        for pf in (False, True):
            x, S, z = ob.kalman_log_likelihood(raw, y, pf)
            rx, rS, rz = kalman.log_likelihood(y, *raw[:4], x0=raw[4], sigma0=raw[5], predict_first=pf)
            assert z == pytest.approx(rz, rel=1e-12) and x == pytest.approx(rx, rel=1e-11) and S == pytest.approx(rS, rel=1e-11)


def test_sv_particle_filter_pinned_by_grid_filter(ob):
    """The stochastic-volatility model (BASELINE configs[2]; no closed form, not in the reference's src/) against an
    independent deterministic answer: the forward recursion on a 3001-point grid of the state (oracle/grid_filter.py).
    The grid answer is converged (two grid sizes agree to 1e-6); the particle estimate is unbiased for it and its variance
    falls like 1/Nx - the same two properties the Kalman likelihood pins for the linear-Gaussian model."""
    from oracle import grid_filter
    SV = [-1.0, 0.95, 0.25]
    _, y = ob.simulate(ob.SV1D, SV, 60, 1998)
    g1 = grid_filter.sv_log_likelihood(y, *SV, n_grid=3001)
    g2 = grid_filter.sv_log_likelihood(y, *SV, n_grid=1501)
    assert abs(g1 - g2) < 1e-6
    K = 96
    for seg in (0, 256):
        z = np.array([ob.Filter(ob.SV1D, SV, 1024, seg=seg, seed=3000 + s).log_likelihood(y) for s in range(K)])
        se = z.std(ddof=1) / np.sqrt(K)
        assert abs(z.mean() + 0.5 * z.var(ddof=1) - g1) < 4.5 * se, (z.mean(), z.var(ddof=1), g1)
        r = np.exp(z - g1)
        assert abs(r.mean() - 1.0) < 4.5 * r.std(ddof=1) / np.sqrt(K)
    z4 = np.array([ob.Filter(ob.SV1D, SV, 4096, seed=4000 + s).log_likelihood(y) for s in range(48)])
    assert z4.var(ddof=1) < 0.6 * z.var(ddof=1) and abs(z4.mean() - g1) < 0.1


def test_ucsv_particle_filter_against_rao_blackwellised_filter(ob):
    """UCSV (BASELINE configs[4]'s model; ssm.jl:215-263; no closed form): the oracle's bootstrap filter and an independent
    Rao-Blackwellised filter (oracle/rbpf_ucsv.py: particles over the two log-volatilities, Kalman for x) are both unbiased
    for p(y), so their means of exp(logZ) agree within Monte-Carlo error - which cross-checks the reference's conventions
    (previous log-volatility in x', gammas as standard deviations, variances exp(log s))."""
    from oracle import rbpf_ucsv
    UC = [0.2, 0.2, 3.0, 0.0, 0.0]
    _, y = ob.simulate(ob.UCSV3D, UC, 40, 1998)
    rng = np.random.default_rng(123)
    zr = np.array([rbpf_ucsv.log_likelihood(y, *UC, n=8000, rng=rng) for _ in range(24)])
    zb = np.array([ob.Filter(ob.UCSV3D, UC, 4096, seed=7000 + s).log_likelihood(y) for s in range(64)])
    c = zr.mean()
    er, eb = np.exp(zr - c), np.exp(zb - c)
    se = np.sqrt(er.var(ddof=1) / er.size + eb.var(ddof=1) / eb.size)
    assert abs(er.mean() - eb.mean()) < 4.5 * se, (er.mean(), eb.mean(), se)
    assert zr.std(ddof=1) < zb.std(ddof=1) * 1.5 and abs(zr.mean() - zb.mean()) < 0.5          # same scale, RB no noisier
    # a wrong convention is detected: x' driven by the CURRENT log-volatility changes the likelihood visibly
    # (power check with the Rao-Blackwellised filter itself: shifting gamma by 50 % moves logZ by far more than the se)
    zw = np.array([rbpf_ucsv.log_likelihood(y, 0.3, 0.3, *UC[2:], n=8000, rng=rng) for _ in range(8)])
    assert abs(zw.mean() - zr.mean()) > 5 * (zr.std(ddof=1) / np.sqrt(zr.size) + zw.std(ddof=1) / np.sqrt(zw.size))
