"""Generates the committed golden fixtures under tests/golden/.

Sources of truth:
  * kalman_*  : numpy restatement (oracle/kalman.py) of the reference's exact Kalman
                log-likelihood src/kalman_filter.jl:29-70 -- the reference's only known-answer
                for this path.  The reference itself cannot be run here (Julia,
                no toolchain; see DESIGN.md "Oracle"), so there are no reference-run outputs.
  * filter_*  : outputs of the CPU oracle (oracle/smc_oracle.c) on small seeded cases; the GPU
                box re-derives them and the HIP path must reproduce them bit for bit.
  * sampler_* : the PMMH pieces (proposal, accept uniform, prior log densities) of the oracle, and whole small sampler
                runs (density_tempered, online SMC^2) of the host mirror over the oracle backend with the device-style
                rejuvenation; the HIP backend must reproduce theta / logZ / ladder bit for bit.
Run:  python tests/golden/make_golden.py      (from the repo root)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from oracle import kalman  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

LG = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]     # README.md:12-22 (A,B,Q,R,x0,sigma0)
SV = [-1.0, 0.95, 0.25]
UC = [0.2, 0.2, 3.0, 0.0, 0.0]          # BASELINE.md C5 data generator

CASES = [
    # name, model, raw, n, seg, seed, stream, T
    ("lg_n64", ob.LG1D, LG, 64, 0, 11, 0, 100),
    ("lg_n1024_c1", ob.LG1D, LG, 1024, 0, 1, 0, 100),        # BASELINE config 1
    ("lg_n1000_seg256", ob.LG1D, LG, 1000, 256, 5, 3, 30),   # 4 segments, ragged tail
    ("lg_n4096_seg512", ob.LG1D, LG, 4096, 512, 9, 0, 20),
    ("sv_n1024", ob.SV1D, SV, 1024, 0, 2, 0, 50),
    ("sv_n3000_seg1024", ob.SV1D, SV, 3000, 1024, 2, 7, 20),
    ("ucsv_n512", ob.UCSV3D, UC, 512, 0, 3, 0, 50),
    ("ucsv_n2500_seg1024", ob.UCSV3D, UC, 2500, 1024, 4, 1, 20),
]


def main():
    ob.build(True)
    # ---- data + Kalman pin ----
    kal = {}
    for T in (100, 200, 1000):
        _, y = ob.simulate(ob.LG1D, LG, T, 1998)
        _, _, kf = kalman.log_likelihood(y, *LG[:4], x0=LG[4], sigma0=LG[5], predict_first=False)
        _, _, kf_lit = kalman.log_likelihood(y, *LG[:4], x0=LG[4], sigma0=LG[5], predict_first=True)
        kal["T%d" % T] = dict(logZ_kf=kf, logZ_kf_literal=kf_lit, y_head=[float(v).hex() for v in y[:8]],
                              y_sum=float(np.sum(y)).hex())
    with open(os.path.join(OUT, "kalman_lg.json"), "w") as f:
        json.dump(dict(params=LG, sim_seed=1998, cases=kal), f, indent=1)
    # ---- filter vectors ----
    arrs = {}
    meta = {}
    for name, model, raw, n, seg, seed, stream, T in CASES:
        _, y = ob.simulate(model, raw, T, 1998)
        flt = ob.Filter(model, raw, n, seg=seg, seed=seed, stream=stream)
        logZ, lm, es = flt.log_likelihood(y, trace=True)
        x, w, a, logw = flt.state()
        C, m, S, hi, lo = flt.weights_raw()
        meta[name] = dict(model=model, raw=raw, n=n, seg=seg, seed=seed, stream=stream, T=T, logZ=float(logZ).hex())
        arrs[name + "/y"] = y
        arrs[name + "/logmu"] = lm
        arrs[name + "/ess"] = es
        arrs[name + "/x"] = x
        arrs[name + "/w"] = w
        arrs[name + "/anc"] = a.astype(np.int32)
        arrs[name + "/C"] = C
        arrs[name + "/m"] = m
        arrs[name + "/S"] = S
        arrs[name + "/S2hi"] = hi
        arrs[name + "/S2lo"] = lo
    # ---- stand-alone normalize / resample ----
    rng = np.random.default_rng(42)
    logw = rng.normal(size=777) * 3.0 - 50.0
    logw[5] = -np.inf
    lmu, w, ess = ob.normalize(logw)
    arrs["normalize/logw"] = logw
    arrs["normalize/w"] = w
    arrs["normalize/out"] = np.array([lmu, ess])
    a = ob.resample(w, 2000, seed=17, stream=2, t=9)
    arrs["resample/a"] = a.astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "filter_vectors.npz"), **arrs)
    with open(os.path.join(OUT, "filter_vectors.json"), "w") as f:
        json.dump(meta, f, indent=1)
    sampler_vectors()
    print("wrote", sorted(os.listdir(OUT)))


def hexes(a):
    return [float(v).hex() for v in np.asarray(a, dtype=np.float64).ravel()]


def sampler_vectors():
    import io
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import sequential_monte_carlo_amd as smc
    from oracle_backend import OracleBackend
    out = {}
    # ---- PMMH pieces (rejuvenate!, smc_samplers.jl:114-129) ----
    rng = np.random.default_rng(7)
    A = rng.normal(size=(4, 4))
    chol = np.linalg.cholesky(A @ A.T + 0.2 * np.eye(4))
    theta = rng.normal(size=4)
    out["pmmh"] = dict(seed=987654321, stream=41, chol=hexes(chol), theta=hexes(theta), scale=1.5,
                       proposals={str(c): hexes(ob.pmmh_propose(987654321, 41, c, theta, chol, 1.5)) for c in (0, 1, 2)},
                       log_uniform={str(c): float(ob.pmmh_log_uniform(987654321, 41, c)).hex() for c in (0, 1, 2)})
    pri = {"uniform": (1, [0.0, 2.0, 0, 0, 0]), "normal": (2, [3.0, 2.0, 0, 0, 0]),
           "truncnormal": (3, [0.0, 1.0, -1.0, 1.0, float(np.log(0.6826894921370859))]), "lognormal": (4, [0.0, 1.0, 0, 0, 0])}
    xs = [-1.5, -0.3, 0.0, 0.4, 1.0, 1.7, 2.0, 2.5]
    out["prior_logpdf"] = {k: dict(family=f, par=par, x=xs, logpdf=hexes([ob.prior_logpdf(f, par, x) for x in xs])) for k, (f, par) in pri.items()}
    # ---- whole sampler runs over the oracle backend (ThetaMap path) ----
    LGd = dict(A=0.5, B=1.0, Q=0.9, R=0.8)
    prior = smc.product_distribution([smc.TruncatedNormal(0, 1, -1, 1), smc.LogNormal(), smc.LogNormal()])
    tmap = smc.ThetaMap(1, [0, -1, 1, 2, -1, -1], [0.0, 1.0, 0.0, 0.0, 0.0, 1.0])

    def mod(th):
        return smc.UnivariateLinearGaussian(A=th[0], B=1.0, Q=th[1], R=th[2])
    _, y = smc.simulate(smc.UnivariateLinearGaussian(**LGd), 16, seed=1998)
    s = smc.SMC(128, 16, mod, prior, 2, 0.5, seed=21, backend=OracleBackend(), theta_map=tmap)
    buf = io.StringIO()
    stages = smc.density_tempered(s, y, verbose=True, out=buf)
    out["density_tempered_lg"] = dict(N=128, M=16, T=16, chain=2, seed=21, text=buf.getvalue(), theta=hexes(s.theta), logZ=hexes(s.logZ),
                                      xi=hexes([st[0] for st in stages]), psteps=int(s.psteps), psteps_skipped=int(s.psteps_skipped))
    s = smc.SMC(128, 16, mod, prior, 2, 0.6, seed=22, backend=OracleBackend(), theta_map=tmap)
    buf = io.StringIO()
    smc.smc2(s, y)
    smc.smc2_run(s, y, 2, 16, window=4, verbose=True, out=buf)
    x, w, _ = s._main.state()
    out["smc2_lg"] = dict(N=128, M=16, T=16, chain=2, seed=22, ess_threshold=0.6, window=4, text=buf.getvalue(), theta=hexes(s.theta),
                          logZ=hexes(s.logZ), omega=hexes(s.omega), x_sum=float(np.sum(x)).hex(), w_head=hexes(w[:, :4]), psteps=int(s.psteps))
    with open(os.path.join(OUT, "sampler_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
