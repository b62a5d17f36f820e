"""Host-side profile of one online SMC^2 run (README model, Ntheta=512 x Nx=1024, T=200, chain 3)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sequential_monte_carlo_amd as smc

M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N, T, chain = 1024, 200, 3
_, y = smc.simulate(smc.UnivariateLinearGaussian(A=0.5, B=1.0, Q=0.9, R=0.8), T, seed=1998)
prior = smc.product_distribution([smc.TruncatedNormal(0, 1, -1, 1), smc.LogNormal(), smc.LogNormal()])
mod = lambda th: smc.UnivariateLinearGaussian(A=th[0], B=1.0, Q=th[1], R=th[2])
backend = smc.smc_samplers.HipBackend(device=0)
def raw_fn(th):
    m = th.shape[0]
    return 1, np.column_stack([th[:, 0], np.ones(m), th[:, 1], th[:, 2], np.zeros(m), np.ones(m)])
def run(seed):
    s = smc.SMC(N, M, mod, prior, chain, 0.5, seed=seed, backend=backend, raw_fn=raw_fn)
    sink = io.StringIO()
    smc.smc2(s, y)
    for t in range(2, T + 1):
        smc.smc2_step(s, y, t, verbose=False, out=sink)
    return s
run(1)
t0 = time.perf_counter(); s = run(2); dt = time.perf_counter() - t0
print("run %.2f ms, psteps %.3e -> %.3e p-steps/s" % (dt * 1e3, s.psteps, s.psteps / dt))
pr = cProfile.Profile(); pr.enable(); run(3); pr.disable()
o = io.StringIO(); pstats.Stats(pr, stream=o).sort_stats("tottime").print_stats(22); print(o.getvalue()[:4500])
