"""How long a lone small filter takes: log_likelihood(Nx, y, model), T = 1000, one filter (the README's shape)."""
import sys; sys.path.insert(0, "/root/repo")
import time
import sequential_monte_carlo_amd as smc
m = smc.UnivariateLinearGaussian(A=0.5, B=1.0, Q=0.9, R=0.8)
_, y = smc.simulate(m, 1000)
for n in (1024, 2048, 4096, 8192, 16384, 65536, 2**18):
    ts = []
    for rep in range(5):
        t0 = time.perf_counter(); x, w, logZ = smc.log_likelihood(n, y, m, seed=5); ts.append((time.perf_counter() - t0) * 1e3)
    print("log_likelihood Nx=%d T=1000: %s ms per call" % (n, " ".join("%.2f" % t for t in ts)), flush=True)
# the same lone filters cut into short segments (seg is an argument of smc_create / log_likelihood): many workgroups, one launch per step
for n in (2048, 4096, 8192):
    for seg in (256, 512, 1024):
        if seg >= n:
            continue
        ts = []
        for rep in range(4):
            t0 = time.perf_counter(); x, w, logZ = smc.log_likelihood(n, y, m, seed=5, seg=seg); ts.append((time.perf_counter() - t0) * 1e3)
        print("log_likelihood Nx=%d seg=%d T=1000: %s ms per call, logZ %.3f" % (n, seg, " ".join("%.2f" % t for t in ts), logZ), flush=True)
