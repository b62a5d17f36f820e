"""Stand-alone normalize / resample (particles.jl:5-19) through the C ABI with host buffers: time per call against n."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from sequential_monte_carlo_amd import _lib as L
rng = np.random.default_rng(1)
for lg in (10, 14, 16, 18, 20, 22, 24):
    n = 1 << lg
    logw = rng.normal(size=n)
    for rep in range(3):
        t0 = time.perf_counter(); lm, w, ess = L.normalize(logw); tn = (time.perf_counter() - t0) * 1e3
    for rep in range(3):
        t0 = time.perf_counter(); a = L.resample(w, seed=3); tr = (time.perf_counter() - t0) * 1e3
    print("n=2^%-2d normalize %.3f ms  resample %.3f ms   (host buffers in and out: %.1f / %.1f MB)" % (lg, tn, tr, 16 * n / 1e6, 12 * n / 1e6), flush=True)
