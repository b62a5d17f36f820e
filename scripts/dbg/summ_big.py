"""Per-step summaries of a big (multi-segment) filter: cost per step of the trailing selection kernels."""
import sys; sys.path.insert(0, "/root/repo")
import time
import numpy as np
import sequential_monte_carlo_amd as smc
from sequential_monte_carlo_amd import _lib as L
LGR = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
m = smc.UnivariateLinearGaussian(A=0.5, B=1.0, Q=0.9, R=0.8)
_, y = smc.simulate(m, 300)
for n in (2**16, 2**20, 2**22):
    h = L.Handle(1, 1, n, seed=3)
    h.set_params(np.array([LGR]))
    for name, ps, mom in (("none", None, False), ("moments", None, True), ("1 level", [0.5], False), ("3 levels + moments", [0.25, 0.5, 0.75], True)):
        h.set_summaries(ps, 0, moments=mom)
        ts = []
        for rep in range(4):
            t0 = time.perf_counter(); h.log_likelihood(y); ts.append((time.perf_counter() - t0) * 1e3)
        print("Nx=%d %-20s %.2f ms per call of 300 steps (%.1f us per step)" % (n, name, min(ts), min(ts) / 0.3), flush=True)
