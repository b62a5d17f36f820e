"""What the per-step summaries cost inside the LDS-resident kernel: a lone filter and a batch of 512, Nx = 1024, T = 1000."""
import sys; sys.path.insert(0, "/root/repo")
import time
import numpy as np
import sequential_monte_carlo_amd as smc
from sequential_monte_carlo_amd import _lib as L
LGR = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
m = smc.UnivariateLinearGaussian(A=0.5, B=1.0, Q=0.9, R=0.8)
_, y = smc.simulate(m, 1000)
for nth in (1, 512):
    h = L.Handle(1, nth, 1024, seed=3)
    h.set_params(np.tile(LGR, (nth, 1)))
    for name, ps, mom in (("none", None, False), ("moments", None, True), ("1 level", [0.5], False), ("3 levels", [0.25, 0.5, 0.75], False),
                          ("3 levels + moments", [0.25, 0.5, 0.75], True), ("7 levels", [0.0, 0.05, 0.25, 0.5, 0.75, 0.999, 1.0], False)):
        h.set_summaries(ps, 0, moments=mom)
        ts = []
        for rep in range(6):
            t0 = time.perf_counter(); h.log_likelihood(y); ts.append((time.perf_counter() - t0) * 1e3)
        print("n_theta=%d %-20s %.2f ms per call of 1000 steps (%.2f us per step)" % (nth, name, min(ts), min(ts)), flush=True)
