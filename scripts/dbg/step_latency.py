"""Latency of the step API (one bootstrap_filter! per call, README.md:33-61) for a lone filter of 1024 particles."""
import sys; sys.path.insert(0, "/root/repo")
import time
import numpy as np
import sequential_monte_carlo_amd as smc
from sequential_monte_carlo_amd import _lib as L
LGR = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
m = smc.UnivariateLinearGaussian(A=0.5, B=1.0, Q=0.9, R=0.8)
_, y = smc.simulate(m, 2001)
for n in (1024, 2**16):
    h = L.Handle(1, 1, n, seed=3)
    h.set_params(np.array([LGR]))
    h.init(float(y[0]))
    for rep in range(3):
        t0 = time.perf_counter()
        for t in range(1, 1001):
            h.step(float(y[t]))
        dt = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    for t in range(1, 1001):
        h.step(float(y[t])); h.quantiles([0.25, 0.5, 0.75]); h.moments()
    d3 = (time.perf_counter() - t0) * 1e3
    print("Nx=%d: smc_step %.1f us per call; step + quantiles + moments %.1f us per observation" % (n, dt, d3), flush=True)
