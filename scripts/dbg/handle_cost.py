"""What a handle costs: smc_create + smc_destroy, and a whole log_likelihood(1024, y[:100], model) call from Python (BASELINE configs[0])."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import sequential_monte_carlo_amd as smc
from sequential_monte_carlo_amd import _lib as L
LG = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
m = smc.UnivariateLinearGaussian(A=0.5, B=1.0, Q=0.9, R=0.8)
_, y = smc.simulate(m, 100)
for n, nth in ((1024, 1), (1024, 512), (2**20, 1)):
    for rep in range(3):
        t0 = time.perf_counter()
        for it in range(20):
            h = L.Handle(1, nth, n, seed=3); h.close()
        dt = (time.perf_counter() - t0) / 20 * 1e3
    print("smc_create + smc_destroy, %d x %d: %.3f ms" % (nth, n, dt), flush=True)
for rep in range(3):
    t0 = time.perf_counter()
    for it in range(20):
        x, w, z = smc.log_likelihood(1024, y, m, seed=5)
    dt = (time.perf_counter() - t0) / 20 * 1e3
print("smc.log_likelihood(1024, y[:100], m) from Python: %.3f ms per call" % dt)
h = L.Handle(1, 1, 1024, seed=3); h.set_params(LG)
t0 = time.perf_counter()
for it in range(50): h.log_likelihood(y)
print("the same on an existing handle: %.3f ms per call" % ((time.perf_counter() - t0) / 50 * 1e3))
