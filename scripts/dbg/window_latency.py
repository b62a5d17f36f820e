"""smc_step_window + smc_step_commit of the online sampler's shape (512 filters of 1024 particles, 16 steps per window)."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import sequential_monte_carlo_amd as smc
from sequential_monte_carlo_amd import _lib as L
LG = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
_, y = L.simulate(1, LG, 4000, 1998)
for nth in (512,):
    h = L.Handle(1, nth, 1024, seed=3)
    h.set_params(np.tile(LG, (nth, 1)))
    h.init(float(y[0]))
    for k in (1, 4, 16):
        t = 1
        for rep in range(3):
            t0 = time.perf_counter()
            for it in range(60):
                h.step_window(y[t:t + k]); h.step_commit(k); t += k
            dt = (time.perf_counter() - t0) / 60 * 1e6
        print("n_theta=%d window of %2d steps + commit: %.1f us per window (%.1f us per step)" % (nth, k, dt, dt / k), flush=True)
