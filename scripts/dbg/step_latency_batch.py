"""smc_step (results through tickets in pinned memory) for batches: latency per call against the number of filters."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from sequential_monte_carlo_amd import _lib as L
LG = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
_, y = L.simulate(1, LG, 1001, 1998)
for nth in (1, 8, 64, 512, 4096):
    h = L.Handle(1, nth, 1024, seed=3)
    h.set_params(np.tile(LG, (nth, 1)))
    h.init(float(y[0]))
    for rep in range(3):
        t0 = time.perf_counter()
        for t in range(1, 301):
            h.step(float(y[t]))
        dt = (time.perf_counter() - t0) / 300 * 1e6
    t0 = time.perf_counter()
    for t in range(301, 601):
        h.step_window(y[t:t + 1]); h.step_commit(1)
    dw = (time.perf_counter() - t0) / 300 * 1e6
    print("n_theta=%-5d smc_step %.1f us per call; step_window(1) + commit %.1f us" % (nth, dt, dw), flush=True)
