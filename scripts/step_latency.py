import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sequential_monte_carlo_amd import _lib as L
LG = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
_, y = L.simulate(1, LG, 2100, 1998)
for nth, nx in ((1, 256), (512, 1024), (1, 1 << 16)):
    h = L.Handle(1, nth, nx, seed=1); h.set_params(np.tile(LG, (nth, 1)))
    h.init(y[0])
    for t in range(1, 100): h.step(y[t])
    t0 = time.perf_counter()
    for t in range(100, 2100): h.step(y[t])
    dt = (time.perf_counter() - t0) / 2000
    print("n_theta %d nx %d: %.2f us per smc_step call (device %.2f us)" % (nth, nx, dt * 1e6, h.elapsed_ms() * 1e3))
    h.close()
