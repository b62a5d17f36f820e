import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from sequential_monte_carlo_amd import _lib as L
import os
MODEL = int(os.environ.get("MODEL", "1"))
LG = {1: [0.5, 1.0, 0.9, 0.8, 0.0, 1.0], 2: [-1.0, 0.95, 0.25], 3: [0.2, 0.2, 3.0, 0.0, 0.0]}[MODEL]
_, y = L.simulate(MODEL, LG, 100, 1998)
for nth in (64, 512, 4096):
    for nx in (2048, 4096):
        h = L.Handle(MODEL, nth, nx, seed=1); h.set_params(np.tile(LG, (nth, 1)))
        h.log_likelihood(y[:8]); h.log_likelihood(y); ms = h.elapsed_ms()
        print("n_theta=%d Nx=%d: %.3e p-steps/s" % (nth, nx, nth * nx * 100 / ms * 1e3), flush=True)
