import sys, os; sys.path.insert(0, "/root/repo")
import numpy as np
from sequential_monte_carlo_amd import _lib as L
MODEL = int(os.environ.get("MODEL", "1"))
RAW = {1: [0.5, 1.0, 0.9, 0.8, 0.0, 1.0], 2: [-1.0, 0.95, 0.25], 3: [0.2, 0.2, 3.0, 0.0, 0.0]}[MODEL]
_, y = L.simulate(MODEL, RAW, 100, 1998)
for nx in (256, 512, 2048):
    for nth in (384, 512, 640, 768, 1024, 1536):
        h = L.Handle(MODEL, nth, nx, seed=1); h.set_params(np.tile(RAW, (nth, 1)))
        h.log_likelihood(y[:8]); h.log_likelihood(y); ms = h.elapsed_ms()
        print("model %d n_theta=%d Nx=%d: %.3e p-steps/s" % (MODEL, nth, nx, nth * nx * 100 / ms * 1e3), flush=True)
