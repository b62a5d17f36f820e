import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sequential_monte_carlo_amd import _lib as L
model = int(sys.argv[1]); nx = int(sys.argv[2]); nth = int(sys.argv[3])
raw = {1: [0.5, 1.0, 0.9, 0.8, 0.0, 1.0], 2: [-1.0, 0.95, 0.25], 3: [0.2, 0.2, 3.0, 0.0, 0.0]}[model]
_, y = L.simulate(model, raw, 200, 1998)
h = L.Handle(model, nth, nx, seed=1); h.set_params(np.tile(raw, (nth, 1)))
h.log_likelihood(y[:10]); z = h.log_likelihood(y)
print("model", model, "nx", nx, "nth", nth, "RES_NP", os.environ.get("SMC_RES_NP"), "ms %.4f" % h.elapsed_ms(), "p-steps/s %.3e" % (nth * nx * 200 / h.elapsed_ms() * 1e3))
