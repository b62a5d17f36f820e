import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from sequential_monte_carlo_amd import _lib as L
UC = [0.2, 0.2, 3.0, 0.0, 0.0]
for model, raw in ((3, UC), (1, [0.5,1,0.9,0.8,0,1])):
  for n, nth in ((2048, 256), (4096, 128), (2048, 1024)):
    _, y = L.simulate(model, raw, 100, 1998)
    h = L.Handle(model, nth, n, seed=1); h.set_params(np.tile(raw, (nth, 1)))
    h.log_likelihood(y); ts = []
    for _ in range(3):
        h.log_likelihood(y); ts.append(h.elapsed_ms())
    print(model, n, nth, "%.3f ms" % min(ts), "%.3e p-steps/s" % (nth * n * 100 / min(ts) * 1e3)); h.close()
