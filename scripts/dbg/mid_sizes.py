import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from sequential_monte_carlo_amd import _lib as L
LG = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
T = 200
_, y = L.simulate(1, LG, T, 1998)
n = int(sys.argv[1]); seg = int(sys.argv[2])
h = L.Handle(1, 1, n, seg=seg, seed=1); h.set_params(LG)
h.log_likelihood(y[:20]); best = 1e9
for _ in range(5):
    z = h.log_likelihood(y); best = min(best, h.elapsed_ms())
print("n=2^%d seg=%d np=%s: %.2f us/step logZ=%.4f" % (int(np.log2(n)), seg, os.environ.get("SMC_NP", "def"), best / T * 1e3, z[0]))
