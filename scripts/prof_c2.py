"""Profiling driver: C2-shaped run (LG, Nx=2^20) with a short series, for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sequential_monte_carlo_amd import _lib as L
T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
wl = sys.argv[3] if len(sys.argv) > 3 else "c2"
LG = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
SV = [-1.0, 0.95, 0.25]; UC = [0.2, 0.2, 3.0, 0.0, 0.0]
if wl in ("c2", "c3"):
    m, raw = (1, LG) if wl == "c2" else (2, SV)
    _, y = L.simulate(m, raw, T, 1998)
    h = L.Handle(m, 1, 1 << 20, seg=seg, seed=1); h.set_params(raw)
else:
    m, raw = (1, LG) if wl == "c4" else (3, UC)
    _, y = L.simulate(m, raw, T, 1998)
    h = L.Handle(m, 512, 1024, seed=1); h.set_params(np.tile(raw, (512, 1)))
h.log_likelihood(y)
z = h.log_likelihood(y)
print("logZ", z[0], "ms", h.elapsed_ms(), "p-steps/s %.3e" % (h.n_theta * h.n_x * T / h.elapsed_ms() * 1e3))
