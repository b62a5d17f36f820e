"""One LG filter per size with the optional outputs switched on (T = 100): traces (sums of squares at every step), ancestors,
the systematic resampler - looking for cliffs against the plain run."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from sequential_monte_carlo_amd import _lib as L
LG = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
T = 100
_, y = L.simulate(1, LG, T, 1998)
for nth, lg in ((1, 14), (1, 18), (1, 20), (1, 22), (1, 24), (512, 10), (64, 16)):
    nx = 1 << lg
    row = []
    for name, flags, trace in (("plain", 0, False), ("trace", 0, True), ("anc", L.FLAG_ANCESTORS, False), ("sys", L.FLAG_SYSTEMATIC, False), ("sys+trace", L.FLAG_SYSTEMATIC, True)):
        h = L.Handle(1, nth, nx, seed=1, flags=flags)
        h.set_params(np.tile(LG, (nth, 1)))
        h.log_likelihood(y[:8], trace=trace); h.log_likelihood(y, trace=trace)
        ms = h.elapsed_ms()
        row.append("%s %.3e" % (name, nth * nx * T / ms * 1e3))
        h.close()
    print("n_theta=%-4d Nx=2^%-2d: %s" % (nth, lg, " | ".join(row)), flush=True)
