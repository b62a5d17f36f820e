"""C2-shaped run with opt-in systematic resampling vs the default multinomial (same box, same call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sequential_monte_carlo_amd import _lib as L
T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
LG = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
_, y = L.simulate(1, LG, T, 1998)
for name, nth, nx, fl in (("c2 multinomial", 1, 1 << 20, 0), ("c2 systematic", 1, 1 << 20, L.FLAG_SYSTEMATIC),
                          ("c2 systematic n=10^6", 1, 1000000, L.FLAG_SYSTEMATIC),
                          ("c4 multinomial", 512, 1024, 0), ("c4 systematic", 512, 1024, L.FLAG_SYSTEMATIC),
                          ("c4 systematic n=1000", 512, 1000, L.FLAG_SYSTEMATIC)):
    h = L.Handle(1, nth, nx, seed=1, flags=fl); h.set_params(np.tile(LG, (nth, 1)))
    h.log_likelihood(y[:10]); z = h.log_likelihood(y)
    print("%-24s logZ %.4f  %.3f ms  %.2f us/step  %.3e p-steps/s" % (name, z[0], h.elapsed_ms(), h.elapsed_ms() / T * 1e3, nth * nx * T / h.elapsed_ms() * 1e3))
    h.close()
