"""Diagnostic: the batched kernel with the same active filters launched as grids of different sizes / with and without the
active-first order (FilterView::order).  The first rounds repeat ONE configuration: the times fall from run to run as the
clocks come up, which is not an effect of the grid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sequential_monte_carlo_amd import _lib as L
LG = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
_, y = L.simulate(1, LG, 200, 1998)
act = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for nth in (act, act, act, act, act + 128, act, 2 * act, act, act + 128, act):
    h = L.Handle(1, nth, 1024, seed=1); h.set_params(np.tile(LG, (nth, 1)))
    h.log_likelihood(y)
    out = []
    for kind in ("none", "front") if nth == act else ("front",):
        m = np.ones(nth, dtype=np.uint8); m[:act] = 0
        h.set_skip(None if kind == "none" else m)
        ts = []
        for _ in range(7):
            h.log_likelihood(y); ts.append(h.elapsed_ms())
        out.append("%s min %.3f med %.3f" % (kind, min(ts), sorted(ts)[3]))
    print("active %d grid %4d: %s" % (act, nth, "  ".join(out)))
    h.close()
