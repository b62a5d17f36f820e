"""How the batched LDS-resident kernel behaves when a fraction of its filters is skipped (proposals outside the
prior's support): random mask vs the same number of active filters packed at the front of the grid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sequential_monte_carlo_amd import _lib as L
LG = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]; UC = [0.2, 0.2, 3.0, 0.0, 0.0]
nth = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rng = np.random.default_rng(0)
for model, raw in ((1, LG), (3, UC)):
    _, y = L.simulate(model, raw, 200, 1998)
    h = L.Handle(model, nth, 1024, seed=1); h.set_params(np.tile(raw, (nth, 1)))
    h.log_likelihood(y)
    for frac in (0.0, 0.25, 0.5, 0.75):
        A = int(round(nth * (1 - frac)))
        res = []
        for kind in ("random", "front"):
            m = np.ones(nth, dtype=np.uint8)
            if kind == "random": m[rng.permutation(nth)[:A]] = 0
            else: m[:A] = 0
            h.set_skip(m if frac > 0 else None)
            ts = []
            for _ in range(5):
                h.log_likelihood(y); ts.append(h.elapsed_ms())
            res.append(min(ts))
        print("model %d nth %d active %4d: random mask %.3f ms  packed-front %.3f ms  (%.3e / %.3e executed p-steps/s)" % (
            model, nth, A, res[0], res[1], A * 1024 * 200 / res[0] * 1e3, A * 1024 * 200 / res[1] * 1e3))
    h.close()
