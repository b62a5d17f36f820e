"""Sanity at sizes beyond the BASELINE configs: Nx = 2^24 single filter, N_theta = 4096 batched."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sequential_monte_carlo_amd import _lib as L
LG = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
_, y = L.simulate(1, LG, 50, 1998)
kz = L.kalman_log_likelihood(LG, y)[0, 2]
for n, seg in ((1 << 22, 2048), (1 << 24, 8192), ((1 << 24) + 12345, 8192)):
    h = L.Handle(1, 1, n, seg=seg, seed=1); h.set_params(LG)
    z, lm, es = h.log_likelihood(y, trace=True); ms = h.elapsed_ms()
    x, w, _ = h.state(want_anc=False)
    print("Nx=%d seg=%d: logZ=%.5f (Kalman %.5f) %.2f ms %.3e p-steps/s  sum w=%.12f ess/N=%.3f" % (
        n, seg, z[0], kz, ms, n * 50 / ms * 1e3, w.sum(), es[-1, 0] / n))
    assert abs(z[0] - kz) < 0.05 and abs(w.sum() - 1) < 1e-9
    h.close()
nth = 4096
h = L.Handle(1, nth, 1024, seed=1); h.set_params(np.tile(LG, (nth, 1)))
_, y2 = L.simulate(1, LG, 200, 1998)
h.log_likelihood(y2[:5]); z = h.log_likelihood(y2); ms = h.elapsed_ms()
kz2 = L.kalman_log_likelihood(LG, y2)[0, 2]
print("Ntheta=4096 x 1024 T=200: %.2f ms %.3e p-steps/s mean logZ %.4f (Kalman %.4f) sd %.4f" % (ms, nth * 1024 * 200 / ms * 1e3, z.mean(), kz2, z.std()))
assert abs(z.mean() + 0.5 * z.var() - kz2) < 0.05
h.close()
nth = 512
h = L.Handle(1, nth, 8192, seed=1); h.set_params(np.tile(LG, (nth, 1)))
h.log_likelihood(y2[:5]); z = h.log_likelihood(y2); ms = h.elapsed_ms()
print("Ntheta=512 x 8192 T=200 (author's 2nd workload shape): resident=%d %.2f ms %.3e p-steps/s" % (h.resident, ms, nth * 8192 * 200 / ms * 1e3))
h.close()
