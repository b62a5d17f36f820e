"""Edge shapes: very long series, a single observation, the largest batch, awkward particle counts - do they run, at what rate."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from sequential_monte_carlo_amd import _lib as L
LG = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
def run(nth, nx, T, trace=False, model=1, raw=LG):
    _, y = L.simulate(model, raw, T, 1998)
    h = L.Handle(model, nth, nx, seed=1); h.set_params(np.tile(raw, (nth, 1)))
    t0 = time.perf_counter(); z = h.log_likelihood(y, trace=trace); dt = time.perf_counter() - t0
    z0 = z[0] if trace else z
    print("n_theta=%-6d Nx=%-9d T=%-7d trace=%d: %.2f ms, %.3e p-steps/s, logZ[0]/T %.4f, seg %d nseg %d" % (nth, nx, T, trace, dt * 1e3, nth * nx * T / dt, z0[0] / T, h.seg, h.nseg), flush=True)
    h.close()
run(1, 1024, 1); run(1, 2**20, 1); run(1, 1024, 200000); run(1, 2**16, 20000, trace=True); run(1, 2**20, 12000)
run(65535, 256, 50); run(20000, 1024, 50); run(1, 3000000, 100); run(1, 1048577, 100); run(3, 5000001, 50)
run(1, 2**20, 2000, model=3, raw=[0.2, 0.2, 3.0, 0.0, 0.0]); run(512, 1000, 200, model=3, raw=[0.2, 0.2, 3.0, 0.0, 0.0])
