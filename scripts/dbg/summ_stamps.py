import sys; sys.path.insert(0, "/root/repo")
import numpy as np
import sequential_monte_carlo_amd as smc
from sequential_monte_carlo_amd import _lib as L
LGR = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
m = smc.UnivariateLinearGaussian(A=0.5, B=1.0, Q=0.9, R=0.8)
_, y = smc.simulate(m, 1000)
for nth in (1, 512):
    h = L.Handle(1, nth, 1024, seed=3)
    h.set_params(np.tile(LGR, (nth, 1)))
    h.set_summaries([0.0, 0.05, 0.25, 0.5, 0.75, 0.999, 1.0], 0, moments=False)
    h.log_likelihood(y); h.log_likelihood(y)
    q, _, _ = h.get_summaries(1000)
    d = q[100:, 0, 1:6] * 10.0   # ns (100 MHz counter)
    per = np.diff(q[100:, 0, 6]) * 10.0
    print("n_theta=%d: phases ns (minmax+barrier | hist+barrier | pick+barrier | compact+barrier | rank+store):" % nth, d.mean(axis=0).round(0), "sum", d.mean(axis=0).sum().round(0), "step period", per.mean().round(0))
