import sys; sys.path.insert(0, "/root/repo")
import numpy as np
import sequential_monte_carlo_amd as smc
from sequential_monte_carlo_amd import _lib as L
LGR = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
m = smc.UnivariateLinearGaussian(A=0.5, B=1.0, Q=0.9, R=0.8)
_, y = smc.simulate(m, 300)
h = L.Handle(1, 1, 2**20, seed=3)
h.set_params(np.array([LGR]))
h.set_summaries([0.25, 0.5, 0.75], 0, moments=True)
for rep in range(3):
    h.log_likelihood(y)
