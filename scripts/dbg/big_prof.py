import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from sequential_monte_carlo_amd import _lib as L
LG = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
lg = int(sys.argv[1])
_, y = L.simulate(1, LG, 200, 1998)
h = L.Handle(1, 1, 1 << lg, seed=1)
h.set_params(LG)
for rep in range(2):
    h.log_likelihood(y)
