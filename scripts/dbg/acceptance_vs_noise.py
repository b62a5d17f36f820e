import io, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
import sequential_monte_carlo_amd as smc
from test_samplers_cpu import LG_TMAP, lg_mod, lg_prior
from test_gpu_api import _KalmanBackend
_, y = smc.simulate(smc.UnivariateLinearGaussian(A=0.5, B=1.0, Q=1.0, R=1.0), 1000, seed=1998)
for name, N, be, tm in (("pf N=1024", 1024, None, LG_TMAP), ("pf N=256", 256, None, LG_TMAP), ("pf N=64", 64, None, LG_TMAP), ("kalman", 1, _KalmanBackend(), None)):
    s = smc.SMC(N, 512, lg_mod, lg_prior(), 3, 0.5, seed=1, backend=be, theta_map=tm)
    b = io.StringIO(); smc.density_tempered(s, y, verbose=True, out=b)
    print(name); print(b.getvalue(), smc.expected_parameters(s))
