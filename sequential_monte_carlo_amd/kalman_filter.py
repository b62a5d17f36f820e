"""Host mirror of src/kalman_filter.jl for the univariate LinearModel, batched on the GPU.

    log_likelihood_kalman(y, model)   -> (x_T, Sigma_T, logZ)       kalman_filter.jl:55-70

`model` may be a list of LinearModels (one lane per parameter row): the O(1)-per-theta inner "filter"
of the IBIS sampler (src/ibis.jl:134-189).  predict_first=True reproduces the reference loop literally
(it predicts before the first update); the default starts from x_1 ~ N(x0, sigma0) like
bootstrap_filter, so that it is the exact value the particle estimate converges to.
"""
import numpy as np

from . import _lib
from .models import LinearModel


def log_likelihood_kalman(y, model, predict_first=False, device=0):
    single = isinstance(model, LinearModel)
    models = [model] if single else list(model)
    if not all(isinstance(m, LinearModel) for m in models):
        raise TypeError("the Kalman filter needs LinearModel(s)")
    out = _lib.kalman_log_likelihood(np.array([m.raw() for m in models]), y, predict_first, device)
    if single:
        return float(out[0, 0]), float(out[0, 1]), float(out[0, 2])
    return out[:, 0], out[:, 1], out[:, 2]
