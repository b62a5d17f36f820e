"""State-space model families of the reference (src/state_space_models.jl), as parameter holders.

The reference's contract is 4 generic functions per model (preallocate / initial_dist / transition /
observation, ssm.jl:30-42) returning Distributions.jl objects that the filters call per particle.
A GPU cannot call host closures, so the drop-in boundary is "enumerated family + parameter row":
each class below carries `model_id` and `raw()` (the row handed to smc_set_params); the per-particle
arithmetic lives in csrc/smc_spec.h (model_initial / model_transition / model_logobs).
"""
import numpy as np

from . import _lib


class StateSpaceModel:
    """abstract type StateSpaceModel   (ssm.jl:9)"""
    model_id = None
    dim = None

    def raw(self):
        raise NotImplementedError


class LinearModel(StateSpaceModel):
    """LinearModel{Float64,...} (ssm.jl:46-58): x[t] ~ N(A x[t-1], Q), y[t] ~ N(B x[t], R),
    x[1] ~ N(x0, sigma0).  Q, R, sigma0 are VARIANCES (ssm.jl:93,102,108)."""
    model_id = _lib.MODEL_LG1D
    dim = 1

    def __init__(self, A, B, Q, R, x0=0.0, sigma0=1.0):
        self.A, self.B, self.Q, self.R, self.x0, self.sigma0 = map(float, (A, B, Q, R, x0, sigma0))
        if not (self.Q > 0 and self.R > 0 and self.sigma0 > 0):
            raise ValueError("Q, R and sigma0 are variances and must be positive")

    def raw(self):
        return [self.A, self.B, self.Q, self.R, self.x0, self.sigma0]


def UnivariateLinearGaussian(*, A, B, Q, R, x0=0.0, sigma0=1.0):
    """UnivariateLinearGaussian(;A,B,Q,R,x0=0.0,sigma0=1.0)   (ssm.jl:74-77)"""
    return LinearModel(A, B, Q, R, x0, sigma0)


def unobserved_components(*, sigma_eps, sigma_eta, x0):
    """local-level UC model (ssm.jl:119-128): A = B = 1, Q = sigma_eps, R = sigma_eta, sigma0 = sigma_eps."""
    return LinearModel(1.0, 1.0, sigma_eps, sigma_eta, x0, sigma_eps)


class StochasticVolatility(StateSpaceModel):
    """x[1] ~ N(mu, sigma^2/(1-rho^2)), x[t] ~ N(mu + rho (x[t-1]-mu), sigma), y[t] ~ N(0, exp(x[t]/2)).
    Not in the reference's src/ (BASELINE config 3; SURVEY A7'); observation follows ssm.jl:244-247."""
    model_id = _lib.MODEL_SV1D
    dim = 1

    def __init__(self, mu, rho, sigma):
        self.mu, self.rho, self.sigma = map(float, (mu, rho, sigma))
        if not (abs(self.rho) < 1 and self.sigma > 0):
            raise ValueError("need |rho| < 1 and sigma > 0")

    def raw(self):
        return [self.mu, self.rho, self.sigma]


class UCSV(StateSpaceModel):
    """UCSV (ssm.jl:215-263): state (x, log s_eps, log s_eta); x' ~ N(x, exp(log_s_eps/2)) with the
    PREVIOUS log-volatility, log-vols random walks with STD-DEV gamma; y ~ N(x, exp(log_s_eta/2))."""
    model_id = _lib.MODEL_UCSV3D
    dim = 3

    def __init__(self, gamma, x0, log_sigma0):
        self.gamma = (float(gamma[0]), float(gamma[1]))
        self.x0 = float(x0)
        self.log_sigma0 = (float(log_sigma0[0]), float(log_sigma0[1]))
        if not (self.gamma[0] > 0 and self.gamma[1] > 0):
            raise ValueError("gamma must be positive")

    def raw(self):
        return [self.gamma[0], self.gamma[1], self.x0, self.log_sigma0[0], self.log_sigma0[1]]


def unobserved_components_stochastic_volatility(*, x0, gamma_eps, gamma_eta, log_sigma_eps, log_sigma_eta):
    """unobserved_components_stochastic_volatility(;x0,γε,γη,log_σε,log_ση)   (ssm.jl:225-227)"""
    return UCSV((gamma_eps, gamma_eta), x0, (log_sigma_eps, log_sigma_eta))


def simulate(model, T, seed=1998):
    """simulate(rng, model, T) -> (x, y)   (ssm.jl:11-26).  x is [T] for scalar states, [T, d] otherwise.
    Host code (no GPU): the spec's Philox / Box-Muller stream keyed by `seed`."""
    x, y = _lib.simulate(model.model_id, model.raw(), int(T), int(seed))
    return (x[0] if model.dim == 1 else x.T.copy()), y


def params_matrix(models):
    """[n_theta][n_raw] parameter rows of a list of models of one family."""
    if isinstance(models, StateSpaceModel):
        models = [models]
    ids = {m.model_id for m in models}
    if len(ids) != 1:
        raise ValueError("all models of a batch must belong to one family")
    return ids.pop(), np.array([m.raw() for m in models], dtype=np.float64)
