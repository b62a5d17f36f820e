"""Exact Kalman log-likelihood of the univariate linear-Gaussian model (numpy/python floats).

TEST INFRASTRUCTURE: the known-answer pin of the oracle.  Restates the reference's scalar
Kalman step  src/kalman_filter.jl:29-53  and its whole-series loop  :55-70 :

    predict   x- = A x ;  S- = A^2 S + Q
    innovate  s  = B^2 S- + R ;  dy = y - B x-
    update    x+ = x- + (S- B) / s * dy ;  S+ = S- - (S- B)^2 / s
    loglik    -1/2 ( log 2pi + log s + dy^2 / s )

Q, R and sigma0 are variances (src/state_space_models.jl:93,102,108).
NOTE the reference starts the recursion from (x0, sigma0) and *predicts before the first
update* (kalman_filter.jl:39-40 are executed for t = 1 too), so y[1] is scored against N(B A x0, B^2(A^2 sigma0+Q)+R),
whereas bootstrap_filter draws x_1 ~ N(x0, sigma0) directly (particles.jl:97).  The two agree
iff the Kalman recursion is started one step earlier; `log_likelihood(..., predict_first=False)`
is the variant consistent with the particle filter and is the one used as the pin.
"""
import math


def kalman_step(A, B, Q, R, x, S, y, predict=True):
    if predict:
        x = A * x
        S = (A * A) * S + Q
    s = (B * B) * S + R
    dy = y - B * x
    xn = x + (S * B) / s * dy
    Sn = S - (S * B) ** 2 / s
    ll = -0.5 * (math.log(2.0 * math.pi) + math.log(s) + dy / s * dy)
    return xn, Sn, ll


def log_likelihood(y, A, B, Q, R, x0=0.0, sigma0=1.0, predict_first=False):
    """Sum of one-step predictive log densities.  predict_first=True is the literal
    kalman_filter.jl:55-70 loop; False skips the predict at t=1 (x_1 ~ N(x0, sigma0))."""
    x, S, logZ = x0, sigma0, 0.0
    for t, yt in enumerate(y):
        x, S, ll = kalman_step(A, B, Q, R, x, S, float(yt), predict=(predict_first or t > 0))
        logZ += ll
    return x, S, logZ
