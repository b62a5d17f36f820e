"""Deterministic grid (quadrature) filter for ONE-dimensional state-space models: the likelihood p(y_1..T) by the forward
recursion on a fine grid of the state.  TEST INFRASTRUCTURE: an independent known answer for the models that have no closed
form - the stochastic-volatility model of BASELINE configs[2] (SURVEY A7': x_1 ~ N(mu, sigma^2/(1-rho^2)),
x_t ~ N(mu + rho (x_{t-1} - mu), sigma), y_t ~ N(0, exp(x_t / 2))) - the way the reference's Kalman filter
(src/kalman_filter.jl:29-70) is one for the linear-Gaussian model.  A bootstrap filter is unbiased for exp(logZ), so
E[exp(logZ_PF - logZ_grid)] = 1 pins the oracle's propagate / weigh / resample chain for a non-Gaussian observation density.
Error of the quadrature itself: compare two grid sizes (the test does)."""
import numpy as np


def _norm_logpdf(x, mean, sd):
    z = (x - mean) / sd
    return -0.5 * (z * z + np.log(2.0 * np.pi)) - np.log(sd)


def sv_log_likelihood(y, mu, rho, sigma, n_grid=3001, width=9.0):
    """log p(y) of the stochastic-volatility model on a uniform grid of n_grid points over mu +- width * stationary sd
    (trapezoid weights)."""
    y = np.asarray(y, dtype=np.float64)
    s0 = sigma / np.sqrt(1.0 - rho * rho)
    x = np.linspace(mu - width * s0, mu + width * s0, n_grid)
    h = x[1] - x[0]
    wq = np.full(n_grid, h)
    wq[0] = wq[-1] = 0.5 * h
    # transition density p(x' | x) on the grid: row = from, column = to
    P = np.exp(_norm_logpdf(x[None, :], mu + rho * (x[:, None] - mu), sigma))
    logp = _norm_logpdf(x, mu, s0)                       # log p(x_1)
    logZ = 0.0
    for t, yt in enumerate(y):
        if t > 0:
            # predict: p(x_t | y_1..t-1) = int p(x_t | x) f(x) dx
            f = np.exp(logp - logp.max())
            pred = (f * wq) @ P
            logp = np.log(np.maximum(pred, 1e-300)) + logp.max()
        logp = logp + _norm_logpdf(yt, 0.0, np.exp(0.5 * x))      # weigh with p(y_t | x_t)
        m = logp.max()
        inc = m + np.log(np.sum(np.exp(logp - m) * wq))           # log p(y_t | y_1..t-1)
        logZ += inc
        logp = logp - inc                                         # filtered density, normalised
    return float(logZ)
