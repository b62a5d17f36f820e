"""Rao-Blackwellised particle filter for the UCSV model (src/state_space_models.jl:215-263), numpy.

TEST INFRASTRUCTURE: an INDEPENDENT unbiased estimator of the same likelihood the bootstrap filter estimates.  Given the two
log-volatility paths the pair (x, y) is linear-Gaussian, so particles carry only (log_s_eps, log_s_eta) plus the Kalman mean
and variance of x (kalman_filter.jl:39-50 with A = B = 1 and time-varying Q, R).  Conventions as in the reference:
  x'         ~ N(x, exp(log_s_eps / 2))   with the PREVIOUS log-volatility        (:233-238)
  log_s_eps' ~ N(log_s_eps, gamma_eps),  log_s_eta' ~ N(log_s_eta, gamma_eta)   gammas are standard deviations (:239-240)
  y          ~ N(x, exp(log_s_eta / 2))                                          (:244-247)
  x_1 ~ N(x0, exp(log_s_eps0 / 2)), log-vols_1 ~ N(log-vols_0, gamma)             (:249-259)
Both filters are unbiased for p(y), so the means of exp(logZ) over independent runs must agree within Monte-Carlo error:
that cross-checks every convention above without a closed form."""
import numpy as np


def log_likelihood(y, gamma_eps, gamma_eta, x0, lse0, lsn0, n=20000, rng=None):
    rng = np.random.default_rng() if rng is None else rng
    y = np.asarray(y, dtype=np.float64)
    lse_prev = np.full(n, lse0)
    lsn_prev = np.full(n, lsn0)
    m = np.full(n, float(x0))
    P = np.zeros(n)
    logZ = 0.0
    for t, yt in enumerate(y):
        if t == 0:
            P = np.full(n, np.exp(lse0))                       # Var x_1 = exp(lse0 / 2)^2
        else:
            P = P + np.exp(lse_prev)                           # predict with the previous log-volatility
        lse = lse_prev + gamma_eps * rng.standard_normal(n)
        lsn = lsn_prev + gamma_eta * rng.standard_normal(n)
        R = np.exp(lsn)
        s = P + R
        dy = yt - m
        logw = -0.5 * (np.log(2.0 * np.pi) + np.log(s) + dy * dy / s)
        K = P / s
        m = m + K * dy
        P = P - K * P
        mx = logw.max()
        w = np.exp(logw - mx)
        logZ += mx + np.log(w.mean())
        a = rng.choice(n, size=n, replace=True, p=w / w.sum())  # multinomial resampling every step, like the reference
        lse_prev, lsn_prev, m, P = lse[a], lsn[a], m[a], P[a]
    return float(logZ)
