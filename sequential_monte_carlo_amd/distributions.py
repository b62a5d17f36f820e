"""The few priors the reference's examples use (README.md:81-85, examples/inflation_example.jl:33-37,
234-239), with the three operations the samplers need: rand, logpdf, insupport
(src/smc_samplers.jl:38,116,123).  Host-side numpy; O(n_theta) work per stage."""
import math

import numpy as np
from scipy.special import ndtr, ndtri

# enumerated families of the device-side PMMH (include/smc_hip.h SMC_PRIOR_*): spec() -> (family, 5 parameters)
PRIOR_UNIFORM, PRIOR_NORMAL, PRIOR_TRUNCNORMAL, PRIOR_LOGNORMAL = 1, 2, 3, 4


class Normal:
    def __init__(self, mu=0.0, sigma=1.0):
        self.mu, self.sigma = float(mu), float(sigma)

    def rand(self, rng):
        return self.mu + self.sigma * rng.standard_normal()

    def rand_v(self, rng, m):
        return self.mu + self.sigma * rng.standard_normal(m)

    def logpdf(self, x):
        z = (x - self.mu) / self.sigma
        return -0.5 * (z * z + math.log(2 * math.pi)) - math.log(self.sigma)

    def insupport(self, x):
        return math.isfinite(x)

    def logpdf_v(self, x):
        z = (x - self.mu) / self.sigma
        return -0.5 * (z * z + math.log(2 * math.pi)) - math.log(self.sigma)

    def insupport_v(self, x):
        return np.isfinite(x)

    def spec(self):
        return PRIOR_NORMAL, [self.mu, self.sigma, 0.0, 0.0, 0.0]


class TruncatedNormal:
    """TruncatedNormal(mu, sigma, lo, hi)"""

    def __init__(self, mu, sigma, lo, hi):
        self.mu, self.sigma, self.lo, self.hi = map(float, (mu, sigma, lo, hi))
        self._a, self._b = ndtr((self.lo - self.mu) / self.sigma), ndtr((self.hi - self.mu) / self.sigma)
        self._logz = math.log(self._b - self._a)

    def rand(self, rng):
        u = self._a + (self._b - self._a) * rng.random()
        return min(max(self.mu + self.sigma * float(ndtri(u)), self.lo), self.hi)

    def rand_v(self, rng, m):
        u = self._a + (self._b - self._a) * rng.random(m)
        return np.clip(self.mu + self.sigma * ndtri(u), self.lo, self.hi)

    def logpdf(self, x):
        if not self.insupport(x):
            return -math.inf
        z = (x - self.mu) / self.sigma
        return -0.5 * (z * z + math.log(2 * math.pi)) - math.log(self.sigma) - self._logz

    def insupport(self, x):
        return self.lo <= x <= self.hi

    def insupport_v(self, x):
        return (x >= self.lo) & (x <= self.hi)

    def logpdf_v(self, x):
        z = (x - self.mu) / self.sigma
        out = -0.5 * (z * z + math.log(2 * math.pi)) - math.log(self.sigma) - self._logz
        return np.where(self.insupport_v(x), out, -np.inf)

    def spec(self):
        return PRIOR_TRUNCNORMAL, [self.mu, self.sigma, self.lo, self.hi, self._logz]


class LogNormal:
    def __init__(self, mu=0.0, sigma=1.0):
        self.mu, self.sigma = float(mu), float(sigma)

    def rand(self, rng):
        return math.exp(self.mu + self.sigma * rng.standard_normal())

    def rand_v(self, rng, m):
        return np.exp(self.mu + self.sigma * rng.standard_normal(m))

    def logpdf(self, x):
        if x <= 0:
            return -math.inf
        z = (math.log(x) - self.mu) / self.sigma
        return -0.5 * (z * z + math.log(2 * math.pi)) - math.log(self.sigma) - math.log(x)

    def insupport(self, x):
        return x > 0 and math.isfinite(x)

    def insupport_v(self, x):
        return (x > 0) & np.isfinite(x)

    def logpdf_v(self, x):
        ok = self.insupport_v(x)
        xs = np.where(ok, x, 1.0)
        z = (np.log(xs) - self.mu) / self.sigma
        return np.where(ok, -0.5 * (z * z + math.log(2 * math.pi)) - math.log(self.sigma) - np.log(xs), -np.inf)

    def spec(self):
        return PRIOR_LOGNORMAL, [self.mu, self.sigma, 0.0, 0.0, 0.0]


class Uniform:
    def __init__(self, lo=0.0, hi=1.0):
        self.lo, self.hi = float(lo), float(hi)

    def rand(self, rng):
        return self.lo + (self.hi - self.lo) * rng.random()

    def rand_v(self, rng, m):
        return self.lo + (self.hi - self.lo) * rng.random(m)

    def logpdf(self, x):
        return -math.log(self.hi - self.lo) if self.insupport(x) else -math.inf

    def insupport(self, x):
        return self.lo <= x <= self.hi

    def insupport_v(self, x):
        return (x >= self.lo) & (x <= self.hi)

    def logpdf_v(self, x):
        return np.where(self.insupport_v(x), -math.log(self.hi - self.lo), -np.inf)

    def spec(self):
        return PRIOR_UNIFORM, [self.lo, self.hi, 0.0, 0.0, 0.0]


def _vec(f):
    return np.vectorize(f, otypes=[float])


class Product:
    """product_distribution([...]): independent components, theta is a vector."""

    def __init__(self, parts):
        self.parts = list(parts)

    # vectorised over an [M, d] array of parameter particles (same arithmetic as the scalar methods)
    def logpdf_many(self, thetas):
        thetas = np.asarray(thetas, dtype=np.float64)
        out = np.zeros(thetas.shape[0])
        for k, p in enumerate(self.parts):
            out = out + (p.logpdf_v(thetas[:, k]) if hasattr(p, "logpdf_v") else _vec(p.logpdf)(thetas[:, k]))
        return out

    def insupport_many(self, thetas):
        thetas = np.asarray(thetas, dtype=np.float64)
        ok = np.ones(thetas.shape[0], dtype=bool)
        for k, p in enumerate(self.parts):
            ok &= np.array([p.insupport(float(t)) for t in thetas[:, k]]) if not hasattr(p, "insupport_v") else p.insupport_v(thetas[:, k])
        return ok

    def __len__(self):
        return len(self.parts)

    def spec(self):
        """(families [d], parameters [d][5]) when every component is one of the enumerated families the device-side
        PMMH knows (include/smc_hip.h SMC_PRIOR_*), else None (then the samplers keep rejuvenate! on the host)."""
        if not all(hasattr(p, "spec") for p in self.parts) or len(self.parts) > 8:
            return None
        sp = [p.spec() for p in self.parts]
        return np.array([f for f, _ in sp], dtype=np.int32), np.array([q for _, q in sp], dtype=np.float64)

    def rand(self, rng):
        return np.array([p.rand(rng) for p in self.parts])

    def rand_many(self, rng, m):
        """m draws at once, [m, d]; component by component (each component vectorised when it can be)"""
        cols = []
        for p in self.parts:
            cols.append(p.rand_v(rng, m) if hasattr(p, "rand_v") else np.array([p.rand(rng) for _ in range(m)]))
        return np.column_stack(cols)

    def logpdf(self, theta):
        return float(sum(p.logpdf(float(t)) for p, t in zip(self.parts, theta)))

    def insupport(self, theta):
        return all(p.insupport(float(t)) for p, t in zip(self.parts, theta))


def product_distribution(parts):
    return Product(parts)
