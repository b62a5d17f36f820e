"""sequential_monte_carlo_amd -- MI355X-native particle-filter hot path of
charlesknipp/sequential_monte_carlo behind the reference's own API names.

Hand-written HIP kernels for gfx950 (csrc/) behind a C ABI (include/smc_hip.h); this package is the
host-side mirror of src/particles.jl, src/state_space_models.jl and src/smc_samplers.jl.
There is no CPU fallback: importing works anywhere, running a filter needs the GPU library.
"""
from .distributions import LogNormal, Normal, TruncatedNormal, Uniform, product_distribution  # noqa: F401
from .kalman_filter import log_likelihood_kalman  # noqa: F401
from .models import (UCSV, LinearModel, StateSpaceModel, StochasticVolatility, UnivariateLinearGaussian,  # noqa: F401
                     simulate, unobserved_components, unobserved_components_stochastic_volatility)
from .particles import (bootstrap_filter, bootstrap_filter_, log_likelihood, normalize, resample, reweight)  # noqa: F401
from .smc_samplers import (SMC, ThetaMap, density_tempered, estimated_trend, expected_parameters, filtered_summaries, rejuvenate_, resample_, smc2, smc2_run, smc2_step)  # noqa: F401

__version__ = "0.1.0"
