"""Diagnostic (profiling build: make -C sequential_monte_carlo_amd/csrc abl): in-kernel accumulators of the opt-in persistent step kernel
on BASELINE configs[1] - run as  SMC_LIB=.../build_abl/libsmchip_abl.so SMC_DBG=1 SMC_PERSIST=1 python scripts/dbg/persist_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, time
from sequential_monte_carlo_amd import _lib as L
raw = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
_, y = L.simulate(1, raw, 600, 1998)
h = L.Handle(1, 1, 1 << 20, seed=5)
h.set_params(raw)
h.log_likelihood(y)
t0 = time.perf_counter(); h.log_likelihood(y); dt = time.perf_counter() - t0
print("c2 shape, T = 600: %.2f us per step (SMC_PERSIST=%s)" % (dt / 600 * 1e6, os.environ.get("SMC_PERSIST", "0")))
h.close()
