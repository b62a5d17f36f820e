"""Throughput of batches of filters (n_theta x Nx, T = 100) for the three model families: looking for cliffs.
usage: batch_sweep.py [model ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sequential_monte_carlo_amd import _lib as L
RAW = {1: [0.5, 1.0, 0.9, 0.8, 0.0, 1.0], 2: [-1.0, 0.95, 0.25], 3: [0.2, 0.2, 3.0, 0.0, 0.0]}
T = 100
models = [int(a) for a in sys.argv[1:]] or [1, 3]
for model in models:
    _, y = L.simulate(model, RAW[model], T, 1998)
    for nx in (256, 1024, 4096, 8192, 16384, 65536, 2**18, 2**20):
        row = []
        for nth in (1, 8, 64, 512, 4096):
            if nth * nx > 2**28 or (nth > 64 and nx >= 2**18):
                row.append("      -   "); continue
            try:
                h = L.Handle(model, nth, nx, seed=1)
            except L.SmcError as e:
                row.append("   error  "); continue
            h.set_params(np.tile(RAW[model], (nth, 1)))
            h.log_likelihood(y[:8]); h.log_likelihood(y)
            ms = h.elapsed_ms()
            row.append("%9.3e%s" % (nth * nx * T / ms * 1e3, "r" if h.resident else " "))
            h.close()
        print("model %d Nx=%-8d n_theta=1,8,64,512,4096: %s" % (model, nx, " ".join(row)), flush=True)
