"""UCSV (three state coordinates): which segment length for which filter size (T = 100, one filter and a batch of 64)."""
import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from sequential_monte_carlo_amd import _lib as L
RAW = [0.2, 0.2, 3.0, 0.0, 0.0]
T = 100
_, y = L.simulate(3, RAW, T, 1998)
for nth in (1, 64):
    for lg in (13, 14, 16, 18, 20, 22):
        nx = 1 << lg
        if nth * nx > 2**24: continue
        row = []
        for seg in (0, 256, 512, 1024, 2048):
            if seg and seg >= nx: row.append("    -     "); continue
            try:
                h = L.Handle(3, nth, nx, seg=seg, seed=1)
            except L.SmcError as e:
                row.append("  error   "); continue
            h.set_params(np.tile(RAW, (nth, 1)))
            h.log_likelihood(y[:8]); h.log_likelihood(y)
            ms = h.elapsed_ms()
            row.append("%9.3e" % (nth * nx * T / ms * 1e3))
            h.close()
        print("UCSV n_theta=%-3d Nx=2^%-2d seg=auto,256,512,1024,2048: %s" % (nth, lg, " ".join(row)), flush=True)
