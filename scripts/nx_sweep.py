"""Throughput of one LG filter as a function of Nx (T=200): where the launch-bound and the ALU-bound regimes meet.
usage: nx_sweep.py [lg2_lo lg2_hi step] [seg ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sequential_monte_carlo_amd import _lib as L
LG = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
T = 200
lo, hi, st = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 24, 2)
segs = [int(a) for a in sys.argv[4:]] or [0]
_, y = L.simulate(1, LG, T, 1998)
for lg in range(lo, hi + 1, st):
    nx = 1 << lg
    for seg in segs:
        try:
            h = L.Handle(1, 1, nx, seg=seg, seed=1)
        except L.SmcError as e:
            print("Nx=2^%d seg=%d: %s" % (lg, seg, e)); continue
        h.set_params(LG)
        h.log_likelihood(y[:8]); z = h.log_likelihood(y)
        ms = h.elapsed_ms()
        print("Nx=2^%-2d seg=%-5d nseg=%-5d resident=%d  %.3f ms  %.2f us/step  %.3e p-steps/s  logZ=%.3f" % (lg, h.seg, h.nseg, h.resident, ms, ms / T * 1e3, nx * T / ms * 1e3, z[0]))
        h.close()
