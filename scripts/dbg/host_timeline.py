"""Diagnostic: where the wall time of one sampler run goes on the host (wrappers around the sampler's building blocks).
usage: host_timeline.py dt|smc2|c5dt [M=512]"""
import os, sys, time, io, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import sequential_monte_carlo_amd as smc
from sequential_monte_carlo_amd import smc_samplers as S, _lib
import bench

algo = sys.argv[1] if len(sys.argv) > 1 else "smc2"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 512
N, T, chain = 1024, 200, 3
y, prior, mod, tmap = bench.sampler_setup(algo)
backend = S.HipBackend(device=0)
acc = collections.OrderedDict()

def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            d = acc.setdefault(label, [0, 0.0]); d[0] += 1; d[1] += time.perf_counter() - t0
    setattr(obj, name, g)

wrap(S, "resample_"); wrap(S, "random_walk_factor"); wrap(S, "_sync_params"); wrap(S, "_rejuvenate_device")
wrap(backend, "rejuvenate", "backend.rejuvenate (device, blocking)")
# the outer level (csrc/smc_outer.hip).  With theta sharded over G ranks: "window" (records of the rank's own slice) divides by G;
# "walk", "temper", "resample" and "rw_factor" are what every rank repeats on all parameter particles
wrap(_lib, "host_outer_window", "outer: window records (per-rank share)"); wrap(_lib, "host_outer_walk", "outer: walk (replicated)")
wrap(_lib, "host_outer_advance", "outer: advance (per-rank share)")
def timed(f, label):
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            d = acc.setdefault(label, [0, 0.0]); d[0] += 1; d[1] += time.perf_counter() - t0
    return g
for nm in ("temper", "resample", "rw_factor", "reweight"):
    setattr(S.LibOuter, nm, staticmethod(timed(getattr(S.LibOuter, nm), "outer: %s (replicated)" % nm)))
wrap(_lib.Handle, "step_window", "Handle.step_window (device, blocking)"); wrap(_lib.Handle, "step_commit"); wrap(_lib.Handle, "permute")
wrap(_lib.Handle, "pmmh_rejuvenate", "Handle.pmmh_rejuvenate (C call)"); wrap(_lib.Handle, "set_params"); wrap(_lib.Handle, "set_streams")
wrap(_lib.Handle, "log_likelihood", "Handle.log_likelihood")

def run(seed):
    s = smc.SMC(N, M, mod, prior, chain, 0.5, seed=seed, backend=backend, theta_map=tmap)
    if algo == "smc2":
        smc.smc2(s, y); smc.smc2_run(s, y, 2, T, verbose=False)
    else:
        smc.density_tempered(s, y, verbose=False, out=io.StringIO())
    return s

for k in range(4): run(k)
acc.clear()
t0 = time.perf_counter(); run(7); tot = time.perf_counter() - t0
print("%s M=%d: %.2f ms" % (algo, M, tot * 1e3))
for k, (n, t) in acc.items():
    print("  %-45s %3d calls %8.3f ms" % (k, n, t * 1e3))
