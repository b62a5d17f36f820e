"""Geometry sweep for the C2 shape (tuning aid): SMC_NP x seg."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from sequential_monte_carlo_amd import _lib as L
    seg = int(sys.argv[2]); T = 200
    LG = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
    _, y = L.simulate(1, LG, T, 1998)
    h = L.Handle(1, 1, 1 << 20, seg=seg, seed=1); h.set_params(LG)
    h.log_likelihood(y[:20]); best = 1e9
    for _ in range(3):
        z = h.log_likelihood(y); best = min(best, h.elapsed_ms())
    a, m = h.time_step_kernel(y, 32)
    print("seg=%d np=%s: %.3f ms/200 steps  %.2f us/step  kernel avg %.2f min %.2f us  %.3e p-steps/s logZ=%.4f" % (
        seg, os.environ.get("SMC_NP", "def"), best, best / T * 1e3, a * 1e3, m * 1e3, (1 << 20) * T / best * 1e3, z[0]))
else:
    for seg in (256, 512, 1024, 2048, 4096):
        for np_ in (1, 2, 4):
            env = dict(os.environ, SMC_NP=str(np_))
            subprocess.call([sys.executable, __file__, "child", str(seg)], env=env)
