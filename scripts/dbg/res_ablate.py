"""Diagnostic (profiling build): k_resident with and without its LDS search (SMC_ABL bit 0) - the upper bound of what a cheaper
search could buy.  SMC_LIB=.../build_abl/libsmchip_abl.so python scripts/dbg/res_ablate.py"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import numpy as np
    from sequential_monte_carlo_amd import _lib as L
    out = {}
    for name, model, raw, nth in (("LG", 1, [0.5, 1.0, 0.9, 0.8, 0.0, 1.0], 512), ("UCSV", 3, [0.2, 0.2, 3.0, 0.0, 0.0], 512), ("LG4096", 1, [0.5, 1.0, 0.9, 0.8, 0.0, 1.0], 4096), ("LG256", 1, [0.5, 1.0, 0.9, 0.8, 0.0, 1.0], 256)):
        _, y = L.simulate(model, raw, 200, 1998)
        h = L.Handle(model, nth, 1024, seed=1); h.set_params(np.tile(raw, (nth, 1)))
        for _ in range(40): h.log_likelihood(y)
        ts = []
        for _ in range(10): h.log_likelihood(y); ts.append(h.elapsed_ms())
        out[name] = min(ts); h.close()
    print(json.dumps(out))
else:
    for abl in ("0", "1"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, SMC_ABL=abl), capture_output=True, text=True)
        print("SMC_ABL=%s" % abl, r.stdout.strip().splitlines()[-1] if r.returncode == 0 else r.stderr[-500:])
