"""Diagnostic: the opt-in persistent step kernel (SMC_PERSIST=1) against one launch per step: bit-identical results, time per step.
usage: persist_real.py   (runs itself twice as child processes: the switch is read once per handle)"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, time
    from sequential_monte_carlo_amd import _lib as L
    out = {}
    for name, model, raw, n, seg, nth, T in (("c2", 1, [0.5, 1.0, 0.9, 0.8, 0.0, 1.0], 1 << 20, 0, 1, 1000), ("sv", 2, [-1.0, 0.95, 0.25], 1 << 20, 0, 1, 300),
                                             ("ucsv", 3, [0.2, 0.2, 3.0, 0.0, 0.0], 1 << 18, 0, 1, 200), ("lg_ragged", 1, [0.5, 1.0, 0.9, 0.8, 0.0, 1.0], 300000, 1024, 1, 200),
                                             ("lg_batch", 1, [0.5, 1.0, 0.9, 0.8, 0.0, 1.0], 40000, 512, 3, 150)):
        _, y = L.simulate(model, raw, T, 1998)
        h = L.Handle(model, nth, n, seg=seg, seed=5)
        h.set_params(np.tile(raw, (nth, 1)))
        z = h.log_likelihood(y)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); z = h.log_likelihood(y); best = min(best, time.perf_counter() - t0)
        x, w, _ = h.state(want_anc=False)
        out[name] = dict(logZ=[float(v).hex() for v in z], xsum=float(x.sum()).hex(), wsum=float(w.sum()).hex(), us_per_step=best / T * 1e6, dev_ms=h.elapsed_ms())
        h.close()
    print(json.dumps(out))
else:
    res = {}
    for mode in ("0", "1"):
        env = dict(os.environ, SMC_PERSIST=mode)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            print("mode", mode, "FAILED:", r.stderr[-1500:]); continue
        res[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    for k in res.get("0", {}):
        a, b = res["0"][k], res.get("1", {}).get(k)
        same = b is not None and all(a[f] == b[f] for f in ("logZ", "xsum", "wsum"))
        print("%-10s launches %.2f us per step | persistent %s us per step | bit-identical: %s" % (k, a["us_per_step"], ("%.2f" % b["us_per_step"]) if b else "n/a", same))
