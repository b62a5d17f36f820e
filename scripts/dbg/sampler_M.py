import sys, io, time
sys.path.insert(0, "/root/repo")
import importlib.util
import numpy as np
import sequential_monte_carlo_amd as smc
spec = importlib.util.spec_from_file_location("bench", "/root/repo/bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
y, prior, mod, tmap = b.sampler_setup("dt")
backend = smc.smc_samplers.HipBackend()
for M in (100, 256, 512, 768, 1024, 2048):
    for algo in ("dt", "smc2"):
        for rep in range(3):
            s = smc.SMC(1024, M, mod, prior, 3, 0.5, seed=11 + rep, backend=backend, theta_map=tmap)
            t0 = time.perf_counter()
            if algo == "dt":
                smc.density_tempered(s, y, verbose=False, out=io.StringIO())
            else:
                smc.smc2(s, y); smc.smc2_run(s, y, 2, len(y), verbose=False)
            dt = time.perf_counter() - t0
        print("%-5s M=%-5d %.2f ms per run, %.3g p-steps/s, posterior mean %s" % (algo, M, dt * 1e3, s.psteps / dt, np.round((s.theta * s.omega[:, None]).sum(axis=0), 3)), flush=True)
