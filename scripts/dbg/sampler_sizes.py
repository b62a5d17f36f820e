"""The samplers at state-particle counts other than 1024 (examples/inflation_example.jl:256 runs SMC(8192, 512, ...) on UCSV):
density_tempered and the online loop over multi-segment inner filters - do they run, how long."""
import sys, io, time
sys.path.insert(0, "/root/repo")
import importlib.util
import numpy as np
import sequential_monte_carlo_amd as smc
spec = importlib.util.spec_from_file_location("bench", "/root/repo/bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
for algo, N, M, T in (("c5dt", 8192, 128, 100), ("c5dt", 1024, 128, 100), ("dt", 8192, 128, 100), ("dt", 16384, 64, 100), ("smc2", 4096, 64, 60), ("c5smc2", 8192, 32, 40)):
    y, prior, mod, tmap = b.sampler_setup("c5dt" if algo.startswith("c5") else "dt")
    y = y[:T]
    backend = smc.smc_samplers.HipBackend()
    for rep in range(2):
        s = smc.SMC(N, M, mod, prior, 3, 0.5, seed=11 + rep, backend=backend, theta_map=tmap)
        t0 = time.perf_counter()
        if algo.endswith("dt"):
            st = smc.density_tempered(s, y, verbose=False, out=io.StringIO())
        else:
            smc.smc2(s, y); smc.smc2_run(s, y, 2, T, verbose=False)
        dt = time.perf_counter() - t0
    pm = (s.theta * s.omega[:, None]).sum(axis=0)
    print("%-7s Nx=%-6d M=%-4d T=%-4d %.1f ms per run, %.3g p-steps/s, posterior mean %s" % (algo, N, M, T, dt * 1e3, s.psteps / dt, np.round(pm, 3)), flush=True)
