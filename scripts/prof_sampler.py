"""Profile of one whole sampler run (bench.py's sampler workloads): wall time, host cProfile, and - when run under
`rocprofv3 --kernel-trace --stats -- python3 scripts/prof_sampler.py <algo> <M> noprof` - the kernel time per run.
usage: prof_sampler.py dt|smc2|c5dt [M=512] [noprof]"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sequential_monte_carlo_amd as smc
import bench

algo = sys.argv[1] if len(sys.argv) > 1 else "smc2"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 512
N, T, chain = 1024, 200, 3
y, prior, mod, tmap = bench.sampler_setup(algo)
backend = smc.smc_samplers.HipBackend(device=0)


def run(seed):
    s = smc.SMC(N, M, mod, prior, chain, 0.5, seed=seed, backend=backend, theta_map=tmap)
    if algo == "smc2":
        smc.smc2(s, y)
        if hasattr(smc, "smc2_run"):
            smc.smc2_run(s, y, 2, T, verbose=False)
        else:
            for t in range(2, T + 1):
                smc.smc2_step(s, y, t, verbose=False, out=io.StringIO())
    else:
        smc.density_tempered(s, y, verbose=False, out=io.StringIO())
    return s


run(1)
t0 = time.perf_counter(); s = run(2); dt = time.perf_counter() - t0
print("%s M=%d: run %.2f ms, psteps %.3e (+%.3e skipped) -> %.3e p-steps/s" % (algo, M, dt * 1e3, s.psteps, s.psteps_skipped, s.psteps / dt))
if "noprof" not in sys.argv:
    pr = cProfile.Profile(); pr.enable(); run(3); pr.disable()
    o = io.StringIO(); pstats.Stats(pr, stream=o).sort_stats("tottime").print_stats(25); print(o.getvalue()[:5000])
