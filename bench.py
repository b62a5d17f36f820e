#!/usr/bin/env python3
"""bench.py -- particle-steps/sec of the bootstrap-filter hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5]

One "step" = one pass of the hot path over one batch of synthetic input = one
log_likelihood(N, y, model) call (src/particles.jl:132-147) for every filter on the rank.

Default workload (BASELINE.json configs[1], the configuration the metric is quoted on):
  c2  univariate linear-Gaussian, Nx = 2^20, T = 1000, ONE filter per GPU.
At N > 1 the filter (theta) axis is what shards: every rank runs its own filter with
its own Philox stream (global theta index = rank) and no data-path collective; the only exchange
is the outer reweight of src/smc_samplers.jl:232 -- one RCCL all-gather of the N logZ scalars per
pass, followed by the replicated normalize.  Weak scaling (per-GPU work fixed).
Other workloads (for DESIGN.md; same code path): c3 SV Nx=2^20 T=5000; c4 LG 512x1024 T=200;
c5 UCSV 512-per-GPU x 1024 T=200 (theta sharded, N_theta = 512*N).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# secondary (ALU) ceiling: 256 CUs x 4 SIMDs; measured by scripts/dbg/valu_rates.hip on this pool: 4.3-4.7 cycles
# per wave64 f64 / 64-bit-integer VALU instruction with 4 waves per SIMD, core clock 2.2-2.4 GHz under VALU load
N_SIMD, VALU_CYCLES_PER_INST, CLOCK_GHZ = 1024, 4.4, 2.3

LG = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
SV = [-1.0, 0.95, 0.25]
UC = [0.2, 0.2, 3.0, 0.0, 0.0]

WORKLOADS = {
    # name: (model, raw, n_theta per GPU, Nx, T, description)
    "c2": (1, LG, 1, 1 << 20, 1000, "LinearGaussian1D bootstrap filter Nx=2^20 T=1000 (BASELINE configs[1])"),
    "c3": (2, SV, 1, 1 << 20, 5000, "StochasticVolatility1D bootstrap filter Nx=2^20 T=5000 (BASELINE configs[2])"),
    "c4": (1, LG, 512, 1024, 200, "LinearGaussian1D batched inner filters Ntheta=512 x Nx=1024 T=200 (configs[3] shape)"),
    "c5": (3, UC, 512, 1024, 200, "UCSV batched inner filters Ntheta=512/GPU x Nx=1024 T=200 (configs[4] shape)"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS) + ["dt", "smc2"])
    ap.add_argument("--seg", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--resampler", default="multinomial", choices=["multinomial", "systematic"],
                    help="multinomial = the reference's resample (default, the judged configuration); systematic = opt-in "
                         "variant (SMC_FLAG_SYSTEMATIC), reported in config.resampler")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo only to rehearse the N>1 path on a 1-GPU box")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    dist = None
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":   # "nccl" is RCCL on ROCm
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from sequential_monte_carlo_amd import _lib as L

    if args.workload in ("dt", "smc2"):
        return bench_density_tempered(args, rank, local_rank, world, dist, torch, algo=args.workload)
    model, raw, nth, nx, T, desc = WORKLOADS[args.workload]
    _, y = L.simulate(model, raw if model != 3 else UC, T, 1998)
    if args.workload == "c5":
        rng = np.random.default_rng(1234 + rank)
        raws = np.tile(raw, (nth, 1))
        raws[:, 0] = raws[:, 1] = rng.uniform(0.05, 0.6, nth)
    else:
        raws = np.tile(raw, (nth, 1))
    on_gpu = dist is not None and args.dist_backend == "nccl"
    h = L.Handle(model, nth, nx, seg=args.seg, seed=1, device=local_rank if on_gpu else 0,
                 flags=L.FLAG_SYSTEMATIC if args.resampler == "systematic" else 0)
    h.set_params(raws)
    h.set_streams(np.arange(nth, dtype=np.uint32) + rank * nth)      # global theta index

    def sync_all():
        h.synchronize()
        if dist is not None:
            if on_gpu:
                torch.cuda.synchronize()
            dist.barrier()

    def one_pass():
        logZ = h.log_likelihood(y)          # inputs (y) are 8 KB; state lives in HBM
        if dist is not None:
            # outer reweight (smc_samplers.jl:232): all-gather the logZ scalars over xGMI, then the
            # replicated normalize on every rank (identical results everywhere)
            mine = torch.from_numpy(logZ).cuda() if on_gpu else torch.from_numpy(logZ)
            allz = torch.empty(world * nth, dtype=torch.float64, device=mine.device)
            dist.all_gather_into_tensor(allz, mine)
            z = allz.cpu().numpy()
            zmax = z.max()
            wts = np.exp(z - zmax)
            _ = zmax + np.log(wts.sum()) - np.log(z.size)
        return logZ

    for _ in range(args.warmup):
        one_pass()
    sync_all()
    t0 = time.perf_counter()
    dev_ms = 0.0
    for _ in range(args.steps):
        logZ = one_pass()
        dev_ms += h.elapsed_ms()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    psteps = float(nth) * nx * T * args.steps * world
    value = psteps / elapsed

    # ---- roofline of the dominant kernel -----------------------------------------------------
    d = {1: 1, 2: 1, 3: 3}[model]
    bytes_per_pstep = 32 + 16 * d                      # 48 B (d=1), 80 B (d=3)
    roof = None
    if rank == 0:
        if h.resident:
            # one k_resident launch does the whole series: HIP-event time of the call
            ms = dev_ms / args.steps
            units = float(nth) * nx * T
            kname = "k_resident"
        else:
            # HIP-event brackets around runs of 8 consecutive k_step launches, averaged per launch
            avg_ms, min_ms = h.time_step_kernel(y[: min(T, 600)], nsample=64)
            ovh = h.event_overhead_ms(64)          # what an empty event bracket reads (reported, NOT subtracted)
            ms = avg_ms
            units = float(nth) * nx
            kname = "k_step"
        achieved = units * bytes_per_pstep / (ms * 1e-3) / 1e9
        traffic, valu = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_%s.json" % args.workload)
        if os.path.exists(pmc) and args.resampler == "multinomial":
            try:
                prof = json.load(open(pmc))
                traffic = prof.get("hbm_bytes_per_launch")
                wi = prof.get("valu_wave_insts_per_launch")
                if wi:
                    # secondary ceiling (SURVEY 8d "FP64 vector ALU"): the launch's VALU instructions
                    # (rocprofv3 SQ_INSTS_VALU) at the measured issue cost on 1024 SIMDs
                    floor_ms = wi * VALU_CYCLES_PER_INST / (N_SIMD * CLOCK_GHZ * 1e9) * 1e3
                    valu = {"bound": "valu", "wave_insts_per_launch": round(wi), "cycles_per_inst": VALU_CYCLES_PER_INST,
                            "clock_ghz": CLOCK_GHZ, "floor_ms": round(floor_ms, 6), "frac": round(floor_ms / ms, 4)}
            except Exception:
                traffic, valu = None, None
        roof = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "launch_ms": round(ms, 6), "algorithmic_bytes_per_launch": units * bytes_per_pstep}
        if valu:
            roof["secondary"] = valu       # the HBM model is SURVEY's accounting; the kernel's real ceiling is the ALU
        if not h.resident:
            roof["empty_event_bracket_ms"] = round(ovh, 6)   # an event pair's own cost, spread over the 8 launches of a bracket
            roof["launches_per_bracket"] = 8

    # ---- CPU baseline: the oracle (scalar port of particles.jl), bounded sample -----------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import binding as ob
        ob.build()
        cpu_cores = 1
        if nth == 1:
            Ts = 96
            f = ob.Filter(model, raw, nx, seg=h.seg, seed=1, stream=0, systematic=args.resampler == "systematic")
            c0 = time.perf_counter()
            f.log_likelihood(y[:Ts])
            cdt = time.perf_counter() - c0
            cval = nx * Ts / cdt
            sample = "same filter (Nx=%d), first %d of T=%d observations, 1 thread" % (nx, Ts, T)
        else:
            # batched workloads: all host cores over the theta axis, mirroring Threads.@threads
            # (src/smc_samplers.jl:112,223); ctypes releases the GIL inside the oracle
            from concurrent.futures import ThreadPoolExecutor
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                cores = os.cpu_count() or 1
            cores = max(1, min(cores, 16))      # the CPU share of a one-GPU box
            per = 2
            ns = min(nth, cores * per * 4)
            chunks = [(k, min(k + per, ns)) for k in range(0, ns, per)]
            c0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=cores) as ex:
                list(ex.map(lambda ab: ob.log_likelihood_batch(model, raws[ab[0]:ab[1]], nx, y, seg=h.seg, seed=1,
                                                               stream0=ab[0]), chunks))
            cdt = time.perf_counter() - c0
            cval = ns * nx * T / cdt
            sample = "first %d of %d filters (Nx=%d, T=%d), %d threads over theta" % (ns, nth, nx, T, cores)
            cpu_cores = cores
        cpu = {"value": round(cval, 1), "unit": "particle-steps/s", "cores": cpu_cores, "kind": "port", "sample": sample,
               "seconds": round(cdt, 2), "julia": julia_baseline(nx)}

    if rank == 0:
        out = {
            "metric": "particle-steps/sec", "value": value, "unit": "particle-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "n_theta_per_gpu": nth, "n_x": nx, "T": T, "seg": h.seg,
                       "resident_kernel": bool(h.resident), "resampler": args.resampler,
                       "sharding": "theta axis, %d filter(s) per GPU" % nth,
                       "logZ_rank0_theta0": float(logZ[0])},
            "device_ms_per_step": dev_ms / args.steps,
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    h.close()
    if dist is not None:
        dist.destroy_process_group()


def julia_baseline(nx):
    """time the repo's Julia restatement of particles.jl if (and only if) a julia with
    Distributions + StatsBase is on this box; the reference package itself is never run."""
    import shutil
    import subprocess
    if shutil.which("julia") is None:
        return "julia unavailable"
    try:
        out = subprocess.run(["julia", os.path.join(ROOT, "julia", "reference_restatement.jl"), str(nx), "8"],
                             capture_output=True, text=True, timeout=600)
        return json.loads(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 else "julia failed: " + out.stderr[-200:]
    except Exception as e:   # noqa: BLE001
        return "julia failed: %s" % e


def bench_density_tempered(args, rank, local_rank, world, dist, torch, algo="dt"):
    """Whole sampler on the README's LG model (README.md:81-98): N_theta = 512 per GPU x Nx = 1024, T = 200,
    chain = 3, ess_threshold = 0.5.  algo "dt": density_tempered (src/smc_samplers.jl:222-281);
    "smc2": the online SMC^2 run of BASELINE configs[3] (smc2 then smc2! for t = 2..T, smc_samplers.jl:288-340;
    PMMH rejuvenation re-filters y[1:t-1] whenever the outer ESS drops below the threshold).
    A "step" is one complete run; every executed inner particle-step is counted."""
    import io
    import sequential_monte_carlo_amd as smc
    from sequential_monte_carlo_amd.distributed import ThetaComm
    on_gpu = dist is not None and args.dist_backend == "nccl"
    comm = None
    if dist is not None:
        comm = ThetaComm(dist, device=torch.device("cuda", local_rank) if on_gpu else None)
    dev = local_rank if on_gpu else 0
    M, N, T, chain = 512 * world, 1024, 200, 3
    _, y = smc.simulate(smc.UnivariateLinearGaussian(A=0.5, B=1.0, Q=0.9, R=0.8), T, seed=1998)
    prior = smc.product_distribution([smc.TruncatedNormal(0, 1, -1, 1), smc.LogNormal(), smc.LogNormal()])

    def mod(th):
        return smc.UnivariateLinearGaussian(A=th[0], B=1.0, Q=th[1], R=th[2])

    backend = smc.smc_samplers.HipBackend(device=dev)

    def raw_fn(th):   # vectorised theta -> (A,B,Q,R,x0,sigma0) rows of the same model
        m = th.shape[0]
        return 1, np.column_stack([th[:, 0], np.ones(m), th[:, 1], th[:, 2], np.zeros(m), np.ones(m)])

    def run(seed):
        s = smc.SMC(N, M, mod, prior, chain, 0.5, seed=seed, backend=backend, comm=comm, raw_fn=raw_fn)
        if algo == "dt":
            stages = smc.density_tempered(s, y, verbose=False, out=io.StringIO())
        else:
            sink = io.StringIO()
            smc.smc2(s, y)
            stages = []
            for t in range(2, T + 1):
                if s.ess < s.ess_min:
                    stages.append(t)            # a resample-move happens inside this step
                smc.smc2_step(s, y, t, verbose=False, out=sink)
        return s, stages

    for k in range(args.warmup):
        run(100 + k)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    psteps = 0
    for k in range(args.steps):
        s, stages = run(k + 1)
        psteps += s.psteps
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        print(json.dumps({
            "metric": "particle-steps/sec", "value": psteps / elapsed, "unit": "particle-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s LG (README.md:81-98) Ntheta=%d (512/GPU) x Nx=1024 T=200 chain=3"
                                   % ("density_tempered" if algo == "dt" else "online SMC^2 (BASELINE configs[3])", M),
                       "stages_last_run": len(stages), "psteps_per_run": s.psteps, "posterior_mean": [float(v) for v in smc.expected_parameters(s)]},
            "roofline": None, "cpu_baseline": None}))
    backend.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
