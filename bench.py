#!/usr/bin/env python3
"""bench.py -- particle-steps/sec of the bootstrap-filter hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5|dt|smc2|c5dt] [--scaling weak|strong]

One "step" = one pass of the hot path over one batch of synthetic input = one
log_likelihood(N, y, model) call (src/particles.jl:132-147) for every filter on the rank.

Default workload (BASELINE.json configs[1], the configuration the metric is quoted on):
  c2  univariate linear-Gaussian, Nx = 2^20, T = 1000, ONE filter per GPU.
At N > 1 the filter (theta) axis is what shards: every rank runs its own filter with
its own Philox stream (global theta index = rank) and no data-path collective; the only exchange
is the outer reweight of src/smc_samplers.jl:232 -- one RCCL all-gather of the N logZ scalars per
pass, followed by the replicated normalize.  Weak scaling (per-GPU work fixed).
Other filter workloads (same code path): c3 SV Nx=2^20 T=5000; c4 LG 512x1024 T=200;
c5 UCSV 512-per-GPU x 1024 T=200 (theta sharded, N_theta = 512*N).

Whole-sampler workloads (every EXECUTED inner particle-step counted; a "step" is one complete run):
  dt     density_tempered, README LG model (README.md:81-98), chain 3
  smc2   online SMC^2 = BASELINE configs[3]
  c5dt   density_tempered over UCSV with the example's prior = BASELINE configs[4]
--scaling weak (default): N_theta = 512 per GPU.  --scaling strong: N_theta fixed (--n-theta, default 4096:
north_star "SMC^2 N_theta = 4096"), split evenly over the ranks.
With the default filter workload the JSON line also carries "aux": the two strong-scaling sampler runs
(smc2 and c5dt at N_theta = 4096) measured after the timed region, so that the driver's N = 1,2,4,8 sweep
yields the north_star scaling curve too (--no-aux skips them).

Launching.  Under torch.distributed.run (RANK/WORLD_SIZE set) this file is one rank.  A bare
`python bench.py --gpus N` (N > 1, no WORLD_SIZE) makes THIS process a launcher: before anything touches
the GPU it starts N fresh child ranks of itself (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, 127.0.0.1),
relays rank 0's JSON line and exits with the worst child status; the launcher never imports torch or the
HIP library.  --dry-launch: the ranks only form the process group and report (CPU test of the launcher).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# secondary (ALU) ceiling: 256 CUs x 4 SIMDs; measured by scripts/dbg/valu_rate.hip on this pool: 4.3-4.7 cycles
# per wave64 f64 / 64-bit-integer VALU instruction with 4 waves per SIMD, core clock 2.2-2.4 GHz under VALU load
N_SIMD, VALU_CYCLES_PER_INST, CLOCK_GHZ = 1024, 4.4, 2.3


def kernel_source_hash():
    """sha256 (first 16 hex digits) over csrc/*.h and *.hip in name order: the same function scripts/pmc_summary.py stamps a
    counter profile with, so that a profile taken from other kernel sources than the running build is flagged stale"""
    import glob
    import hashlib
    here = os.path.join(ROOT, "sequential_monte_carlo_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(here, "*.h")) + glob.glob(os.path.join(here, "*.hip"))):
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return h.hexdigest()[:16]

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
SAMPLERS = ("dt", "smc2", "c5dt")
# one step of c2 / c3 is 13 / 66 ms of GPU work, of c4 / c5 0.7 / 1.4 ms, of the samplers 7 - 12 ms
DEFAULT_WARMUP = {"c2": 3, "c3": 1, "c4": 60, "c5": 30, "dt": 6, "smc2": 5, "c5dt": 5}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=None,
                    help="untimed warm-up steps; default per workload (DEFAULT_WARMUP): enough to precede the timed region by "
                         ">= 40 ms of GPU work - after idling the first ~50 ms run at lower clocks (k_resident 0.72 -> 0.64 ms)")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS) + list(SAMPLERS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="sampler workloads: weak = 512 parameter particles per GPU; strong = --n-theta in total")
    ap.add_argument("--n-theta", type=int, default=4096, help="total parameter particles of a strong-scaling sampler run")
    ap.add_argument("--window", type=int, default=16, help="smc2: propagation steps per device call (smc2_run); 1 = one smc2! per call")
    ap.add_argument("--seg", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the strong-scaling sampler runs appended to a filter workload's line")
    ap.add_argument("--resampler", default="multinomial", choices=["multinomial", "systematic"],
                    help="multinomial = the reference's resample (default, the judged configuration); systematic = opt-in "
                         "variant (SMC_FLAG_SYSTEMATIC), reported in config.resampler")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo only to rehearse the N>1 path on a 1-GPU box / on CPU")
    ap.add_argument("--dry-launch", action="store_true",
                    help="form the process group, report the ranks seen, run nothing (no GPU needed with gloo)")
    args = ap.parse_args(argv)
    if args.warmup is None:
        args.warmup = DEFAULT_WARMUP[args.workload]
    return args


# ---- launcher: `python bench.py --gpus N` without torch.distributed.run -------------------------------------
def launch(args, argv):
    """Start args.gpus fresh ranks of this file and relay rank 0's stdout.  Nothing here touches the GPU,
    imports torch or loads libsmchip.so: the children are new processes, never an exec of an initialised one."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    if out0:
        sys.stdout.write(out0)
        sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write("bench.py launcher: ranks failed (rank, exit code): %s\n" % bad)
        return 1
    return 0


class Ctx:
    """What a rank knows about the job."""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = self.torch = None
        self.backend = None
        if self.world > 1:
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            if args.dist_backend == "nccl":   # "nccl" is RCCL on ROCm
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            self.backend = dist.get_backend()
        self.on_gpu = self.dist is not None and args.dist_backend == "nccl"
        self.device = self.local_rank if self.on_gpu else 0
        self.ranks_seen = self.dist.get_world_size() if self.dist is not None else 1

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        tt = self.torch.tensor([seconds], dtype=self.torch.float64,
                               device=self.torch.device("cuda", self.local_rank) if self.on_gpu else "cpu")
        self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
        return float(tt.item())

    def barrier(self):
        if self.dist is not None:
            if self.on_gpu:
                self.torch.cuda.synchronize(self.local_rank)
                self.dist.barrier(device_ids=[self.local_rank])
            else:
                self.dist.barrier()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch(args, argv))
    ctx = Ctx(args)
    if ctx.world != args.gpus and ctx.world > 1:
        args.gpus = ctx.world
    if args.dry_launch:
        if ctx.dist is not None:
            ctx.dist.barrier()
        if ctx.rank == 0:
            print(json.dumps({"dry_launch": True, "n_gpus": ctx.world, "ranks_seen": ctx.ranks_seen, "dist_backend": ctx.backend}))
        if ctx.dist is not None:
            ctx.dist.destroy_process_group()
        return
    if args.workload in SAMPLERS:
        out = bench_sampler(args, ctx, args.workload, args.scaling, args.steps, args.warmup)
    else:
        out = bench_filter(args, ctx)
    if ctx.rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    # the secondary sampler runs failed (an exception, or a collective that never returned): the main line is out, the exit
    # status says so.  A stalled helper thread sits inside a collective: no orderly teardown is possible then.
    aux_bad = getattr(ctx, "aux_failed", False)
    if getattr(ctx, "aux_stalled", False):
        os._exit(3)
    if ctx.dist is not None:
        ctx.dist.destroy_process_group()
    if aux_bad:
        sys.exit(3)


# ---- filter workloads (c2 c3 c4 c5) ----------------------------------------------------------------------
def bench_filter(args, ctx):
    from sequential_monte_carlo_amd import _lib as L
    rank, world, dist, torch, on_gpu = ctx.rank, ctx.world, ctx.dist, ctx.torch, ctx.on_gpu
    model, raw, nth, nx, T, desc = WORKLOADS[args.workload]
    _, y = L.simulate(model, raw if model != 3 else UC, T, 1998)
    if args.workload == "c5":
        rng = np.random.default_rng(1234 + rank)
        raws = np.tile(raw, (nth, 1))
        raws[:, 0] = raws[:, 1] = rng.uniform(0.05, 0.6, nth)
    else:
        raws = np.tile(raw, (nth, 1))
    h = L.Handle(model, nth, nx, seg=args.seg, seed=1, device=ctx.device,
                 flags=L.FLAG_SYSTEMATIC if args.resampler == "systematic" else 0)
    h.set_params(raws)
    h.set_streams(np.arange(nth, dtype=np.uint32) + rank * nth)      # global theta index

    def sync_all():
        h.synchronize()
        ctx.barrier()

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
    elapsed = ctx.max_over_ranks(time.perf_counter() - t0)

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
            # HIP-event brackets around runs of 32 consecutive k_step launches, averaged per launch
            avg_ms, min_ms = h.time_step_kernel(y[: min(T, 1000)], nsample=24)
            ovh = h.event_overhead_ms(64)          # what an empty event bracket reads (reported, NOT subtracted)
            ms = avg_ms
            units = float(nth) * nx
            kname = "k_step"
        achieved = units * bytes_per_pstep / (ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                "launch_ms": round(ms, 6), "algorithmic_bytes_per_launch": units * bytes_per_pstep}
        # PMC counters cannot be read inside this process: they come from the committed rocprofv3 --pmc passes
        # (scripts/collect_pmc.sh -> profiles/pmc_<workload>.json), labelled with the commit they were taken at.
        pmc = os.path.join(ROOT, "profiles", "pmc_%s.json" % args.workload)
        if os.path.exists(pmc) and args.resampler == "multinomial":
            try:
                prof = json.load(open(pmc))
                roof["traffic"] = prof.get("hbm_bytes_per_launch")
                roof["traffic_source"] = "profiles/pmc_%s.json@%s sources %s (separate rocprofv3 --pmc passes, not this run)" % (
                    args.workload, prof.get("commit", "unknown"), prof.get("kernel_source_sha16", "unstamped"))
                # counters of another build of the kernels say nothing about this one
                roof["traffic_stale"] = prof.get("kernel_source_sha16") != kernel_source_hash()
                if roof["traffic"]:
                    roof["counter_gbs"] = round(roof["traffic"] / (ms * 1e-3) / 1e9, 1)   # counter bytes / THIS run's launch time
                wi = prof.get("valu_wave_insts_per_launch")
                if wi:
                    # secondary ceiling (SURVEY 8d "FP64 vector ALU"): the launch's VALU instructions
                    # (rocprofv3 SQ_INSTS_VALU) at the measured issue cost on 1024 SIMDs
                    floor_ms = wi * VALU_CYCLES_PER_INST / (N_SIMD * CLOCK_GHZ * 1e9) * 1e3
                    roof["secondary"] = {"bound": "valu", "wave_insts_per_launch": round(wi), "cycles_per_inst": VALU_CYCLES_PER_INST,
                                         "clock_ghz": CLOCK_GHZ, "floor_ms": round(floor_ms, 6), "frac": round(floor_ms / ms, 4),
                                         "source": roof["traffic_source"]}
                    # the HBM model is SURVEY's accounting; the resource that binds this kernel is the vector ALU
                    roof["binding_resource"] = "valu" if floor_ms / ms > achieved / HBM_PEAK_GBS else "hbm"
            except Exception:   # noqa: BLE001
                pass
        if not h.resident:
            roof["empty_event_bracket_ms"] = round(ovh, 6)   # an event pair's own cost, spread over the launches of a bracket
            roof["launches_per_bracket"] = 32 if min(T, 1000) - 1 >= 512 else 8

    # ---- CPU baseline: the oracle (scalar port of particles.jl), bounded sample -----------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, h, model, raw, raws, nth, nx, T, y)

    h.close()
    out = None
    if rank == 0:
        out = {
            "metric": "particle-steps/sec", "value": value, "unit": "particle-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "n_theta_per_gpu": nth, "n_x": nx, "T": T, "seg": h.seg,
                       "resident_kernel": bool(h.resident), "resampler": args.resampler,
                       "sharding": "theta axis, %d filter(s) per GPU" % nth,
                       "logZ_rank0_theta0": float(logZ[0])},
            "ranks_seen": ctx.ranks_seen, "dist_backend": ctx.backend,
            "device_ms_per_step": dev_ms / args.steps,
            "roofline": roof, "cpu_baseline": cpu,
        }
    if not args.no_aux:
        aux = run_aux(args, ctx)
        if out is not None:
            out["aux"] = aux
    return out


def cpu_baseline(args, h, model, raw, raws, nth, nx, T, y):
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
    return {"value": round(cval, 1), "unit": "particle-steps/s", "cores": cpu_cores, "kind": "port", "sample": sample,
            "seconds": round(cdt, 2), "julia": julia_baseline(nx)}


def julia_baseline(nx):
    """time the repo's Julia restatement of particles.jl if (and only if) a julia with
    Distributions + StatsBase is on this box; the reference package itself is never run."""
    import shutil
    if shutil.which("julia") is None:
        return "julia unavailable"
    try:
        out = subprocess.run(["julia", os.path.join(ROOT, "julia", "reference_restatement.jl"), str(nx), "8"],
                             capture_output=True, text=True, timeout=600)
        return json.loads(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 else "julia failed: " + out.stderr[-200:]
    except Exception as e:   # noqa: BLE001
        return "julia failed: %s" % e


def run_aux(args, ctx, timeout_s=300.0):
    """The north_star scaling workloads (strong scaling, N_theta = 4096 in total) appended to a filter line.
    They run in a helper thread: if a collective of this secondary measurement ever stalled, the main line
    (already measured) must still be printed - after `timeout_s` the rank reports the timeout and the process
    leaves through os._exit(3) once the line is out; an exception in a run is reported in the line and makes the exit status 3."""
    res = {}

    def work():
        if ctx.on_gpu:
            ctx.torch.cuda.set_device(ctx.local_rank)     # the current device is per thread: collectives and barriers use it
        for algo in ("smc2", "c5dt"):
            try:
                r = bench_sampler(args, ctx, algo, "strong", steps=2, warmup=1)
                if r is not None:
                    res[algo + "_strong"] = {k: r[k] for k in ("value", "unit", "ms_per_step", "n_gpus", "scaling", "config")}
            except Exception as e:   # noqa: BLE001
                res[algo + "_strong"] = {"error": "%s: %s" % (type(e).__name__, e)}
                ctx.aux_failed = True

    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        res["error"] = "aux runs did not finish within %.0f s" % timeout_s
        if ctx.rank == 0:
            sys.stderr.write("bench.py: %s\n" % res["error"])
        ctx.aux_stalled = True
    return res


# ---- whole-sampler workloads (dt smc2 c5dt) ---------------------------------------------------------------
def sampler_setup(algo):
    """(y, prior, model closure, ThetaMap, true-parameter description) of a sampler workload."""
    import sequential_monte_carlo_amd as smc
    T = 200
    if algo == "c5dt":
        # examples/inflation_example.jl:227-237: UCSV(theta1, theta2, (theta3, theta4)), gamma_eps = gamma_eta = theta1
        _, y = smc.simulate(smc.UCSV((0.2, 0.2), 3.0, (0.0, 0.0)), T, seed=1998)
        prior = smc.product_distribution([smc.Uniform(0.0, 1.0), smc.Normal(3.0, 2.0), smc.Uniform(0.0, 2.0), smc.Uniform(0.0, 2.0)])
        tmap = smc.ThetaMap(smc.UCSV.model_id, [0, 0, 1, 2, 3], [0.0] * 5)

        def mod(th):
            return smc.UCSV((th[0], th[0]), th[1], (th[2], th[3]))
        return y, prior, mod, tmap
    _, y = smc.simulate(smc.UnivariateLinearGaussian(A=0.5, B=1.0, Q=0.9, R=0.8), T, seed=1998)
    prior = smc.product_distribution([smc.TruncatedNormal(0, 1, -1, 1), smc.LogNormal(), smc.LogNormal()])
    tmap = smc.ThetaMap(smc.LinearModel.model_id, [0, -1, 1, 2, -1, -1], [0.0, 1.0, 0.0, 0.0, 0.0, 1.0])

    def mod(th):
        return smc.UnivariateLinearGaussian(A=th[0], B=1.0, Q=th[1], R=th[2])
    return y, prior, mod, tmap


def bench_sampler(args, ctx, algo, scaling, steps, warmup):
    """Whole sampler, N_theta x Nx = 1024, T = 200, chain = 3, ess_threshold = 0.5 (README.md:98).
    "dt": density_tempered (src/smc_samplers.jl:222-281) on the README LG model; "c5dt": the same over UCSV with the
    example's prior (BASELINE configs[4]); "smc2": the online SMC^2 run of BASELINE configs[3] (smc2 then smc2! for
    t = 2..T, smc_samplers.jl:288-340; PMMH rejuvenation re-filters y[1:t-1] whenever the outer ESS drops below the
    threshold).  A "step" is one complete run; every executed inner particle-step is counted."""
    import io
    import sequential_monte_carlo_amd as smc
    from sequential_monte_carlo_amd.distributed import ThetaComm
    dist, torch, world = ctx.dist, ctx.torch, ctx.world
    comm = None
    if dist is not None:
        comm = ThetaComm(dist, device=torch.device("cuda", ctx.local_rank) if ctx.on_gpu else None)
    if scaling == "strong":
        M = int(args.n_theta)
        if M % world:
            raise ValueError("--n-theta (%d) must be a multiple of the number of ranks (%d)" % (M, world))
    else:
        M = 512 * world
    N, T, chain = 1024, 200, 3
    y, prior, mod, tmap = sampler_setup(algo)
    backend = smc.smc_samplers.HipBackend(device=ctx.device, resampler=args.resampler)

    def run(seed):
        s = smc.SMC(N, M, mod, prior, chain, 0.5, seed=seed, backend=backend, comm=comm, theta_map=tmap)
        if algo in ("dt", "c5dt"):
            stages = smc.density_tempered(s, y, verbose=False, out=io.StringIO())
        else:
            smc.smc2(s, y)
            calls = s._calls
            # `for t in 2:T smc²!(smc,y,t) end` (smc_samplers.jl:308-340), several propagation steps per device call
            smc.smc2_run(s, y, 2, T, window=args.window, verbose=False)
            stages = [0] * ((s._calls - calls) // (chain + 2 if s.device_pmmh else chain + 1))     # resample-move rounds
        return s, stages

    for k in range(warmup):
        run(100 + k)
    ctx.barrier()
    t0 = time.perf_counter()
    psteps = skipped = 0
    for k in range(steps):
        s, stages = run(k + 1)
        psteps += s.psteps
        skipped += s.psteps_skipped
    ctx.barrier()
    elapsed = ctx.max_over_ranks(time.perf_counter() - t0)
    out = None
    if ctx.rank == 0:
        names = {"dt": "density_tempered LG (README.md:81-98)", "smc2": "online SMC^2 LG (BASELINE configs[3])",
                 "c5dt": "density_tempered UCSV, prior U(0,1)xN(3,2)xU(0,2)xU(0,2) (BASELINE configs[4])"}
        out = {
            "metric": "particle-steps/sec", "value": psteps / elapsed, "unit": "particle-steps/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s Ntheta=%d (%d/GPU) x Nx=%d T=%d chain=%d" % (names[algo], M, M // world, N, T, chain),
                       "stages_last_run": len(stages), "psteps_per_run": s.psteps,
                       "psteps_skipped_out_of_support_per_run": s.psteps_skipped,
                       # device work beyond the counted steps: the dropped steps of speculated windows and the prefix a partial commit re-runs
                       "psteps_speculated_per_run": s.psteps_speculated, "device_pmmh": bool(s.device_pmmh),
                       "posterior_mean": [float(v) for v in smc.expected_parameters(s)]},
            "ranks_seen": ctx.ranks_seen, "dist_backend": ctx.backend,
            "roofline": None, "cpu_baseline": None}
    backend.close()
    return out


if __name__ == "__main__":
    main()
