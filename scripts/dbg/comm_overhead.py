"""Diagnostic: what the sharded code path of the online sampler costs on top of the unsharded one, measured with ONE rank
(torch.distributed / RCCL with a world of one, and the C-level communicator smc_comm_*): every collective a multi-GPU run
makes is made here too (all-gather of segment records per window, all-gather of the (logw, logZ) slices and all-to-all of
filter slots per resample!, all-gather of the moved slices per rejuvenate!), with no peer to wait for - a lower bound of
what each collective adds per rank.  usage: comm_overhead.py [M=512]"""
import os, sys, time, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import sequential_monte_carlo_amd as smc
from sequential_monte_carlo_amd import smc_samplers as S, _lib
import bench

M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N, T, chain = 1024, 200, 3
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from sequential_monte_carlo_amd.distributed import ThetaComm
comms = {"none": None, "torch.distributed/RCCL, world 1": ThetaComm(dist, device=torch.device("cuda", 0)),
         "smc_comm_* (RCCL from C), world 1": _lib.Comm(_lib.comm_unique_id(), 0, 1, device=0)}
for algo in ("smc2", "c5dt"):
    y, prior, mod, tmap = bench.sampler_setup(algo)
    backend = S.HipBackend(device=0)
    for name, comm in comms.items():
        calls = [0]
        if comm is not None and not hasattr(comm, "_wrapped"):
            for nm in ("all_gather", "exchange_slots"):
                f = getattr(comm, nm)
                def g(*a, _f=f, **k):
                    calls[0] += 1
                    return _f(*a, **k)
                setattr(comm, nm, g)
            comm._wrapped = True
        def run(seed):
            s = smc.SMC(N, M, mod, prior, chain, 0.5, seed=seed, backend=backend, theta_map=tmap, comm=comm)
            if algo == "smc2":
                smc.smc2(s, y); smc.smc2_run(s, y, 2, T, verbose=False)
            else:
                smc.density_tempered(s, y, verbose=False, out=io.StringIO())
            return s
        for k in range(3): run(k)
        best = 1e9
        for k in range(5):
            calls[0] = 0
            t0 = time.perf_counter(); run(7); best = min(best, time.perf_counter() - t0)
        print("%s M=%d comm = %-40s %.2f ms per run, %d collectives" % (algo, M, name, best * 1e3, calls[0]))
    backend.close()
dist.destroy_process_group()
