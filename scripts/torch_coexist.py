"""Checks that libsmchip.so and torch's HIP runtime / RCCL coexist in one process (world_size=1 nccl)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from sequential_monte_carlo_amd import _lib as L
LG = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
_, y = L.simulate(1, LG, 50, 1998)
h = L.Handle(1, 4, 4096, seed=1, device=0); h.set_params(np.tile(LG, (4, 1)))
z = h.log_likelihood(y)
mine = torch.from_numpy(z).cuda(); allz = torch.empty(4, dtype=torch.float64, device="cuda")
dist.all_gather_into_tensor(allz, mine)
torch.cuda.synchronize(); dist.barrier()
assert np.array_equal(allz.cpu().numpy(), z)
z2 = h.log_likelihood(y); assert np.array_equal(z, z2)
print("torch + RCCL + libsmchip coexist OK", z[:2], torch.__version__)
dist.destroy_process_group()
