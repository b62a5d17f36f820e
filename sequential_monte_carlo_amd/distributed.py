"""theta-axis sharding over the GPUs of one node (one process per GPU, torch.distributed).

The only exchange on the path is the outer `reweight` of src/smc_samplers.jl:232,249,265,338: one
all-gather of the per-rank slices of logZ (n_theta doubles in total, 32 KB at n_theta = 4096) per
batched evaluation, after which every rank runs the identical O(n_theta) host logic.  Backend "nccl"
is RCCL over xGMI on the GPU box; "gloo" on CPU for the world_size-2 tests."""
import numpy as np


class ThetaComm:
    def __init__(self, dist, device=None):
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device

    def slice(self, M):
        if M % self.world:
            raise ValueError("n_theta (%d) must be a multiple of the number of ranks (%d)" % (M, self.world))
        per = M // self.world
        return self.rank * per, (self.rank + 1) * per

    def all_gather(self, local):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64))
        if self.device is not None:
            t = t.to(self.device)
        out = torch.empty(t.numel() * self.world, dtype=torch.float64, device=t.device)
        self.dist.all_gather_into_tensor(out, t)
        return out.cpu().numpy()
