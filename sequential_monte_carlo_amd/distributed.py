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
        self._bufs = {}

    def slice(self, M):
        if M % self.world:
            raise ValueError("n_theta (%d) must be a multiple of the number of ranks (%d)" % (M, self.world))
        per = M // self.world
        return self.rank * per, (self.rank + 1) * per

    def all_gather(self, local):
        """every rank's `local` (equal lengths), concatenated in rank order, on every rank.  The device and pinned
        staging buffers are kept per length: a sampler run makes dozens of these tiny latency-bound exchanges."""
        import torch
        src = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64).ravel())
        n = src.numel()
        if self.device is None:
            out = torch.empty(n * self.world, dtype=torch.float64)
            self.dist.all_gather_into_tensor(out, src)
            return out.numpy()
        bufs = self._bufs.get(n)
        if bufs is None:
            bufs = (torch.empty(n, dtype=torch.float64, device=self.device),
                    torch.empty(n * self.world, dtype=torch.float64, device=self.device),
                    torch.empty(n * self.world, dtype=torch.float64).pin_memory())
            self._bufs[n] = bufs
        send, recv, host = bufs
        send.copy_(src)
        self.dist.all_gather_into_tensor(recv, send)
        host.copy_(recv)                  # a device-to-host copy on the current stream: waits for the collective
        return host.numpy().copy()


    # ---- online SMC^2 with sharded theta: resample!(smc) moves whole filters between ranks ---------------
    def _export(self, h, idx):
        import torch
        if hasattr(h, "export_slots"):                      # test backends bring their own (CPU tensors)
            return h.export_slots(idx)
        words = h.slot_bytes() // 8
        buf = torch.empty((len(idx), words), dtype=torch.int64, device=self.device if self.device is not None else "cuda")
        if len(idx):
            h.pack_slots(idx, buf.data_ptr())               # device -> device, no host staging
        return buf

    def _import(self, h, idx, buf):
        if hasattr(h, "import_slots"):
            return h.import_slots(idx, buf)
        if len(idx):
            h.unpack_slots(idx, buf.data_ptr())

    def exchange_slots(self, h, a, M):
        """After the outer resample drew global ancestors a[0..M): local slot m - lo of every rank becomes a
        value copy of global slot a[m] (src/smc_samplers.jl:74-84), wherever that slot lives.  One
        all-to-all of packed filter states (x cloud, weights, records) over xGMI."""
        import torch
        per = M // self.world
        lo = self.rank * per
        a = np.asarray(a, dtype=np.int64)
        send_idx, send_counts = [], []
        for r in range(self.world):
            src = a[r * per:(r + 1) * per]
            mine = src[(src // per) == self.rank] - lo
            send_idx.append(mine)
            send_counts.append(int(mine.size))
        owners = a[lo:lo + per] // per
        recv_counts = [int((owners == s).sum()) for s in range(self.world)]
        dest_idx = np.concatenate([np.nonzero(owners == s)[0] for s in range(self.world)]).astype(np.int32)
        sendbuf = self._export(h, np.concatenate(send_idx).astype(np.int32))
        if sendbuf.is_cuda and self.dist.get_backend() == "gloo":
            # rehearsal of the N > 1 path on a box with one GPU (several ranks on the same card, `bench.py --dist-backend gloo`,
            # tests/test_gpu_api.py): gloo moves host memory, so the packed filters are staged through it
            torch.cuda.synchronize(sendbuf.device)
            host_send = sendbuf.cpu()
            host_recv = torch.empty((int(sum(recv_counts)), sendbuf.shape[1]), dtype=sendbuf.dtype)
            self.dist.all_to_all_single(host_recv, host_send, output_split_sizes=recv_counts, input_split_sizes=send_counts)
            recvbuf = host_recv.to(sendbuf.device)
            torch.cuda.synchronize(sendbuf.device)
            self._import(h, dest_idx, recvbuf)
            return
        recvbuf = torch.empty((int(sum(recv_counts)), sendbuf.shape[1]), dtype=sendbuf.dtype, device=sendbuf.device)
        self.dist.all_to_all_single(recvbuf, sendbuf, output_split_sizes=recv_counts, input_split_sizes=send_counts)
        if recvbuf.is_cuda:
            # the collective is asynchronous on RCCL's stream; the unpack kernel runs on the handle's own
            # HIP stream, so make the received bytes final first
            torch.cuda.synchronize(recvbuf.device)
        self._import(h, dest_idx, recvbuf)
