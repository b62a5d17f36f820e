import sys, time
sys.path.insert(0, "/root/repo")
from sequential_monte_carlo_amd import _lib as L
import torch
tc = td = 0.0
for it in range(40):
    t0 = time.perf_counter(); h = L.Handle(1, 1, 1024, seed=3); t1 = time.perf_counter(); h.close(); t2 = time.perf_counter()
    if it >= 10: tc += t1 - t0; td += t2 - t1
print("smc_create %.3f ms, smc_destroy %.3f ms" % (tc / 30 * 1e3, td / 30 * 1e3))
# the pieces, through torch's bindings of the same runtime
def tm(f, n=30):
    for _ in range(5): f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
print("torch.cuda.Stream() create+destroy %.1f us" % tm(lambda: torch.cuda.Stream()))
print("torch.cuda.Event() create+record+destroy %.1f us" % tm(lambda: torch.cuda.Event(enable_timing=True).record()))
print("torch.empty(64 KB, cuda) alloc+free through the caching allocator %.1f us (cached)" % tm(lambda: torch.empty(8192, device='cuda')))
print("pinned 32 B alloc+free %.1f us" % tm(lambda: torch.empty(4, pin_memory=True)))
