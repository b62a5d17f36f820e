"""Can the C-level communicator (smc_comm_* over RCCL) be rehearsed with two ranks on ONE GPU?  (RCCL normally refuses two
ranks on the same device.)  Rank 0 writes the unique id to a file; both ranks open device 0.  Bounded by the caller's timeout."""
import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
from sequential_monte_carlo_amd import _lib as L
rank, path = int(sys.argv[1]), sys.argv[2]
if rank == 0:
    uid = L.comm_unique_id()
    open(path + ".tmp", "wb").write(uid); os.rename(path + ".tmp", path)
else:
    for _ in range(600):
        if os.path.exists(path): break
        time.sleep(0.05)
    uid = open(path, "rb").read()
try:
    c = L.Comm(uid, rank, 2, device=0)
except Exception as e:
    print("rank %d: smc_comm_create refused: %s" % (rank, e), flush=True)
    sys.exit(3)
out = c.all_gather(np.arange(4.0) + 10 * rank)
print("rank %d: all_gather ->" % rank, out, flush=True)
lm, w, ess, allw = c.outer_reweight(np.log(np.arange(1.0, 9.0) + 8 * rank))
print("rank %d: outer_reweight logmu %.17g ess %.17g" % (rank, lm, ess), flush=True)
c.close()
