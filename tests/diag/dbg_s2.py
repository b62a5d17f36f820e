import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import binding as ob
from sequential_monte_carlo_amd import _lib as L
LG = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
x, y = ob.simulate(ob.LG1D, LG, 3, 1998)
for flags in (L.FLAG_NO_RESIDENT, 0):
    h = L.Handle(L.MODEL_LG1D, 1, 1024, seed=7, flags=flags); h.set_params(LG)
    Z, lm, es = h.log_likelihood(y, trace=True)
    C, m, S, hi, lo = h.weights_raw()
    q = np.diff(np.concatenate([[0], C[0]])).astype(object)
    s2 = int(sum(int(v) * int(v) for v in q))
    f = ob.Filter(ob.LG1D, LG, 1024, seed=7); z, olm, oes = f.log_likelihood(y, trace=True); oC, om, oS, ohi, olo = f.weights_raw()
    print("exact  ", hex(s2 >> 64), hex(s2 & (2**64 - 1)))
    print("gpu    ", hex(int(hi[0, 0])), hex(int(lo[0, 0])), "ess", es[:, 0])
    print("oracle ", hex(int(ohi[0])), hex(int(olo[0])), "ess", oes)
