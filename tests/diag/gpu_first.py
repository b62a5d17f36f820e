"""GPU diagnostic: math parity, filter parity vs oracle (prints WHERE things differ), rough timing.
Development aid; the judged parity tests live in tests/ (pytest -m gpu)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from oracle import binding as ob
from sequential_monte_carlo_amd import _lib as L

print("devices", L.device_count(), L.lib().smc_version())
rng = np.random.default_rng(0)
N = 1 << 20


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


# ---- math parity
xs = np.concatenate([rng.uniform(-708, 5, N // 2), rng.uniform(-2, 2, N // 2)])
g = L.device_math(0, xs)
o = ob.exp(xs[:20000])
print("exp mismatches (20k)", int((bits(g[:20000]) != bits(o)).sum()))
hostv = np.array([L.lib().smc_host_exp(float(v)) for v in xs[:20000]])
print("exp host-vs-dev", int((bits(g[:20000]) != bits(hostv)).sum()))
xs = np.concatenate([rng.uniform(0, 1, N // 2), np.exp(rng.uniform(-700, 700, N // 2))])
g = L.device_math(1, xs)
o = ob.log(xs[:20000])
print("log mismatches", int((bits(g[:20000]) != bits(o)).sum()))
xs = np.exp(rng.uniform(-50, 50, N))
g = L.device_math(2, xs)
print("sqrt mismatches vs numpy", int((bits(g) != bits(np.sqrt(xs))).sum()))
a = rng.normal(size=N) * np.exp(rng.uniform(-20, 20, N))
b = rng.normal(size=N) * np.exp(rng.uniform(-20, 20, N))
g = L.device_math(5, a, b)
print("div mismatches vs numpy", int((bits(g) != bits(a / b)).sum()))
wa = rng.integers(0, 2**64, N, dtype=np.uint64)
wb = rng.integers(0, 2**64, N, dtype=np.uint64)
z0 = L.device_math(3, wa.view(np.float64), wb.view(np.float64))
z1 = L.device_math(4, wa.view(np.float64), wb.view(np.float64))
mm = 0
for i in range(20000):
    ua, ub = int(wa[i]), int(wb[i])
    r0, r1 = ob.box_muller([ua & 0xFFFFFFFF, ua >> 32, ub & 0xFFFFFFFF, ub >> 32])
    mm += int(bits(np.array([r0]))[0] != bits(z0[i:i + 1])[0]) + int(bits(np.array([r1]))[0] != bits(z1[i:i + 1])[0])
print("box-muller mismatches (20k)", mm, "z mean/std", z0.mean(), z0.std(), z1.std())


def cmp(name, model, raw, n, T, seg, flags=L.FLAG_ANCESTORS, ntheta=1):
    x, y = ob.simulate(model, raw, T, 1998)
    h = L.Handle(model, ntheta, n, seg=seg, seed=7, flags=flags)
    h.set_params(np.tile(raw, (ntheta, 1)))
    t0 = time.time()
    Z, lm, es = h.log_likelihood(y, trace=True)
    dt = time.time() - t0
    xs_, w_, a_ = h.state()
    C_, m_, S_, hi_, lo_ = h.weights_raw()
    bad = 0
    for th in range(min(ntheta, 3)):
        f = ob.Filter(model, raw, n, seg=seg, seed=7, stream=th)
        z, olm, oes = f.log_likelihood(y, trace=True)
        ox, ow, oa, _ = f.state()
        oC, om, oS, ohi, olo = f.weights_raw()
        d = dict(logZ=Z[th] == z, logmu=np.array_equal(lm[:, th], olm), ess=np.array_equal(es[:, th], oes),
                 x=np.array_equal(xs_[:, th], ox), w=np.array_equal(w_[th], ow), anc=np.array_equal(a_[th], oa),
                 C=np.array_equal(C_[th], oC), m=np.array_equal(m_[th], om), S=np.array_equal(S_[th], oS),
                 S2=np.array_equal(hi_[th], ohi) and np.array_equal(lo_[th], olo))
        if not all(d.values()):
            bad += 1
            print("  MISMATCH", name, "theta", th, d, "logZ gpu/orc", Z[th], z)
            k = np.nonzero(lm[:, th] != olm)[0]
            if k.size:
                print("   first logmu diff at t=", k[0], lm[k[0], th], olm[k[0]])
    print("%-36s resident=%d seg=%d nseg=%d  %s  wall %.1f ms dev %.3f ms" % (
        name, h.resident, h.seg, h.nseg, "OK" if not bad else "FAIL", dt * 1e3, h.elapsed_ms()))
    sys.stdout.flush()
    h.close()


LG = [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]
SV = [-1.0, 0.95, 0.25]
UC = [0.2, 0.2, 3.0, 0.0, 0.0]
for flags in (L.FLAG_ANCESTORS | L.FLAG_NO_RESIDENT, L.FLAG_ANCESTORS):
    cmp("LG n=1024 T=50", ob.LG1D, LG, 1024, 50, 0, flags)
    cmp("LG n=1000 T=20 (ragged)", ob.LG1D, LG, 1000, 20, 0, flags)
    cmp("LG n=300 seg512 T=20", ob.LG1D, LG, 300, 20, 512, flags)
    cmp("SV n=2048 T=30", ob.SV1D, SV, 2048, 30, 0, flags)
    cmp("UCSV n=1024 T=30", ob.UCSV3D, UC, 1024, 30, 0, flags)
    cmp("LG n=4096 T=10 ntheta=4", ob.LG1D, LG, 4096, 10, 0, flags, ntheta=4)
cmp("LG n=5000 seg256 T=20 (20 segs)", ob.LG1D, LG, 5000, 20, 256)
cmp("LG n=10000 seg1024 T=10", ob.LG1D, LG, 10000, 10, 1024)
cmp("LG n=65536 seg2048 T=10", ob.LG1D, LG, 65536, 10, 2048)
cmp("SV n=20000 seg4096 T=10", ob.SV1D, SV, 20000, 10, 4096)
cmp("UCSV n=30000 seg8192 T=10", ob.UCSV3D, UC, 30000, 10, 8192)
cmp("UCSV n=9000 seg512 T=10 ntheta=3", ob.UCSV3D, UC, 9000, 10, 512, ntheta=3)
# ---- speed
for seg in (1024, 2048, 4096):
    n = 1 << 20
    T = 200
    x, y = ob.simulate(ob.LG1D, LG, T, 1998)
    h = L.Handle(L.MODEL_LG1D, 1, n, seg=seg, seed=1)
    h.set_params(LG)
    h.log_likelihood(y[:20])
    Z = h.log_likelihood(y)
    ms = h.elapsed_ms()
    print("LG n=2^20 T=%d seg=%d: %.3f ms  -> %.3e p-steps/s, %.2f us/step, logZ=%.4f" % (
        T, seg, ms, n * T / ms * 1e3, ms / T * 1e3, Z[0]))
    h.close()
n, nth, T = 1024, 512, 200
x, y = ob.simulate(ob.LG1D, LG, T, 1998)
for flags in (0, L.FLAG_NO_RESIDENT):
    h = L.Handle(L.MODEL_LG1D, nth, n, seed=1, flags=flags)
    h.set_params(np.tile(LG, (nth, 1)))
    h.log_likelihood(y[:10])
    Z = h.log_likelihood(y)
    ms = h.elapsed_ms()
    print("LG batched 512x1024 T=200 resident=%d: %.3f ms -> %.3e p-steps/s" % (h.resident, ms, n * nth * T / ms * 1e3))
    h.close()
