"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol the
header declares, its host build of the numerical spec equals the oracle bit for bit, and the
entry points fail loudly (never fall back) without a GPU."""
import ctypes as C
import json
import sys
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_library_exports_every_declared_symbol(L):
    hdr = open(os.path.join(ROOT, "include", "smc_hip.h")).read()
    declared = set(re.findall(r"\b(smc_[A-Za-z0-9_]+)\s*\(", hdr))
    declared.discard("smc_filter_s")
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    lib = L.lib()
    for name in declared:
        assert getattr(lib, name) is not None
    assert b"gfx950" in lib.smc_version()


def test_library_embeds_gfx950_code_object(L):
    blob = open(L.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"k_step" in blob and b"k_resident" in blob


def test_host_math_equals_oracle(L, ob):
    rng = np.random.default_rng(0)
    lib = L.lib()
    for x in np.concatenate([rng.uniform(-708, 10, 3000), [0.0, -745.0, 709.5, np.inf, -np.inf]]):
        assert bits([lib.smc_host_exp(float(x))])[0] == bits(ob.exp([x]))[0]
    for x in np.concatenate([np.exp(rng.uniform(-740, 700, 3000)), [0.0, 1.0, 5e-324, np.inf]]):
        assert bits([lib.smc_host_log(float(x))])[0] == bits(ob.log([x]))[0]
    for _ in range(2000):
        w = [int(v) for v in rng.integers(0, 2**32, 4)]
        z0, z1 = C.c_double(), C.c_double()
        lib.smc_host_box_muller((C.c_uint32 * 4)(*w), C.byref(z0), C.byref(z1))
        assert (z0.value, z1.value) == ob.box_muller(w)
        out = (C.c_uint32 * 4)()
        lib.smc_host_philox4x32_10((C.c_uint32 * 4)(*w), (C.c_uint32 * 2)(w[0], w[3]), out)
        assert list(out) == ob.philox(w, [w[0], w[3]])


@pytest.mark.parametrize("model,raw", [(1, [0.5, 1.0, 0.9, 0.8, 0.0, 1.0]), (2, [-1.0, 0.95, 0.25]),
                                       (3, [0.2, 0.2, 3.0, 0.0, 0.0])])
def test_simulate_equals_oracle(L, ob, model, raw):
    x, y = L.simulate(model, raw, 300, 1998)
    ox, oy = ob.simulate(model, raw, 300, 1998)
    assert np.array_equal(bits(x), bits(ox)) and np.array_equal(bits(y), bits(oy))
    assert L.lib().smc_model_dim(model) == ob.MODEL_DIM[model] and L.lib().smc_model_nraw(model) == ob.MODEL_NRAW[model]


def test_auto_seg_equals_oracle(L, ob):
    for n in (1, 2, 255, 256, 257, 1000, 1024, 1025, 8192, 8193, 1 << 20, 1 << 21, (1 << 21) + 1, 1 << 24, (1 << 24) + 1, 1 << 25):
        assert L.lib().smc_auto_seg(n) == ob.lib().orc_auto_seg(n)
        assert (n + L.lib().smc_auto_seg(n) - 1) // L.lib().smc_auto_seg(n) <= 4096     # smc_create's segment limit


def test_systematic_targets_are_exact(L):
    """The kernels' division-free T_j = floor((j Dtot + mulhi64(u, Dtot)) / n) against Python's exact integers:
    powers of two and awkward n, tiny and huge Dtot, every region of j."""
    rng = np.random.default_rng(7)
    cases = [(1, 1, 0, 0, 1), (5, 3, 2**63 + 11, 0, 3), ((1 << 63) - 1, (1 << 31) - 1, 2**64 - 1, (1 << 31) - 1 - 8192, 8192),
             ((1 << 62) + 12345, 1 << 20, 123456789123456789, (1 << 20) - 2048, 2048), (977, 1 << 20, 2**63, 4096, 4096)]
    for _ in range(300):
        n = int(rng.integers(1, 1 << 31)) if rng.random() < 0.7 else 1 << int(rng.integers(0, 31))
        D = int(rng.integers(0, 1 << 63)) >> int(rng.integers(0, 62))
        nk = int(min(n, rng.integers(1, 8193)))
        j0 = int(rng.integers(0, n - nk + 1))
        cases.append((D, n, int(rng.integers(0, 1 << 64, dtype=np.uint64)), j0, nk))
    for D, n, u, j0, nk in cases:
        got = L.sys_targets(D, n, u, j0, nk)
        v0 = (u * D) >> 64
        want = [((j0 + k) * D + v0) // n for k in range(nk)]
        assert [int(g) for g in got] == want, (D, n, u, j0, nk)
        assert not D or max(want) < D


def test_argument_errors_are_reported_not_fatal(L):
    lib = L.lib()
    h = C.c_void_p()
    assert lib.smc_create(99, 1, 16, 0, 1, 0, 0, C.byref(h)) == -1 and b"unknown model" in lib.smc_last_error()
    assert lib.smc_create(1, 0, 16, 0, 1, 0, 0, C.byref(h)) == -1
    assert lib.smc_create(1, 1, 16, 300, 1, 0, 0, C.byref(h)) == -1 and b"power of two" in lib.smc_last_error()
    assert lib.smc_simulate(7, None, 3, 1, None, None) == -1
    assert lib.smc_step(None, 0.0, None, None) == -1
    assert lib.smc_destroy(None) == 0


def test_no_gpu_means_error_not_fallback(L):
    """Without a HIP device the filter entry points must fail loudly (no CPU path exists)."""
    if L.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(L.SmcError):
        L.Handle(L.MODEL_LG1D, 1, 64)
    with pytest.raises(L.SmcError):
        L.normalize(np.zeros(4))
    with pytest.raises(L.SmcError):
        L.resample(np.ones(4) / 4)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sequential_monte_carlo_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in src.replace("the oracle", "").replace("oracle's", "").replace(
                    "vs oracle", "").replace("against the oracle", ""), fn


def test_bench_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` without torch.distributed.run: the launcher process starts two fresh ranks
    (before anything touches a GPU), they form a gloo group, and rank 0's JSON line comes back through the parent."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--dry-launch"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line == {"dry_launch": True, "n_gpus": 2, "ranks_seen": 2, "dist_backend": "gloo"}
    # a failing rank makes the launcher fail (bad flag -> argparse exits 2 in every rank)
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--workload", "nope"],
                         capture_output=True, text=True, timeout=120, env=env)
    assert bad.returncode != 0


def _header_signatures():
    """{name: (return type, [argument types])} of every function include/smc_hip.h declares, const and names dropped"""
    hdr = open(os.path.join(ROOT, "include", "smc_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    sigs = {}
    for ret, name, args in re.findall(r"\b(int|double|void|const char\s*\*)\s+(smc_\w+)\s*\(([^;{}]*?)\)\s*;", hdr, flags=re.S):
        types = []
        for a in args.split(","):
            a = re.sub(r"\bconst\b", " ", a).strip()
            if a in ("void", ""):
                continue
            m = re.match(r"(.*?)(\b\w+\b)?\s*(\[\d*\])?\s*$", a)          # type, [parameter name], [array suffix]
            ty = a if "*" in a and a.rstrip().endswith("*") else m.group(1)
            ty = re.sub(r"\s+", "", ty) + ("*" if m.group(3) else "")
            types.append(ty)
        sigs[name] = (re.sub(r"\s+|const", "", ret), types)
    return sigs


JULIA_TO_C = {
    "Cint": {"int"}, "Int64": {"int64_t"}, "UInt64": {"uint64_t"}, "UInt32": {"uint32_t"}, "Float64": {"double"},
    "Cstring": {"char*"}, "Ptr{Cvoid}": {"smc_handle", "smc_comm", "void*"}, "Ref{Ptr{Cvoid}}": {"smc_handle*", "smc_comm*"},
    "Ptr{Float64}": {"double*"}, "Ptr{Int32}": {"int32_t*"}, "Ptr{UInt8}": {"uint8_t*"}, "Ptr{UInt64}": {"uint64_t*"},
    "Ptr{Int64}": {"int64_t*"}, "Ptr{UInt32}": {"uint32_t*"},
}


def test_julia_binding_matches_header():
    """Julia cannot run in this image, so every `ccall((:name, LIBSMC), Ret, (ArgTypes...), ...)` of julia/hip_backend.jl
    is checked textually against include/smc_hip.h: the symbol exists, the return type, the arity and every argument
    type agree; and types are defined before the first method that names them (the include-order bug of round 1)."""
    src = open(os.path.join(ROOT, "julia", "hip_backend.jl")).read()
    sigs = _header_signatures()
    calls = re.findall(r"ccall\(\(:(\w+),\s*LIBSMC\),\s*([\w{}]+),\s*\(([^()]*)\)", src, flags=re.S)
    assert len(calls) >= 25
    seen = set()
    for name, ret, args in calls:
        assert name in sigs, "ccall to undeclared symbol %s" % name
        cret, cargs = sigs[name]
        assert cret in JULIA_TO_C[ret], (name, ret, cret)
        jargs = [a.strip() for a in args.replace("\n", " ").split(",") if a.strip()]
        assert len(jargs) == len(cargs), (name, jargs, cargs)
        for j, c in zip(jargs, cargs):
            assert c in JULIA_TO_C[j], (name, j, c)
        seen.add(name)
    # the entry points a sampler host needs are all bound
    need = {"smc_create", "smc_destroy", "smc_set_params", "smc_init", "smc_step", "smc_log_likelihood", "smc_get_state",
            "smc_permute", "smc_pmmh_configure", "smc_pmmh_rejuvenate", "smc_step_window", "smc_step_commit", "smc_normalize",
            "smc_resample", "smc_get_quantiles", "smc_comm_unique_id", "smc_comm_create", "smc_outer_reweight",
            "smc_comm_all_gather", "smc_comm_exchange_slots", "smc_last_error"}
    assert need <= seen, need - seen
    # definition order: a struct is defined before any method signature or field names it
    for ty in ("HipFilter", "HipParticles", "HipWeights", "HipSampler", "HipComm"):
        first_def = re.search(r"^(?:mutable )?struct %s\b" % ty, src, flags=re.M).start()
        uses = [m.start() for m in re.finditer(r"::%s\b" % ty, src)]
        assert uses and min(uses) > first_def, ty
    # AbstractVector contract of the two views: size and getindex for both
    assert re.search(r"Base\.getindex\(p::HipParticles", src) and re.search(r"Base\.getindex\(w::HipWeights", src)
    assert re.search(r"Base\.size\(p::Union\{HipParticles,HipWeights\}\)", src)


def test_host_reweight_is_normalize(L):
    """smc_host_reweight == normalize (particles.jl:5-15): logmu = logsumexp(logw) - log n, w = softmax, ess = 1 / sum w^2;
    smc_host_outer_steps == that many smc²! host halves one after the other, stopping below the ESS threshold."""
    from scipy.special import logsumexp
    rng = np.random.default_rng(2)
    for n in (1, 2, 17, 512, 4096):
        logw = rng.normal(size=n) * 4 - 300
        lm, w, ess = L.host_reweight(logw)
        assert lm == pytest.approx(logsumexp(logw) - np.log(n), rel=1e-13) and abs(w.sum() - 1) < 1e-12
        assert np.allclose(w, np.exp(logw - logsumexp(logw)), rtol=1e-12) and ess == pytest.approx(1 / np.sum(w * w), rel=1e-12)
    lm, w, ess = L.host_reweight(np.zeros(64))
    assert lm == 0.0 and ess == pytest.approx(64.0) and np.all(w == 1 / 64)
    lm, w, ess = L.host_reweight(np.array([-np.inf, 0.0, -np.inf]))
    assert np.array_equal(w, [0.0, 1.0, 0.0]) and ess == 1.0 and lm == pytest.approx(-np.log(3))
    lm, w, ess = L.host_reweight(np.full(5, -np.inf))
    assert lm == -np.inf and ess == 0.0 and np.all(w == 0.2)
    M, k = 48, 6
    lik = rng.normal(size=(k, M)) * 0.7 - 1.5
    omega0, logZ0 = np.full(M, 1.0 / M), rng.normal(size=M)
    om, lz, ess, j = L.host_outer_steps(omega0, logZ0, lik, ess_min=0.0)
    assert j == k and ess.shape == (k,)
    o, z = omega0.copy(), logZ0.copy()
    for i in range(k):
        _, o, e = L.host_reweight(np.array([L.lib().smc_host_log(float(v)) for v in o]) + lik[i])
        z = z + lik[i]
        assert e == ess[i]
    assert np.array_equal(bits(o), bits(om)) and np.array_equal(bits(z), bits(lz))
    thr = float(np.sort(ess)[2])                                   # stops after the first step below the threshold
    om2, lz2, ess2, j2 = L.host_outer_steps(omega0, logZ0, lik, ess_min=thr + 1e-9)
    first = int(np.argmax(ess < thr + 1e-9)) + 1
    assert j2 == first and np.array_equal(ess2, ess[:first]) and np.array_equal(bits(lz2), bits(logZ0 + lik[:first].sum(axis=0))) or j2 == first


_HV_SNIPPET = r"""
import sys, json, numpy as np
sys.path.insert(0, %r)
from sequential_monte_carlo_amd import _lib as L
rng = np.random.default_rng(11)
out = []
for n in (1, 7, 64, 515, 4096):
    om = rng.random(n); om[0] = 0.0
    if n > 8: om[3] = 5e-320; om[5] = 2.3e-308
    om /= om.sum()
    lz = rng.normal(size=n)
    lik = rng.normal(size=(5, n)) * 2.0
    if n > 8: lik[1, 2] = -np.inf; lik[2, 4] = -900.0; lik[3, 6] = 710.0
    o, z, e, j = L.host_outer_steps(om, lz, lik, 0.0)
    lw = rng.normal(size=n) * 40; lw[n // 2] = -np.inf
    lm, w, es = L.host_reweight(lw)
    out.append([o.view(np.uint64).tolist(), z.view(np.uint64).tolist(), e.view(np.uint64).tolist(), int(j),
                float(lm).hex(), np.asarray(w).view(np.uint64).tolist(), float(es).hex()])
print(json.dumps(out))
"""


def test_host_vector_path_equals_scalar_path():
    """The outer reweight's elementwise halves run as 4-wide vector code on hosts with AVX2 + FMA (smc_capi.hip, hv_*);
    SMC_HOST_SCALAR=1 forces the scalar sp_log / sp_exp calls: the same bits, weights of zero, subnormal weights, -inf and
    far-apart log-weights and lengths that are no multiple of the vector width included."""
    import subprocess
    code = _HV_SNIPPET % ROOT
    runs = []
    for env_extra in ({}, {"SMC_HOST_SCALAR": "1"}):
        env = dict(os.environ, **env_extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        runs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert runs[0] == runs[1]


def test_host_resample_sorted_is_numpy_choice_sorted(L):
    """smc_host_resample_sorted == sort(Generator.choice(n, m, p=w)) for the same uniforms (numpy looks its uniforms up in
    cumsum(p) / cumsum(p)[-1] with side="right"): the index draw of resample!(smc), smc_samplers.jl:74-84."""
    for trial in range(60):
        g = np.random.default_rng(trial)
        n = int(g.integers(1, 700))
        w = g.random(n) ** 6
        if trial % 3 == 0:
            w[g.integers(0, n, n // 3 + 1)] = 0.0
        if w.sum() == 0.0:
            w[-1] = 1.0
        w = w / w.sum()
        r1, r2 = np.random.default_rng(100 + trial), np.random.default_rng(100 + trial)
        a = L.host_resample_sorted(w, np.sort(r1.random(n)))
        assert a.dtype == np.int32 and np.array_equal(a, np.sort(r2.choice(n, size=n, replace=True, p=w)))
        assert np.all(w[a] > 0)                                   # never an ancestor of weight zero
    with pytest.raises(RuntimeError):
        L.host_resample_sorted(np.array([0.5, 0.5]), np.array([0.7, 0.2]))      # not sorted
    with pytest.raises(RuntimeError):
        L.host_resample_sorted(np.zeros(4), np.array([0.1]))


def test_exchange_plan_c_equals_python_and_is_consistent(L):
    """smc_comm_plan_exchange (the host arithmetic of the C-level all-to-all of filter slots) for worlds of 1..8 ranks:
    identical to the plan distributed.ThetaComm.exchange_slots derives, every rank's sends match the other ranks' receives,
    and replaying the plan on plain arrays performs resample!(smc) (slot m <- slot a[m], smc_samplers.jl:74-84)."""
    import ctypes as C
    rng = np.random.default_rng(9)
    lib = L.lib()
    i32p, i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    for world in (1, 2, 3, 4, 8):
        per = 6
        M = per * world
        for trial in range(4):
            a = np.sort(rng.integers(0, M, M)).astype(np.int32) if trial % 2 else rng.integers(0, M, M).astype(np.int32)
            plans = []
            for rank in range(world):
                send_idx, dest_idx = np.zeros(M, np.int32), np.zeros(per, np.int32)
                send_cnt, recv_cnt, ns = np.zeros(world, np.int64), np.zeros(world, np.int64), C.c_int64()
                assert lib.smc_comm_plan_exchange(a.ctypes.data_as(i32p), M, rank, world, send_idx.ctypes.data_as(i32p),
                                                  send_cnt.ctypes.data_as(i64p), C.byref(ns), dest_idx.ctypes.data_as(i32p),
                                                  recv_cnt.ctypes.data_as(i64p)) == 0
                send_idx = send_idx[:ns.value]
                # the Python twin (distributed.py)
                lo = rank * per
                p_send = [a[r * per:(r + 1) * per][(a[r * per:(r + 1) * per] // per) == rank] - lo for r in range(world)]
                owners = a[lo:lo + per] // per
                p_dest = np.concatenate([np.nonzero(owners == s)[0] for s in range(world)])
                assert np.array_equal(send_idx, np.concatenate(p_send)) and [len(q) for q in p_send] == list(send_cnt)
                assert np.array_equal(dest_idx, p_dest) and [int((owners == s).sum()) for s in range(world)] == list(recv_cnt)
                plans.append((send_idx, send_cnt, dest_idx, recv_cnt))
            # replay on plain data: slot value = global slot id
            data = [np.arange(r * per, (r + 1) * per) for r in range(world)]
            new = [d.copy() for d in data]
            for r in range(world):                       # receiver
                _, _, dest_idx, recv_cnt = plans[r]
                off = 0
                for s in range(world):                   # sender
                    send_idx, send_cnt, _, _ = plans[s]
                    start = int(send_cnt[:r].sum())
                    chunk = data[s][send_idx[start:start + int(send_cnt[r])]]
                    assert len(chunk) == recv_cnt[s]
                    new[r][dest_idx[off:off + len(chunk)]] = chunk
                    off += len(chunk)
            assert np.array_equal(np.concatenate(new), a)
    assert lib.smc_comm_plan_exchange(None, 4, 0, 2, None, None, None, None, None) == -1
