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
    for n in (1, 2, 255, 256, 257, 1000, 1024, 1025, 8192, 8193, 4096, 4097, 1 << 19, (1 << 19) + 1, 1 << 20, 1 << 21, (1 << 21) + 1, 1 << 24, (1 << 24) + 1, 1 << 25, (1 << 25) + 1,
              1 << 26, (1 << 26) + 1, 1 << 27):
        for model in (1, 2, 3):
            assert L.lib().smc_auto_seg(model, n) == ob.lib().orc_auto_seg(model, n)
            assert (n + L.lib().smc_auto_seg(model, n) - 1) // L.lib().smc_auto_seg(model, n) <= 16384     # smc_create's segment limit


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
    "Ptr{Int64}": {"int64_t*"}, "Ptr{UInt32}": {"uint32_t*"}, "Ptr{Cint}": {"int*"},
}


def test_julia_binding_matches_header():
    """Julia cannot run in this image, so every `ccall((:name, LIBSMC), Ret, (ArgTypes...), ...)` of julia/hip_backend.jl
    is checked textually against include/smc_hip.h: the symbol exists, the return type, the arity and every argument
    type agree; types are defined before the first method that names them (the include-order bug of round 1); and the sampler
    entry points are methods on the GPU-backed `HipSMC` with the REFERENCE's signatures (src/smc_samplers.jl:74,103,163,222,288,
    308), so that existing callers reach them without a change at the call site."""
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
    need = {"smc_create", "smc_destroy", "smc_set_params", "smc_set_streams", "smc_init", "smc_step", "smc_log_likelihood", "smc_get_state",
            "smc_permute", "smc_pmmh_configure", "smc_pmmh_rejuvenate", "smc_step_window", "smc_step_commit", "smc_normalize",
            "smc_resample", "smc_get_quantiles", "smc_comm_unique_id", "smc_comm_create", "smc_outer_reweight",
            "smc_comm_all_gather", "smc_comm_exchange_slots", "smc_last_error",
            # the outer level: every host takes its numbers from the same library routines
            "smc_host_reweight", "smc_host_outer_temper", "smc_host_outer_resample", "smc_host_rw_factor", "smc_host_outer_window",
            "smc_host_outer_walk", "smc_host_outer_advance",
            # priors / model closures outside the enumerated families: host-side proposals, skipped filters, accepted clouds copied
            "smc_set_skip", "smc_copy_from"}
    assert need <= seen, need - seen
    # the reference's sampler signatures, as methods on HipSMC (= SMC{SSM,XT,θT,HipKernel}: more specific than the reference's)
    assert re.search(r"^const HipSMC\{SSM,XT,θT\} = SMC\{SSM,XT,θT,HipKernel\}", src, flags=re.M)
    for sig in (r"function resample!\(smc::HipSMC\)",                                                              # :74
                r"function rejuvenate!\(smc::HipSMC, y::Vector\{Float64\}, ξ::Float64, verbose::Bool\)",               # :103
                r"rejuvenate!\(smc::HipSMC, y::Vector\{Float64\}, verbose::Bool\) = rejuvenate!\(smc, y, 1\.0, verbose\)",   # :148
                r"function exchange!\(smc::HipSMC, y::Vector\{Float64\}, verbose::Bool\)",                            # :163
                r"function density_tempered\(smc::HipSMC, y::Vector\{Float64\}, verbose=true\)",                      # :222
                r"function smc²\(smc::HipSMC, y::Vector\{Float64\}\)",                                                # :288
                r"function smc²!\(smc::HipSMC, y::Vector\{Float64\}, t::Int64, verbose::Bool=true\)",                 # :308
                r"function expected_parameters\(smc::HipSMC\)",                                                      # :61
                r"function HipSMC\(N::Int64, M::Int64, model::SSM, prior::Sampleable, chain::Int64, ess_threshold::Float64, min_ar::Float64=-1\.0;",
                r"function SMC\(N::Int64, M::Int64, model::SSM, prior::MultivariateDistribution, chain::Int64, ess_threshold::Float64,",
                r"function HipSampler\(θ::Vector, model, prior;",
                r"function Base\.getproperty\(smc::HipSMC, s::Symbol\)"):
        assert re.search(sig, src), sig
    # no method takes the extra positional sampler argument of round 2 any more
    assert not re.search(r"\(smc::SMC, hs::HipSampler", src)
    # definition order: a struct is defined before any method signature or field names it
    for ty in ("HipFilter", "HipParticles", "HipWeights", "HipSampler", "HipComm", "HipKernel"):
        first_def = re.search(r"^(?:mutable )?struct %s\b" % ty, src, flags=re.M).start()
        uses = [m.start() for m in re.finditer(r"::%s\b" % ty, src)]
        assert uses and min(uses) > first_def, ty
    assert min(m.start() for m in re.finditer(r"::HipSMC\b", src)) > src.index("const HipSMC{SSM,XT,θT}")
    # AbstractVector contract of the two views: size and getindex for both
    assert re.search(r"Base\.getindex\(p::HipParticles", src) and re.search(r"Base\.getindex\(w::HipWeights", src)
    assert re.search(r"Base\.size\(p::Union\{HipParticles,HipWeights\}\)", src)


def test_host_reweight_is_normalize(L, ob):
    """smc_host_reweight == normalize (particles.jl:5-15): logmu = logsumexp(logw) - log n, w = softmax, ess = 1 / sum w^2 - and,
    bit for bit, the oracle's whole-vector restatement (orc_outer_reweight); dead entries (-inf, NaN) carry no weight."""
    from scipy.special import logsumexp
    rng = np.random.default_rng(2)
    for n in (1, 2, 7, 8, 9, 17, 512, 4096, 5000):
        logw = rng.normal(size=n) * 4 - 300
        lm, w, ess = L.host_reweight(logw)
        assert lm == pytest.approx(logsumexp(logw) - np.log(n), rel=1e-13) and abs(w.sum() - 1) < 1e-12
        assert np.allclose(w, np.exp(logw - logsumexp(logw)), rtol=1e-10, atol=1e-13) and ess == pytest.approx(1 / np.sum(w * w), rel=1e-10)   # 48-bit fixed point relative to the largest weight
        olm, ow, oess = ob.outer_reweight(logw)
        assert lm == olm and ess == oess and np.array_equal(bits(w), bits(ow))
        lm2, w2, ess2 = L.host_reweight(logw, want_w=False)
        assert w2 is None and (lm2, ess2) == (lm, ess)
    lm, w, ess = L.host_reweight(np.zeros(64))
    assert lm == 0.0 and ess == pytest.approx(64.0) and np.all(w == 1 / 64)
    lm, w, ess = L.host_reweight(np.array([-np.inf, 0.0, np.nan]))
    assert np.array_equal(w, [0.0, 1.0, 0.0]) and ess == 1.0 and lm == pytest.approx(-np.log(3))
    lm, w, ess = L.host_reweight(np.full(5, -np.inf))
    assert lm == -np.inf and ess == 0.0 and np.all(w == 0.0)
    # far-apart log-weights: what lies 2^-48 below the largest weight of its segment, or 2^-64 below the largest segment, is 0
    lw = np.array([0.0, -20.0, -30.0, -40.0] + [-1e3] * 8 + [-30.0])      # exp(-40) < 2^-48 < exp(-30)
    lm, w, ess = L.host_reweight(lw)
    assert w[0] > w[1] > w[2] > 0 and w[3] == 0 and np.all(w[4:12] == 0) and w[12] > 0
    assert np.array_equal(bits(w), bits(ob.outer_reweight(lw)[1]))


def test_outer_records_are_shardable(L, ob):
    """The outer level's sums are integer sums over fixed segments of 8 entries: the records a rank computes for the segments
    it holds are the records of the whole vector, whatever the number of ranks - so smc_host_outer_combine over the concatenated
    records IS smc_host_reweight of the concatenated vector; the window walk (smc_samplers.jl:323-338) by records ==
    the oracle's step-by-step walk, stopping below the ESS threshold included."""
    rng = np.random.default_rng(3)
    assert L.lib().smc_outer_seg() == L.OUTER_SEG == 8
    for n in (8, 64, 200, 4096):
        lw = rng.normal(size=n) * 6
        lw[rng.integers(0, n, 3)] = -np.inf
        lm, _, ess = L.host_reweight(lw)
        whole = L.host_outer_records(lw)
        assert L.host_outer_combine(whole, n) == (lm, ess)
        for world in (2, 4, 8):
            if n % (world * 8):
                continue
            per = n // world
            parts = np.concatenate([L.host_outer_records(lw[r * per:(r + 1) * per]) for r in range(world)])
            assert np.array_equal(parts, whole)
        k = 6
        lz = rng.normal(size=n)
        lik = rng.normal(size=(k, n)) * 0.7 - 1.5
        if n > 8:
            lik[1, 2] = -np.inf; lik[2, 4] = -900.0; lik[3, 6] = 710.0
        rec = L.host_outer_window(lw, lik)
        e_all, j_all = L.host_outer_walk(rec, n, 0.0)
        assert j_all == k
        for ess_min in (0.0, float(np.sort(e_all)[2]) + 1e-9, float(e_all.max()) + 1.0):
            e, j = L.host_outer_walk(rec, n, ess_min)
            lw2, lz2 = L.host_outer_advance(lw, lz, lik, j)
            olw, olz, oe, oj = ob.outer_steps(lw, lz, lik, ess_min)
            assert j == oj and np.array_equal(bits(e), bits(oe)) and np.array_equal(bits(lw2), bits(olw)) and np.array_equal(bits(lz2), bits(olz))
            first = int(np.argmax(e_all < ess_min)) + 1 if np.any(e_all < ess_min) else k
            assert j == first and np.array_equal(e, e_all[:j])
        if n % 16 == 0:      # two ranks, each walking its own half: the records side by side are the whole vector's
            h = n // 2
            both = np.concatenate([L.host_outer_window(lw[:h], lik[:, :h]), L.host_outer_window(lw[h:], lik[:, h:])], axis=1)
            assert np.array_equal(both, rec)


def test_outer_temper_resample_rw_factor_library_equals_oracle(L, ob):
    """The tempering bisection (smc_samplers.jl:240-266), the index draw of resample!(smc) (:74-84) and the random-walk factor
    (:87-101) of the library against the oracle's independent restatements, bit for bit; plus what they must be: the exponent
    puts the ESS at ess_min to the bisection's resolution, the ancestors are Multinomial(n, w) (chi-square) in ascending order,
    L L' is the reference's proposal covariance."""
    rng = np.random.default_rng(4)
    for n in (5, 24, 32, 512, 4096):
        lz = rng.normal(size=n) * 3 - 100
        for xi in (0.0, 0.3, 0.9):
            a, b = L.host_outer_temper(lz, xi, n * 0.5), ob.outer_temper(lz, xi, n * 0.5)
            assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2] and np.array_equal(bits(a[3]), bits(b[3]))
            if a[2]:
                assert xi < a[0] < 1.0 and abs(a[1] - n * 0.5) < 0.02 * n and L.host_reweight(a[3])[2] == a[1]
            else:
                assert a[0] == 1.0 and a[1] >= n * 0.5 - 0.02 * n
        lw = rng.normal(size=n) * 2
        a, b = L.host_outer_resample(lw, n, 777 + n), ob.outer_resample(lw, n, 777 + n)
        assert a.dtype == np.int32 and np.array_equal(a, b) and np.all(np.diff(a) >= 0) and 0 <= a.min() and a.max() < n
        lw[::3] = -np.inf
        a = L.host_outer_resample(lw, 2 * n + 1, 5)
        assert a.size == 2 * n + 1 and np.all(np.isfinite(lw[a])) and np.array_equal(a, ob.outer_resample(lw, 2 * n + 1, 5))
        assert np.array_equal(L.host_outer_resample(np.full(n, -np.inf), n, 1), np.arange(n))      # no weight at all: the identity
        for d in (1, 2, 3, 4, 8):
            if n <= d + 1:
                continue
            th = rng.normal(size=(n, d)) @ rng.normal(size=(d, d))
            (La, ua), (Lb, ub) = L.host_rw_factor(th), ob.rw_factor(th)
            assert ua == ub == (d == 1) and np.array_equal(bits(La), bits(Lb))
            if d > 1:
                assert np.allclose(La @ La.T, 2.83 ** 2 / d * np.cov(th.T) + 1e-10 * np.eye(d), rtol=1e-10) and np.all(np.triu(La, 1) == 0)
            else:
                assert La[0, 0] == pytest.approx(2.83 ** 2 * np.var(th[:, 0], ddof=1) + 1e-10, rel=1e-12)
    n = 64
    lw = rng.normal(size=n)
    w = L.host_reweight(lw)[1]
    cnt = np.zeros(n)
    for seed in range(400):
        cnt += np.bincount(L.host_outer_resample(lw, n, seed), minlength=n)
    chi2 = float(np.sum((cnt - 400 * n * w) ** 2 / (400 * n * w)))
    assert 30 < chi2 < 110, chi2                                   # 63 degrees of freedom


_HV_SNIPPET = r"""
import sys, json, numpy as np
sys.path.insert(0, %r)
from sequential_monte_carlo_amd import _lib as L
rng = np.random.default_rng(11)
out = []
for n in (1, 7, 64, 515, 4096):
    lw0 = rng.normal(size=n) * 3
    lz = rng.normal(size=n)
    lik = rng.normal(size=(5, n)) * 2.0
    if n > 8: lik[1, 2] = -np.inf; lik[2, 4] = -900.0; lik[3, 6] = 710.0; lik[4, 1] = np.nan
    rec = L.host_outer_window(lw0, lik)
    e, j = L.host_outer_walk(rec, n, 0.0)
    lw = rng.normal(size=n) * 40; lw[n // 2] = -np.inf
    lm, w, es = L.host_reweight(lw)
    t = L.host_outer_temper(lz * 30, 0.1, n * 0.5)
    out.append([rec.ravel().tolist(), e.view(np.uint64).tolist(), int(j), float(lm).hex(), np.asarray(w).view(np.uint64).tolist(),
                float(es).hex(), float(t[0]).hex(), float(t[1]).hex()])
print(json.dumps(out))
"""


def test_host_vector_path_equals_scalar_path():
    """The outer level's elementwise work (exp(logw) = p 2^k, the fixed-point weights and their sums) runs as vector code on
    hosts with AVX2 + FMA (csrc/smc_outer.hip); SMC_HOST_SCALAR=1 forces the scalar build of the same source: the same bits,
    with -inf / NaN entries, far-apart log-weights and lengths that are no multiple of the vector width or of a segment."""
    import subprocess
    code = _HV_SNIPPET % ROOT
    runs = []
    for env_extra in ({}, {"SMC_HOST_SCALAR": "1"}):
        env = dict(os.environ, **env_extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        runs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert runs[0] == runs[1]


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


def test_julia_binding_blocks_and_brackets_balance():
    """No Julia in the image: at least the gross structure of julia/hip_backend.jl is checked - every bracket closes, every block
    opener outside brackets (function, if, for, while, struct, try, ...) has its `end` (comments and string literals skipped)."""
    import re
    src = open(os.path.join(ROOT, "julia", "hip_backend.jl"), encoding="utf-8").read()
    out, i = [], 0
    while i < len(src):
        c = src[i]
        if c == "#":
            while i < len(src) and src[i] != "\n":
                i += 1
            continue
        if c == '"':
            i += 1
            while i < len(src) and src[i] != '"':
                i += 2 if src[i] == "\\" else 1
            i += 1
            out.append('""')
            continue
        out.append(c)
        i += 1
    code = "".join(out)
    openers = {"function", "if", "for", "while", "struct", "try", "let", "do", "begin", "quote", "module", "macro"}
    close = {")": "(", "]": "[", "}": "{"}
    brackets, blocks = [], []
    for m in re.finditer(r"[A-Za-z_²θξωμ!][A-Za-z_0-9²θξωμ!]*|[()\[\]{}]", code):
        t, line = m.group(0), code.count("\n", 0, m.start()) + 1
        if t in "([{":
            brackets.append((t, line))
        elif t in ")]}":
            assert brackets and brackets[-1][0] == close[t], "bracket mismatch at line %d" % line
            brackets.pop()
        elif not brackets and t in openers:
            blocks.append((t, line))
        elif not brackets and t == "end":
            assert blocks, "`end` without a block at line %d" % line
            blocks.pop()
    assert not brackets and not blocks, (brackets[:3], blocks[:3])
