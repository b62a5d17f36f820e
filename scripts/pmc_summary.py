"""Per-launch averages of the rocprofv3 --pmc passes of scripts/collect_pmc.sh for the workload's dominant kernel
(k_step for c2 / c3, k_resident for c4 / c5) -> the JSON bench.py reads from profiles/pmc_<workload>.json."""
import csv, glob, hashlib, json, os, sys
wl, commit, root = sys.argv[1], sys.argv[2], sys.argv[3]


def kernel_source_hash():
    """sha256 over the kernel sources of the running build (csrc/*.h, *.hip in name order): bench.py compares it with the hash of
    ITS build and flags counter-derived fields as stale when they differ"""
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sequential_monte_carlo_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(here, "*.h")) + glob.glob(os.path.join(here, "*.hip"))):
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return h.hexdigest()[:16]


want = "k_step<" if wl in ("c2", "c3") else "k_resident<"
tot, cnt, kname = {}, {}, None
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if want not in k:
            continue
        kname = k
        c = r["Counter_Name"]
        tot[c] = tot.get(c, 0.0) + float(r["Counter_Value"])
        cnt[c] = cnt.get(c, 0) + 1
# a dispatch appears once per counter: per-launch average = sum / number of dispatches seen for that counter
avg = {c: tot[c] / cnt[c] for c in tot}
nx, nth, d = (1 << 20, 1, 1) if wl in ("c2", "c3") else (1024, 512, 1 if wl == "c4" else 3)
units = nx * nth * (1 if wl in ("c2", "c3") else 60)          # particle-steps per launch (prof_c2.py: T = 60)
out = {"workload": "%s (scripts/prof_c2.py 60 0 %s), kernel %s" % (wl, wl, kname), "commit": commit,
       "command": "rocprofv3 --kernel-trace --pmc <group> --output-format csv -- python3 scripts/prof_c2.py 60 0 %s ; one pass per group (scripts/collect_pmc.sh)" % wl,
       "kernel_source_sha16": kernel_source_hash(),
       "launches_averaged": max(cnt.values()) if cnt else 0, "counters_per_launch": avg}
if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
    out["FETCH_SIZE_KB_per_launch"], out["WRITE_SIZE_KB_per_launch"] = avg["FETCH_SIZE"], avg["WRITE_SIZE"]
    out["correction"] = ("gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced 16-B-per-lane reads (MI355X_MICROARCH.md, HBM "
                         "section) -> read bytes = 2*FETCH_SIZE*1024; WRITE_SIZE exact for 16-B-per-lane stores; the 8-byte gathers of x "
                         "are an uncalibrated width")
    out["hbm_bytes_per_launch"] = int(2 * avg["FETCH_SIZE"] * 1024 + avg["WRITE_SIZE"] * 1024)
    out["algorithmic_bytes_per_launch"] = units * (32 + 16 * d)
if "SQ_INSTS_VALU" in avg:
    out["valu_wave_insts_per_launch"] = avg["SQ_INSTS_VALU"]
    out["valu_insts_per_particle_step"] = avg["SQ_INSTS_VALU"] * 64 / units
if "SQ_WAIT_ANY" in avg and "SQ_WAVE_CYCLES" in avg:
    out["wait_fraction_of_wave_cycles"] = avg["SQ_WAIT_ANY"] / avg["SQ_WAVE_CYCLES"]
if "SQ_LDS_BANK_CONFLICT" in avg and avg.get("SQ_LDS_IDX_ACTIVE"):
    out["lds_bank_conflict_fraction"] = avg["SQ_LDS_BANK_CONFLICT"] / avg["SQ_LDS_IDX_ACTIVE"]
print(json.dumps(out, indent=1))
