cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SMC_LIB=$PWD/sequential_monte_carlo_amd/csrc/build_abl/libsmchip_abl.so
for abl in 0 1 2 4 8 16 64 95; do
  SMC_ABL=$abl rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/ablpmc/$abl -o p -- python3 scripts/prof_c2.py 40 0 c2 > /dev/null 2>&1
  python3 - <<PY
import csv,glob
tot={};cnt={}
for f in glob.glob("gpurun_out/ablpmc/$abl/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_step<" in r["Kernel_Name"]:
            c=r["Counter_Name"]; tot[c]=tot.get(c,0)+float(r["Counter_Value"]); cnt[c]=cnt.get(c,0)+1
print("abl=$abl", {c: round(tot[c]/cnt[c]*64/(1<<20),1) for c in tot})
PY
done
