#!/bin/bash
# Collects the PMC counters bench.py quotes (profiles/pmc_<workload>.json) on the GPU box: separate rocprofv3 --pmc
# passes with --kernel-trace only (MI355X_MICROARCH.md "rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE do not fit
# one pass), then scripts/pmc_summary.py turns the per-dispatch CSVs into one JSON per workload, stamped with the commit.
# usage (from the repo root, on the box):  bash scripts/collect_pmc.sh <commit-id> [c2 c4 ...]
set -e
COMMIT=${1:-unknown}; shift || true
WLS=${@:-c2 c4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for wl in $WLS; do
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    tag=$(echo $grp | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmc_${wl}/$tag -o p -- python3 scripts/prof_c2.py 60 0 $wl > gpurun_out/pmc_${wl}_$tag.log 2>&1
  done
  python3 scripts/pmc_summary.py $wl $COMMIT gpurun_out/pmc_${wl} > gpurun_out/pmc_${wl}.json
  cat gpurun_out/pmc_${wl}.json
done
