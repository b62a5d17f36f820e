#!/bin/bash
# Everything the judged numbers come from, in one GPU-box call (run from the repo root through gpurun):
#   bench lines of every workload, the rocprofv3 --kernel-trace --stats summary of the SAME bench command, and the PMC
#   passes (scripts/collect_pmc.sh).  Results land in gpurun_out/$ROUND/ ; copy what is to be judged into profiles/.
# usage: bash scripts/refresh_profiles.sh <commit-id> [round tag, default r03]
COMMIT=${1:-unknown}
ROUND=${2:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$ROUND; mkdir -p $O
bash scripts/collect_pmc.sh $COMMIT c2 c4 c3 c5 > $O/pmc.log 2>&1
cp gpurun_out/pmc_c2.json gpurun_out/pmc_c4.json gpurun_out/pmc_c3.json gpurun_out/pmc_c5.json $O/
# bench reads profiles/pmc_<wl>.json for the counter-derived fields: use the ones just collected
cp gpurun_out/pmc_c2.json gpurun_out/pmc_c4.json gpurun_out/pmc_c3.json gpurun_out/pmc_c5.json profiles/
python3 bench.py > $O/bench_c2.json 2> $O/bench_c2.err
echo "bench c2 done"
for wl in c3 c4 c5; do python3 bench.py --workload $wl --no-aux > $O/bench_$wl.json 2> $O/bench_$wl.err; done
for wl in dt smc2 c5dt; do python3 bench.py --workload $wl --steps 5 > $O/bench_$wl.json 2> $O/bench_$wl.err; done
echo "bench others done"
python3 bench.py --resampler systematic --no-aux --no-cpu-baseline > $O/bench_c2_systematic.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c2 -o kt -- python3 bench.py --steps 3 --no-aux --no-cpu-baseline > $O/kt_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c4 -o kt -- python3 bench.py --workload c4 --steps 3 --no-aux --no-cpu-baseline > $O/kt_c4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_smc2 -o kt -- python3 bench.py --workload smc2 --steps 3 > $O/kt_smc2.log 2>&1
find $O -name "*kernel_stats.csv" | head
head -c 600 $O/bench_c2.json; echo
