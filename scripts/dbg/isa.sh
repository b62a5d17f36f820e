#!/bin/bash
# Diagnostic: dump the gfx950 ISA of one model TU and print instruction statistics of one kernel.
# usage: scripts/dbg/isa.sh <model 1|2|3> <mangled-name-prefix>
set -e
M=${1:-1}; K=${2:-_ZN3smc6k_stepILi1ELi512ELi2ELb1ELb0E}
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fPIC -DSMC_MODEL=$M \
  --cuda-device-only -S -o /tmp/isa/model$M.s /root/repo/sequential_monte_carlo_amd/csrc/smc_model.hip 2>/dev/null
awk -v k="^$K.*:" '$0 ~ k {f=1} f{print} /^\.Lfunc_end/{if(f)exit}' /tmp/isa/model$M.s > /tmp/isa/k.s
echo "lines $(wc -l < /tmp/isa/k.s)  VALU $(grep -cE '^\s+v_' /tmp/isa/k.s)  v_mov_b32 $(grep -cE '^\s+v_mov_b32_e32' /tmp/isa/k.s)  branches $(grep -cE 's_cbranch' /tmp/isa/k.s)"
awk -v k="^$K.*:" '$0 ~ k {f=1} f&&/; (NumVgprs|ScratchSize|Occupancy|NumSgprs|LDSByteSize):/{print; n++} n>=5{exit}' /tmp/isa/model$M.s
