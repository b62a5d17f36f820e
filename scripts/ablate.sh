#!/bin/bash
# ablation sweep of k_step (profiling-only build). bits: 0 search, 1 gather, 2 no-normal(+philox), 3 pick philox,
# 4 epilogue, 6 box-muller only (keep philox)
export SMC_LIB=$PWD/sequential_monte_carlo_amd/csrc/build_abl/libsmchip_abl.so
for abl in 0 1 2 4 8 16 64 95; do
  SMC_ABL=$abl SMC_NP=${NP:-2} python scripts/tune.py child ${SEG:-2048} | sed "s/^/abl=$abl /"
done
