#!/bin/bash
# What a cheaper numerical specification would buy (timing only; `make -C sequential_monte_carlo_amd/csrc exp EXPTAG=... EXPDEF=...`):
# Philox4x32 with 7 instead of 10 rounds, pick numbers as 53-bit doubles times the mass instead of 64 x 64 -> 128-bit products.
for lib in "" _r7 _f _r7f; do
  if [ -z "$lib" ]; then unset SMC_LIB; else export SMC_LIB=$PWD/sequential_monte_carlo_amd/csrc/build_exp/libsmchip_exp$lib.so; fi
  for wl in c2 c4; do
    python bench.py --workload $wl --no-aux --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('lib=%-5s %s value %.4g ms_per_step %.5f launch_ms %s logZ %s' % ('${lib:-base}', d['config']['workload'][:30], d['value'], d['ms_per_step'], d.get('roofline', {}).get('launch_ms'), d.get('logZ', d.get('config', {}).get('logZ'))))
"
  done
done
