#!/bin/bash
# tools/k1h_weight_sweep.sh — K1h's share of a workgroup's blocks for a wave alone on its SIMD (NTC_K1H_LONE_WEIGHT, sixteenths of a paired
# wave's share): the default bench command per weight, hash / fix-up / apply milliseconds per step
cd ${GRAFT_REPO_ROOT:-.}
for w in ${WEIGHTS:-16 18 20 22 24 28}; do
  NTC_K1H_LONE_WEIGHT=$w python bench.py --no-cpu-baseline --no-live-pmc "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('weight $w: %.4f ms/step  %.3f T  hash %.4f  fixup %.4f  apply %.4f  frac %.3f' % (d['ms_per_step'], d['value']/1e12, r['hash_ms'], r['fixup_ms'], r['apply_ms'], r['frac']))"
done
