#!/bin/bash
# tools/k1h_ab.sh [bench args] — the in-tree library against tools/lib_k1h_r04.so (the round-4 build: six waves per CU) inside ONE lease,
# interleaved (new, old, new, old): hash / fix-up / apply ms per step of the median repeat
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc --no-nodefer $*"
for i in 1 2; do
  for lib in "" "--lib tools/lib_k1h_r04.so"; do
    timeout 600 python bench.py $ARGS $lib 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('%-28s %.4f ms/step (min %.4f max %.4f)  %.3f T  hash %.4f  fixup %.4f  apply %.4f  frac %.3f  sclk %s' % ('$lib' or 'in-tree', j['ms_per_step'], j['ms_per_step_min'], j['ms_per_step_max'], j['value']/1e12, r['hash_ms'], r['fixup_ms'], r['apply_ms'], r['frac'], j['sclk_mhz']['after']))"
  done
done
