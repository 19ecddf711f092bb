#!/bin/bash
# tools/k1h_ab.sh "<lib> <lib> ..." [bench args] — the in-tree library against experiment builds (tools/k1h_variant.sh -> tools/lib_k1h_<name>.so) inside
# ONE lease, interleaved, two rounds: hash / fix-up / apply ms per step of the median repeat
LIBS=$1; shift
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc --no-nodefer $*"
for i in 1 2; do
  for lib in "" $LIBS; do
    if [ -n "$lib" ]; then L="--lib $lib"; else L=""; fi
    timeout 600 python bench.py $ARGS $L 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('%-32s %.4f ms/step (min %.4f max %.4f)  %.3f T  hash %.4f  fixup %.4f  apply %.4f  frac %.3f  sclk %s' % ('$lib' or 'in-tree', j['ms_per_step'], j['ms_per_step_min'], j['ms_per_step_max'], j['value']/1e12, r['hash_ms'], r['fixup_ms'], r['apply_ms'], r['frac'], j['sclk_mhz']['after']))"
  done
done
