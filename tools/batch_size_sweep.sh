#!/bin/bash
# tools/batch_size_sweep.sh — hash / fix-up / apply ms per step against the reads per step (tiled device batches, config 2), one lease
for R in 500000 1000000 2000000 4000000 6000000 10000000; do
  python bench.py --steps 20 --warmup 5 --reads-per-step $R --no-cpu-baseline --no-live-pmc --no-nodefer --repeats 3 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('%9d reads  %.4f ms/step  %.3f T  hash %.4f  fixup %.4f  apply %.4f' % ($R, j['ms_per_step'], j['value']/1e12, r['hash_ms'], r['fixup_ms'], r['apply_ms']))"
done
