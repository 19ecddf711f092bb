#!/bin/bash
# tools/ab_run.sh <lib names...> — bench every tools/lib_<name>.so ("cur" = the in-tree library) on dist g and u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for n in "$@"; do
  for d in ${AB_DISTS:-g u}; do
    if [ "$n" = cur ]; then L=""; else L="--lib $ROOT/tools/lib_$n.so"; fi
    python $ROOT/bench.py --no-cpu-baseline --no-live-pmc --dist $d $L ${AB_ARGS:-} | python -c "
import sys,json
j=json.loads(sys.stdin.readlines()[-1]); print('%-10s dist=%s  %.1f G kmers/s  step %.4f ms  hash %.4f ms  apply %.3f ms total' % ('$n','$d',j['value']/1e9,j['ms_per_step'],j['roofline']['avg_launch_ms'],j['sketch_apply']['total_ms']))"
  done
done
