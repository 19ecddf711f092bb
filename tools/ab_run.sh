#!/bin/bash
# tools/ab_run.sh <lib names...> — bench every tools/lib_<name>.so ("cur" = the in-tree library) on dist g and u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for n in "$@"; do
  for d in g u; do
    if [ "$n" = cur ]; then unset NTCARD_HIP_LIB; else export NTCARD_HIP_LIB=$ROOT/tools/lib_$n.so; fi
    python $ROOT/bench.py --no-cpu-baseline --dist $d ${AB_ARGS:-} | python -c "
import sys,json
j=json.loads(sys.stdin.readlines()[-1]); print('%-10s dist=%s  %.1f G kmers/s  launch %.4f ms' % ('$n','$d',j['value']/1e9,j['roofline']['avg_launch_ms']))"
  done
done
