#!/bin/bash
# tools/kstats.sh <out tag> <lib name | cur>... — rocprofv3 kernel-trace stats (average duration per kernel) of the driver's bench command for several builds in one lease
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  if [ "$n" = cur ]; then L=""; else L="--lib $ROOT/tools/lib_$n.so"; fi
  rm -rf /tmp/ks_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$n -o t -- python $ROOT/bench.py --no-cpu-baseline --no-live-pmc ${KS_NODEFER:---no-nodefer} --steps 20 --warmup 5 --repeats ${KS_REPEATS:-3} $L ${KS_ARGS:-} > $OUT/$n.log 2>&1
  f=$(find /tmp/ks_$n -name '*kernel_stats.csv' | head -1)
  cp "$f" $OUT/${n}_kernel_stats.csv
  echo "== $n"; python - "$f" <<'PY'
import csv,sys
for i,r in enumerate(csv.reader(open(sys.argv[1]))):
    if i==0: continue
    name=r[0][:60]; calls=int(r[1]); tot=float(r[2])/1e6; avg=float(r[3])/1e3
    if tot>0.5: print("%-62s calls %5d  total %9.3f ms  avg %9.1f us"%(name,calls,tot,avg))
PY
done
