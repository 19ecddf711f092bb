#!/bin/bash
# tools/k1h_counters.sh — stall / instruction-cache counters of the K1h launch (one --pmc pass per group); the list of what this box offers goes to
# gpurun_out/pmc_avail.txt.  Usage (GPU box): bash tools/k1h_counters.sh "GROUP1 COUNTERS" "GROUP2 COUNTERS" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/k1h_counters
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $ROOT/gpurun_out/pmc_avail.txt 2>&1
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-live-pmc --repeats 1 --no-nodefer"
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -o p -- python $ROOT/bench.py $ARGS > $OUT/g$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:48]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in agg.items():
    if "k1h" in k or "split" in k or "count_kernel" in k:
        print("==", k)
        for c,vals in sorted(v.items()): print("  %-32s %16.1f  n=%d"%(c,sum(vals)/len(vals),len(vals)))
PY
