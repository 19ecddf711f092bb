#!/bin/bash
# tools/pmc_ab.sh <out tag> <kernel substring> <lib name | cur>... — one rocprofv3 --pmc pass (PMC_SET, default an LDS / issue set) of the driver's bench command per build,
# means per dispatch of the kernels whose name holds the substring
TAG=$1; KSUB=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SET=${PMC_SET:-SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES}
for n in "$@"; do
  if [ "$n" = cur ]; then L=""; else L="--lib $ROOT/tools/lib_$n.so"; fi
  rm -rf /tmp/pm_$n
  rocprofv3 --pmc $SET --output-format csv -d /tmp/pm_$n -o p -- python $ROOT/bench.py --no-cpu-baseline --no-live-pmc --no-nodefer --steps 20 --warmup 5 --repeats 1 $L > $OUT/pmc_$n.log 2>&1
  echo "== $n"; python - /tmp/pm_$n "$KSUB" <<'PY'
import csv,sys,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if sys.argv[2] in row["Kernel_Name"]: agg[row["Kernel_Name"][:50]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in agg.items():
    print(k)
    for c,vals in sorted(v.items()): print("   %-26s %16.0f n=%d"%(c,sum(vals)/len(vals),len(vals)))
PY
done
