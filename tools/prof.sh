#!/bin/bash
# tools/prof.sh <tag> [bench args...] — rocprofv3 kernel trace + PMC passes of bench.py on the GPU box.
# Summaries land in gpurun_out/prof_<tag>/ ; copy what should be judged into profiles/.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# PROF_DEFAULT=1: profile exactly the default bench command (what BENCH_rNN.json is measured with)
if [ "${PROF_DEFAULT:-0}" = 1 ]; then ARGS="--no-cpu-baseline --no-live-pmc $*"; else ARGS="--steps 4 --warmup 1 --reads-per-step 4000000 --no-cpu-baseline --no-live-pmc $*"; fi
# PROF_PHASE=bench: only the unprofiled bench line; PROF_PHASE=counters: only the rocprofv3 passes (a bench run right behind PMC passes of
# another command has measured up to 30 % slow — the clocks stay in the profiling state for a while — so the lines of all tags are taken first)
if [ "${PROF_PHASE:-all}" != counters ]; then
  python $ROOT/bench.py $ARGS 2> $OUT/bench.err | tail -1 > $OUT/bench.json
fi
if [ "${PROF_PHASE:-all}" = bench ]; then exit 0; fi
# the kernel trace runs the SAME command as the bench line (all repeats of the timed region: warmup + repeats x steps launches, so that the
# average duration of the hash kernel is the one the bench line's HIP events see), without the second engine of roofline_nodefer
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/bench.py $ARGS --no-nodefer > $OUT/trace.log 2>&1
# the PMC passes: ONE timed region, so that dispatches / (steps + warmup) is the per-step figure
ARGS="$ARGS --repeats 1 --no-nodefer"
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$n -o p -- python $ROOT/bench.py $ARGS > $OUT/pmc_$n.log 2>&1
done
# compact summary: per-kernel stats + per-counter mean over the hash kernel's dispatches
python - <<PY
import csv, glob, os, collections
out="$OUT"
for f in glob.glob(out+"/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats", f)
    for i,row in enumerate(csv.reader(open(f))):
        if i<8: print(",".join(row))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in agg.items():
    if "sketch" in k or "nthash" in k or "split" in k or "count" in k or "k1h" in k:
        print("== counters (mean per dispatch)", k)
        for c,vals in sorted(v.items()): print("  %-28s %16.1f  n=%d"%(c,sum(vals)/len(vals),len(vals)))
PY
