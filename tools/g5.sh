cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_ic -o p -- python $R/bench.py --steps 4 --warmup 1 --reads-per-step 4000000 --no-cpu-baseline --no-live-pmc --dist u > $R/gpurun_out/pmc_ic.log 2>&1
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_ic/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:50]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in agg.items():
    if "sketch" in k:
        print(k)
        for c,vals in sorted(v.items()): print("  %-28s %16.1f n=%d"%(c,sum(vals)/len(vals),len(vals)))
PY
