#!/bin/bash
# tools/dbg/pmc.sh <binary + args...> — rocprofv3 PMC passes of a stand-alone harness; summary to gpurun_out/pmc_dbg.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_dbg
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ifetch|icache|inst_cache|SQC" | head -40 > $OUT/avail.txt
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_IFETCH SQ_IFETCH_LEVEL" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$n -o p -- $ROOT/"$@" > $OUT/pmc_$n.log 2>&1
done
python - <<PY
import csv, glob, collections
out="$OUT"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:50]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out+"/../pmc_dbg.txt","w") as fo:
    fo.write(open(out+"/avail.txt").read())
    for k,v in agg.items():
        fo.write("== counters (mean per dispatch) %s\n"%k)
        for c,vals in sorted(v.items()): fo.write("  %-28s %16.1f  n=%d\n"%(c,sum(vals)/len(vals),len(vals)))
PY
