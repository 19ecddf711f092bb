cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -i -E "icache|ifetch|inst_cache|SQC_" | head -40 > $R/gpurun_out/counters_icache.txt
cat $R/gpurun_out/counters_icache.txt | cut -c1-160
