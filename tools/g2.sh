cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_apply -o t -- python $R/bench.py --no-cpu-baseline --no-live-pmc $BARGS > $R/gpurun_out/prof_apply.log 2>&1
find $R/gpurun_out/prof_apply -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-160
