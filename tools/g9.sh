cd $GRAFT_REPO_ROOT
for a in "--bitslice" "--bitslice --always-log" "--bitslice --direct-atomics" "--dist u --bitslice --direct-atomics" "--always-log"; do
python bench.py --no-cpu-baseline --no-live-pmc $a | python -c "
import sys,json
j=json.loads(sys.stdin.readlines()[-1]); print('%-40s %.1f G kmers/s  step %.4f ms  hash %.4f ms  frac %.3f apply %.3f' % ('$a',j['value']/1e9,j['ms_per_step'],j['roofline']['avg_launch_ms'],j['roofline']['frac'],j['sketch_apply']['total_ms']))"
done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trg -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-live-pmc --bitslice --always-log > /dev/null 2>&1
grep -E "sketch_" $(find /tmp/trg -name "*kernel_stats.csv" | head -1) | cut -c1-140
