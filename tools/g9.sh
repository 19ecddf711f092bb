cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "bit_sliced or (fullsize and 4)" 2>&1 | tail -2
for a in "--dist u --bitslice" "--bitslice" "--dist u" ""; do
python bench.py --no-cpu-baseline --no-live-pmc $a | python -c "
import sys,json
j=json.loads(sys.stdin.readlines()[-1]); print('%-24s %.1f G kmers/s  step %.4f ms  hash %.4f ms  frac %.3f apply %.3f' % ('$a',j['value']/1e9,j['ms_per_step'],j['roofline']['avg_launch_ms'],j['roofline']['frac'],j['sketch_apply']['total_ms']))"
done
