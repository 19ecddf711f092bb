cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fullsize_gpu.py -x -q -k "hit_log or wraparound or sketch_device_batch or fullsize or bit_sliced" 2>&1 | tail -4
for a in "" "--dist u" "--config 4" "--config 5" "--always-log" "--direct-atomics" "--dist u --direct-atomics"; do
python bench.py --no-cpu-baseline --no-live-pmc $a | python -c "
import sys,json
j=json.loads(sys.stdin.readlines()[-1]); print('%-28s %.1f G kmers/s  step %.4f ms  hash %.4f ms  apply %.3f ms total (%d)' % ('$a',j['value']/1e9,j['ms_per_step'],j['roofline']['avg_launch_ms'],j['sketch_apply']['total_ms'],j['sketch_apply']['applies']))"
done
