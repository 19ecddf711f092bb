set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "hit_log or sketch_device_batch or simple_and_tuned or wraparound" 2>&1 | tail -15
for d in g u; do
 timeout 300 python bench.py --no-cpu-baseline --no-live-pmc --dist $d 2>&1 | tail -1 > gpurun_out/bench_log_$d.json; cat gpurun_out/bench_log_$d.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value']/1e9, j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac'], j['sketch_apply'])"
 timeout 300 python bench.py --no-cpu-baseline --no-live-pmc --dist $d --direct-atomics 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('direct', j['value']/1e9, j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac'])"
done
