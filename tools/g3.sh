cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bit_sliced" 2>&1 | tail -4
AB_DISTS="u g" AB_ARGS="--bitslice --always-log" bash tools/ab_run.sh timers cur
