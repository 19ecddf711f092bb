cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bit_sliced or sketch_device_batch" 2>&1 | tail -8
AB_DISTS="u g" bash tools/ab_run.sh timers cur
