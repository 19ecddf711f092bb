cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "hit_log or wraparound or sketch_device_batch" 2>&1 | tail -4
AB_DISTS="u g" bash tools/ab_run.sh cur
