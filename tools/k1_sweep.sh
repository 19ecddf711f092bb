#!/bin/bash
# tools/k1_sweep.sh — the workloads the general kernel K1 takes (A/B of K1 changes): one line each
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
run() { echo -n "$* : "; timeout 300 python $ROOT/bench.py "$@" --no-cpu-baseline --no-live-pmc 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%.3f T  %.3f ms/step  hash %.3f apply %.3f  %s' % (d['value']/1e12, d['ms_per_step'], r.get('hash_ms') or 0, r.get('apply_ms') or 0, d['sketch_apply']['mode_at_end']))"; }
run --steps 20 --warmup 5 --config 4
run --steps 20 --warmup 5 --config 5
run --steps 20 --warmup 5 --layout rows --lane-kernel
run --steps 20 --warmup 5 --layout rows --lane-kernel --dist u
run --steps 20 --warmup 5 --k 20
run --steps 20 --warmup 5 --k 64
run --steps 10 --warmup 2 --read-len 250 --layout rows
run --steps 10 --warmup 2 --read-len 100 --layout rows
