#!/bin/bash
# tools/perf_sweep.sh — bench.py over the shapes DESIGN.md §5 quotes (one line each); run on the GPU box
cd ${GRAFT_REPO_ROOT:-.}
for a in "" "--dist u" "--s-bits 11" "--k 64" "--k 16" "--k 128" "--k 96 --read-len 250" "--k 12 --gap 2" "--k 32 --gap 8" "--klist 32,64,96,128" "--klist 16,24,32,48" "--klist 21,31,41,51,61,71" \
         "--read-len 100" "--read-len 200" "--read-len 250" "--read-len 300 --reads-per-step 5000000" "--read-len 76" "--read-len 50" "--dist u --read-len 100" "--dist u --read-len 250" "--r-bits 24" "--dist u --r-bits 24"; do
python bench.py --no-cpu-baseline --no-live-pmc $a | python -c "
import sys,json
j=json.loads(sys.stdin.readlines()[-1]); print('%-44s %7.1f G kmers/s  step %.4f ms  hash %.4f ms launches %d apply %.3f ms (%d) %s  frac %.3f' % ('$a',j['value']/1e9,j['ms_per_step'],j['roofline']['avg_launch_ms'],j['roofline']['launches'],j['sketch_apply']['total_ms'],j['sketch_apply']['applies'],j['sketch_apply']['mode_at_end'],j['roofline']['frac']))"
done
