#!/bin/bash
# tools/r03_sweep.sh — the round-3 table of DESIGN.md §5: one bench line per workload (no CPU baseline, no live PMC)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_sweep.txt
: > $OUT
run() { echo "== $*" >> $OUT; timeout 300 python $ROOT/bench.py "$@" --no-cpu-baseline --no-live-pmc 2>/dev/null | tail -1 >> $OUT; }
run --steps 20 --warmup 5
run --steps 20 --warmup 5 --dist u
run --steps 20 --warmup 5 --layout rows
run --steps 20 --warmup 5 --layout rows --dist u
run --steps 20 --warmup 5 --layout rows --lane-kernel
run --steps 20 --warmup 5 --s-bits 11
run --steps 20 --warmup 5 --config 4
run --steps 20 --warmup 5 --config 5
for L in 50 76 100 125 128 143 159 200 250 300; do run --steps 10 --warmup 2 --read-len $L; done
for L in 100 143 250; do run --steps 10 --warmup 2 --read-len $L --layout rows; done
run --steps 20 --warmup 5 --reads-per-step 1000000
run --steps 20 --warmup 5 --reads-per-step 1000000 --layout rows
run --steps 10 --warmup 2 --reads-per-step 40000000
run --steps 20 --warmup 5 --k 20
run --steps 20 --warmup 5 --k 64
run --steps 20 --warmup 5 --r-bits 24
python - <<PY
import json
for line in open("$OUT"):
    line=line.strip()
    if line.startswith("=="): name=line; continue
    try: d=json.loads(line)
    except ValueError: print(name, "FAILED"); continue
    r=d["roofline"]
    print("%-60s %.3f T  %.3f ms/step  hash %.3f apply %.3f  frac %.3f hashfrac %.3f  %s / %s" % (name[3:], d["value"]/1e12, d["ms_per_step"], r.get("hash_ms") or 0, r.get("apply_ms") or 0, r["frac"], d.get("roofline_hash",{}).get("frac",0), d["config"].get("layout"), d["sketch_apply"]["mode_at_end"]))
PY
