#!/bin/bash
# tools/k1h_ab.sh <variant names...> — average duration of the tiled kernels per variant library ("cur" = in-tree), from rocprofv3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  if [ "$n" = cur ]; then L=""; else L="--lib $ROOT/tools/lib_k1h_$n.so"; fi
  for d in ${AB_DISTS:-u}; do
    rm -rf /tmp/ab_$n
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$n -o t -- python $ROOT/bench.py --no-cpu-baseline --no-live-pmc --steps 10 --warmup 2 --dist $d $L ${AB_ARGS:-} > /tmp/ab_$n.log 2>&1
    f=$(find /tmp/ab_$n -name '*kernel_stats.csv' | head -1)
    python - "$n" "$d" "$f" <<'PY'
import csv, sys
n, d, f = sys.argv[1:4]
out = []
for row in csv.DictReader(open(f)):
    nm = row["Name"]
    for key in ("sketch_k1h", "k1h_fix", "k1h_slow", "sketch_ts", "split_kernel", "count_kernel"):
        if key in nm:
            out.append("%s %.1f us x%s" % (key, float(row["AverageNs"]) / 1000, row["Calls"]))
print("%-10s dist=%s: %s" % (n, d, ";  ".join(out)))
PY
  done
done
