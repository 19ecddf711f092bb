#!/usr/bin/env python3
"""tools/collect_prof.py <tag> <round> [bench args...] — turn gpurun_out/prof_<tag>/ (written by tools/prof.sh on the
GPU box) into the tracked evidence: profiles/<round>_<tag>/{kernel_stats.csv, summary.txt, bench.json} and an entry of
profiles/traffic_pmc.json keyed by the workload (what bench.py reports as roofline.traffic)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, rnd = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles", f"{rnd}_{tag}")
    os.makedirs(dst, exist_ok=True)
    ks = glob.glob(src + "/trace/**/*kernel_stats.csv", recursive=True)
    lines = []
    if ks:
        shutil.copy(ks[0], os.path.join(dst, "kernel_stats.csv"))
        lines.append("== kernel stats (rocprofv3 --kernel-trace --stats)")
        lines += [",".join(r)[:220] for i, r in enumerate(csv.reader(open(ks[0]))) if i < 10]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(src + "/pmc_*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    bj = os.path.join(src, "bench.json")
    bench = None
    if os.path.exists(bj):
        shutil.copy(bj, os.path.join(dst, "bench.json"))
        try:
            bench = json.loads(open(bj).read().strip().splitlines()[-1])
        except ValueError:
            bench = None
    # A bench step is one batch; the engine may cut a batch into several dispatches of the hash kernel (the first batch
    # of an engine is split into a small head, whose hit log is sampled to choose the update mode, and the rest), so the
    # per-step figure is the SUM over all dispatches divided by the number of steps the command ran (warm-up + timed).
    n_steps = (bench["steps"] + bench["warmup"]) if bench else None
    # (the kernel trace runs every repeat of the timed region: warmup + repeats x steps; the PMC passes run one)
    n_trace = (bench["warmup"] + bench.get("repeats", 1) * bench["steps"]) if bench else None
    if ks and n_trace:
        for r in csv.DictReader(open(ks[0])):
            if "sketch_" in r["Name"] or "k1h_" in r["Name"]:
                lines.append("   -> %s: %d dispatches over %d bench steps (%d warm-up + %d x %d timed) = %.4f ms per step" %
                             (r["Name"][:48], int(r["Calls"]), n_trace, bench["warmup"], bench.get("repeats", 1), bench["steps"], float(r["TotalDurationNs"]) / n_trace / 1e6))
    hash_kernels, apply_kernels = {}, {}
    for k, v in agg.items():
        if any(t in k for t in ("sketch_", "k1h_", "split_kernel", "split_packed_kernel", "count_kernel", "finalize")):
            lines.append(f"== counters (separate --pmc passes), mean per dispatch | per bench step: {k}")
            for c, vals in sorted(v.items()):
                per_step = ("%18.1f" % (sum(vals) / n_steps)) if n_steps else "-"
                lines.append("  %-28s %18.1f  n=%d | %s" % (c, sum(vals) / len(vals), len(vals), per_step))
            if "sketch_" in k or "k1h_" in k:
                hash_kernels[k] = {c: sum(vals) / (n_steps or len(vals)) for c, vals in v.items()}
            elif any(t in k for t in ("split_kernel", "split_packed_kernel", "count_kernel", "log_")):
                apply_kernels[k] = {c: sum(vals) / (n_steps or len(vals)) for c, vals in v.items()}
    open(os.path.join(dst, "summary.txt"), "w").write("\n".join(lines) + "\n")
    # traffic entry: 2*FETCH_SIZE + WRITE_SIZE (KB) per bench step of the hash kernels (K1h + K1f, or K1) AND of the sketch-update kernels (split / count /
    # log_*), which is what bench.py's live passes report as roofline.traffic (round 6: the table used to hold the hash kernels' share only)
    if bench and hash_kernels:
        allk = list(hash_kernels.values()) + list(apply_kernels.values())
        fetch = sum(v.get("FETCH_SIZE", 0.0) for v in allk)
        write = sum(v.get("WRITE_SIZE", 0.0) for v in allk)
        fetch_h = sum(v.get("FETCH_SIZE", 0.0) for v in hash_kernels.values())
        write_h = sum(v.get("WRITE_SIZE", 0.0) for v in hash_kernels.values())
        key = bench["config"]["traffic_key"]
        tj = os.path.join(ROOT, "profiles", "traffic_pmc.json")
        table = json.load(open(tj))
        table[key] = {"fetch_kb": fetch, "write_kb": write, "traffic_bytes": int((2 * fetch + write) * 1024),
                      "traffic_bytes_hash": int((2 * fetch_h + write_h) * 1024),
                      "source": f"profiles/{rnd}_{tag}/summary.txt",
                      "valu_insts_per_launch": sum(v.get("SQ_INSTS_VALU", 0.0) for v in allk),
                      "valu_insts_per_launch_hash": sum(v.get("SQ_INSTS_VALU", 0.0) for v in hash_kernels.values())}
        json.dump(table, open(tj, "w"), indent=1)
        print("traffic entry", key, table[key])
    print("wrote", dst)


if __name__ == "__main__":
    main()
