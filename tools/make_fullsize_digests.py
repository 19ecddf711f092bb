#!/usr/bin/env python3
"""tools/make_fullsize_digests.py — full-size goldens for BASELINE.json configs 2, 4 and 5 (100 M synthetic 150 bp reads).

Runs the REAL reference (oracle/_ref/ref_tool fullsize: /root/reference/ntcard.cpp's ntRead / stRead / outDefault,
compiled where it lies by oracle/Makefile) over the repo's synthetic read stream (orc_gen_reads, bit-identical to the
device generator K0) in the build container and commits, per k: F1, the sha1 of the raw uint16 t_Counter planes and
the reference's own <prefix>_k<K>.hist bytes (tests/golden/fullsize/).  The -m gpu test
tests/test_fullsize_gpu.py regenerates the same reads on the device and compares digests and .hist bytes.

  python tools/make_fullsize_digests.py [n_reads] [name ...]   (default 100000000 reads, all configs; ~4 minutes on 8 cores,
                                                                ~3 GB of RAM; with names: only those, merged into digests.json)
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "oracle", "_ref", "ref_tool")
OUT = os.path.join(ROOT, "tests", "golden", "fullsize")

CONFIGS = [  # (name, dist, klist, gap, s_bits[, r_bits, n_reads])
    ("cfg2", 1, [32], 0, 7),
    ("cfg2u", 0, [32], 0, 7),
    ("cfg4", 1, [32, 64, 96, 128], 0, 7),
    ("cfg5", 1, [12], 2, 7),
    ("cfg3s", 1, [32], 0, 11),  # config 3's sampling (the >= 50 GB branch of ntcard.cpp:427-431: sBits = 11), one GPU's share of reads
    ("cfg5b", 1, [32], 8, 7),   # SURVEY 8(d)'s other spaced seed: -k 32 -g 8 (round 5)
    ("cfg2r24", 1, [32], 0, 7, 24, 20_000_000),  # a sketch of another size (-b: rBits = 24), 20 M reads (round 5)
]


def sha1_file(path):
    h = hashlib.sha1()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    only = set(sys.argv[2:])
    os.makedirs(OUT, exist_ok=True)
    threads = os.cpu_count() or 1
    meta = {"n_reads": n, "read_len": 150, "seed": 1, "r_bits": 27, "s_bits": 7, "cov_max": 1000, "generator": "orc_gen_reads == K0",
            "made_by": "oracle/_ref/ref_tool fullsize (the reference's ntRead/stRead/outDefault)", "configs": {}}
    dj = os.path.join(OUT, "digests.json")
    if only and os.path.exists(dj):
        old = json.load(open(dj))
        assert old["n_reads"] == n, "partial regeneration must keep n_reads"
        meta["configs"] = old["configs"]
    with tempfile.TemporaryDirectory() as tmp:
        for name, dist, klist, gap, s_bits, *rest in CONFIGS:
            if only and name not in only:
                continue
            r_bits, n_cfg = (rest + [27, n])[:2] if rest else (27, n)
            prefix = os.path.join(tmp, name)
            cmd = [TOOL, "fullsize", "1", str(n_cfg), "150", str(dist), ",".join(map(str, klist)), str(gap), str(r_bits), str(s_bits), str(threads), prefix]
            out = subprocess.run(cmd, stdout=subprocess.PIPE, check=True).stdout.decode()
            f1 = {int(l.split()[0][2:]): int(l.split()[1][3:]) for l in out.strip().splitlines()}
            ent = {"dist": dist, "klist": klist, "gap": gap, "s_bits": s_bits, "planes": []}
            if rest:
                ent["r_bits"], ent["n_reads"] = r_bits, n_cfg
            for k in klist:
                hist = open(f"{prefix}_k{k}.hist", "rb").read()
                gold = f"{name}_k{k}.hist"
                open(os.path.join(OUT, gold), "wb").write(hist)
                ent["planes"].append({"k": k, "f1": f1[k], "t_counter_sha1": sha1_file(f"{prefix}_k{k}.tcounter"),
                                      "hist_sha1": hashlib.sha1(hist).hexdigest(), "hist_file": gold})
                os.remove(f"{prefix}_k{k}.tcounter")
            meta["configs"][name] = ent
            print(name, json.dumps(ent["planes"]))
        # BASELINE config 5 names nthll as well: the reference's nthll ntRead (nthll.cpp:92-105) over the same stream -> its register file and the line its main prints
        if not only or "hll" in only:
            hll_tool = os.path.join(ROOT, "oracle", "_ref", "ref_hll_tool")
            regs_path = os.path.join(OUT, "hll_k32_b16.regs")
            line = subprocess.run([hll_tool, "fullsize", "1", str(n), "150", "1", "32", "16", str(threads), regs_path], stdout=subprocess.PIPE, check=True).stdout.decode()
            meta["configs"]["hll"] = {"dist": 1, "k": 32, "n_bits": 16, "regs_file": "hll_k32_b16.regs", "regs_sha1": sha1_file(regs_path), "line": line,
                                      "made_by": "oracle/_ref/ref_hll_tool fullsize (the reference's nthll ntRead + the estimate of its main)"}
            print("hll", json.dumps(meta["configs"]["hll"]))
    with open(os.path.join(OUT, "digests.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()
