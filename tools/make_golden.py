#!/usr/bin/env python3
"""Generate tests/golden/* from the REAL reference (oracle/_ref, built from /root/reference).

Runs only in the build container.  Fixtures are data: inputs (our own synthetic reads) and the
reference's outputs on them.  No reference source text is stored.

  python tools/make_golden.py
"""
import gzip
import hashlib
import json
import os
import random
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def rseq(rng, n, pn=0.02, plow=0.1, alphabet_bad="NnRYKMSWBDHV-.*Xx"):
    s = []
    for _ in range(n):
        r = rng.random()
        if r < pn:
            s.append(rng.choice(alphabet_bad))
        elif r < pn + plow:
            s.append(rng.choice("acgtu"))
        else:
            s.append(rng.choice("ACGTU"))
    return "".join(s)


def hash_vectors():
    rng = random.Random(20240928)
    seqs = [
        "ACGTACACTGGACTGAGTCT",  # the reference's own known-answer input (UnitTests.cpp:39)
        "GAGTGTCAAACATTCAGACAACAGCAGGGGTGCTCTGGAATCCTATGTGAGGAACAAACATTCAGGCCACAGTAG",
        "ACGTNACGTACGTACGTacgtacguACGTNNACGTACGTACGT",
        "",
        "A",
        "N" * 40,
        "ACGT" * 50,
        "T" * 140,
    ]
    for n, pn in ((150, 0.0), (150, 0.01), (150, 0.05), (151, 0.3), (33, 0.0), (64, 0.02), (300, 0.005), (1000, 0.002)):
        seqs.append(rseq(rng, n, pn))
    out = {"seqs": seqs, "nthash": [], "sthash": []}
    bseqs = [s.encode() for s in seqs]
    for k in (1, 4, 12, 20, 31, 32, 33, 34, 35, 64, 70, 96, 128, 200):
        res = orc.ref_hash(bseqs, k, 1)
        out["nthash"].append({"k": k, "pos": [p.tolist() for p, _ in res],
                              "hash": [["%016x" % int(x) for x in h[:, 0]] for _, h in res]})
    res3 = orc.ref_hash([bseqs[0]], 20, 3)
    out["kat_h3"] = ["%d" % int(x) for x in res3[0][1][0]]
    for k, g in ((12, 2), (12, 4), (13, 3), (20, 8), (32, 8), (64, 10), (33, 1)):
        res = orc.ref_sthash(bseqs, k, g)
        out["sthash"].append({"k": k, "gap": g, "pos": [p.tolist() for p, _ in res],
                              "hash": [["%016x" % int(x) for x in h[:, 0]] for _, h in res]})
    with open(os.path.join(GOLD, "hash_vectors.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))


def small_reads():
    """5 000 x 70 bp reads from a 40 kbp genome (our generator, dist=1) -> FASTQ fixture."""
    n, L = 5000, 70
    slots = orc.gen_reads(7, 0, n, L, 72, 1, genome_len=40_000)
    reads = [slots[i * 72: i * 72 + L].tobytes() for i in range(n)]
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * L) for i, r in enumerate(reads))
    path = os.path.join(GOLD, "reads_small.fq.gz")
    with open(path, "wb") as raw:
        with gzip.GzipFile(fileobj=raw, mode="wb", mtime=0, compresslevel=9) as gz:
            gz.write(fq)
    return reads, fq


def run_ref_cli(args, inputs, workdir):
    cmd = [orc.REF_NTCARD] + args + inputs
    subprocess.check_call(cmd, cwd=workdir, stderr=subprocess.DEVNULL)


def cli_goldens(reads, fq):
    manifest = {}
    with tempfile.TemporaryDirectory() as d:
        fqp = os.path.join(d, "reads.fq")
        with open(fqp, "wb") as f:
            f.write(fq)
        cases = [
            ("k12", ["-k", "12", "-p", "out"], ["out_k12.hist"]),
            ("k12_g2", ["-k", "12", "-g", "2", "-p", "out"], ["out_k12.hist"]),
            ("k32", ["-k", "32", "-p", "out"], ["out_k32.hist"]),
            ("k20_c50", ["-k", "20", "-c", "50", "-p", "out"], ["out_k20.hist"]),
            ("multi", ["-k", "16,24,32,48", "-p", "out"], ["out_k16.hist", "out_k24.hist", "out_k32.hist", "out_k48.hist"]),
            ("k24_s11_r22", ["-k", "24", "-r", "22", "-p", "out"], ["out_k24.hist"]),
        ]
        for name, args, outs in cases:
            for o in outs:
                if os.path.exists(os.path.join(d, o)):
                    os.remove(os.path.join(d, o))
            run_ref_cli(args, [fqp], d)
            for o in outs:
                dst = f"ref_{name}__{o}"
                data = open(os.path.join(d, o), "rb").read()
                with open(os.path.join(GOLD, dst), "wb") as f:
                    f.write(data)
                manifest[dst] = {"args": args, "md5": hashlib.md5(data).hexdigest()}
        # compact (-o) form
        run_ref_cli(["-k", "12,20", "-c", "20", "-o", "compact.tsv"], [fqp], d)
        data = open(os.path.join(d, "compact.tsv"), "rb").read()
        with open(os.path.join(GOLD, "ref_compact__k12_20_c20.tsv"), "wb") as f:
            f.write(data)
        manifest["ref_compact__k12_20_c20.tsv"] = {"args": ["-k", "12,20", "-c", "20", "-o", "compact.tsv"],
                                                  "md5": hashlib.md5(data).hexdigest()}
    with open(os.path.join(GOLD, "cli_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


def sketch_goldens(reads):
    """t_Counter digests + estimator in/out of the reference's ntRead/stRead/compEst (whitebox)."""
    out = []
    for klist, gap, rb, sb in (([32], 0, 18, 7), ([12], 2, 16, 7), ([16, 24, 32, 48], 0, 17, 7), ([20], 0, 18, 11),
                               ([33], 0, 16, 5), ([12], 0, 27, 7)):
        rc, rf = orc.ref_sketch(reads, klist, gap, rb, sb)
        ent = {"klist": klist, "gap": gap, "r_bits": rb, "s_bits": sb, "f1": [int(x) for x in rf], "planes": []}
        for ki in range(len(klist)):
            F0, fm = orc.ref_est(rc[ki], rb, sb)
            p = orc.value_hist(rc[ki], rb)
            nzv = [[int(s), int(v), int(p[s, v])] for s in range(2) for v in np.nonzero(p[s])[0]]
            ent["planes"].append({
                "fnv1a64": "%016x" % orc.fnv1a64(rc[ki]),
                "nonzero": [int(np.count_nonzero(rc[ki][0])), int(np.count_nonzero(rc[ki][1]))],
                "sum": [int(rc[ki][0].astype(np.uint64).sum()), int(rc[ki][1].astype(np.uint64).sum())],
                "p_nonzero": nzv,
                "F0": F0,
                "f_1_1000": [float(x) for x in fm[1:1001]],
                "f_tail_max": float(np.max(fm[1001:])),
            })
        out.append(ent)
    with open(os.path.join(GOLD, "sketch_goldens.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))


def hll_goldens(reads, fq):
    """nthll: registers (digest) + the printed estimate of the real nthll code, and the CLI's stdout line"""
    out = {"cases": []}
    for k, nb in ((32, 16), (12, 10), (64, 16), (20, 12), (70, 8)):
        regs, line = orc.ref_hll(reads, k, nb)
        out["cases"].append({"k": k, "n_bits": nb, "fnv1a64": "%016x" % orc.fnv1a64(regs), "max": int(regs.max()),
                             "sum": int(regs.astype(np.uint64).sum()), "line": line.decode()})
    with tempfile.TemporaryDirectory() as d:
        fqp = os.path.join(d, "reads.fq")
        with open(fqp, "wb") as f:
            f.write(fq)
        out["cli_k32"] = subprocess.check_output([orc.REF_NTHLL, "-k", "32", fqp], stderr=subprocess.DEVNULL).decode()
        out["cli_k20_b12"] = subprocess.check_output([orc.REF_NTHLL, "-k", "20", "-b", "12", fqp], stderr=subprocess.DEVNULL).decode()
    with open(os.path.join(GOLD, "nthll_goldens.json"), "w") as f:
        json.dump(out, f, indent=1)


def main():
    if not orc.have_ref():
        sys.exit("oracle/_ref is missing: run `make -C oracle ref` in the build container first")
    os.makedirs(GOLD, exist_ok=True)
    hash_vectors()
    reads, fq = small_reads()
    cli_goldens(reads, fq)
    sketch_goldens(reads)
    hll_goldens(reads, fq)
    print("golden fixtures written to", GOLD)
    for fn in sorted(os.listdir(GOLD)):
        print("  %8d  %s" % (os.path.getsize(os.path.join(GOLD, fn)), fn))


if __name__ == "__main__":
    main()
