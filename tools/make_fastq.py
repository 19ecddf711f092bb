#!/usr/bin/env python3
"""tools/make_fastq.py <out_prefix> <n_files> <reads_per_file> [read_len] — synthetic FASTQ files from the
oracle's read generator (dist g), for end-to-end CLI timing on the GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402

prefix, n_files, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
L = int(sys.argv[4]) if len(sys.argv) > 4 else 150
stride = (L + 3) & ~3
if stride == L:
    stride += 4
for fi in range(n_files):
    slots = orc.gen_reads(11, fi * n, n, L, stride, 1, genome_len=100_000_000).reshape(n, stride)[:, :L]
    # record = "@r\n" + seq + "\n+\n" + qual + "\n"
    rec = np.empty((n, 3 + L + 3 + L + 1), dtype=np.uint8)
    rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
    rec[:, 3:3 + L] = slots
    rec[:, 3 + L:6 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 6 + L:6 + 2 * L] = ord("I")
    rec[:, 6 + 2 * L] = ord("\n")
    rec.tofile("%s_%d.fq" % (prefix, fi))
print("wrote", n_files, "files x", n, "reads")
