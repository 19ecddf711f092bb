#!/usr/bin/env python3
"""tools/fuzz_hll.py [n_cases] [seed] — randomised nthll parity: engine registers vs oracle registers"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.cuda.init()
import orc
import ntcard_amd as nt
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
for case in range(n_cases):
    k, nb = rng.choice([rng.randint(1, 120), 21, 32, 64]), rng.choice([4, 8, 12, 16])
    L, n, pn = rng.choice([rng.randint(1, 260), 100, 150]), rng.choice([100, 3000, 40000]), rng.choice([0, 0.001, 0.05])
    mode = rng.choice(["equal", "ragged"])
    lens = [L] * n if mode == "equal" else [rng.randint(0, L) for _ in range(n)]
    reads = ["".join(rng.choice("N") if rng.random() < pn else rng.choice("ACGTacgt") for _ in range(l)).encode() for l in lens]
    with nt.HllEngine(k, nb) as e:
        h = len(reads) // 2
        e.submit_reads(reads[:h]); e.submit_reads(reads[h:])
        regs, f1 = e.finish()
    oregs, _ = orc.hll_reads(reads, k, nb)
    if not np.array_equal(regs, oregs):
        print("MISMATCH case", case, k, nb, L, n, pn, mode, int(np.count_nonzero(regs != oregs))); sys.exit(1)
print("hll fuzz OK:", n_cases, "cases")
