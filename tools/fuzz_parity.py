#!/usr/bin/env python3
"""tools/fuzz_parity.py [n_cases] [seed] — randomised parity sweep: engine vs oracle on random shapes
(read length / raggedness / dirt / k list / gap / sBits / rBits / submit pattern / sketch-update mode and log size).  Prints the first mismatch and exits 1.
NTC_BIN_MIN=1024 in the environment sends the length bins of the larger ragged cases through the tiled kernels (default: bins from 32 Ki reads)."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.cuda.init()
import orc
import ntcard_amd as nt

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def rseq(n, pn, plow):
    out = []
    for _ in range(n):
        r = rng.random()
        out.append(rng.choice("NnRYKM-*") if r < pn else (rng.choice("acgtu") if r < pn + plow else rng.choice("ACGTU")))
    return "".join(out).encode()


for case in range(n_cases):
    gap = 0
    if rng.random() < 0.3:
        k = rng.randint(4, 90)
        gap = rng.randrange(k % 2 or 2, max(k - 1, 3), 2) if k > 3 else 0
        if gap >= k or gap % 2 != k % 2: gap = 0
        klist = [k]
    else:
        klist = sorted(set(rng.choice([rng.randint(1, 40), rng.randint(12, 32), rng.randint(1, 200), rng.choice([12, 16, 31, 32, 33, 64, 96, 128])]) for _ in range(rng.choice([1, 1, 2, 4, 6]))))
    s_bits, r_bits = rng.choice([2, 3, 5, 7, 7, 8, 11]), rng.choice([12, 16, 18, 23])
    mode = rng.choice(["equal", "equal", "two", "ragged", "long"])
    L = rng.choice([rng.randint(1, 300), 100, 150, 151, 250])
    n = rng.choice([70, 500, 3000, 3000, 40000])  # (40000: with NTC_BIN_MIN=1024 in the environment the length bins of a ragged batch take the tiled kernels, several per launch)
    pn = rng.choice([0, 0, 0.001, 0.02, 0.2])
    if mode == "equal": lens = [L] * n
    elif mode == "two": lens = [L if rng.random() < 0.9 else max(1, L - rng.randint(1, 30)) for _ in range(n)]
    elif mode == "ragged": lens = [rng.randint(0, L) for _ in range(n)]
    else: lens = [rng.choice([L, 1000, 5000, 70000]) for _ in range(max(3, n // 50))]
    reads = [rseq(l, pn, 0.1) for l in lens]
    cuts = sorted(rng.sample(range(len(reads) + 1), min(len(reads) + 1, rng.choice([0, 1, 3, 6]))))
    flags, log_entries = rng.choice([(0, 0), (nt.FLAG_ALWAYS_LOG, 0), (nt.FLAG_DIRECT_ATOMICS, 0),
                                     (nt.FLAG_ALWAYS_LOG | nt.FLAG_PARTITION_ALWAYS, 0),
                                     (nt.FLAG_ALWAYS_LOG | nt.FLAG_PARTITION_ALWAYS, 1 << rng.choice([14, 16, 18]))])
    with nt.Engine(klist, gap=gap, r_bits=r_bits, s_bits=s_bits, flags=flags, log_entries=log_entries) as e:
        prev = 0
        for c in cuts + [len(reads)]:
            e.submit_reads(reads[prev:c]); prev = c
        tc, ph, f1 = e.finish(counters=True)
    oc, of1 = orc.sketch_reads(reads, klist, gap, r_bits, s_bits)
    ok = np.array_equal(f1, of1) and np.array_equal(tc, oc)
    if ok and mode == "equal" and L >= 1 and gap == 0:
        # the same reads in the tiled layout: the tiled kernel pair K1h + K1f where it is built (every k of the list in 12 .. 32, sBits >= 7:
        # then a fallback would be an error), the device-side re-layout + K1 otherwise; batches cut at the same places
        k1c = all(12 <= k <= 32 for k in klist) and s_bits >= 7 and (len(klist) << (r_bits + 1)) <= (1 << 32) and r_bits + 1 + s_bits - 7 <= 32 and not (flags & nt.FLAG_LANE_KERNEL)
        with nt.Engine(klist, gap=gap, r_bits=r_bits, s_bits=s_bits, flags=flags | (nt.FLAG_REQUIRE_TILED if k1c else 0), log_entries=log_entries) as e:
            prev, keep = 0, []
            for c in cuts + [len(reads)]:
                if c > prev:
                    t = torch.from_numpy(nt.tile_reads(reads[prev:c], L)).cuda()
                    keep.append(t)
                    e.submit_tiled_device(t.data_ptr(), c - prev, L)
                prev = c
            tc2, ph2, f12 = e.finish(counters=True)
        ok = np.array_equal(f12, of1) and np.array_equal(tc2, oc)
        if not ok: f1, tc = f12, tc2
        mode = "equal+tiled(K1h)" if k1c else "equal+tiled(K1)"
    desc = "case %d: klist=%s gap=%d s=%d r=%d mode=%s L=%d n=%d pn=%g cuts=%s flags=%d log=%d" % (case, klist, gap, s_bits, r_bits, mode, L, len(reads), pn, cuts, flags, log_entries)
    if not ok:
        print("MISMATCH", desc, "f1", list(f1), list(of1), "diff counters", int(np.count_nonzero(tc != oc)))
        sys.exit(1)
    if case % 20 == 0: print("ok", desc, flush=True)
print("fuzz parity OK:", n_cases, "cases")
