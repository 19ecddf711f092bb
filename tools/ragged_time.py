#!/usr/bin/env python3
"""tools/ragged_time.py — kernel time on host-submitted batches: equal-length vs variable-length reads, through the tiled kernels (K1h + K1f:
equal lengths as one tiled batch, mixed lengths as ragged tiled batches per bin of ceil(len / 16) — round 5) and through K1 alone (NTC_FLAG_LANE_KERNEL: row
slots, the path mixed lengths took before)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc
import ntcard_amd as nt
n, L = 8_000_000, 150
slots = orc.gen_reads(3, 0, n, L, 152, 1, genome_len=100_000_000).reshape(n, 152)
rng = np.random.default_rng(1)
for name, lens in (("equal 150", np.full(n, 150)), ("95% 150, 5% shorter", np.where(rng.random(n) < 0.95, 150, rng.integers(50, 150, n))),
                   ("uniform 100..150", rng.integers(100, 151, n))):
    offs = np.zeros(n + 1, dtype=np.uint64); offs[1:] = np.cumsum(lens)
    bases = np.empty(int(offs[-1]), dtype=np.uint8)
    idx = np.arange(152)[None, :] < lens[:, None]
    bases[:] = slots[idx]
    for kern, flags in (("K1h + K1f (tiles)", 0), ("K1 (row slots)", nt.FLAG_LANE_KERNEL)):
        with nt.Engine([32], r_bits=27, s_bits=7, flags=flags) as e:
            e.submit(bases, offs)  # warm-up with the whole batch: staging buffers, K1h's hand-over arrays and the clocks are as in a long run
            e.sync()
            e.reset()
            e.set_profiling(True)
            e.submit(bases, offs)
            e.sync()
            ms, launches = e.kernel_time()
            ms += e.fixup_time()
            _, _, f1 = e.finish(counters=False, p_hist=True)
        print("%-22s %-18s hash kernels %.3f ms (%d launches) for %d k-mers -> %.1f G k-mers/s" % (name, kern, ms, launches, int(f1[0]), f1[0] / ms / 1e6), flush=True)
