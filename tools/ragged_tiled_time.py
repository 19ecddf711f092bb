#!/usr/bin/env python3
"""tools/ragged_tiled_time.py — hash / fix-up / apply ms per 10 M reads for RAGGED tiled device batches (reads of 100 .. 150 bp, uniformly distributed
lengths: bins C = 7 .. 10 of ceil(len / 16), each one ntc_submit_tiled_ragged_device batch) against one equal-length 150 bp batch, and against the same
reads in ragged row slots through K1 (the path they took before round 5)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ntcard_amd as nt

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
rng = np.random.default_rng(3)
lens = rng.integers(100, 151, size=n)
# build the bins directly as tiled arrays (numpy): reads = random ACGT
bins = []
for C in range(7, 11):
    sel = np.sort(lens[(lens > 16 * (C - 1)) & (lens <= 16 * C)])[::-1]
    m = len(sel)
    if m == 0:
        continue
    ntl = (m + 2047) // 2048
    a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(ntl * 2048, 16 * C))]
    mask = np.arange(16 * C)[None, :] >= np.concatenate([sel, np.full(ntl * 2048 - m, 16 * C)])[:, None]
    a[mask] = ord("A")
    tl = np.concatenate([sel - 16 * (C - 1), np.zeros(ntl * 2048 - m, dtype=sel.dtype)]).reshape(ntl, 2048)
    tails = np.stack([(tl > d).sum(axis=1) for d in range(16)], axis=1).astype(np.int32)
    tiles = np.ascontiguousarray(a.reshape(ntl, 2048, C, 16).transpose(0, 2, 1, 3)).reshape(-1)
    bins.append((C, m, torch.from_numpy(tiles).cuda(), torch.from_numpy(tails.reshape(-1)).cuda()))
kmers = int(sum(np.maximum(lens - 31, 0)))
for name in ("ragged tiles, one call per bin", "ragged tiles, the bins in one call"):
    with nt.Engine([32], r_bits=27, s_bits=7, flags=nt.FLAG_REQUIRE_TILED | nt.FLAG_DEFER_REDO) as e:
        for rep in range(2):
            e.reset(); e.set_profiling(True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                if "one call per bin" in name:
                    for C, m, dt, dl in bins:
                        e.submit_tiled_ragged_device(dt.data_ptr(), m, C, dl.data_ptr())
                else:
                    e.submit_tiled_bins_device([(dt.data_ptr(), m, 16 * C, dl.data_ptr()) for C, m, dt, dl in bins])
            e.flush(); e.sync(); dt_s = time.perf_counter() - t0
        ker, _ = e.kernel_time(); fix = e.fixup_time(); app, _ = e.apply_time()
        _, _, f1 = e.finish()
        assert int(f1[0]) == 5 * kmers, (int(f1[0]), 5 * kmers)
    print("%-36s %d reads x 5: %.3f ms per pass  %.3f T k-mers/s  (hash %.3f fix-up %.3f apply %.3f ms per pass)" % (name, n, dt_s / 5 * 1e3, 5 * kmers / dt_s / 1e12, ker / 5, fix / 5, app / 5), flush=True)
