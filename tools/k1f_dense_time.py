#!/usr/bin/env python3
"""tools/k1f_dense_time.py — what the fix-up kernels cost when non-base bytes are dense (the suspect lists overflow and K1f takes its slow path):
hash / fix-up milliseconds per 10 M reads of 150 bp at several rates of 'N', tiled device batches of 2 M reads."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ntcard_amd as nt

n, L = 2_000_000, 150
rng = np.random.default_rng(3)
for p_bad in (0.0, 0.0005, 0.002, 0.005, 0.02, 0.1):
    arr = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, L))]
    if p_bad:
        arr = np.where(rng.random((n, L)) < p_bad, np.uint8(ord("N")), arr).astype(np.uint8)
    t = torch.from_numpy(np.ascontiguousarray(np.pad(arr, ((0, (-n) % 2048), (0, (-L) % 16)), constant_values=ord("A"))
                                              .reshape(-1, 2048, (L + 15) // 16, 16).transpose(0, 2, 1, 3)).reshape(-1)).cuda()
    for teams in (0,):
        with nt.Engine([32], r_bits=27, s_bits=7, flags=nt.FLAG_REQUIRE_TILED | nt.FLAG_DEFER_REDO) as e:
            for _ in range(3):
                e.submit_tiled_device(t.data_ptr(), n, L)
            e.flush(); e.sync(); e.reset(); e.set_profiling(True)
            for _ in range(10):
                e.submit_tiled_device(t.data_ptr(), n, L)
            e.sync()
            ker, launches = e.kernel_time()
            fix = e.fixup_time()
        print("N rate %.4f  %s: hash %.3f ms  fix-up %.3f ms per 10 M reads" % (p_bad, "K1h + K1f", ker / 10 * 5, fix / 10 * 5), flush=True)
