#!/usr/bin/env python3
"""tools/hll_time.py — nthll mode throughput on device-resident reads (10 M x 150 bp per submit)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ntcard_amd as nt
n, L, stride = 10_000_000, 150, 152
bs = []
for i in range(4):
    d = torch.empty(n * stride + 16, dtype=torch.uint8, device="cuda")
    nt.gen_reads_device(d.data_ptr(), 5, i * n, n, L, stride, 1, genome_len=100_000_000)
    bs.append(d)
for k in (32, 64):
    with nt.HllEngine(k, 16) as e:
        e.submit_device(bs[0].data_ptr(), n, L, stride)  # warm-up: registers + thresholds
        e.sync()
        t0 = time.perf_counter()
        for d in bs[1:]:
            e.submit_device(d.data_ptr(), n, L, stride)
        e.sync()
        dt = time.perf_counter() - t0
        regs, f1 = e.finish()
    print("nthll k=%d: %.1f G k-mers/s (%.3f ms per 10 M reads), est %d" % (k, 3 * n * (L - k + 1) / dt / 1e9, dt / 3 * 1e3, nt.hll_estimate(regs, 16)))
