#!/usr/bin/env python3
"""tools/mode_sweep.py — K1 with the hit log vs direct atomics over genome sizes (how often the sampled k-mers of a batch repeat):
the data behind log_decide_kernel's thresholds.  python tools/mode_sweep.py [k] [gap]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import ntcard_amd as nt

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
gap = int(sys.argv[2]) if len(sys.argv) > 2 else 0
R, L, steps = 10_000_000, 150, 6
stride = 152
buf = torch.empty(R * stride, dtype=torch.uint8, device="cuda")
for glen in (100_000, 1_000_000, 10_000_000, 100_000_000, 1_000_000_000):
    nt.gen_reads_device(buf.data_ptr(), 1, 0, R, L, stride, 1, genome_len=glen)
    torch.cuda.synchronize()
    row = []
    for name, flags in (("adaptive", 0), ("log", nt.FLAG_ALWAYS_LOG), ("direct", nt.FLAG_DIRECT_ATOMICS)):
        with nt.Engine([k], gap=gap, r_bits=27, s_bits=7, flags=flags | nt.FLAG_LANE_KERNEL) as e:
            e.submit_device(buf.data_ptr(), R, L, stride)
            e.flush(); e.sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                e.submit_device(buf.data_ptr(), R, L, stride)
            e.flush(); e.sync()
            dt = (time.perf_counter() - t0) / steps * 1e3
            row.append(f"{name} {dt:.3f} ms" + (f" (chose {e.update_mode()})" if name == "adaptive" else ""))
    print(f"k={k} gap={gap} genome {glen:>11}: " + ", ".join(row), flush=True)
