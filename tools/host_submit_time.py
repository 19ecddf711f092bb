#!/usr/bin/env python3
"""tools/host_submit_time.py — PCIe-inclusive rate of the host-buffer entry point ntc_submit (pack -> pinned -> H2D -> K1)"""
import sys, os, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc
import ntcard_amd as nt
n, L = 4_000_000, 150
slots = orc.gen_reads(3, 0, n, L, 152, 1, genome_len=100_000_000).reshape(n, 152)
bases = np.ascontiguousarray(slots[:, :L]).reshape(-1)
offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
for threads in (1, 4, 8):
    with nt.Engine([32], r_bits=27, s_bits=7) as e:
        e.submit(bases[: 150 * 100000], offs[:100001])  # warm-up: allocations, pinning
        e.sync()
        t0 = time.perf_counter()
        def work():
            for _ in range(3):
                e.submit(bases, offs)
        ts = [threading.Thread(target=work) for _ in range(threads)]
        [t.start() for t in ts]; [t.join() for t in ts]
        e.sync()
        dt = time.perf_counter() - t0
        _, _, f1 = e.finish(counters=False, p_hist=True)
    reads = 3 * threads * n
    print("ntc_submit, %d caller thread(s): %.1f M reads/s = %.1f G k-mers/s = %.1f GB/s of bases" % (threads, reads / dt / 1e6, reads * (L - 31) / dt / 1e9, reads * L / dt / 1e9))
