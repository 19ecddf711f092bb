#!/usr/bin/env python3
"""tools/full_parity.py — BASELINE.json configs 2, 4 and 5 at FULL size: the GPU engine's t_Counter image, F1 and the
rendered .hist of 100 M synthetic 150 bp reads against the CPU oracle (OpenMP) on the same generator.
Takes a few minutes of host time; not part of the pytest suite.   python tools/full_parity.py [n_reads]"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import orc
import ntcard_amd as nt

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
L, STRIDE, CH, RB, SB = 150, 152, 4_000_000, 27, 7
cores = os.cpu_count() or 1


def hist_bytes(ph, f1):
    F0, f = nt.estimate(ph, RB, SB, 1000)
    path = "/tmp/full_parity.hist"
    nt.write_hist(path, int(f1), F0, f, 1000)
    return open(path, "rb").read()


for name, klist, gap in (("config 2: k=32", [32], 0), ("config 4: k=32,64,96,128", [32, 64, 96, 128], 0), ("config 5: k=12 gap 2", [12], 2)):
    t0 = time.time()
    with nt.Engine(klist, gap=gap, r_bits=RB, s_bits=SB) as e:
        d = torch.empty(10_000_000 * STRIDE + 16, dtype=torch.uint8, device="cuda")
        for first in range(0, N, 10_000_000):
            n = min(10_000_000, N - first)
            nt.gen_reads_device(d.data_ptr(), 1, first, n, L, STRIDE, 1, genome_len=100_000_000)
            e.submit_device(d.data_ptr(), n, L, STRIDE)
            e.sync()
        tc, ph, f1 = e.finish(counters=True)
    t_gpu = time.time() - t0
    t0 = time.time()
    oc = np.zeros((len(klist), 2, 1 << RB), dtype=np.uint16)
    of1 = np.zeros(len(klist), dtype=np.uint64)
    offs = np.arange(CH + 1, dtype=np.uint64) * np.uint64(L)
    for first in range(0, N, CH):
        n = min(CH, N - first)
        slots = orc.gen_reads(1, first, n, L, STRIDE, 1, genome_len=100_000_000)
        bases = np.ascontiguousarray(slots.reshape(n, STRIDE)[:, :L]).reshape(-1)
        of1 += orc.sketch_update(oc, bases, offs[: n + 1], klist, gap, RB, SB, threads=cores)
    t_cpu = time.time() - t0
    same_f1 = bool(np.array_equal(f1, of1))
    same_tc = bool(np.array_equal(tc, oc))
    same_hist = all(hist_bytes(ph[ki], f1[ki]) == hist_bytes(orc.value_hist(oc[ki], RB), of1[ki]) for ki in range(len(klist)))
    print("%s, %d reads: F1 %s  t_Counter (%d MiB) %s  .hist %s   [sha1 %s]  gpu path %.1f s (incl. generation), oracle %.1f s on %d threads"
          % (name, N, "IDENTICAL" if same_f1 else "DIFFERENT", tc.nbytes >> 20, "IDENTICAL" if same_tc else "DIFFERENT",
             "IDENTICAL" if same_hist else "DIFFERENT", hashlib.sha1(tc.tobytes()).hexdigest()[:16], t_gpu, t_cpu, cores), flush=True)
    print("   F1 =", [int(x) for x in f1])
    assert same_f1 and same_tc and same_hist
print("full-size parity OK")
