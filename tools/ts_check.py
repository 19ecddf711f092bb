#!/usr/bin/env python3
"""tools/ts_check.py — quick GPU check of the tiled streaming kernel K1c against the oracle, then a timing run.
   python tools/ts_check.py [--no-perf] [--reads N]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import ntcard_amd as nt
import orc

ap = argparse.ArgumentParser()
ap.add_argument("--no-perf", action="store_true")
ap.add_argument("--reads", type=int, default=10_000_000)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--dist", default="g")
args = ap.parse_args()
dev = torch.device("cuda:0")


def check_reads(reads, L, k=32, r_bits=18, s_bits=7, tag="", flags=0):
    tiles = torch.from_numpy(nt.tile_reads(reads, L)).to(dev)
    with nt.Engine([k], r_bits=r_bits, s_bits=s_bits, device=0, flags=flags | nt.FLAG_REQUIRE_TILED) as e:
        e.submit_tiled_device(tiles.data_ptr(), len(reads), L)
        tc, ph, f1 = e.finish(counters=True)
    oc, of1 = orc.sketch_reads(reads, [k], 0, r_bits, s_bits)
    ok = np.array_equal(f1, of1) and np.array_equal(tc, oc)
    print(f"{'ok  ' if ok else 'FAIL'} {tag}: n={len(reads)} L={L} s={s_bits} F1 {int(f1[0])} vs {int(of1[0])}, increments {int(tc.sum())} vs {int(oc.sum())}", flush=True)
    if not ok:
        d = np.argwhere(tc != oc)
        print("   first diffs:", d[:5].tolist(), [(int(tc[tuple(i)]), int(oc[tuple(i)])) for i in d[:5]])
    return ok


def gen_host(n, L, dist, genome_len=200_000, seed=1):
    stride = (L + 3) & ~3
    sl = orc.gen_reads(seed, 0, n, L, stride, dist, genome_len=genome_len)
    return [sl[i * stride: i * stride + L].tobytes() for i in range(n)]


allok = True
# generator equality: tiled K0 == row-major K0 re-tiled
n, L = 5000, 150
tb = nt.tiled_bytes(n, L)
t = torch.empty(tb, dtype=torch.uint8, device=dev)
nt.gen_reads_tiled_device(t.data_ptr(), 1, 0, n, L, 1, genome_len=200_000)
torch.cuda.synchronize()
reads = gen_host(n, L, 1)
ok = np.array_equal(t.cpu().numpy(), nt.tile_reads(reads, L))
print("ok  " if ok else "FAIL", "tiled generator == oracle generator")
allok &= ok
for k in (12, 13, 15, 16, 17, 20, 21, 24, 25, 27, 29, 31):
    for (n, L, dist, s) in ((5000, 150, 1, 7), (3000, k, 0, 7), (2500, k + 1, 1, 8), (2100, 97, 1, 11)):
        allok &= check_reads(gen_host(n, L, dist), L, k=k, s_bits=s, tag=f"k={k} dist={dist}")
for (n, L, dist, s) in [(5000, 150, 1, 7), (2048, 150, 0, 7), (4096, 100, 1, 7), (100, 159, 1, 7), (3000, 33, 1, 7), (3000, 32, 0, 7), (6000, 150, 1, 8), (6000, 150, 1, 11),
                        (2500, 250, 1, 7), (1, 150, 1, 7), (70000, 150, 1, 7), (1_300_000, 150, 1, 7)]:
    allok &= check_reads(gen_host(n, L, dist), L, s_bits=s, tag=f"dist={dist}")
# adversarial: many N, IUPAC, lower case, U, poly-A, all-N reads
rng = np.random.default_rng(5)
alpha = np.frombuffer(b"ACGTacgtUuNnRYKM.-*", dtype=np.uint8)
for (n, L, p_bad) in [(3000, 150, 0.02), (2100, 150, 0.3), (2048, 64, 0.005), (500, 150, 1.0)]:
    base = rng.integers(0, 4, size=(n, L))
    arr = alpha[base]
    bad = rng.random((n, L)) < p_bad
    arr = np.where(bad, alpha[rng.integers(4, len(alpha), size=(n, L))], arr).astype(np.uint8)
    rd = [arr[i].tobytes() for i in range(n)]
    rd[0] = b"A" * L
    rd[1] = b"N" * L
    rd[2] = b"acgu" * (L // 4) + b"a" * (L % 4)
    allok &= check_reads(rd, L, tag=f"adversarial p_bad={p_bad}")
allok &= check_reads([b"A" * 150] * 4096, 150, tag="poly-A x4096 (every lane pushes at the same steps)")
allok &= check_reads(gen_host(5000, 150, 1), 150, tag="direct atomics", flags=nt.FLAG_DIRECT_ATOMICS)
print("ALL OK" if allok else "SOME FAILED", flush=True)

if not args.no_perf:
    R, L, K = args.reads, 150, args.steps
    dist = 1 if args.dist == "g" else 0
    nb = 4
    bufs = []
    for s in range(nb):
        b = torch.empty(nt.tiled_bytes(R, L), dtype=torch.uint8, device=dev)
        nt.gen_reads_tiled_device(b.data_ptr(), 1, s * R, R, L, dist, 100_000_000)
        bufs.append(b)
    torch.cuda.synchronize()
    with nt.Engine([32], r_bits=27, s_bits=7, device=0, flags=nt.FLAG_REQUIRE_TILED) as e:
        for _ in range(2):
            e.submit_tiled_device(bufs[0].data_ptr(), R, L)
        e.flush()
        e.sync()
        e.reset()
        e.set_profiling(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(K):
            e.submit_tiled_device(bufs[s % nb].data_ptr(), R, L)
        e.flush()
        e.sync()
        dt = time.perf_counter() - t0
        ker_ms, launches = e.kernel_time()
        ap_ms, applies = e.apply_time()
        _, ph, f1 = e.finish()
    print(f"perf: {K} steps x {R} reads: {dt * 1e3 / K:.3f} ms/step wall, hash kernel {ker_ms / K:.3f} ms/step, apply {ap_ms / K:.3f} ms/step ({applies} applies), "
          f"{int(f1[0]) / dt / 1e12:.3f} T k-mers/s, F1={int(f1[0])}")
