#!/usr/bin/env python3
"""tools/fuzz_tiled.py [n_cases] [seed] — randomised parity sweep of the TILED device path (K1h + K1f) against
the oracle: k (every built variant, the spaced seed included), read length, batch sizes from one read to hundreds of tiles (many waves, block
ranges that split tiles, the share-out by SIMD), sBits / rBits, rate and kind of non-base bytes (table-slot bytes 1, 3, 4, 5, 7 included),
small hit logs (region switches, applies in mid-run), several submits per engine with deferred fix-ups; every third case RAGGED: reads of mixed lengths in
length bins, submitted bin by bin (ntc_submit_tiled_ragged_device) or several bins per call (ntc_submit_tiled_bins_device: one launch per k over up to 8 bins,
more go in groups), an equal-length bin among them now and then.  Prints the first mismatch, exits 1."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

torch.cuda.init()
import orc
import ntcard_amd as nt

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
alpha = np.frombuffer(b"ACGTacgtUuNnRYKM.-*", dtype=np.uint8)
odd = np.array([1, 3, 4, 5, 7], dtype=np.uint8)



def ragged_case(case):
    """-> (ok, description)"""
    if rng.random() < 0.2:
        klist, gap = ([12], 2) if rng.random() < 0.5 else ([32], 8)
    else:
        klist, gap = sorted(set(int(x) for x in rng.integers(12, 33, size=int(rng.choice([1, 1, 2, 3]))))), 0
    s_bits = int(rng.choice([7, 7, 8, 11]))
    r_bits = int(rng.choice([12, 16, 20, 23]))
    p_bad = float(rng.choice([0.0, 0.0005, 0.005, 0.05]))
    flags = nt.FLAG_REQUIRE_TILED | (nt.FLAG_DEFER_REDO if rng.random() < 0.5 else 0)
    log_entries = int(rng.choice([0, 1 << 18, 1 << 20]))
    n_sub = int(rng.choice([1, 2, 4, 7]))
    lo = int(rng.integers(1, 120))
    hi = lo + int(rng.choice([0, 5, 30, 60, 150]))
    all_reads, calls, keep = [], [], []
    for _ in range(n_sub):
        n = int(rng.choice([1, 300, 5000, 40_000, 150_000]))
        lens = rng.integers(lo, hi + 1, size=n)
        big = alpha[rng.integers(0, 4, size=(n, hi))]
        if p_bad:
            big = np.where(rng.random((n, hi)) < p_bad, alpha[rng.integers(4, len(alpha), size=(n, hi))], big).astype(np.uint8)
        reads = [big[i, :lens[i]].tobytes() for i in range(n)]
        all_reads += reads
        bins = []
        for C in sorted(set(int(x) for x in (lens + 15) // 16), reverse=True):
            sel = [r for r in reads if (len(r) + 15) // 16 == C]
            if rng.random() < 0.15 and len(set(len(r) for r in sel)) == 1:  # an equal-length bin
                bins.append((torch.from_numpy(nt.tile_reads(sel, len(sel[0]))).cuda(), len(sel), len(sel[0]), None))
            else:
                tiles, tails, _ = nt.tile_reads_ragged(sel, C)
                bins.append((torch.from_numpy(tiles).cuda(), len(sel), 16 * C, torch.from_numpy(tails.reshape(-1).astype(np.int32)).cuda()))
        keep.append(bins)
        calls.append(("bins" if rng.random() < 0.7 else "each", bins))
    oc, of1 = orc.sketch_reads(all_reads, klist, gap, r_bits, s_bits)
    desc = f"case {case} RAGGED: klist={klist} gap={gap} s={s_bits} r={r_bits} len={lo}..{hi} reads={[sum(b[1] for b in c[1]) for c in calls]} bins={[len(c[1]) for c in calls]} how={[c[0] for c in calls]} p_bad={p_bad} flags={flags} log={log_entries}"
    with nt.Engine(klist, gap=gap, r_bits=r_bits, s_bits=s_bits, flags=flags, log_entries=log_entries) as e:
        for how, bins in calls:
            if how == "bins":
                e.submit_tiled_bins_device([(t.data_ptr(), n, L, (d.data_ptr() if d is not None else 0)) for t, n, L, d in bins])
            else:
                for t, n, L, d in bins:
                    if d is None:
                        e.submit_tiled_device(t.data_ptr(), n, L)
                    else:
                        e.submit_tiled_ragged_device(t.data_ptr(), n, L // 16, d.data_ptr())
        tc, ph, f1 = e.finish(counters=True)
    ok = np.array_equal(f1, of1) and np.array_equal(tc, oc)
    if not ok:
        print("MISMATCH", desc, "f1", f1, of1, "counters differ at", int((tc != oc).sum()))
    return ok, desc


for case in range(n_cases):
    if case % 3 == 2:
        ok, desc = ragged_case(case)
        if not ok:
            sys.exit(1)
        if case % 5 == 0 or case % 5 == 2:
            print("ok", desc, flush=True)
        continue
    teams = False  # (round 3's kernel K1c, NTC_FLAG_TILED_TEAMS, was retired in round 5)
    gap = 0
    if not teams and rng.random() < 0.15:
        klist, gap = ([12], 2) if rng.random() < 0.6 else ([32], 8)
    else:
        klist = sorted(set(int(x) for x in rng.integers(12, 33, size=int(rng.choice([1, 1, 1, 2, 3])))))
    s_bits = int(rng.choice([7, 7, 8, 9, 11, 14]))
    r_bits = int(rng.choice([12, 16, 20, 23]))
    L = int(rng.choice([int(rng.integers(max(klist), 401)), 100, 150, 151, 250]))
    p_bad = float(rng.choice([0.0, 0.0005, 0.005, 0.05]))
    slot_bytes = (not teams) and rng.random() < 0.2
    flags = nt.FLAG_REQUIRE_TILED | (nt.FLAG_DEFER_REDO if rng.random() < 0.6 else 0)
    log_entries = int(rng.choice([0, 1 << 18, 1 << 20]))
    n_sub = int(rng.choice([1, 1, 2, 5, 11]))
    sizes = [int(rng.choice([1, 70, 2048, 2049, 30_000, 200_000, 500_000])) for _ in range(n_sub)]
    if sum(sizes) * L > 120_000_000:
        sizes = [min(s, 120_000_000 // (L * n_sub)) or 1 for s in sizes]
    counters = np.zeros((len(klist), 2, 1 << r_bits), dtype=np.uint16)
    of1 = np.zeros(len(klist), dtype=np.uint64)
    bufs, arrs = [], []
    for n in sizes:
        arr = alpha[rng.integers(0, 4, size=(n, L))]
        if p_bad:
            arr = np.where(rng.random((n, L)) < p_bad, alpha[rng.integers(4, len(alpha), size=(n, L))], arr).astype(np.uint8)
        if slot_bytes:
            m = max(1, n // 500)
            arr[rng.integers(0, n, size=m), rng.integers(0, L, size=m)] = odd[rng.integers(0, 5, size=m)]
        arrs.append(arr)
        bufs.append(torch.from_numpy(nt.tile_reads([arr[i].tobytes() for i in range(n)], L) if n <= 4096 else
                                     np.ascontiguousarray(np.pad(arr, ((0, (-n) % 2048), (0, (-L) % 16)), constant_values=ord("A"))
                                                          .reshape(-1, 2048, (L + 15) // 16, 16).transpose(0, 2, 1, 3)).reshape(-1)).cuda())
        offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
        of1 += orc.sketch_update(counters, np.ascontiguousarray(arr).reshape(-1), offs, klist, gap, r_bits, s_bits)
    desc = f"case {case}: klist={klist} gap={gap} s={s_bits} r={r_bits} L={L} sizes={sizes} p_bad={p_bad} slot_bytes={slot_bytes} teams={teams} flags={flags} log={log_entries}"
    with nt.Engine(klist, gap=gap, r_bits=r_bits, s_bits=s_bits, flags=flags, log_entries=log_entries) as e:
        for arr, t in zip(arrs, bufs):
            e.submit_tiled_device(t.data_ptr(), arr.shape[0], L)
        tc, ph, f1 = e.finish(counters=True)
    if not (np.array_equal(f1, of1) and np.array_equal(tc, counters)):
        print("MISMATCH", desc, "f1", f1, of1, "counters differ at", int((tc != counters).sum()))
        sys.exit(1)
    if case % 5 == 0:
        print("ok", desc, flush=True)
print(f"fuzz tiled OK: {n_cases} cases")
