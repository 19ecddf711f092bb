"""K1h on the CPU: the GENERATED instruction list of the one-wave-per-tile kernel (ntcard_amd/csrc/gen_k1h.py) runs on the wave
emulator (k1h_asm.Emu) and, together with a Python model of the fix-up kernels K1f, must reproduce the oracle's counters and F1
exactly (tests/k1h_model.py).  This is what pins the generator's logic before the kernel reaches a GPU: register assignment, the
list scheduler's reordering, the chunk cursor, the ring, the queue, the hit log with its region switches, suspects and ties.
"""
import numpy as np
import pytest

import k1h_model as km
import orc


def tile_array(arr):
    n, L = arr.shape
    C16, ntl = (L + 15) // 16, (n + 2047) // 2048
    a = np.full((ntl * 2048, C16 * 16), ord("A"), dtype=np.uint8)
    a[:n, :L] = arr
    return np.ascontiguousarray(a.reshape(ntl, 2048, C16, 16).transpose(0, 2, 1, 3)).reshape(-1)


def run(n, L, k, p_bad=0.0, r_bits=14, n_waves=2, seed=1, s_bits=7, gap=0, **kw):
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(b"ACGTacgtUuNnRYKM.-*", dtype=np.uint8)
    arr = alpha[rng.integers(0, 4, size=(n, L))]
    if p_bad:
        arr = np.where(rng.random((n, L)) < p_bad, alpha[rng.integers(4, len(alpha), size=(n, L))], arr).astype(np.uint8)
    res = km.run_k1h(tile_array(arr), n, L, k, r_bits=r_bits, n_waves=n_waves, s_bits=s_bits, gap=gap, **kw)
    reads = [arr[i].tobytes() for i in range(n)]
    fk, f1_sub = km.k1f_model(reads, L, k, r_bits, s_bits, res["dirty"], res["tie"], res["sus"], res["sus_overflow"], gap=gap)
    got = np.bincount(np.concatenate([res["keys"], np.array(fk, dtype=np.uint32)]).astype(np.int64), minlength=2 << r_bits).astype(np.uint32) + res["sketch"]
    oc, of1 = orc.sketch_reads(reads, [k], gap, r_bits, s_bits)
    assert res["f1"] - f1_sub == int(of1[0])
    assert np.array_equal(got, oc[0].reshape(-1).astype(np.uint32))
    return res


def run_ragged(n, C, k, p_bad=0.0, r_bits=14, n_waves=2, seed=1, s_bits=7, gap=0, lo=None, **kw):
    """a ragged batch: reads of 16 C - 15 .. 16 C bases (or lo .. 16 C), every tile sorted longest first, tails[tile][d] = its reads with more than d
    bases in their last 16-base piece"""
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(b"ACGTacgtUuNnRYKM.-*", dtype=np.uint8)
    lens = rng.integers(max(lo or 16 * C - 15, 16 * C - 15), 16 * C + 1, size=n)
    ntl = (n + 2047) // 2048
    reads, tails = [], np.zeros((ntl, 16), dtype=np.uint32)
    arr = np.full((n, 16 * C), ord("A"), dtype=np.uint8)
    for t in range(ntl):
        ls = np.sort(lens[t * 2048:(t + 1) * 2048])[::-1]
        for d in range(16):
            tails[t, d] = int(np.count_nonzero(ls - 16 * (C - 1) > d))
        for j, ln in enumerate(ls):
            row = alpha[rng.integers(0, 4, size=ln)]
            if p_bad:
                row = np.where(rng.random(ln) < p_bad, alpha[rng.integers(4, len(alpha), size=ln)], row).astype(np.uint8)
            arr[t * 2048 + j, :ln] = row
            reads.append(row.tobytes())
    res = km.run_k1h(tile_array(arr), n, 16 * C, k, r_bits=r_bits, n_waves=n_waves, s_bits=s_bits, gap=gap, tails=tails, **kw)
    fk, f1_sub = km.k1f_model(reads, 16 * C, k, r_bits, s_bits, res["dirty"], res["tie"], res["sus"], res["sus_overflow"], gap=gap)
    got = np.bincount(np.concatenate([res["keys"], np.array(fk, dtype=np.uint32)]).astype(np.int64), minlength=2 << r_bits).astype(np.uint32) + res["sketch"]
    oc, of1 = orc.sketch_reads(reads, [k], gap, r_bits, s_bits)
    assert res["f1"] - f1_sub == int(of1[0])
    assert np.array_equal(got, oc[0].reshape(-1).astype(np.uint32))
    return res


@pytest.mark.parametrize("n,C,k,p_bad,n_waves", [(2048, 10, 32, 0.0, 2), (5000, 10, 32, 0.003, 3), (3000, 3, 32, 0.01, 2), (2100, 4, 25, 0.01, 2),
                                                 (2500, 2, 12, 0.02, 2), (4100, 5, 17, 0.0, 4), (2048, 3, 31, 0.02, 1), (2048, 2, 20, 0.01, 2), (2300, 1, 12, 0.0, 2)])
def test_k1h_emulated_ragged_batches(n, C, k, p_bad, n_waves):
    """reads of unequal length (ntRead takes any string, ntcard.cpp:173-189): a batch of reads 16 C - 15 .. 16 C bases long, tiles sorted longest first,
    the steps that end in the last piece masked to the prefix of the tile that is long enough (gen_k1h.Gen.tail_step)"""
    run_ragged(n, C, k, p_bad, n_waves=n_waves, seed=n + C)


@pytest.mark.parametrize("n,L,k,p_bad,n_waves", [
    (2048, 40, 32, 0.0, 2),       # one tile shared by two waves (the second one fills its window with two masked blocks)
    (4097, 47, 32, 0.02, 3),      # a partial last tile of one read
    (2100, 64, 25, 0.01, 2), (2049, 33, 20, 0.0, 2), (3000, 30, 16, 0.01, 2), (2500, 20, 12, 0.02, 2),  # other k: window start chunk, phase, table groups
    (1, 150, 32, 0.0, 2), (2048, 160, 32, 0.001, 2),  # a virtual chunk behind the read (blocks = chunks + 1)
])
def test_k1h_emulated_matches_oracle(n, L, k, p_bad, n_waves):
    run(n, L, k, p_bad, n_waves=n_waves)


def test_k1h_emulated_150bp_two_tiles():
    res = run(5000, 150, 32, 0.002)
    assert len(res["sus"]) > 0 and not res["sus_overflow"]


def test_k1h_emulated_log_regions_and_direct_atomics():
    res = run(4096, 100, 31, 0.005, log_region_cap=256, log_regions=6)  # region switches, then out of regions: device atomics
    assert res["sketch"].any() and res["sk_dirty"] == 1              # ... which the wave reports in the engine's "direct atomics happened" word (round 6)
    res = run(2048, 80, 32, use_log=False)
    assert res["sketch"].any() and res["sk_dirty"] == 1
    res = run(2048, 80, 32, use_log=False, sk_dirty_word=False)        # an engine without a log passes no such word
    assert res["sk_dirty"] == 0
    res = run(2048, 80, 32)                                           # everything logged: the word stays clear
    assert not res["sketch"].any() and res["sk_dirty"] == 0


def test_k1h_emulated_suspect_overflow_and_dense_non_bases():
    res = run(4097, 47, 32, 0.02, n_waves=3, sus_cap=16)
    assert res["sus_overflow"]                                   # K1f's slow path (the model's)
    run(2100, 150, 32, 0.3)
    run(500, 150, 32, 1.0, n_waves=1)


@pytest.mark.parametrize("s_bits", [8, 9, 11])
def test_k1h_emulated_larger_s_bits(s_bits):
    """sBits >= 8 (>= 50 GB of input, ntcard.cpp:427-431): the walk tests 8-bit prefixes of ntComp's patterns, the resolve pass the rest"""
    run(6000, 150, 32, 0.002, s_bits=s_bits, r_bits=12)


@pytest.mark.parametrize("k,gap,L,s_bits", [(12, 2, 150, 7), (12, 2, 40, 7), (32, 8, 100, 7), (20, 4, 64, 9), (13, 1, 30, 7)])
def test_k1h_emulated_spaced_seeds(k, gap, L, s_bits):
    """stRead / NTMSM64 with ntcard's one seed "1" x (k-g)/2 "0" x g "1" x rest (ntcard.cpp:160-171,407-413): two more terms per strand and
    step in the walk, a resolve table without the don't-care positions, a start state without them"""
    run(3000, L, k, 0.004, gap=gap, s_bits=s_bits, r_bits=12)


@pytest.mark.parametrize("k", [13, 14, 15, 17, 19, 23, 24, 27, 28, 29, 30])
def test_k1h_emulated_other_k(k):
    """the library is built with every k of 12 .. 32 (gen_k1h.VARIANTS): phases (k - 1) mod 16 and table sizes the cases above do not touch"""
    run(2100, k + 29, k, 0.01, n_waves=2, seed=k)


def test_k1h_generator_budgets():
    """every variant the library is built with fits the machine: 255 VGPRs, SGPRs below the compiler's reserved ones, six wave areas + table
    + the SIMD numbers inside a CU's 160 KiB of LDS"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ntcard_amd", "csrc"))
    import gen_k1h
    assert gen_k1h.S_END <= 100 and gen_k1h.V_CQMASK4 <= 254
    for k, gap in gen_k1h.VARIANTS:
        assert gen_k1h.TABLE_OFF + gen_k1h.table_bytes(k) <= gen_k1h.LDS_BYTES
        for sb in (7, 8):
            prog = gen_k1h.Gen(k, sb, gap).build()
            assert 3500 < prog.n_insts() < 6200  # (x ~8 bytes: inside the 64 KiB instruction cache two CUs share)
