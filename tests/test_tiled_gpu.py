"""GPU parity of the TILED path (ntc_submit_tiled_device -> K1h + K1f, ntc_sketch_k1h.hip / gen_k1h.py): the one-wave-per-tile kernel and its
fix-up kernels against the CPU oracle on seeded and adversarial inputs (bit-exact), and against the real reference's full-size digests.

ntHashIterator's N semantics (ntHashIterator.hpp:59-86) are K1f's job (K1h leaves every window near a non-ACGTU byte to it), so the inputs lean on
that: reads with many non-ACGTU bytes, all-N reads, lower case / U, partial tiles, read lengths around the 16-base chunks.
"""
import hashlib
import json
import os
import random

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize")


@pytest.fixture(scope="module")
def nt():
    assert torch.cuda.is_available(), "GPU tests need a HIP device (run on the MI355X box)"
    import ntcard_amd
    return ntcard_amd


def gen_host(n, L, dist, genome_len=200_000, seed=1):
    stride = (L + 3) & ~3
    sl = orc.gen_reads(seed, 0, n, L, stride, dist, genome_len=genome_len)
    return [sl[i * stride: i * stride + L].tobytes() for i in range(n)]


class _DevArray:
    """zero-copy view of a device int32 array for torch.as_tensor (__cuda_array_interface__)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i4", "data": (int(ptr), False), "version": 2}


VARIANT_FLAGS = 0  # (rounds 3-4 ran every test of this module a second time through K1c, NTC_FLAG_TILED_TEAMS; round 5 retired that kernel)


def run_tiled(nt, reads, L, k=32, r_bits=18, s_bits=7, flags=0, pieces=1, log_entries=0):
    # (K1h's table word holds r_bits + 1 + s_bits - 7 bits: a configuration beyond 32 is K1's, after a re-layout — the one case here that may fall back)
    req = nt.FLAG_REQUIRE_TILED if r_bits + 1 + s_bits - 7 <= 32 else 0
    with nt.Engine([k], r_bits=r_bits, s_bits=s_bits, flags=flags | VARIANT_FLAGS | req, log_entries=log_entries) as e:
        step = (len(reads) + pieces - 1) // pieces
        keep = []
        for i in range(0, len(reads), step):
            part = reads[i:i + step]
            t = torch.from_numpy(nt.tile_reads(part, L)).cuda()
            keep.append(t)
            e.submit_tiled_device(t.data_ptr(), len(part), L)
        return e.finish(counters=True)


def check(nt, reads, L, **kw):
    tc, ph, f1 = run_tiled(nt, reads, L, **kw)
    oc, of1 = orc.sketch_reads(reads, [kw.get("k", 32)], 0, kw.get("r_bits", 18), kw.get("s_bits", 7))
    assert np.array_equal(f1, of1), (f1, of1)
    assert np.array_equal(tc, oc)
    assert np.array_equal(ph[0], orc.value_hist(oc[0], kw.get("r_bits", 18)))


def test_tiled_generator_matches_oracle(nt):
    n, L = 5000, 150
    t = torch.empty(nt.tiled_bytes(n, L), dtype=torch.uint8, device="cuda")
    nt.gen_reads_tiled_device(t.data_ptr(), 1, 0, n, L, 1, genome_len=200_000)
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy(), nt.tile_reads(gen_host(n, L, 1), L))


@pytest.mark.parametrize("n,L,dist,s_bits", [
    (5000, 150, 1, 7), (2048, 150, 0, 7), (4096, 100, 1, 7), (100, 159, 1, 7), (3000, 33, 1, 7), (3000, 32, 0, 7), (6000, 150, 1, 8),
    (6000, 150, 1, 11), (2500, 250, 1, 7), (1, 150, 1, 7), (70000, 150, 1, 7), (4097, 47, 1, 9), (2049, 48, 0, 7), (3000, 49, 1, 24),
])
def test_tiled_kernel_matches_oracle(nt, n, L, dist, s_bits):
    check(nt, gen_host(n, L, dist), L, s_bits=s_bits)


@pytest.mark.parametrize("n,L,p_bad", [(3000, 150, 0.02), (2100, 150, 0.3), (2048, 64, 0.005), (500, 150, 1.0), (5000, 97, 0.001)])
def test_tiled_kernel_non_acgt_bytes(nt, n, L, p_bad):
    """N / IUPAC / punctuation anywhere, lower case and U as bases, all-N and poly-A reads: F1 and every counter"""
    rng = np.random.default_rng(int(p_bad * 1000) + L)
    alpha = np.frombuffer(b"ACGTacgtUuNnRYKMBDHVSW.-*@`bf", dtype=np.uint8)  # '@', '`', 'B', 'b', 'F', 'f' differ from a base letter in one bit
    arr = alpha[rng.integers(0, 4, size=(n, L))]
    bad = rng.random((n, L)) < p_bad
    arr = np.where(bad, alpha[rng.integers(4, len(alpha), size=(n, L))], arr).astype(np.uint8)
    reads = [arr[i].tobytes() for i in range(n)]
    reads[0] = b"A" * L
    reads[1] = b"N" * L
    reads[2] = (b"acgu" * L)[:L]
    reads[3] = b"N" + reads[3][1:]
    reads[4] = reads[4][:-1] + b"N"
    reads[5] = reads[5][:31] + b"N" + reads[5][32:]
    check(nt, reads, L)


def test_tiled_every_lane_pushes_at_once(nt):
    """identical reads: every lane has a candidate at the same steps (the strand queues run full and the walkers have to wait)"""
    rng = random.Random(4)
    one = "".join(rng.choice("ACGT") for _ in range(150)).encode()
    check(nt, [one] * 6000, 150, s_bits=7)
    check(nt, [b"A" * 150] * 4096, 150)


def test_tiled_batches_and_modes(nt):
    """several submits into one engine, direct atomics instead of the hit log, the partitioned apply"""
    reads = gen_host(9000, 150, 1)
    check(nt, reads, 150, pieces=3)
    check(nt, reads, 150, flags=nt.FLAG_DIRECT_ATOMICS)
    check(nt, reads, 150, flags=nt.FLAG_ALWAYS_LOG | nt.FLAG_PARTITION_ALWAYS, pieces=2)


@pytest.mark.parametrize("log_entries,pieces", [(0, 1), (1 << 18, 3), (1 << 20, 2)])
def test_packed_runs_between_two_partition_passes(nt, log_entries, pieces):
    """round 6: with two partition passes (r_bits = 24: 5 + 5 bits above the 2^15-counter slices) the first pass writes what it leaves of a key three to a
    64-bit word and the second reads such words — words of one and two keys (small rounds), a read repeated 3000 times (its few counters overflow their
    runs of a small log: the direct fall-back from both passes), several updates per engine (the first one writes, the later ones add)"""
    reads = gen_host(3000, 150, 1) + [gen_host(1, 150, 0, seed=9)[0]] * 3000 + gen_host(500, 150, 0, seed=5)
    check(nt, reads, 150, r_bits=24, flags=nt.FLAG_ALWAYS_LOG | nt.FLAG_PARTITION_ALWAYS, log_entries=log_entries, pieces=pieces)


def test_first_apply_writes_and_later_ones_add(nt):
    """round 6: the first sketch update behind a reset WRITES its counts into the zeroed sketch (count_kernel, no read) unless some kernel has incremented the
    sketch itself — K1h waves out of log regions, K1f's overflow / slow path, a partition run that overflowed; K1f's suspects are log entries of regions of
    their own.  Every order of (logged | direct) x (first | later apply), across flushes and a reset, against the oracle"""
    rng = np.random.default_rng(6)
    alpha = np.frombuffer(b"ACGTacgtNn", dtype=np.uint8)
    n, L = 40_000, 150
    arr = alpha[rng.integers(0, 4, size=(n, L))]
    arr = np.where(rng.random((n, L)) < 0.004, alpha[rng.integers(4, len(alpha), size=(n, L))], arr).astype(np.uint8)
    reads = [arr[i].tobytes() for i in range(n)]
    parts = [reads[:15_000], reads[15_000:22_000], reads[22_000:]]
    tiles = [torch.from_numpy(nt.tile_reads(p, L)).cuda() for p in parts]
    oc, of1 = orc.sketch_reads(reads, [32], 0, 18, 7)
    oc2, of12 = orc.sketch_reads(parts[1] + parts[2], [32], 0, 18, 7)
    # log_entries: default (everything logged, one apply), 2^18 (several applies inside a batch), 2^14 (64 regions of 256: the waves run out of regions and count
    # with device atomics before the first apply: it must add, not write)
    for le in (0, 1 << 18, 1 << 14):
        for flags in (nt.FLAG_PARTITION_ALWAYS | nt.FLAG_REQUIRE_TILED, nt.FLAG_PARTITION_ALWAYS | nt.FLAG_REQUIRE_TILED | nt.FLAG_DEFER_REDO, nt.FLAG_REQUIRE_TILED):
            with nt.Engine([32], r_bits=18, s_bits=7, flags=flags, log_entries=le) as e:
                e.submit_tiled_device(tiles[0].data_ptr(), len(parts[0]), L)
                e.flush()                                            # first apply
                e.submit_tiled_device(tiles[1].data_ptr(), len(parts[1]), L)
                e.flush()                                            # adds
                e.flush()                                            # nothing pending
                e.submit_tiled_device(tiles[2].data_ptr(), len(parts[2]), L)
                tc, ph, f1 = e.finish(counters=True)
                assert np.array_equal(f1, of1) and np.array_equal(tc, oc), (le, flags)
                e.reset()                                            # the sketch is zero again: the next apply is a first one
                e.submit_tiled_device(tiles[1].data_ptr(), len(parts[1]), L)
                e.submit_tiled_device(tiles[2].data_ptr(), len(parts[2]), L)
                tc, ph, f1 = e.finish(counters=True)
                assert np.array_equal(f1, of12) and np.array_equal(tc, oc2), (le, flags, "after reset")
    # the address of the counters handed out: the caller may have added to them, so no apply may overwrite
    with nt.Engine([32], r_bits=18, s_bits=7, flags=nt.FLAG_PARTITION_ALWAYS | nt.FLAG_REQUIRE_TILED) as e:
        sk, ncnt, _ = e.device_state()
        view = torch.as_tensor(_DevArray(sk, ncnt), device="cuda")
        view[12345] += 7
        torch.cuda.synchronize()
        e.submit_tiled_device(tiles[0].data_ptr(), len(parts[0]), L)
        e.submit_tiled_device(tiles[1].data_ptr(), len(parts[1]), L)
        e.submit_tiled_device(tiles[2].data_ptr(), len(parts[2]), L)
        tc, ph, f1 = e.finish(counters=True)
    want = oc.reshape(-1).copy()
    want[12345] += 7
    assert np.array_equal(tc.reshape(-1), want)


@pytest.mark.parametrize("k", list(range(12, 32)))
def test_tiled_kernel_every_k(nt, k):
    """K1h is generated for k = 12 .. 32 (one assembly body per k): reads of
    exactly k and k + 1 bases, 97 and 150 bp, non-ACGTU bytes, sBits 7 / 8 / 11"""
    rng = np.random.default_rng(k)
    alpha = np.frombuffer(b"ACGTacgtUuNnRYKM.-*", dtype=np.uint8)
    for n, L, s_bits, p_bad in ((3000, 150, 7, 0.002), (2100, k, 7, 0.0), (2500, k + 1, 8, 0.01), (2049, 97, 11, 0.0)):
        arr = alpha[rng.integers(0, 4, size=(n, L))]
        if p_bad:
            arr = np.where(rng.random((n, L)) < p_bad, alpha[rng.integers(4, len(alpha), size=(n, L))], arr).astype(np.uint8)
        check(nt, [arr[i].tobytes() for i in range(n)], L, k=k, s_bits=s_bits)


@pytest.mark.parametrize("seed", range(8))
def test_tiled_kernel_random_shapes(nt, seed):
    """random read length (32 .. 400), read count (partial last tile, one read, several tiles per team), sBits and rate of
    non-ACGTU bytes: K1h's chunk / ring / window arithmetic has no special case for 150 bp"""
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.integers(12, 33))
    L = int(rng.integers(k, 401))
    n = int(rng.choice([1, 63, 2047, 2048, 2049, 5000, 9000]))
    s_bits = int(rng.choice([7, 8, 9, 12]))
    p_bad = float(rng.choice([0.0, 0.001, 0.02]))
    alpha = np.frombuffer(b"ACGTacgtUuNnRYKM.-*", dtype=np.uint8)
    arr = alpha[rng.integers(0, 4, size=(n, L))]
    if p_bad:
        arr = np.where(rng.random((n, L)) < p_bad, alpha[rng.integers(4, len(alpha), size=(n, L))], arr).astype(np.uint8)
    check(nt, [arr[i].tobytes() for i in range(n)], L, k=k, s_bits=s_bits)


def test_tiled_kernel_k_lists(nt):
    """a list of k within 16 .. 32 is one K1h launch per k over the same tiles (ntRead's loop over kList, ntcard.cpp:147-158): planes and
    F1 per k as the oracle's; reads shorter than some of the k"""
    for klist, L, n in (([21, 25, 31], 150, 5000), ([32, 16, 24], 100, 2100), ([17, 29], 20, 3000), ([20, 32], 19, 100), ([12, 15, 13], 14, 2500)):
        reads = gen_host(n, L, 1)
        t = torch.from_numpy(nt.tile_reads(reads, L)).cuda()
        with nt.Engine(klist, r_bits=16, s_bits=7, flags=nt.FLAG_REQUIRE_TILED | VARIANT_FLAGS) as e:
            e.submit_tiled_device(t.data_ptr(), n, L)
            e.submit_tiled_device(t.data_ptr(), n // 2, L)  # a prefix of a tiled buffer is a batch: the slots behind its last read are ignored
            tc, ph, f1 = e.finish(counters=True)
        oc, of1 = orc.sketch_reads(reads + reads[: n // 2], klist, 0, 16, 7)
        assert np.array_equal(f1, of1) and np.array_equal(tc, oc), klist


def tile_array(arr):
    """(n, L) uint8 array of reads -> tiled layout (numpy only: millions of reads)"""
    n, L = arr.shape
    C16, ntl = (L + 15) // 16, (n + 2047) // 2048
    a = np.full((ntl * 2048, C16 * 16), ord("A"), dtype=np.uint8)
    a[:n, :L] = arr
    return np.ascontiguousarray(a.reshape(ntl, 2048, C16, 16).transpose(0, 2, 1, 3)).reshape(-1)


@pytest.mark.parametrize("L,k", [(12, 12), (14, 12), (16, 16), (16, 12), (20, 16), (33, 16), (47, 12), (32, 32), (40, 25)])
def test_tiled_many_tiles_per_team_short_reads(nt, L, k):
    """3.2 M reads of 1 .. 3 chunks with 1.5 % non-ACGTU bytes: every team of waves walks >= 3 tiles whose blocks are (nearly) whole
    tiles, so the packer runs several TILES ahead of the resolvers — the dirty-piece queue's tile tags, the suspects' tile tags
    and the ring-slot phase (seq * C) % 5 all wrap within one launch.  F1 and every counter vs the oracle."""
    n = 3_200_000
    rng = np.random.default_rng(L * 100 + k)
    alpha = np.frombuffer(b"ACGTacgtUuNnRYKM.-*", dtype=np.uint8)
    arr = alpha[rng.integers(0, 4, size=(n, L))]
    arr = np.where(rng.random((n, L)) < 0.015, alpha[rng.integers(4, len(alpha), size=(n, L))], arr).astype(np.uint8)
    t = torch.from_numpy(tile_array(arr)).cuda()
    with nt.Engine([k], r_bits=20, s_bits=7, flags=nt.FLAG_REQUIRE_TILED | VARIANT_FLAGS) as e:
        e.submit_tiled_device(t.data_ptr(), n, L)
        tc, ph, f1 = e.finish(counters=True)
    counters = np.zeros((1, 2, 1 << 20), dtype=np.uint16)
    offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    of1 = orc.sketch_update(counters, np.ascontiguousarray(arr).reshape(-1), offs, [k], 0, 20, 7)
    assert np.array_equal(f1, of1), (f1, of1)
    assert np.array_equal(tc, counters)


def test_tiled_layout_falls_back_for_other_configurations(nt):
    """a configuration the tiled kernel is not built for is re-laid out on the device and takes the general kernel"""
    reads = gen_host(5000, 150, 1)
    t = torch.from_numpy(nt.tile_reads(reads, 150)).cuda()
    for klist, gap, s_bits in (([40], 0, 7), ([11], 0, 7), ([32, 64], 0, 7), ([20], 4, 7), ([32], 0, 5)):  # (k = 12 / gap = 2 has a tiled kernel since round 4)
        with nt.Engine(klist, gap=gap, r_bits=16, s_bits=s_bits) as e:
            e.submit_tiled_device(t.data_ptr(), len(reads), 150)
            tc, ph, f1 = e.finish(counters=True)
        oc, of1 = orc.sketch_reads(reads, klist, gap, 16, s_bits)
        assert np.array_equal(f1, of1) and np.array_equal(tc, oc)
        with pytest.raises(nt.NtcError):
            with nt.Engine(klist, gap=gap, r_bits=16, s_bits=s_bits, flags=nt.FLAG_REQUIRE_TILED) as e:
                e.submit_tiled_device(t.data_ptr(), len(reads), 150)


@pytest.mark.parametrize("name,R", [("cfg2", 10_000_000), ("cfg2u", 10_000_000), ("cfg3s", 10_000_000), ("cfg2", 50_000_000), ("cfg5", 10_000_000),
                                    ("cfg2r24", 10_000_000), ("cfg5b", 10_000_000)])  # (cfg2r24: rBits = 24, 20 M reads — a sketch of another size through NTC_FLAG_REQUIRE_TILED)
def test_tiled_fullsize_matches_reference_goldens(nt, name, R, tmp_path):
    """100 M synthetic reads through the tiled kernel against the digests of the REAL reference (see test_fullsize_gpu.py); the
    50 M-read batches make every team of waves walk ~48 tiles in one launch (tile tags of deferred work wrap many times)"""
    with open(os.path.join(GOLD, "digests.json")) as f:
        meta = json.load(f)
    cfg = meta["configs"][name]
    n, L, rb, sb, cov = cfg.get("n_reads", meta["n_reads"]), meta["read_len"], cfg.get("r_bits", meta["r_bits"]), cfg.get("s_bits", meta["s_bits"]), meta["cov_max"]
    buf = torch.empty(nt.tiled_bytes(R, L), dtype=torch.uint8, device="cuda")
    with nt.Engine(cfg["klist"], gap=cfg["gap"], r_bits=rb, s_bits=sb, flags=nt.FLAG_REQUIRE_TILED | VARIANT_FLAGS) as e:
        for first in range(0, n, R):
            m = min(R, n - first)
            nt.gen_reads_tiled_device(buf.data_ptr(), meta["seed"], first, m, L, cfg["dist"], genome_len=100_000_000)
            e.submit_tiled_device(buf.data_ptr(), m, L)  # the buffer is regenerated in place: stream order is all the contract asks for
        tc, ph, f1 = e.finish(counters=True)
    pl = cfg["planes"][0]
    assert int(f1[0]) == pl["f1"]
    assert hashlib.sha1(np.ascontiguousarray(tc[0]).data).hexdigest() == pl["t_counter_sha1"]
    F0, f = nt.estimate(ph[0], rb, sb, cov)
    out = tmp_path / f"{name}.hist"
    nt.write_hist(out, f1[0], F0, f, cov)
    assert out.read_bytes() == open(os.path.join(GOLD, pl["hist_file"]), "rb").read()


@pytest.mark.parametrize("klist,s_bits,L,sizes", [([16, 24, 32, 48], 7, 150, [5000, 70, 2049]), ([32, 64, 96, 128], 7, 150, [9000, 4100]), ([12, 40], 11, 61, [3000, 1]),
                                                 ([33, 20, 100], 8, 250, [2500]), ([31, 32, 33, 34, 35, 36], 7, 100, [4096, 100])])
def test_tiled_batches_with_a_mixed_k_list(nt, klist, s_bits, L, sizes):
    """a k list of which only a part is K1h's (12 .. 32): K1h + K1f take their k from the tiles, K1 stages the SAME tiles for the others (a.tiled: no
    re-layout pass) — device batches of several sizes (partial last waves, one read), reads with N, and the same reads through ntc_submit; multi-k
    through ntRead's loop over kList (ntcard.cpp:147-158)"""
    rng = np.random.default_rng(sum(klist) + L)
    parts = [_ragged_reads(rng, n, L, L, 0.003) for n in sizes]
    allr = sum(parts, [])
    oc, of1 = orc.sketch_reads(allr, klist, 0, 18, s_bits)
    keep = [torch.from_numpy(nt.tile_reads(p, L)).cuda() for p in parts]
    for flags in (0, nt.FLAG_DEFER_REDO, nt.FLAG_ALWAYS_LOG):
        with nt.Engine(klist, r_bits=18, s_bits=s_bits, flags=flags) as e:
            for p, t in zip(parts, keep):
                e.submit_tiled_device(t.data_ptr(), len(p), L)
            tc, ph, f1 = e.finish(counters=True)
        assert np.array_equal(f1, of1), (flags, f1, of1)
        assert np.array_equal(tc, oc), flags
    with nt.Engine(klist, r_bits=18, s_bits=s_bits) as e:  # host reads: packed into tiles (equal length), both kernels from there
        for p in parts:
            e.submit_reads(p)
        tc, ph, f1 = e.finish(counters=True)
    assert np.array_equal(f1, of1) and np.array_equal(tc, oc)
    with pytest.raises(nt.NtcError):  # the validation flag asks for K1h on EVERY k
        with nt.Engine(klist, r_bits=18, s_bits=s_bits, flags=nt.FLAG_REQUIRE_TILED) as e:
            e.submit_tiled_device(keep[0].data_ptr(), len(parts[0]), L)


def test_mixed_k_list_with_reads_too_long_for_tiled_staging(nt):
    """`-k 32,64` on equal-length 4 kb reads.  K1 cannot stage 64 slots of 4 kb next to its tables, so the host batch must take row slots
    (chunked) instead of tiles, and a device-resident tiled batch of such reads is refused with NOTHING counted (K1h used to count its k first)"""
    rng = np.random.default_rng(11)
    L, n, klist = 4096, 1500, [32, 64]
    reads = _ragged_reads(rng, n, L, L, 0.001)
    oc, of1 = orc.sketch_reads(reads, klist, 0, 18, 7)
    with nt.Engine(klist, r_bits=18, s_bits=7) as e:
        e.submit_reads(reads)
        tc, ph, f1 = e.finish(counters=True)
    assert np.array_equal(f1, of1) and np.array_equal(tc, oc)
    tiles = torch.from_numpy(nt.tile_reads(reads, L)).cuda()
    with nt.Engine(klist, r_bits=18, s_bits=7) as e:
        with pytest.raises(nt.NtcError):
            e.submit_tiled_device(tiles.data_ptr(), n, L)
        tc, ph, f1 = e.finish(counters=True)
        assert not f1.any() and not tc.any(), "a refused batch left counts behind"
        e.submit_reads(reads)  # the engine is still usable
        tc, ph, f1 = e.finish(counters=True)
    assert np.array_equal(f1, of1) and np.array_equal(tc, oc)
    with nt.Engine([32], r_bits=18, s_bits=7, flags=nt.FLAG_REQUIRE_TILED) as e:  # K1h alone has no such limit
        e.submit_tiled_device(tiles.data_ptr(), n, L)
        tc, ph, f1 = e.finish(counters=True)
    o32, f32 = orc.sketch_reads(reads, [32], 0, 18, 7)
    assert np.array_equal(f1, f32) and np.array_equal(tc, o32)


def test_tiled_fullsize_config4_mixed_list(nt, tmp_path):
    """BASELINE config 4 (k = 32, 64, 96, 128 on 100 M reads) from TILED batches: k = 32 through K1h + K1f, the other three through K1 staging the tiles —
    F1, the sha1 of every raw t_Counter plane and the .hist bytes of the REAL reference for all four k"""
    with open(os.path.join(GOLD, "digests.json")) as f:
        meta = json.load(f)
    cfg = meta["configs"]["cfg4"]
    n, L, rb, sb, cov = meta["n_reads"], meta["read_len"], meta["r_bits"], meta["s_bits"], meta["cov_max"]
    R = 10_000_000
    buf = torch.empty(nt.tiled_bytes(R, L), dtype=torch.uint8, device="cuda")
    with nt.Engine(cfg["klist"], gap=cfg["gap"], r_bits=rb, s_bits=sb) as e:
        for first in range(0, n, R):
            m = min(R, n - first)
            nt.gen_reads_tiled_device(buf.data_ptr(), meta["seed"], first, m, L, cfg["dist"], genome_len=100_000_000)
            e.submit_tiled_device(buf.data_ptr(), m, L)
        tc, ph, f1 = e.finish(counters=True)
    for ki, pl in enumerate(cfg["planes"]):
        assert int(f1[ki]) == pl["f1"]
        assert hashlib.sha1(np.ascontiguousarray(tc[ki]).data).hexdigest() == pl["t_counter_sha1"]
        F0, f = nt.estimate(ph[ki], rb, sb, cov)
        out = tmp_path / f"cfg4_{ki}.hist"
        nt.write_hist(out, f1[ki], F0, f, cov)
        assert out.read_bytes() == open(os.path.join(GOLD, pl["hist_file"]), "rb").read()


def test_tiled_deferred_fixups(nt):
    """NTC_FLAG_DEFER_REDO: the caller leaves its batches alone until sync, so the engine collects up to eight K1h launches and sends ONE K1f over
    all of them (blockIdx.y = the launch).  Eleven batches of different sizes and read lengths with non-ACGTU bytes: the ninth forces a K1f in
    mid-run; one batch holds reference-table slot bytes (its launch alone takes K1f's slow path); a flush in between; more batches behind it"""
    rng = np.random.default_rng(77)
    alpha = np.frombuffer(b"ACGTacgtUuNnRYKM.-*", dtype=np.uint8)
    parts, bufs = [], []
    for i in range(11):
        n, L = 30_000 + 4000 * i, (150, 97, 64)[i % 3]
        arr = alpha[rng.integers(0, 4, size=(n, L))]
        arr = np.where(rng.random((n, L)) < 0.003, alpha[rng.integers(4, len(alpha), size=(n, L))], arr).astype(np.uint8)
        if i == 4:
            arr[rng.integers(0, n, size=50), rng.integers(0, L, size=50)] = np.array([1, 3, 4, 5, 7], dtype=np.uint8)[rng.integers(0, 5, size=50)]
        parts.append(arr)
        bufs.append(torch.from_numpy(tile_array(arr)).cuda())
    order = list(range(11)) + [0, 1]
    with nt.Engine([32], r_bits=20, s_bits=7, flags=nt.FLAG_REQUIRE_TILED | nt.FLAG_DEFER_REDO | VARIANT_FLAGS) as e:
        for i in order[:11]:
            e.submit_tiled_device(bufs[i].data_ptr(), parts[i].shape[0], parts[i].shape[1])
        e.flush()
        for i in order[11:]:  # and on after a flush
            e.submit_tiled_device(bufs[i].data_ptr(), parts[i].shape[0], parts[i].shape[1])
        tc, ph, f1 = e.finish(counters=True)
    counters = np.zeros((1, 2, 1 << 20), dtype=np.uint16)
    of1 = np.zeros(1, dtype=np.uint64)
    for i in order:
        arr = parts[i]
        offs = np.arange(arr.shape[0] + 1, dtype=np.uint64) * np.uint64(arr.shape[1])
        of1 += orc.sketch_update(counters, np.ascontiguousarray(arr).reshape(-1), offs, [32], 0, 20, 7)
    assert np.array_equal(f1, of1) and np.array_equal(tc, counters)


def test_tiled_reference_table_slot_bytes_and_suspect_overflow(nt):
    """bytes 1, 3, 4, 5, 7 are bases to the reference (its seed table's first row, nthash.hpp:32) and no letters to the packing kernels: K1f
    notices them and re-derives every window near them from the bytes; a read set dense with N overflows the suspect list — same slow path"""
    rng = np.random.default_rng(5)
    n, L = 6000, 150
    arr = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, L))]
    odd = np.array([1, 3, 4, 5, 7], dtype=np.uint8)
    arr = np.where(rng.random((n, L)) < 0.002, odd[rng.integers(0, 5, size=(n, L))], arr).astype(np.uint8)
    if True:
        check(nt, [arr[i].tobytes() for i in range(n)], L)
    dense = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, size=(400_000, 64))].astype(np.uint8)
    t = torch.from_numpy(tile_array(dense)).cuda()
    counters = np.zeros((1, 2, 1 << 20), dtype=np.uint16)
    offs = np.arange(dense.shape[0] + 1, dtype=np.uint64) * np.uint64(64)
    of1 = orc.sketch_update(counters, np.ascontiguousarray(dense).reshape(-1), offs, [32], 0, 20, 7)
    for cap in ("", "64"):  # the engine's own list (sized for every candidate: fast path), then a short one: the launch overflows -> slow path
        if cap:
            os.environ["NTC_K1H_SUS_CAP"] = cap
        try:
            with nt.Engine([32], r_bits=20, s_bits=7, flags=nt.FLAG_REQUIRE_TILED | VARIANT_FLAGS) as e:
                e.submit_tiled_device(t.data_ptr(), dense.shape[0], 64)
                tc, ph, f1 = e.finish(counters=True)
        finally:
            os.environ.pop("NTC_K1H_SUS_CAP", None)
        assert np.array_equal(f1, of1) and np.array_equal(tc, counters), cap


@pytest.mark.parametrize("n,L,p_bad,s_bits", [(6000, 150, 0.003, 7), (2049, 12, 0.01, 7), (3000, 13, 0.0, 8), (70000, 100, 0.001, 11), (400_000, 64, 0.2, 7)])
def test_tiled_spaced_seed_k12_gap2(nt, n, L, p_bad, s_bits):
    """stRead with ntcard's -g seed (ntcard.cpp:160-171,407-413) through the tiled layout: K1h's (k = 12, gap = 2) variant, K1f with the
    spaced closed form (the last case overflows the suspect list: slow path)"""
    rng = np.random.default_rng(n + L)
    alpha = np.frombuffer(b"ACGTacgtUuNnRYKM.-*", dtype=np.uint8)
    arr = alpha[rng.integers(0, 4, size=(n, L))]
    if p_bad:
        arr = np.where(rng.random((n, L)) < p_bad, alpha[rng.integers(4, len(alpha), size=(n, L))], arr).astype(np.uint8)
    t = torch.from_numpy(tile_array(arr)).cuda()
    with nt.Engine([12], gap=2, r_bits=18, s_bits=s_bits, flags=nt.FLAG_REQUIRE_TILED) as e:
        e.submit_tiled_device(t.data_ptr(), n, L)
        tc, ph, f1 = e.finish(counters=True)
    counters = np.zeros((1, 2, 1 << 18), dtype=np.uint16)
    offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    of1 = orc.sketch_update(counters, np.ascontiguousarray(arr).reshape(-1), offs, [12], 2, 18, s_bits)
    assert np.array_equal(f1, of1), (f1, of1)
    assert np.array_equal(tc, counters)


def _ragged_reads(rng, n, lo, hi, p_bad=0.0):
    alpha = np.frombuffer(b"ACGTacgtUuNnRYKM.-*", dtype=np.uint8)
    lens = rng.integers(lo, hi + 1, size=n)
    out = []
    for ln in lens:
        row = alpha[rng.integers(0, 4, size=ln)]
        if p_bad:
            row = np.where(rng.random(ln) < p_bad, alpha[rng.integers(4, len(alpha), size=ln)], row).astype(np.uint8)
        out.append(row.tobytes())
    return out


@pytest.mark.parametrize("n,C,k,p_bad,s_bits", [(5000, 10, 32, 0.002, 7), (2048, 10, 32, 0.0, 7), (70000, 10, 32, 0.001, 7), (3000, 3, 32, 0.01, 7), (4100, 5, 17, 0.0, 7),
                                               (2500, 2, 12, 0.02, 7), (2048, 2, 20, 0.01, 8), (9000, 7, 25, 0.003, 11), (300000, 10, 32, 0.0005, 7)])
def test_tiled_ragged_batches(nt, n, C, k, p_bad, s_bits):
    """ntc_submit_tiled_ragged_device: reads of 16 C - 15 .. 16 C bases, tiles sorted longest first, tails[tile][16] — K1h masks the windows behind
    every read's end, K1f knows every read's own length (ntRead takes any string, ntcard.cpp:173-189)"""
    rng = np.random.default_rng(n + C + k)
    reads = _ragged_reads(rng, n, 16 * C - 15, 16 * C, p_bad)
    tiles, tails, order = nt.tile_reads_ragged(reads, C)
    dt, dl = torch.from_numpy(tiles).cuda(), torch.from_numpy(tails.reshape(-1).astype(np.int32)).cuda()
    for flags in (0, nt.FLAG_DEFER_REDO):
        with nt.Engine([k], r_bits=18, s_bits=s_bits, flags=flags | nt.FLAG_REQUIRE_TILED) as e:
            e.submit_tiled_ragged_device(dt.data_ptr(), n, C, dl.data_ptr())
            e.submit_tiled_ragged_device(dt.data_ptr(), min(n, 2048), C, dl.data_ptr())  # a prefix of a ragged batch is a ragged batch
            tc, ph, f1 = e.finish(counters=True)
        sub = [reads[i] for i in order[:min(n, 2048)]]
        oc, of1 = orc.sketch_reads(reads + sub, [k], 0, 18, s_bits)
        assert np.array_equal(f1, of1), (f1, of1)
        assert np.array_equal(tc, oc)


@pytest.mark.parametrize("klist,s_bits,sizes", [([32], 7, [(10, 9000), (9, 5000), (8, 2048), (7, 100)]), ([20, 32], 8, [(10, 3000), (2, 4000), (1, 2500)]),
                                               ([25], 7, [(c, 1500 + 37 * c) for c in range(2, 12)])])
def test_tiled_bins_in_one_launch(nt, klist, s_bits, sizes):
    """ntc_submit_tiled_bins_device: several length bins — ragged ones and an equal-length one — share ONE K1h launch per k (K1hMulti: every bin its own
    arguments, bit arrays, suspect list and a share of the workgroups); bins shorter than a k of the list sit that k out; more bins than a launch takes
    go in groups.  Counters and F1 equal the oracle's over all reads, with and without deferred fix-ups"""
    rng = np.random.default_rng(len(sizes) + klist[0])
    bins, all_reads = [], []
    for C, n in sizes:
        reads = _ragged_reads(rng, n, 16 * C - 15, 16 * C, 0.004)
        tiles, tails, order = nt.tile_reads_ragged(reads, C)
        bins.append((torch.from_numpy(tiles).cuda(), n, 16 * C, torch.from_numpy(tails.reshape(-1).astype(np.int32)).cuda()))
        all_reads += reads
    uni = _ragged_reads(rng, 3333, 45, 45, 0.01)  # an equal-length bin (no tails) in the same call
    bins.append((torch.from_numpy(nt.tile_reads(uni, 45)).cuda(), len(uni), 45, None))
    all_reads += uni
    oc, of1 = orc.sketch_reads(all_reads, klist, 0, 18, s_bits)
    for flags in (0, nt.FLAG_DEFER_REDO):
        with nt.Engine(klist, r_bits=18, s_bits=s_bits, flags=flags | nt.FLAG_REQUIRE_TILED) as e:
            e.submit_tiled_bins_device([(t.data_ptr(), n, L, (d.data_ptr() if d is not None else 0)) for t, n, L, d in bins])
            tc, ph, f1 = e.finish(counters=True)
        assert np.array_equal(f1, of1), (f1, of1)
        assert np.array_equal(tc, oc)


@pytest.mark.parametrize("klist", [[32, 64], [16, 24, 32, 48], [40, 20]])
def test_ragged_tiles_under_a_mixed_k_list(nt, monkeypatch, klist):
    """round 6: ragged tiled batches under a k list of which only a part is K1h's — K1h + K1f take their k from the tiles, K1 stages the SAME tiles and learns
    every read's length from a slot table derived from the tiles' prefix tables (tails): length bins through ntc_submit_tiled_bins_device and
    ntc_submit_tiled_ragged_device, reads shorter than the larger k, and the same reads through ntc_submit (host packer: bins on tiles, the rest in row slots)"""
    rng = np.random.default_rng(sum(klist))
    bins, all_reads = [], []
    for C, n in ((10, 5000), (9, 3000), (4, 2500), (2, 2100)):
        reads = _ragged_reads(rng, n, 16 * C - 15, 16 * C, 0.004)
        tiles, tails, order = nt.tile_reads_ragged(reads, C)
        bins.append((torch.from_numpy(tiles).cuda(), n, 16 * C, torch.from_numpy(tails.reshape(-1).astype(np.int32)).cuda()))
        all_reads += reads
    oc, of1 = orc.sketch_reads(all_reads, klist, 0, 18, 7)
    for flags in (0, nt.FLAG_DEFER_REDO, nt.FLAG_ALWAYS_LOG):
        with nt.Engine(klist, r_bits=18, s_bits=7, flags=flags) as e:
            e.submit_tiled_bins_device([(t.data_ptr(), n, L, d.data_ptr()) for t, n, L, d in bins[:2]])
            for t, n, L, d in bins[2:]:
                e.submit_tiled_ragged_device(t.data_ptr(), n, L // 16, d.data_ptr())
            tc, ph, f1 = e.finish(counters=True)
        assert np.array_equal(f1, of1), (flags, f1, of1)
        assert np.array_equal(tc, oc), flags
    monkeypatch.setenv("NTC_BIN_MIN", "1024")  # (the host packer's bar for a bin: 32 Ki reads by default)
    mixed = [all_reads[i] for i in rng.permutation(len(all_reads))]
    with nt.Engine(klist, r_bits=18, s_bits=7) as e:
        e.submit_reads(mixed)
        tc, ph, f1 = e.finish(counters=True)
    assert np.array_equal(f1, of1) and np.array_equal(tc, oc)


@pytest.mark.parametrize("cap", ["64", "7"])
def test_tiled_bins_with_overflowing_suspect_lists(nt, monkeypatch, cap):
    """several length bins in one launch, reads dense with N, suspect lists far too short (NTC_K1H_SUS_CAP): every bin's K1f finds ITS regions overflowed and
    takes the slow path; twice per engine, with and without deferred fix-ups"""
    monkeypatch.setenv("NTC_K1H_SUS_CAP", cap)
    rng = np.random.default_rng(int(cap))
    bins, allr = [], []
    for C, n in ((10, 9000), (9, 4000), (6, 7000), (3, 3000)):
        reads = _ragged_reads(rng, n, 16 * C - 15, 16 * C, 0.02)
        tiles, tails, _ = nt.tile_reads_ragged(reads, C)
        bins.append((torch.from_numpy(tiles).cuda(), n, 16 * C, torch.from_numpy(tails.reshape(-1).astype(np.int32)).cuda()))
        allr += reads
    oc, of1 = orc.sketch_reads(allr + allr, [25], 0, 18, 7)
    for flags in (0, nt.FLAG_DEFER_REDO):
        with nt.Engine([25], r_bits=18, s_bits=7, flags=flags | nt.FLAG_REQUIRE_TILED) as e:
            for _ in range(2):
                e.submit_tiled_bins_device([(t.data_ptr(), n, L, d.data_ptr()) for t, n, L, d in bins])
            tc, ph, f1 = e.finish(counters=True)
        assert np.array_equal(f1, of1) and np.array_equal(tc, oc)


def test_host_batches_of_mixed_lengths_one_after_the_other(nt):
    """several ntc_submit calls on one engine, each a multi-bin launch of a different size and bin composition: a bin's suspect list is indexed by the wave's
    number in the launch, so regions that THIS launch's waves of the bin did not write still hold the counts of an earlier launch — K1f must visit only
    the regions of the workgroups that walked the bin (round 5: found by the CLI on trimmed FASTQ, 0.01 % of the counters off)"""
    rng = np.random.default_rng(11)
    batches = [_ragged_reads(rng, n, 40, 95, 0.003) for n in (9000, 7000, 12000, 5000, 9000)]
    done = []
    for flags in (nt.FLAG_REQUIRE_TILED, nt.FLAG_REQUIRE_TILED | nt.FLAG_DEFER_REDO):
        with nt.Engine([25], r_bits=18, s_bits=7, flags=flags) as e:
            done = []
            for b in batches:
                e.submit_reads(b)
                done += b
                tc, ph, f1 = e.finish(counters=True)
                oc, of1 = orc.sketch_reads(done, [25], 0, 18, 7)
                assert np.array_equal(f1, of1) and np.array_equal(tc, oc), (len(done), int((tc != oc).sum()))


def test_host_submit_of_mixed_lengths_takes_ragged_tiles(nt):
    """ntc_submit over adapter-trimmed-like reads (100 .. 150 bp, a few shorter than k, one long sequence): binned by ceil(len / 16), bins of >= 1024 reads
    as ragged tiles, the rest in row slots — counters and F1 equal the oracle's"""
    rng = np.random.default_rng(5)
    reads = _ragged_reads(rng, 60000, 100, 150, 0.002) + _ragged_reads(rng, 300, 20, 40, 0.0) + _ragged_reads(rng, 50, 200, 230, 0.01) + _ragged_reads(rng, 2, 5000, 9000, 0.001)
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    oc, of1 = orc.sketch_reads(reads, [32], 0, 18, 7)
    for flags in (nt.FLAG_REQUIRE_TILED, 0):  # (the validation flag lowers the size a bin needs for the tiled kernels from 512 Ki reads to 1024)
        with nt.Engine([32], r_bits=18, s_bits=7, flags=flags) as e:
            e.submit_reads(reads)
            tc, ph, f1 = e.finish(counters=True)
        assert np.array_equal(f1, of1) and np.array_equal(tc, oc)
