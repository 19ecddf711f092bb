"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs (bit-exact: integer/byte work), against the committed golden fixtures produced by the
real reference, and — at sizes the oracle cannot reach quickly — through size-independent properties.
"""
import gzip
import json
import os
import random
import threading

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def nt():
    assert torch.cuda.is_available(), "GPU tests need a HIP device (run on the MI355X box)"
    import ntcard_amd
    return ntcard_amd


def to_slots(reads, stride=None):
    L = max((len(r) for r in reads), default=0)
    stride = stride or max(4, (L + 3) & ~3)
    buf = np.full(len(reads) * stride + 16, 10, dtype=np.uint8)
    for i, r in enumerate(reads):
        buf[i * stride: i * stride + len(r)] = np.frombuffer(r, dtype=np.uint8)
    return buf, stride


def rseq(rng, n, pn=0.02, plow=0.1, bad="NnRYKMSWBDHV-.*Xx\r"):
    out = []
    for _ in range(n):
        r = rng.random()
        if r < pn:
            out.append(rng.choice(bad))
        elif r < pn + plow:
            out.append(rng.choice("acgtu"))
        else:
            out.append(rng.choice("ACGTU"))
    return "".join(out).encode()


def small_reads(golden_dir):
    with gzip.open(os.path.join(golden_dir, "reads_small.fq.gz"), "rb") as f:
        lines = f.read().split(b"\n")
    return [lines[i] for i in range(1, len(lines) - 1, 4)]


# ------------------------------------------------------------------------------------------------
def test_generator_matches_oracle(nt):
    for dist, glen in ((0, 0), (1, 100_000), (1, 100_000_000)):
        n, L, stride = 3000, 150, 152
        d = torch.empty(n * stride, dtype=torch.uint8, device="cuda")
        nt.gen_reads_device(d.data_ptr(), 42, 1_000_000, n, L, stride, dist, genome_len=max(glen, L))
        torch.cuda.synchronize()
        ref = orc.gen_reads(42, 1_000_000, n, L, stride, dist, genome_len=max(glen, L))
        assert np.array_equal(d.cpu().numpy(), ref)


@pytest.mark.parametrize("L", [150, 151, 70, 33, 253])
def test_hash_dump_matches_oracle(nt, L):
    """K1d: every canonical hash of every clean window, vs ntHashIterator semantics in the oracle"""
    rng = random.Random(1000 + L)
    reads = [rseq(rng, L, pn=rng.choice([0.0, 0.0, 0.01, 0.05, 0.3])) for _ in range(333)]
    reads[0] = b"N" * L
    reads[1] = (b"ACGT" * 100)[:L]
    reads[2] = (b"acgu" * 100)[:L]
    buf, stride = to_slots(reads)
    d = torch.from_numpy(buf).cuda()
    for k in (1, 2, 3, 4, 5, 12, 20, 31, 32, 33, 34, 35, 64, 70, 96, 128, 200):
        maxw = max(L - k + 1, 1)
        dh = torch.zeros(len(reads) * maxw, dtype=torch.int64, device="cuda")
        dc = torch.full((len(reads),), -1, dtype=torch.int32, device="cuda")
        nt.hash_dump_device(d.data_ptr(), len(reads), L, stride, k, 0, maxw, dh.data_ptr(), dc.data_ptr())
        torch.cuda.synchronize()
        hh = dh.cpu().numpy().view(np.uint64).reshape(len(reads), maxw)
        cc = dc.cpu().numpy()
        for i, r in enumerate(reads):
            oh, _ = orc.hash_read(r, k)
            assert cc[i] == len(oh), (L, k, i, cc[i], len(oh))
            assert np.array_equal(hh[i, : len(oh)], oh), (L, k, i)


def test_golden_hash_vectors_through_production_kernel(nt, golden_dir):
    """hash_vectors.json holds what the REFERENCE's ntHashIterator / stHashIterator enumerate (ntHashIterator.hpp:59-86,
    stHashIterator.hpp:60-87, nthash.hpp:641-646; made by tools/make_golden.py with oracle/_ref/ref_tool) for 16
    sequences of length 0..1000.  The validation build of the production kernel K1 (every window passes the filter,
    so every value comes out of its closed-form resolve stage) must reproduce every 64-bit value, spaced seeds
    included; the simple kernel K1d is checked on the plain vectors as well."""
    with open(os.path.join(golden_dir, "hash_vectors.json")) as f:
        vec = json.load(f)
    seqs = [s.encode() for s in vec["seqs"]]

    def dump(seq, k, gap, k1):
        L = len(seq)
        buf, stride = to_slots([seq], stride=max(4, (L + 3) & ~3))
        buf[buf == 10] = ord("A")
        d = torch.from_numpy(buf).cuda()
        maxw = max(L - k + 1, 1)
        dh = torch.zeros(maxw, dtype=torch.int64, device="cuda")
        dc = torch.full((1,), -1, dtype=torch.int32, device="cuda")
        nt.hash_dump_device(d.data_ptr(), 1, L, stride, k, gap, maxw, dh.data_ptr(), dc.data_ptr(), k1=k1)
        torch.cuda.synchronize()
        n = int(dc[0])
        return ["%016x" % int(x) for x in dh.cpu().numpy().view(np.uint64)[:n]]

    for ent in vec["sthash"]:
        for s, hs in zip(seqs, ent["hash"]):
            assert dump(s, ent["k"], ent["gap"], True) == hs, (ent["k"], ent["gap"], len(s))
            assert dump(s, ent["k"], ent["gap"], False) == hs  # the plain entry point forwards spaced seeds to K1
    for ent in vec["nthash"]:
        for s, hs in zip(seqs, ent["hash"]):
            assert dump(s, ent["k"], 0, True) == hs, (ent["k"], len(s))
            if len(s) <= 600:  # the simple kernel parks 256 slots per block in LDS: 640-byte slots at most
                assert dump(s, ent["k"], 0, False) == hs, (ent["k"], len(s))


def test_known_answer_on_device(nt):
    # vendor/ntHash/unittest/UnitTests.cpp:39,45 — the reference's own invariant value
    buf, stride = to_slots([b"ACGTACACTGGACTGAGTCT"])
    d = torch.from_numpy(buf).cuda()
    dh = torch.zeros(1, dtype=torch.int64, device="cuda")
    dc = torch.zeros(1, dtype=torch.int32, device="cuda")
    nt.hash_dump_device(d.data_ptr(), 1, 20, stride, 20, 0, 1, dh.data_ptr(), dc.data_ptr())
    torch.cuda.synchronize()
    assert int(dc[0]) == 1
    assert int(dh.cpu().numpy().view(np.uint64)[0]) == 10434435546371013747


@pytest.mark.parametrize("dist,klist,r_bits,s_bits", [
    (1, [32], 20, 7), (0, [32], 20, 7), (1, [32], 22, 11), (1, [12], 18, 7), (1, [33], 20, 7),
    (1, [31], 19, 5), (1, [16, 24, 32, 48], 19, 7), (1, [32, 64, 96, 128], 20, 7), (0, [150], 16, 2),
    (1, [12, 20, 31, 33, 47, 64], 18, 7),   # more than one fused launch group (4 + 2)
    (1, [100, 120, 140, 150], 16, 7),       # tables too big to fuse all four: the group is split
])
def test_sketch_device_batch_matches_oracle(nt, dist, klist, r_bits, s_bits):
    n, L, stride = 20_000, 150, 152
    d = torch.empty(n * stride + 16, dtype=torch.uint8, device="cuda")
    nt.gen_reads_device(d.data_ptr(), 9, 0, n, L, stride, dist, genome_len=300_000)
    torch.cuda.synchronize()
    host = d[: n * stride].cpu().numpy().reshape(n, stride)
    reads = [host[i, :L].tobytes() for i in range(n)]
    with nt.Engine(klist, r_bits=r_bits, s_bits=s_bits) as e:
        e.submit_device(d.data_ptr(), n, L, stride)
        tc, ph, f1 = e.finish(counters=True)
    oc, of1 = orc.sketch_reads(reads, klist, 0, r_bits, s_bits)
    assert np.array_equal(f1, of1)
    assert np.array_equal(tc, oc)
    for ki in range(len(klist)):
        assert np.array_equal(ph[ki], orc.value_hist(oc[ki], r_bits))


def test_simple_and_tuned_kernels_agree(nt):
    """independently written kernels (ntc_kernels.hip vs ntc_sketch_hf.hip) and both forms of the sketch update
    (hit log + partitioned apply vs one device atomic per sampled k-mer) on dirty, ragged input"""
    rng = random.Random(4242)
    reads = [rseq(rng, rng.choice([40, 100, 149, 150, 150, 150, 151, 200]), pn=rng.choice([0, 0, 0.003, 0.05])) for _ in range(6000)]
    for klist, sb in (([32], 7), ([25, 61], 3)):
        res = []
        for flags in (0, nt.FLAG_SIMPLE_KERNEL, nt.FLAG_DIRECT_ATOMICS):
            with nt.Engine(klist, r_bits=19, s_bits=sb, flags=flags) as e:
                e.submit_reads(reads)
                res.append(e.finish(counters=True))
        for other in res[1:]:
            assert np.array_equal(res[0][2], other[2]) and np.array_equal(res[0][0], other[0])
        oc, of1 = orc.sketch_reads(reads, klist, 0, 19, sb)
        assert np.array_equal(res[0][2], of1) and np.array_equal(res[0][0], oc)


def test_host_submit_ragged_and_chunked(nt):
    """ntc_submit: raw host buffers, reads of any length (empty, shorter than k, long sequences that
    the shim splits into overlapping chunks), dirty bytes everywhere"""
    rng = random.Random(77)
    reads = [rseq(rng, rng.choice([0, 1, 5, 31, 32, 33, 100, 150, 151, 250]), pn=rng.choice([0, 0.01, 0.1])) for _ in range(3000)]
    reads += [rseq(rng, n, pn=0.001) for n in (257, 300, 1000, 5000, 20_000, 70_000)]
    rng.shuffle(reads)
    for klist in ([32], [12, 33, 64], [128]):
        with nt.Engine(klist, r_bits=18, s_bits=4) as e:
            e.submit_reads(reads)
            tc, ph, f1 = e.finish(counters=True)
        oc, of1 = orc.sketch_reads(reads, klist, 0, 18, 4)
        assert np.array_equal(f1, of1), (klist, f1, of1)
        assert np.array_equal(tc, oc), klist
    # short-read-only batch (uniform, non-chunked path) and an all-too-short batch
    shorts = [rseq(rng, 75, pn=0.01) for _ in range(1000)]
    with nt.Engine([20], r_bits=16, s_bits=3) as e:
        e.submit_reads(shorts)
        e.submit_reads([b"ACGT", b"", b"ACGTACGT"])
        tc, ph, f1 = e.finish(counters=True)
    oc, of1 = orc.sketch_reads(shorts, [20], 0, 16, 3)
    assert np.array_equal(f1, of1) and np.array_equal(tc, oc)


@pytest.mark.parametrize("r_bits,klist,log_entries", [
    (27, [32], 0),            # default geometry: two partition passes (8 + 5 bits) above 2^15-counter slices
    (27, [32], 1 << 16),      # log much smaller than the batch: several applies + per-wave overflow to direct atomics
    (20, [32, 40, 50], 0),    # one partition pass, key space not a power of two (3 planes)
    (14, [20], 0),            # the whole sketch is one slice: no partition pass
    (12, [15, 16, 17, 18, 19], 1 << 14),
    (30, [32], 1 << 22),      # rBits = 30: two 8-bit passes
])
def test_hit_log_geometries_match_direct_atomics(nt, r_bits, klist, log_entries):
    """ntComp's increment is deferred (hit log -> partition -> LDS histogram, ntc_apply.hip); every log geometry must
    give exactly the counters of the literal one-atomic-per-hit form (ntcard.cpp:142-143), across several submits"""
    n, L, stride = 60_000, 150, 152
    d = torch.empty(n * stride + 16, dtype=torch.uint8, device="cuda")
    nt.gen_reads_device(d.data_ptr(), 77, 0, n, L, stride, 1, genome_len=200_000)
    res = []
    for flags, le in ((nt.FLAG_DIRECT_ATOMICS, 0), (nt.FLAG_ALWAYS_LOG, log_entries), (0, log_entries),
                      (nt.FLAG_ALWAYS_LOG | nt.FLAG_PARTITION_ALWAYS, log_entries)):  # the last one runs A1/A2/A3 on these small logs
        with nt.Engine(klist, r_bits=r_bits, s_bits=5, flags=flags, log_entries=le) as e:
            e.submit_device(d.data_ptr(), 20_032, L, stride)
            e.submit_device(d.data_ptr() + 20_032 * stride, n - 20_032, L, stride)
            e.flush()
            e.submit_device(d.data_ptr(), 6400, L, stride)
            _, ph, f1 = e.finish()
            sk, ncnt, _ = e.device_state()
            torch.cuda.synchronize()
            raw = torch.as_tensor(_DevArray(sk, ncnt), device="cuda").clone()
            res.append((ph.copy(), f1.copy(), raw))
    for other in res[1:]:
        assert np.array_equal(res[0][1], other[1])
        assert np.array_equal(res[0][0], other[0])
        assert torch.equal(res[0][2], other[2])   # the uint32 counters themselves (before the uint16 wrap)
    assert int(res[0][2].sum(dtype=torch.int64)) > 0


def test_deferred_pass_over_several_buffers(nt):
    """several device batches in different buffers, different tails, a change of slot geometry, a reset in between, no ntc_sync in
    between -> counters and F1 equal the oracle's over everything submitted after the reset (written for round 2's K1b, whose
    deferred pass kept slot ADDRESSES across batches; kept as a test of ntc_submit_device + ntc_reset)"""
    rng = random.Random(99)

    def batch(n, L, stride, pn):
        reads = [rseq(rng, L, pn=pn, plow=0.05) for _ in range(n)]
        buf, _ = to_slots(reads, stride=stride)
        buf[buf == 10] = ord("A")
        return reads, torch.from_numpy(buf).cuda()
    dropped = batch(2048 * 2 + 5, 150, 152, 0.01)
    batches = [batch(2048 * 2 + 300, 150, 152, 0.002), batch(2048 + 1, 150, 152, 0.02), batch(2048 * 3, 150, 152, 0.0),
               batch(2048 * 2 + 77, 148, 156, 0.004), batch(2048 + 900, 150, 152, 0.001)]
    with nt.Engine([32], r_bits=20, s_bits=7, flags=nt.FLAG_ALWAYS_LOG | nt.FLAG_DEFER_REDO) as e:
        e.submit_device(dropped[1].data_ptr(), len(dropped[0]), 150, 152)
        e.reset()
        for reads, d in batches:
            L = len(reads[0])
            e.submit_device(d.data_ptr(), len(reads), L, 156 if L == 148 else 152)
        tc, ph, f1 = e.finish(counters=True)
    allreads = [r for reads, _ in batches for r in reads]
    oc, of1 = orc.sketch_reads(allreads, [32], 0, 20, 7)
    assert np.array_equal(f1, of1)
    assert np.array_equal(tc, oc)


def test_submit_device_buffer_may_be_reused_in_stream_order(nt):
    """ntc_submit_device's contract: the buffer belongs to the caller again as soon as the stream has passed the call, so
    overwriting the buffer right behind the submit (same stream) must not change a counter.  (Rounds 2-4 had a kernel, K1b, that
    handed reads to a later pass by address; the test stays as the contract's.)  With NTC_FLAG_DEFER_REDO the caller promises to
    keep the buffer."""
    rng = random.Random(7)
    reads = [rseq(rng, 150, pn=0.01, plow=0.05) for _ in range(2048 * 2 + 333)]
    buf, _ = to_slots(reads, stride=152)
    buf[buf == 10] = ord("A")
    oc, of1 = orc.sketch_reads(reads, [32], 0, 20, 7)
    for flags in (nt.FLAG_ALWAYS_LOG, 0):
        d = torch.from_numpy(buf).cuda()
        with nt.Engine([32], r_bits=20, s_bits=7, flags=flags) as e:  # engine and torch both use the null stream here
            e.submit_device(d.data_ptr(), len(reads), 150, 152)
            d.fill_(ord("C"))
            tc, ph, f1 = e.finish(counters=True)
        assert np.array_equal(f1, of1) and np.array_equal(tc, oc)
    d = torch.from_numpy(buf).cuda()
    with nt.Engine([32], r_bits=20, s_bits=7, flags=nt.FLAG_ALWAYS_LOG | nt.FLAG_DEFER_REDO) as e:
        e.submit_device(d.data_ptr(), len(reads), 150, 152)
        e.submit_device(d.data_ptr(), 2048, 150, 152)
        tc, ph, f1 = e.finish(counters=True)
    oc2, of2 = orc.sketch_reads(reads + reads[:2048], [32], 0, 20, 7)
    assert np.array_equal(f1, of2) and np.array_equal(tc, oc2)


def test_more_than_2_32_counters_do_not_alias(nt):
    """3 values of k at rBits = 30 are 3 x 2^31 counters: beyond 32-bit hit-log keys, so the engine increments directly — through
    every k's own plane pointer.  Each k's value histogram must equal that of a single-k engine."""
    n, L, stride = 30_000, 150, 152
    d = torch.empty(n * stride + 16, dtype=torch.uint8, device="cuda")
    nt.gen_reads_device(d.data_ptr(), 3, 0, n, L, stride, 1, genome_len=100_000)
    klist = [20, 24, 28]
    with nt.Engine(klist, r_bits=30, s_bits=4) as e:
        e.submit_device(d.data_ptr(), n, L, stride)
        _, ph, f1 = e.finish()
    for i, k in enumerate(klist):
        with nt.Engine([k], r_bits=30, s_bits=4) as e:
            e.submit_device(d.data_ptr(), n, L, stride)
            _, ph1, f11 = e.finish()
        assert int(f1[i]) == int(f11[0]) and np.array_equal(ph[i], ph1[0]), k


def test_partition_runs_longer_than_one_count_pass(nt):
    """a large log over a small sketch: a (workgroup, digit) run of the partition holds more than 65535 keys, which the 16-bit
    LDS histogram of the count pass has to take in pieces (it used to spin forever)"""
    n, L, stride = 3_000_000, 150, 152
    d = torch.empty(n * stride + 16, dtype=torch.uint8, device="cuda")
    nt.gen_reads_device(d.data_ptr(), 11, 0, n, L, stride, 0)
    res = []
    for flags, le in ((nt.FLAG_DIRECT_ATOMICS, 0), (nt.FLAG_ALWAYS_LOG | nt.FLAG_PARTITION_ALWAYS | nt.FLAG_LANE_KERNEL, 1 << 28)):
        with nt.Engine([32], r_bits=17, s_bits=2, flags=flags, log_entries=le) as e:
            e.submit_device(d.data_ptr(), n, L, stride)
            _, ph, f1 = e.finish()
            res.append((ph.copy(), f1.copy()))
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][0], res[1][0])


@pytest.mark.parametrize("s_bits,dist,want_mode", [(7, 1, 1), (7, 0, 0), (11, 1, 1), (11, 0, 0), (9, 1, 1)])
def test_adaptive_mode_probes_once(nt, s_bits, dist, want_mode):
    """lane-per-read engines decide the update mode ONCE from a sample of the first ~2^20 logged entries: repeat-rich
    reads (dist g) -> direct atomics, uniform reads -> hit log; a batch is cut into head + rest at most once, whatever
    sBits is (regression: at sBits = 11 the head never logged enough for the probe and EVERY batch was cut into
    0.65 M-slot pieces)."""
    n, L, stride = 6_000_000, 150, 152
    d = torch.empty(n * stride + 16, dtype=torch.uint8, device="cuda")
    nt.gen_reads_device(d.data_ptr(), 5, 0, n, L, stride, dist, genome_len=3_000_000)
    with nt.Engine([32], r_bits=27, s_bits=s_bits, flags=nt.FLAG_LANE_KERNEL) as e:
        e.set_profiling(True)
        for _ in range(4):
            e.submit_device(d.data_ptr(), n, L, stride)
        e.flush()
        _, launches = e.kernel_time()
        assert e.update_mode() == want_mode
        assert 4 <= launches <= 5, launches


@pytest.mark.parametrize("L,stride,s_bits,pn", [
    (150, 152, 7, 0.0005), (150, 152, 11, 0.0), (150, 160, 7, 0.01), (128, 128, 7, 0.0), (156, 156, 8, 0.002),
    (32, 128, 7, 0.0), (33, 132, 7, 0.001), (95, 140, 7, 0.0), (96, 144, 9, 0.0), (97, 148, 7, 0.0005), (159, 160, 7, 0.0),
])
def test_equal_length_row_slots_match_oracle(nt, L, stride, s_bits, pn):
    """equal-length k = 32 batches in row slots (K1 with the log forced; the shapes are those of round 2's bit-sliced row-slot kernel K1b:
    window counts around 16-window blocks, slot strides 128..160, lower case / U, all-N reads)"""
    rng = random.Random(L * 31 + stride)
    n = 2048 * 3 + 777
    reads = [rseq(rng, L, pn=pn, plow=0.05) for _ in range(n)]
    reads[5] = b"N" * L
    reads[2048 + 9] = (b"acgu" * 64)[:L]
    reads[4096] = (b"ACGT" * 64)[:L - 1] + b"n"
    buf, _ = to_slots(reads, stride=stride)
    buf[buf == 10] = ord("A")  # padding bytes are base letters (ntc_submit_device's contract for the fast path)
    d = torch.from_numpy(buf).cuda()
    with nt.Engine([32], r_bits=20, s_bits=s_bits, flags=nt.FLAG_ALWAYS_LOG) as e:
        e.submit_device(d.data_ptr(), n, L, stride)
        tc, ph, f1 = e.finish(counters=True)
    oc, of1 = orc.sketch_reads(reads, [32], 0, 20, s_bits)
    assert np.array_equal(f1, of1)
    assert np.array_equal(tc, oc)


def test_log_and_adaptive_modes_agree_at_size(nt):
    """2 M genome-like reads (1 % substitutions, 0.05 % N) in row slots: K1 with the hit log forced vs the adaptive engine, counters compared on the device"""
    n, L, stride = 2_000_000, 150, 152
    d = torch.empty(n * stride + 16, dtype=torch.uint8, device="cuda")
    nt.gen_reads_device(d.data_ptr(), 5, 0, n, L, stride, 1, genome_len=3_000_000)
    res = []
    for flags in (nt.FLAG_ALWAYS_LOG, 0):
        with nt.Engine([32], r_bits=24, s_bits=7, flags=flags) as e:
            e.submit_device(d.data_ptr(), n, L, stride)
            _, ph, f1 = e.finish()
            sk, ncnt, _ = e.device_state()
            torch.cuda.synchronize()
            res.append((ph.copy(), f1.copy(), torch.as_tensor(_DevArray(sk, ncnt), device="cuda").clone()))
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][0], res[1][0])
    assert torch.equal(res[0][2], res[1][2])


class _DevArray:
    """zero-copy view of a device int32 array for torch.as_tensor (__cuda_array_interface__)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i4", "data": (int(ptr), False), "version": 2}


def test_golden_hist_from_reference(nt, golden_dir, tmp_path):
    """the reference CLI's own .hist outputs (default rBits = 27) reproduced byte for byte"""
    reads = small_reads(golden_dir)
    cases = [("ref_k12__out_k12.hist", [12], 27, 7, 1000), ("ref_k32__out_k32.hist", [32], 27, 7, 1000),
             ("ref_k20_c50__out_k20.hist", [20], 27, 7, 50), ("ref_k24_s11_r22__out_k24.hist", [24], 22, 7, 1000)]
    for name, klist, rb, sb, cov in cases:
        with nt.Engine(klist, r_bits=rb, s_bits=sb) as e:
            e.submit_reads(reads)
            _, ph, f1 = e.finish()
        F0, f = nt.estimate(ph[0], rb, sb, cov)
        out = tmp_path / name
        nt.write_hist(out, f1[0], F0, f, cov)
        assert out.read_bytes() == open(os.path.join(golden_dir, name), "rb").read(), name
    with nt.Engine([16, 24, 32, 48], r_bits=27, s_bits=7) as e:
        e.submit_reads(reads)
        _, ph, f1 = e.finish()
    for ki, k in enumerate([16, 24, 32, 48]):
        F0, f = nt.estimate(ph[ki], 27, 7, 1000)
        out = tmp_path / f"m_k{k}.hist"
        nt.write_hist(out, f1[ki], F0, f, 1000)
        assert out.read_bytes() == open(os.path.join(golden_dir, f"ref_multi__out_k{k}.hist"), "rb").read(), k


def test_golden_sketch_digests(nt, golden_dir):
    reads = small_reads(golden_dir)
    with open(os.path.join(golden_dir, "sketch_goldens.json")) as f:
        gold = json.load(f)
    for ent in gold:
        if ent["r_bits"] > 22:
            continue
        with nt.Engine(ent["klist"], gap=ent["gap"], r_bits=ent["r_bits"], s_bits=ent["s_bits"]) as e:
            e.submit_reads(reads)
            tc, ph, f1 = e.finish(counters=True)
        assert [int(x) for x in f1] == ent["f1"]
        for ki, pl in enumerate(ent["planes"]):
            assert "%016x" % orc.fnv1a64(tc[ki]) == pl["fnv1a64"]
            nz = [[int(s), int(v), int(ph[ki][s, v])] for s in range(2) for v in np.nonzero(ph[ki][s])[0]]
            assert nz == pl["p_nonzero"]


@pytest.mark.parametrize("k,gap", [(12, 2), (12, 4), (13, 3), (20, 8), (32, 8), (33, 1), (64, 10), (31, 29)])
def test_gap_seeds_match_oracle(nt, k, gap):
    """stRead / stHashIterator path (ntcard.cpp:160-171): spaced seed 1^((k-g)/2) 0^g 1^((k-g)/2)"""
    rng = random.Random(k * 100 + gap)
    reads = [rseq(rng, rng.choice([k - 1, k, k + 1, 70, 150, 151]), pn=rng.choice([0, 0, 0.01, 0.05])) for _ in range(4000)]
    reads += [rseq(rng, n, pn=0.002) for n in (300, 2000)]
    with nt.Engine([k], gap=gap, r_bits=18, s_bits=5) as e:
        e.submit_reads(reads)
        tc, ph, f1 = e.finish(counters=True)
    oc, of1 = orc.sketch_reads(reads, [k], gap, 18, 5)
    assert np.array_equal(f1, of1)
    assert np.array_equal(tc, oc)


@pytest.mark.parametrize("L,k,gap", [
    (150, 12, 2), (150, 13, 3), (150, 20, 8), (150, 32, 8), (151, 33, 1), (149, 31, 29), (150, 64, 10), (70, 12, 4),
    (250, 32, 0), (1000, 32, 0), (1000, 33, 5), (32, 32, 0), (33, 32, 0), (35, 32, 0), (36, 32, 0), (37, 33, 0),
    (31, 32, 0), (40, 12, 2), (13, 12, 2), (12, 12, 2), (163, 150, 0), (64, 61, 0), (68, 64, 0), (300, 200, 0),
])
def test_equal_length_batches_match_oracle(nt, L, k, gap):
    """equal-length waves take the closed-form start and (with -g) the rolling spaced-seed step; shapes around the
    group / block boundaries, staging without register prefetch (long slots), reads shorter than k"""
    rng = random.Random(L * 1000 + k * 7 + gap)
    reads = [rseq(rng, L, pn=rng.choice([0, 0, 0.003, 0.02])) for _ in range(3000)]
    with nt.Engine([k], gap=gap, r_bits=18, s_bits=5) as e:
        e.submit_reads(reads[:1000])
        e.submit_reads(reads[1000:])
        tc, ph, f1 = e.finish(counters=True)
    oc, of1 = orc.sketch_reads(reads, [k], gap, 18, 5)
    assert np.array_equal(f1, of1)
    assert np.array_equal(tc, oc)


def test_golden_gap_hist_from_reference(nt, golden_dir, tmp_path):
    """the reference CLI's `-k 12 -g 2` output (the shape of its own check-dna-gap target, Makefile.am:53-54,71-72)"""
    reads = small_reads(golden_dir)
    with nt.Engine([12], gap=2, r_bits=27, s_bits=7) as e:
        e.submit_reads(reads)
        _, ph, f1 = e.finish()
    F0, f = nt.estimate(ph[0], 27, 7, 1000)
    out = tmp_path / "gap_k12.hist"
    nt.write_hist(out, f1[0], F0, f, 1000)
    assert out.read_bytes() == open(os.path.join(golden_dir, "ref_k12_g2__out_k12.hist"), "rb").read()


def test_uint16_wraparound(nt):
    rng = random.Random(3)
    seq = "".join(rng.choice("ACGT") for _ in range(4000)).encode()
    h, pos = orc.hash_read(seq, 32)
    i = next(i for i, x in enumerate(h) if orc.lib().orc_sample_of(int(x), 7) < 2)
    km = seq[pos[i]: pos[i] + 32]
    with nt.Engine([32], r_bits=12, s_bits=7) as e:
        e.submit_reads([km] * 65537)
        tc, ph, f1 = e.finish(counters=True)
    assert int(f1[0]) == 65537 and int(tc.sum()) == 1  # 65536 increments wrap to 0, +1
    assert int(ph[0].sum()) == 2 << 12


def test_sketch_dump_and_merge(nt):
    """§8(f)-3: a dumped t_Counter image merged into another engine == one run over all reads (mod 2^16)"""
    rng = random.Random(5)
    reads = [rseq(rng, rng.choice([60, 100, 150]), pn=0.01) for _ in range(6000)]
    klist = [16, 33]
    with nt.Engine(klist, r_bits=16, s_bits=3) as a:
        a.submit_reads(reads[:2500])
        tc_a, _, f1_a = a.finish(counters=True)
    with nt.Engine(klist, r_bits=16, s_bits=3) as b:
        b.submit_reads(reads[2500:])
        b.merge_counters(tc_a, f1_a)
        b.submit_reads([])
        tc, ph, f1 = b.finish(counters=True)
    oc, of1 = orc.sketch_reads(reads, klist, 0, 16, 3)
    assert np.array_equal(f1, of1) and np.array_equal(tc, oc)
    for ki in range(len(klist)):
        assert np.array_equal(ph[ki], orc.value_hist(oc[ki], 16))


def test_batching_order_and_reset_invariance(nt):
    """results do not depend on batch boundaries or submit order (commutative atomics)"""
    n, L, stride = 50_000, 150, 152
    d = torch.empty(n * stride + 16, dtype=torch.uint8, device="cuda")
    nt.gen_reads_device(d.data_ptr(), 11, 0, n, L, stride, 1, genome_len=1_000_000)
    with nt.Engine([32], r_bits=22, s_bits=7) as e:
        e.submit_device(d.data_ptr(), n, L, stride)
        a_tc, a_ph, a_f1 = e.finish(counters=True)
        e.reset()
        cuts = [0, 64 * 3, 64 * 3 + 17 * 64, 30_016, n]  # device batches must start on 16-byte aligned slots
        order = [2, 0, 3, 1]
        for j in order:
            lo, hi = cuts[j], cuts[j + 1]
            e.submit_device(d.data_ptr() + lo * stride, hi - lo, L, stride)
        b_tc, b_ph, b_f1 = e.finish(counters=True)
    assert np.array_equal(a_f1, b_f1) and np.array_equal(a_tc, b_tc) and np.array_equal(a_ph, b_ph)


def test_concurrent_submit_is_thread_safe(nt):
    rng = random.Random(5)
    parts = [[rseq(rng, 150, pn=0.005) for _ in range(2000)] for _ in range(4)]
    with nt.Engine([32], r_bits=18, s_bits=5) as e:
        ts = [threading.Thread(target=e.submit_reads, args=(p,)) for p in parts]
        [t.start() for t in ts]
        [t.join() for t in ts]
        tc, ph, f1 = e.finish(counters=True)
    oc, of1 = orc.sketch_reads([r for p in parts for r in p], [32], 0, 18, 5)
    assert np.array_equal(f1, of1) and np.array_equal(tc, oc)


def test_full_size_properties(nt):
    """config-2 shape at a size the CPU cannot check quickly: 4 M x 150 bp, k = 32, rBits = 27.
    F1 closed form (uniform reads have no dirty bytes), increments == sum of counters, and the
    value histogram accounts for every bucket; sharded == whole (the multi-GPU merge identity)."""
    n, L, stride, k = 4_000_000, 150, 152, 32
    d = torch.empty(n * stride + 16, dtype=torch.uint8, device="cuda")
    nt.gen_reads_device(d.data_ptr(), 1, 0, n, L, stride, 0)
    with nt.Engine([k], r_bits=27, s_bits=7) as e:
        e.submit_device(d.data_ptr(), n, L, stride)
        _, ph, f1 = e.finish()
        assert int(f1[0]) == n * (L - k + 1)
        assert int(ph[0].sum()) == 2 << 27
        hits = int((ph[0].astype(np.uint64) * np.arange(65536, dtype=np.uint64)).sum())
        assert abs(hits / (n * (L - k + 1)) - 2.0 ** -6) < 2e-4  # two samples of ~2^-7 each
        sk, ncnt, _ = e.device_state()
        assert ncnt == 2 << 27
    half = n // 2
    with nt.Engine([k], r_bits=27, s_bits=7) as e1, nt.Engine([k], r_bits=27, s_bits=7) as e2:
        e1.submit_device(d.data_ptr(), half, L, stride)
        e2.submit_device(d.data_ptr() + half * stride, n - half, L, stride)
        _, p1, f1a = e1.finish()
        _, p2, f1b = e2.finish()
    assert int(f1a[0] + f1b[0]) == n * (L - k + 1)
    h1 = int((p1[0].astype(np.uint64) * np.arange(65536, dtype=np.uint64)).sum())
    h2 = int((p2[0].astype(np.uint64) * np.arange(65536, dtype=np.uint64)).sum())
    assert h1 + h2 == hits


def test_native_merge_devices_matches_single_engine(nt):
    """ntc_merge_devices (C ABI, one host process): reads sharded over min(2, device_count) devices x 2 engines each,
    merged by the 16-bit slice exchange (peer copies; plain copies between engines of one device) == one engine over all reads:
    identical t_Counter, value histogram, F1 and therefore .hist (config 3 at test size, sBits = 11)"""
    ndev = min(2, torch.cuda.device_count())
    n, L, stride = 48_000, 150, 152
    per = n // (2 * ndev)
    want = None
    bufs = []
    for dev in range(ndev):
        with torch.cuda.device(dev):
            d = torch.empty(n * stride + 16, dtype=torch.uint8, device=f"cuda:{dev}")
            nt.gen_reads_device(d.data_ptr(), 21, 0, n, L, stride, 1, genome_len=500_000, device=dev)
            bufs.append(d)
    with nt.Engine([32, 45], r_bits=22, s_bits=11, device=0) as e:
        e.submit_device(bufs[0].data_ptr(), n, L, stride)
        want = e.finish(counters=True)
    engines = [nt.Engine([32, 45], r_bits=22, s_bits=11, device=dev) for dev in range(ndev) for _ in range(2)]
    try:
        for j, e in enumerate(engines):
            dev = j // 2
            e.submit_device(bufs[dev].data_ptr() + j * per * stride, per if j < len(engines) - 1 else n - j * per, L, stride)
        nt.merge_devices(engines)
        got = engines[0].finish(counters=True)
        assert np.array_equal(got[2], want[2]) and np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        for e in engines[1:]:
            _, ph, f1 = e.finish()
            assert int(f1.sum()) == 0 and int(ph[:, :, 1:].sum()) == 0  # the merged-in engines start from zero again
    finally:
        for e in engines:
            e.close()
    # nthll registers merge with max
    hs = [nt.HllEngine(32, 16, device=0) for _ in range(2)]
    try:
        hs[0].submit_device(bufs[0].data_ptr(), n // 2, L, stride)
        hs[1].submit_device(bufs[0].data_ptr() + (n // 2) * stride, n - n // 2, L, stride)
        nt.merge_devices(hs)
        regs, f1 = hs[0].finish()
    finally:
        for h in hs:
            h.close()
    with nt.HllEngine(32, 16, device=0) as h:
        h.submit_device(bufs[0].data_ptr(), n, L, stride)
        regs1, f11 = h.finish()
    assert f1 == f11 and np.array_equal(regs, regs1)


@pytest.mark.parametrize("klist,gap", [([256], 0), ([400], 0), ([600], 0), ([64, 300, 500], 0), ([301], 101), ([255, 256, 257], 0)])
def test_k_beyond_255(nt, klist, gap):
    """k up to ntc_max_k() = 600 (the closed-form table stays in LDS, fewer waves per CU): ragged reads up to 1500 bp, a 20 kbp
    sequence cut into overlapping slots by the host shim, reads shorter than k, N bytes; multi-k and gap seeds"""
    rng = random.Random(3)
    reads = [rseq(rng, rng.randint(200, 1500), pn=0.001) for _ in range(200)] + [rseq(rng, 20000, pn=0.0005), rseq(rng, 600), rseq(rng, 599), rseq(rng, 300), b"", rseq(rng, 5)]
    with nt.Engine(klist, gap=gap, r_bits=14, s_bits=3) as e:
        e.submit_reads(reads)
        tc, ph, f1 = e.finish(counters=True)
    oc, of1 = orc.sketch_reads(reads, klist, gap, 14, 3)
    assert np.array_equal(f1, of1) and np.array_equal(tc, oc)
    assert nt._abi.lib().ntc_max_k() == 600
    with pytest.raises(nt.NtcError):
        nt.Engine([601], r_bits=14, s_bits=3)


@pytest.mark.parametrize("n_eng,r_bits", [(3, 8), (5, 8), (7, 13), (8, 20)])
def test_native_merge_slices_and_wraps(nt, n_eng, r_bits):
    """ntc_merge_devices' 16-bit slice exchange with engine counts that do not divide the counters (short and 16-byte-rounded
    slices), and with per-engine counters and sums beyond 65535 (the exchange carries the low halves only; t_Counter wraps
    there, ntcard.cpp:142-143,439).  All engines on device 0: the same code path as across devices except for the copy call."""
    n, L, stride = 30_000, 150, 152
    d = torch.empty(n * stride + 16, dtype=torch.uint8, device="cuda")
    nt.gen_reads_device(d.data_ptr(), 77, 0, n, L, stride, 0)
    passes = 30 if r_bits <= 8 else 2
    cuts = [n * j // n_eng // 4 * 4 for j in range(n_eng)] + [n]  # 16-byte aligned shares
    with nt.Engine([32], r_bits=r_bits, s_bits=2) as e:
        for _ in range(passes):
            e.submit_device(d.data_ptr(), n, L, stride)
        want = e.finish(counters=True)
    if r_bits <= 8:
        assert n * (L - 31) * passes // 2 // (2 << r_bits) > 65535  # the counters really wrap
    engines = [nt.Engine([32], r_bits=r_bits, s_bits=2) for _ in range(n_eng)]
    try:
        for j, e in enumerate(engines):
            for _ in range(passes):
                e.submit_device(d.data_ptr() + cuts[j] * stride, cuts[j + 1] - cuts[j], L, stride)
        nt.merge_devices(engines)
        got = engines[0].finish(counters=True)
        assert np.array_equal(got[2], want[2]) and np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        # the merged engine keeps counting
        engines[0].submit_device(d.data_ptr(), 1000, L, stride)
        more = engines[0].finish(counters=True)
    finally:
        for e in engines:
            e.close()
    with nt.Engine([32], r_bits=r_bits, s_bits=2) as e:
        for _ in range(passes):
            e.submit_device(d.data_ptr(), n, L, stride)
        e.submit_device(d.data_ptr(), 1000, L, stride)
        want2 = e.finish(counters=True)
    assert np.array_equal(more[0], want2[0]) and np.array_equal(more[2], want2[2])


def test_native_merge_keeps_its_buffers(nt):
    """the exchange buffers, copy streams and events of ntc_merge_devices live in the engines: a second merge of the same group creates
    nothing (n^2 streams and events and two device buffers per call), and is still right"""
    n, L, stride = 20_000, 150, 152
    d = torch.empty(n * stride + 16, dtype=torch.uint8, device="cuda")
    nt.gen_reads_device(d.data_ptr(), 5, 0, n, L, stride, 1, genome_len=300_000)
    engines = [nt.Engine([32], r_bits=18, s_bits=7) for _ in range(3)]
    try:
        for rnd in range(3):
            for j, e in enumerate(engines):
                e.submit_device(d.data_ptr() + (j * 6000) * stride, 6000, L, stride)
            nt.merge_devices(engines)
            if rnd == 0:
                first = [e.merge_allocations() for e in engines]
                assert all(x > 0 for x in first)
            else:
                assert [e.merge_allocations() for e in engines] == first
        got = engines[0].finish(counters=True)
    finally:
        for e in engines:
            e.close()
    with nt.Engine([32], r_bits=18, s_bits=7) as e:
        for rnd in range(3):
            for j in range(3):
                e.submit_device(d.data_ptr() + (j * 6000) * stride, 6000, L, stride)
        want = e.finish(counters=True)
    assert np.array_equal(got[2], want[2]) and np.array_equal(got[0], want[0])


def test_value_hist_device_matches_numpy(nt):
    rng = np.random.default_rng(3)
    c = rng.integers(0, 70000, size=1 << 20, dtype=np.uint32)
    c[::7] = 0
    d = torch.from_numpy(c.view(np.int32)).cuda()
    h = torch.zeros(65536, dtype=torch.int32, device="cuda")
    nt.value_hist_device(d.data_ptr(), d.numel(), h.data_ptr())
    nt.value_hist_device(d[: 1 << 18].data_ptr(), 1 << 18, h.data_ptr())  # accumulates
    torch.cuda.synchronize()
    want = np.bincount(c & 0xFFFF, minlength=65536) + np.bincount(c[: 1 << 18] & 0xFFFF, minlength=65536)
    assert np.array_equal(h.cpu().numpy().astype(np.int64), want)


def test_merge_step_kernels_match_numpy(nt):
    """the device steps of the one-process-per-GPU merge (ntc_narrow_u16_device, ntc_sum_slices_u16_device, ntc_value_hist_u16_device: what
    parallel.exchange_and_sum_u16 runs around the RCCL all-to-all): low halves, wrapping 16-bit sums of slices, histogram of uint16 counters"""
    rng = np.random.default_rng(11)
    n, world = (1 << 18) + 8, 5
    c = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    d = torch.from_numpy(c.view(np.int32)).cuda()
    low = torch.empty(n, dtype=torch.int16, device="cuda")
    nt.narrow_u16_device(d.data_ptr(), n, low.data_ptr())
    assert np.array_equal(low.cpu().numpy().view(np.uint16), (c & 0xFFFF).astype(np.uint16))
    sl = rng.integers(0, 1 << 16, size=(world, n), dtype=np.uint32).astype(np.uint16)
    sl[:, :16] = 0xFFFF  # sums that wrap several times
    ds = torch.from_numpy(sl.view(np.int16).reshape(-1)).cuda()
    nt.sum_slices_u16_device(ds.data_ptr(), n, world, n - 3)  # (a length that is no multiple of 8: the scalar tail)
    got = ds.cpu().numpy().view(np.uint16).reshape(world, n)
    want = sl.astype(np.uint32).sum(axis=0).astype(np.uint16)
    assert np.array_equal(got[0, : n - 3], want[: n - 3]) and np.array_equal(got[0, n - 3:], sl[0, n - 3:]) and np.array_equal(got[1:], sl[1:])
    h = torch.zeros(65536, dtype=torch.int32, device="cuda")
    u = rng.integers(0, 70000, size=n - 5, dtype=np.uint32).astype(np.uint16)
    u[::5] = 0
    du = torch.from_numpy(np.concatenate([u, np.zeros(5, np.uint16)]).view(np.int16)).cuda()
    nt.value_hist_u16_device(du.data_ptr(), n - 5, h.data_ptr())
    nt.value_hist_u16_device(du.data_ptr(), 4096, h.data_ptr())  # accumulates
    torch.cuda.synchronize()
    assert np.array_equal(h.cpu().numpy().astype(np.int64), np.bincount(u, minlength=65536) + np.bincount(u[:4096], minlength=65536))


def test_bench_under_torchrun_single_rank(tmp_path):
    """the multi-GPU code path of bench.py (RCCL init, all-to-all slice exchange, histogram reduce) with one rank"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--reads-per-step", "1000000", "--no-cpu-baseline", "--no-live-pmc"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(r.stdout.decode().strip().splitlines()[-1])
    ref = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--reads-per-step", "1000000",
                          "--no-cpu-baseline", "--no-live-pmc"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=root)
    k = json.loads(ref.stdout.decode().strip().splitlines()[-1])
    assert j["f1_total"] == k["f1_total"] and j["sampled_increments"] == k["sampled_increments"] and j["n_gpus"] == 1
    assert j["merge"]["mode"] == "slices"
    # the merge that ships hits (round 6: ntc_log_export_device -> all-to-all of keys -> ntc_log_replace_device -> the engine's own sketch update), and sBits = 11,
    # whose default it is
    for extra in (["--merge", "owner"], ["--s-bits", "11"]):
        r = subprocess.run(cmd + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        j = json.loads(r.stdout.decode().strip().splitlines()[-1])
        ref = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--reads-per-step", "1000000",
                              "--no-cpu-baseline", "--no-live-pmc"] + [a for a in extra if a not in ("--merge", "owner")], stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, timeout=600, cwd=root)
        k = json.loads(ref.stdout.decode().strip().splitlines()[-1])
        assert j["merge"]["mode"] == "owner" and j["merge"]["keys_sent"] == j["sampled_increments"], j["merge"]
        assert j["f1_total"] == k["f1_total"] and j["sampled_increments"] == k["sampled_increments"]


@pytest.mark.parametrize("n_eng,r_bits", [(2, 18), (4, 16), (8, 21)])
def test_owner_mode_between_engines_on_one_device(nt, n_eng, r_bits):
    """the device steps of the merge that ships hits, without a communicator: N engines on this device hash N shares of the reads; every engine's pending log
    is split by counter-range owner (ntc_log_export_device), owner j gets part j of every engine as its pending log (ntc_log_replace_device) and its own
    sketch update counts them: range j of engine j's sketch == range j of ONE engine's sketch over all the reads (uint32 counters: before the wrap)"""
    rng = np.random.default_rng(n_eng)
    alpha = np.frombuffer(b"ACGTacgtNn", dtype=np.uint8)
    n, L = 6000 * n_eng, 150
    arr = alpha[rng.integers(0, 4, size=(n, L))]
    arr = np.where(rng.random((n, L)) < 0.003, alpha[rng.integers(4, len(alpha), size=(n, L))], arr).astype(np.uint8)
    reads = [arr[i].tobytes() for i in range(n)]
    share = n // n_eng
    tiles = [torch.from_numpy(nt.tile_reads(reads[i * share:(i + 1) * share], L)).cuda() for i in range(n_eng)]
    with nt.Engine([32], r_bits=r_bits, s_bits=7) as one:
        for t in tiles:
            one.submit_tiled_device(t.data_ptr(), share, L)
        one.flush()
        sk, ncnt, _ = one.device_state()
        torch.cuda.synchronize()
        want = torch.as_tensor(_DevArray(sk, ncnt), device="cuda").clone()
    engs = [nt.Engine([32], r_bits=r_bits, s_bits=7, flags=nt.FLAG_DEFER_REDO if i % 2 else 0) for i in range(n_eng)]
    try:
        for e, t in zip(engs, tiles):
            e.submit_tiled_device(t.data_ptr(), share, L)
        parts = []
        for e in engs:
            counts = e.log_export(n_eng)
            offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
            buf = torch.empty(max(int(offs[-1]), 1), dtype=torch.int32, device="cuda")
            assert e.log_export(n_eng, buf.data_ptr(), offs[:-1]) == counts
            torch.cuda.synchronize()
            parts.append([buf[int(offs[p]):int(offs[p + 1])] for p in range(n_eng)])
        assert sum(int(p.numel()) for ps in parts for p in ps) == int(want.to(torch.int64).sum())
        rng_len = ncnt // n_eng
        for j, e in enumerate(engs):
            mine = torch.cat([parts[i][j] for i in range(n_eng)])
            assert bool(((mine.to(torch.int64) // rng_len) == j).all())
            e.log_replace(mine.data_ptr(), mine.numel())
            e.flush()
            sk, _, _ = e.device_state()
            torch.cuda.synchronize()
            got = torch.as_tensor(_DevArray(sk, ncnt), device="cuda")
            assert torch.equal(got[j * rng_len:(j + 1) * rng_len], want[j * rng_len:(j + 1) * rng_len]), j
            assert int(got.to(torch.int64).sum()) == int(mine.numel())  # nothing outside its range
        with pytest.raises(nt.NtcError):  # the sketch holds counts now: hits can no longer be shipped (the caller merges counters)
            engs[0].log_export(n_eng)
    finally:
        for e in engs:
            e.close()


def test_bench_gpus_flag_launches_real_ranks(tmp_path):
    """`python bench.py --gpus N` (the form the driver uses) must run N RCCL ranks, not one: N = min(2, device_count) ranks through
    bench.py's own re-launch under torch.distributed.run, the merged result compared with the sums of N one-rank runs over the same
    read-index ranges (rank r of N hashes reads [r * R * K, (r + 1) * R * K)); more GPUs than the node has must fail loudly."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = min(2, torch.cuda.device_count())
    common = ["--steps", "2", "--warmup", "1", "--reads-per-step", "1000000", "--no-cpu-baseline", "--no-live-pmc"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n)] + common, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert j["n_gpus"] == n and (n == 1 or j["config"]["rccl_ranks"] == n)
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common + ["--steps", str(2 * n)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         timeout=900, cwd=root, env=env)  # one rank over the union of the N ranks' read ranges
    k = json.loads(one.stdout.decode().strip().splitlines()[-1])
    assert j["f1_total"] == k["f1_total"] and j["sampled_increments"] == k["sampled_increments"]
    too_many = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(torch.cuda.device_count() + 1)] + common,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd=root, env=env)
    assert too_many.returncode != 0 and b"HIP device" in too_many.stderr


def test_randomised_shapes_small():
    """a short run of the randomised sweep (tools/fuzz_parity.py: random k lists, gaps, lengths, dirt, submit patterns)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "60", "2024"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:]
