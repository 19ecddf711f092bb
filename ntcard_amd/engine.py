"""Python mirror of the reference seam over the C ABI (ctypes); torch is plumbing only
(device memory, streams, torch.distributed for the multi-GPU sketch merge).

Names follow the reference: an `Engine` owns t_Counter and F1 (ntcard.cpp:433-439), `submit*`
is a batch of ntRead/stRead calls (ntcard.cpp:147-171), `finish` returns what compEst consumes
(ntcard.cpp:237-247), `estimate` is compEst's recurrence (ntcard.cpp:249-274) and `write_hist`
is outDefault's body (ntcard.cpp:291-294).
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import NtcConfig, NtcError, check

FLAG_SIMPLE_KERNEL = 1  # NTC_FLAG_SIMPLE_KERNEL: run the simple validation kernel
FLAG_LANE_KERNEL = 32  # NTC_FLAG_LANE_KERNEL: the lane-per-read kernel K1 takes every batch (tiled ones are re-laid out as row slots)
FLAG_ALWAYS_LOG = 8  # NTC_FLAG_ALWAYS_LOG: never switch from the hit log to direct atomics
FLAG_PARTITION_ALWAYS = 16  # NTC_FLAG_PARTITION_ALWAYS: small logs go through the partition passes too (validation)
FLAG_DEFER_REDO = 128  # NTC_FLAG_DEFER_REDO: submit_device buffers stay unchanged until sync(); the handed-back reads of several batches share one pass
FLAG_REQUIRE_TILED = 64  # NTC_FLAG_REQUIRE_TILED: submit_tiled_device fails instead of falling back to the general kernel
FLAG_DIRECT_ATOMICS = 2  # NTC_FLAG_DIRECT_ATOMICS: no hit log, one device atomic per sampled k-mer
SIZE_RULE_BYTES = 50_000_000_000  # ntcard.cpp:430: total input < 50 GB => sBits = 7


def s_bits_for_input(total_bytes, requested=11):
    """The reference's size rule (ntcard.cpp:427-431); the caller applies it before Engine()."""
    return 7 if total_bytes < SIZE_RULE_BYTES else requested


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Engine:
    def __init__(self, klist, gap=0, r_bits=27, s_bits=7, device=0, stream=None, ext_sketch=None, ext_f1=None, flags=0, log_entries=0):
        self._lib = _abi.lib()
        self.klist = [int(k) for k in klist]
        self.gap, self.r_bits, self.s_bits, self.device = int(gap), int(r_bits), int(s_bits), int(device)
        self._karr = (C.c_uint32 * len(self.klist))(*self.klist)
        cfg = NtcConfig()
        cfg.n_k = len(self.klist)
        cfg.k = C.cast(self._karr, C.POINTER(C.c_uint32))
        cfg.gap, cfg.r_bits, cfg.s_bits, cfg.device = self.gap, self.r_bits, self.s_bits, self.device
        cfg.stream = C.c_void_p(stream) if stream else None
        self._keep = (ext_sketch, ext_f1)  # keep torch tensors alive
        cfg.ext_sketch = C.c_void_p(ext_sketch.data_ptr()) if ext_sketch is not None else None
        cfg.ext_f1 = C.c_void_p(ext_f1.data_ptr()) if ext_f1 is not None else None
        cfg.flags = int(flags)
        cfg.log_entries = int(log_entries)
        h = C.c_void_p()
        check(self._lib.ntc_create(C.byref(cfg), C.byref(h)))
        self._h = h

    # -- lifecycle ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.ntc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def reset(self):
        check(self._lib.ntc_reset(self._h))

    # -- the seam ----------------------------------------------------------------------------
    def submit(self, bases, offsets):
        """bases: uint8 array / bytes of concatenated reads; offsets: uint64[n+1]."""
        b = np.frombuffer(bases, dtype=np.uint8) if isinstance(bases, (bytes, bytearray)) else np.ascontiguousarray(bases, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        if b.size == 0:
            b = np.zeros(1, dtype=np.uint8)
        check(self._lib.ntc_submit(self._h, _np_ptr(b), _np_ptr(o), len(o) - 1))

    def submit_spans(self, buf, starts, lens):
        """reads as spans of one host buffer: read i = buf[starts[i] : starts[i] + lens[i]]"""
        b = np.frombuffer(buf, dtype=np.uint8) if isinstance(buf, (bytes, bytearray)) else np.ascontiguousarray(buf, dtype=np.uint8)
        st = np.ascontiguousarray(starts, dtype=np.uint64)
        ln = np.ascontiguousarray(lens, dtype=np.uint32)
        assert len(st) == len(ln)
        check(self._lib.ntc_submit_spans(self._h, _np_ptr(b), _np_ptr(st), _np_ptr(ln), len(st)))

    def submit_reads(self, reads):
        offs = np.zeros(len(reads) + 1, dtype=np.uint64)
        if reads:
            offs[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
        self.submit(b"".join(reads), offs)

    def submit_device(self, d_slots_ptr, n_reads, read_len, stride):
        check(self._lib.ntc_submit_device(self._h, C.c_void_p(d_slots_ptr), n_reads, read_len, stride))

    def submit_tiled_device(self, d_tiles_ptr, n_reads, read_len):
        """a device-resident batch in the tiled layout (include/ntcard_hip.h: ntc_submit_tiled_device)"""
        check(self._lib.ntc_submit_tiled_device(self._h, C.c_void_p(d_tiles_ptr), n_reads, read_len))

    def submit_tiled_ragged_device(self, d_tiles_ptr, n_reads, n_chunks, d_tails_ptr):
        """a device-resident RAGGED batch: reads of 16 n_chunks - 15 .. 16 n_chunks bases in the tiled layout, every tile sorted longest first,
        d_tails = uint32[n_tiles][16] (include/ntcard_hip.h: ntc_submit_tiled_ragged_device; tile_reads_ragged builds both)"""
        check(self._lib.ntc_submit_tiled_ragged_device(self._h, C.c_void_p(d_tiles_ptr), n_reads, n_chunks, C.c_void_p(d_tails_ptr)))

    def submit_tiled_bins_device(self, bins):
        """several device-resident tiled batches in one call (include/ntcard_hip.h: ntc_submit_tiled_bins_device); bins: (d_tiles_ptr, n_reads, read_len,
        d_tails_ptr or 0) — d_tails_ptr = 0: an equal-length batch, else a ragged one with read_len = 16 x its chunks"""
        n = len(bins)
        tiles = (C.c_void_p * n)(*[b[0] for b in bins])
        nr = (C.c_uint64 * n)(*[b[1] for b in bins])
        rl = (C.c_uint32 * n)(*[b[2] for b in bins])
        tails = (C.c_void_p * n)(*[(b[3] or None) for b in bins])
        check(self._lib.ntc_submit_tiled_bins_device(self._h, n, tiles, nr, rl, tails))

    def sync(self):
        check(self._lib.ntc_sync(self._h))

    def flush(self):
        """apply the pending hit log to the device sketch (asynchronous on the engine's stream)"""
        check(self._lib.ntc_flush(self._h))

    def finish(self, counters=False, p_hist=True):
        """-> (t_counter uint16[nk,2,2^r] | None, p_hist uint32[nk,2,65536] | None, f1 uint64[nk])"""
        nk = len(self.klist)
        tc = np.zeros((nk, 2, 1 << self.r_bits), dtype=np.uint16) if counters else None
        ph = np.zeros((nk, 2, 65536), dtype=np.uint32) if p_hist else None
        f1 = np.zeros(nk, dtype=np.uint64)
        check(self._lib.ntc_finish(self._h, _np_ptr(tc) if counters else None, _np_ptr(ph) if p_hist else None, _np_ptr(f1)))
        return tc, ph, f1

    def merge_counters(self, t_counter, f1=None):
        """add a dumped t_Counter image (uint16[nk,2,2^r], from finish(counters=True)) and its F1 into this engine"""
        tc = np.ascontiguousarray(t_counter, dtype=np.uint16)
        assert tc.size == len(self.klist) * (2 << self.r_bits)
        f = np.ascontiguousarray(f1, dtype=np.uint64) if f1 is not None else None
        check(self._lib.ntc_merge_counters(self._h, _np_ptr(tc), _np_ptr(f) if f is not None else None))

    def device_state(self):
        sk, n, f1 = C.c_void_p(), C.c_uint64(), C.c_void_p()
        check(self._lib.ntc_device_state(self._h, C.byref(sk), C.byref(n), C.byref(f1)))
        return sk.value, n.value, f1.value

    def log_export(self, n_parts, d_keys_ptr=None, part_offset=None):
        """the pending hit log split by counter-range owner (include/ntcard_hip.h: ntc_log_export_device) -> counts per owner (list of int); with d_keys_ptr
        (device uint32 buffer) and part_offset (n_parts ints) the keys are written as well.  NtcError(NTC_ERR_STATE) when the sketch already holds counts."""
        counts = (C.c_uint64 * n_parts)()
        offs = (C.c_uint64 * n_parts)(*[int(x) for x in part_offset]) if part_offset is not None else None
        check(self._lib.ntc_log_export_device(self._h, n_parts, C.c_void_p(d_keys_ptr) if d_keys_ptr else None, offs, counts))
        return [int(c) for c in counts]

    def log_replace(self, d_keys_ptr, n_keys):
        """the n_keys counter indices at d_keys_ptr (device uint32) become the pending hit log (ntc_log_replace_device)"""
        check(self._lib.ntc_log_replace_device(self._h, C.c_void_p(d_keys_ptr) if n_keys else None, n_keys))

    def set_profiling(self, on=True):
        check(self._lib.ntc_set_profiling(self._h, 1 if on else 0))

    def kernel_time(self):
        ms, n = C.c_double(), C.c_uint64()
        check(self._lib.ntc_kernel_time(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def update_mode(self):
        """0: hit log + partitioned apply, 1: direct atomics (waits for the stream)"""
        m = C.c_uint32()
        check(self._lib.ntc_update_mode(self._h, C.byref(m)))
        return m.value

    def merge_allocations(self):
        """buffers / streams / events ntc_merge_devices has created for this engine (kept between merges)"""
        n = C.c_uint64()
        check(self._lib.ntc_merge_allocations(self._h, C.byref(n)))
        return n.value

    def fixup_time(self):
        """milliseconds of the deferred K1f launches (FLAG_DEFER_REDO engines: one per up to 8 K1h launches, on the engine's stream; 0 otherwise)"""
        ms = C.c_double()
        check(self._lib.ntc_fixup_time(self._h, C.byref(ms)))
        return ms.value

    def apply_time(self):
        ms, n = C.c_double(), C.c_uint64()
        check(self._lib.ntc_apply_time(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value


class HllEngine(Engine):
    """nthll: uint8 M[1<<n_bits] of max leading-zero runs (nthll.cpp:92-105,212-243)"""

    def __init__(self, k, n_bits=16, device=0, stream=None):
        self._lib = _abi.lib()
        self.klist, self.gap, self.n_bits, self.device = [int(k)], 0, int(n_bits), int(device)
        h = C.c_void_p()
        check(self._lib.ntc_hll_create(int(k), int(n_bits), int(device), C.c_void_p(stream) if stream else None, C.byref(h)))
        self._h = h

    def finish(self):
        regs = np.zeros(1 << self.n_bits, dtype=np.uint8)
        f1 = np.zeros(1, dtype=np.uint64)
        check(self._lib.ntc_hll_finish(self._h, _np_ptr(regs), _np_ptr(f1)))
        return regs, int(f1[0])


def hll_estimate(regs, n_bits=16):
    est = C.c_double()
    r = np.ascontiguousarray(regs, dtype=np.uint8)
    check(_abi.lib().ntc_hll_estimate(_np_ptr(r), int(n_bits), C.byref(est)))
    return est.value


def merge_devices(engines):
    """sum (mod 2^16, like t_Counter) of the engines' sketches and F1 into engines[0] by a 16-bit slice exchange over peer copies;
    the others are reset (ntc_merge_devices)"""
    arr = (C.c_void_p * len(engines))(*[e._h for e in engines])
    check(_abi.lib().ntc_merge_devices(arr, len(engines)))


# -- stateless entry points ---------------------------------------------------------------------
def estimate(p_hist_k, r_bits, s_bits, cov_max=1000):
    """compEst for one k from p[2][65536] -> (F0, f[0..cov_max])"""
    p = np.ascontiguousarray(p_hist_k, dtype=np.uint32)
    assert p.shape == (2, 65536)
    cov_max = min(int(cov_max), 65535)
    f0 = C.c_double()
    f = np.zeros(cov_max + 1, dtype=np.float64)
    check(_abi.lib().ntc_estimate(_np_ptr(p), r_bits, s_bits, cov_max, C.byref(f0), _np_ptr(f)))
    return f0.value, f


def write_hist(path, f1, F0, f, cov_max=1000):
    f = np.ascontiguousarray(f, dtype=np.float64)
    check(_abi.lib().ntc_write_hist(str(path).encode(), int(f1), float(F0), _np_ptr(f), min(int(cov_max), 65535)))


def gen_reads_device(d_ptr, seed, first, n, read_len, stride, dist, genome_len=100_000_000, device=0, stream=None):
    check(_abi.lib().ntc_gen_reads_device(device, C.c_void_p(stream) if stream else None, C.c_void_p(d_ptr), seed, first, n,
                                          read_len, stride, dist, genome_len))


def tiled_bytes(n_reads, read_len):
    """size of a batch in the tiled layout"""
    return int(_abi.lib().ntc_tiled_bytes(n_reads, read_len))


def gen_reads_tiled_device(d_ptr, seed, first, n, read_len, dist, genome_len=100_000_000, device=0, stream=None):
    """the reads of gen_reads_device (bit-identical bases) in the tiled layout"""
    check(_abi.lib().ntc_gen_reads_tiled_device(device, C.c_void_p(stream) if stream else None, C.c_void_p(d_ptr), seed, first, n,
                                                read_len, dist, genome_len))


def tile_reads(reads, read_len=None):
    """host-side packing of equal-length reads (list of bytes) into the tiled layout -> uint8 array (tests, small inputs)"""
    n = len(reads)
    L = read_len if read_len is not None else (len(reads[0]) if n else 0)
    C16 = (L + 15) // 16
    nt = (n + 2047) // 2048
    out = np.full((nt, C16, 2048, 16), ord("A"), dtype=np.uint8)
    if n:
        a = np.full((nt * 2048, C16 * 16), ord("A"), dtype=np.uint8)
        a[:n, :L] = np.frombuffer(b"".join(reads), dtype=np.uint8).reshape(n, L)
        out[:] = a.reshape(nt, 2048, C16, 16).transpose(0, 2, 1, 3)
    return out.reshape(-1)


def tile_reads_ragged(reads, n_chunks):
    """host-side packing of a RAGGED batch (reads of 16 n_chunks - 15 .. 16 n_chunks bases, list of bytes) -> (tiles uint8 array, tails uint32
    [n_tiles, 16], order): the reads are sorted longest first (order[j] = index of the read in slot j), tails[t, d] = reads of tile t with more than d
    bases in their last piece (tests, small inputs)"""
    n, C16 = len(reads), int(n_chunks)
    lens = np.array([len(r) for r in reads], dtype=np.int64)
    assert n == 0 or (lens.min() > 16 * (C16 - 1) and lens.max() <= 16 * C16)
    order = np.argsort(-lens, kind="stable")
    nt = (n + 2047) // 2048
    a = np.full((nt * 2048, C16 * 16), ord("A"), dtype=np.uint8)
    for j, i in enumerate(order):
        a[j, :lens[i]] = np.frombuffer(reads[i], dtype=np.uint8)
    tails = np.zeros((nt, 16), dtype=np.uint32)
    tl = lens[order] - 16 * (C16 - 1)
    for t in range(nt):
        part = tl[t * 2048:(t + 1) * 2048]
        for d in range(16):
            tails[t, d] = int(np.count_nonzero(part > d))
    tiles = np.ascontiguousarray(a.reshape(nt, 2048, C16, 16).transpose(0, 2, 1, 3)).reshape(-1)
    return tiles, tails, order


def value_hist_device(d_counters_ptr, n, d_hist_ptr, device=0, stream=None):
    """accumulate the value histogram (counter & 0xffff) of n device uint32 counters into a device uint32[65536]"""
    check(_abi.lib().ntc_value_hist_device(device, C.c_void_p(stream) if stream else None, C.c_void_p(d_counters_ptr), n,
                                           C.c_void_p(d_hist_ptr)))


def narrow_u16_device(d_counters_ptr, n, d_out_ptr, device=0, stream=None):
    """d_out (uint16[n]) = the low halves of n device uint32 counters (t_Counter wraps at 16 bits: all the merge has to move)"""
    check(_abi.lib().ntc_narrow_u16_device(device, C.c_void_p(stream) if stream else None, C.c_void_p(d_counters_ptr), n, C.c_void_p(d_out_ptr)))


def sum_slices_u16_device(d_slices_ptr, stride, n_slices, length, device=0, stream=None):
    """slice 0 += slices 1 .. n_slices - 1 with wrapping 16-bit adds; slice r = uint16[r * stride : r * stride + length]"""
    check(_abi.lib().ntc_sum_slices_u16_device(device, C.c_void_p(stream) if stream else None, C.c_void_p(d_slices_ptr), stride, n_slices, length))


def value_hist_u16_device(d_counters_ptr, n, d_hist_ptr, device=0, stream=None):
    """accumulate the value histogram of n device uint16 counters into a device uint32[65536]"""
    check(_abi.lib().ntc_value_hist_u16_device(device, C.c_void_p(stream) if stream else None, C.c_void_p(d_counters_ptr), n, C.c_void_p(d_hist_ptr)))


def hash_dump_device(d_slots_ptr, n_reads, read_len, stride, k, gap, max_win, d_hash_ptr, d_count_ptr, device=0, stream=None, k1=False):
    """every canonical (spaced-seed when gap != 0) hash of every clean window; k1=True: out of the production kernel K1"""
    fn = _abi.lib().ntc_hash_dump_k1_device if k1 else _abi.lib().ntc_hash_dump_device
    check(fn(device, C.c_void_p(stream) if stream else None, C.c_void_p(d_slots_ptr), n_reads,
                                          read_len, stride, k, gap, max_win, C.c_void_p(d_hash_ptr), C.c_void_p(d_count_ptr)))
