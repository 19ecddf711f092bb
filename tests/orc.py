"""ctypes binding of oracle/libntc_oracle.so — the CPU checker (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libntc_oracle.so")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
REF_TOOL = os.path.join(REF_DIR, "ref_tool")
REF_NTCARD = os.path.join(REF_DIR, "ntcard_ref")

_lib = None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    src = os.path.join(ORACLE_DIR, "ntc_oracle.c")
    if not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        build_oracle()
    L = C.CDLL(LIB_PATH)
    u8, u32, u64, sz = C.c_uint8, C.c_uint32, C.c_uint64, C.c_size_t
    p = C.c_void_p
    L.orc_seed.restype = u64
    L.orc_seed.argtypes = [u8]
    L.orc_seed_comp.restype = u64
    L.orc_seed_comp.argtypes = [u8]
    L.orc_srol.restype = u64
    L.orc_srol.argtypes = [u64, C.c_uint]
    L.orc_sror1.restype = u64
    L.orc_sror1.argtypes = [u64]
    L.orc_window_hash.restype = C.c_int
    L.orc_window_hash.argtypes = [C.c_char_p, C.c_uint, C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_uint)]
    L.orc_hash_read.restype = sz
    L.orc_hash_read.argtypes = [C.c_char_p, sz, C.c_uint, p, p, sz]
    L.orc_gap_positions.restype = sz
    L.orc_gap_positions.argtypes = [C.c_uint, C.c_uint, p]
    L.orc_sthash_read.restype = sz
    L.orc_sthash_read.argtypes = [C.c_char_p, sz, C.c_uint, p, sz, p, p, sz]
    L.orc_multihash.restype = u64
    L.orc_multihash.argtypes = [u64, C.c_uint, C.c_uint]
    L.orc_sample_of.restype = C.c_uint
    L.orc_sample_of.argtypes = [u64, C.c_uint]
    L.orc_sketch_update.restype = None
    L.orc_sketch_update.argtypes = [p, p, p, u64, p, u32, u32, u32, u32, p, C.c_int]
    L.orc_sketch_update_sharded.restype = None
    L.orc_sketch_update_sharded.argtypes = [p, p, p, u64, p, u32, u32, u32, p, C.c_int]
    L.orc_value_hist.restype = None
    L.orc_value_hist.argtypes = [p, u32, p]
    L.orc_comp_est_p.restype = None
    L.orc_comp_est_p.argtypes = [p, u32, u32, u32, C.POINTER(C.c_double), p]
    L.orc_comp_est.restype = None
    L.orc_comp_est.argtypes = [p, u32, u32, u32, C.POINTER(C.c_double), p]
    L.orc_format_hist.restype = sz
    L.orc_format_hist.argtypes = [u64, C.c_double, p, u32, C.c_char_p, sz]
    L.orc_hll_update.restype = None
    L.orc_hll_update.argtypes = [p, u32, p, p, u64, u32, C.c_int]
    L.orc_hll_estimate.restype = C.c_double
    L.orc_hll_estimate.argtypes = [p, u32]
    L.orc_gen_reads.restype = None
    L.orc_gen_reads.argtypes = [u64, u64, u64, u32, u32, u32, u64, p]
    L.orc_fnv1a64.restype = u64
    L.orc_fnv1a64.argtypes = [p, sz]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def hash_read(seq: bytes, k: int):
    """-> (hashes u64[n], pos u32[n]) for every clean k-window, in order."""
    L = lib()
    cap = max(len(seq), 1)
    h = np.zeros(cap, dtype=np.uint64)
    pos = np.zeros(cap, dtype=np.uint32)
    n = L.orc_hash_read(seq, len(seq), k, _ptr(h), _ptr(pos), cap)
    return h[:n].copy(), pos[:n].copy()


def gap_positions(k: int, gap: int):
    L = lib()
    out = np.zeros(max(gap, 1), dtype=np.uint32)
    n = L.orc_gap_positions(k, gap, _ptr(out))
    return out[:n].copy()


def sthash_read(seq: bytes, k: int, gap: int):
    L = lib()
    gp = gap_positions(k, gap)
    cap = max(len(seq), 1)
    h = np.zeros(cap, dtype=np.uint64)
    pos = np.zeros(cap, dtype=np.uint32)
    n = L.orc_sthash_read(seq, len(seq), k, _ptr(gp), len(gp), _ptr(h), _ptr(pos), cap)
    return h[:n].copy(), pos[:n].copy()


def concat_reads(reads):
    """list[bytes] -> (bases uint8[], offsets uint64[n+1])"""
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    if reads:
        offs[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8).copy()
    if bases.size == 0:
        bases = np.zeros(1, dtype=np.uint8)
    return bases, offs


def sketch_update(counters, bases, offsets, klist, gap, r_bits, s_bits, f1=None, threads=0):
    """counters: uint16 [nk,2,1<<r_bits] (updated in place).  Returns f1 (uint64[nk])."""
    L = lib()
    kl = np.asarray(klist, dtype=np.uint32)
    if f1 is None:
        f1 = np.zeros(len(kl), dtype=np.uint64)
    assert counters.dtype == np.uint16 and counters.flags.c_contiguous
    L.orc_sketch_update(_ptr(counters), _ptr(bases), _ptr(offsets), len(offsets) - 1, _ptr(kl),
                        len(kl), gap, r_bits, s_bits, _ptr(f1), threads)
    return f1


def sketch_update_sharded(counters, bases, offsets, klist, r_bits, s_bits, f1=None, threads=1):
    """plain k-mers, per-k precomputed tables, one thread per shard (the timed cpu_baseline form); same results"""
    L = lib()
    kl = np.asarray(klist, dtype=np.uint32)
    if f1 is None:
        f1 = np.zeros(len(kl), dtype=np.uint64)
    assert counters.dtype == np.uint16 and counters.flags.c_contiguous
    L.orc_sketch_update_sharded(_ptr(counters), _ptr(bases), _ptr(offsets), len(offsets) - 1, _ptr(kl),
                                len(kl), r_bits, s_bits, _ptr(f1), threads)
    return f1


def sketch_reads(reads, klist, gap=0, r_bits=27, s_bits=7, threads=0):
    counters = np.zeros((len(klist), 2, 1 << r_bits), dtype=np.uint16)
    bases, offs = concat_reads(reads)
    f1 = sketch_update(counters, bases, offs, klist, gap, r_bits, s_bits, threads=threads)
    return counters, f1


def hll_reads(reads, k, n_bits=16, threads=0):
    """-> (regs uint8[1<<n_bits], estimate) as nthll.cpp computes them"""
    L = lib()
    regs = np.zeros(1 << n_bits, dtype=np.uint8)
    bases, offs = concat_reads(reads)
    L.orc_hll_update(_ptr(regs), n_bits, _ptr(bases), _ptr(offs), len(reads), k, threads)
    return regs, float(L.orc_hll_estimate(_ptr(regs), n_bits))


def value_hist(counters_k, r_bits):
    L = lib()
    p = np.zeros((2, 65536), dtype=np.uint32)
    L.orc_value_hist(_ptr(np.ascontiguousarray(counters_k)), r_bits, _ptr(p))
    return p


def comp_est_p(p, r_bits, s_bits, limit=65535):
    L = lib()
    f0 = C.c_double(0.0)
    fm = np.zeros(65536, dtype=np.float64)
    L.orc_comp_est_p(_ptr(np.ascontiguousarray(p, dtype=np.uint32)), r_bits, s_bits, limit, C.byref(f0), _ptr(fm))
    return f0.value, fm


def format_hist(f1, F0, f_mean, cov_max=1000):
    L = lib()
    cap = 64 * (cov_max + 4)
    buf = C.create_string_buffer(cap)
    n = L.orc_format_hist(int(f1), F0, _ptr(f_mean), cov_max, buf, cap)
    return buf.raw[:n]


def hist_from_counters(counters_k, f1, r_bits, s_bits, cov_max=1000):
    p = value_hist(counters_k, r_bits)
    F0, fm = comp_est_p(p, r_bits, s_bits, limit=cov_max)
    return format_hist(f1, F0, fm, cov_max)


def gen_reads(seed, first, n, read_len, stride, dist, genome_len=100_000_000):
    L = lib()
    out = np.zeros(n * stride, dtype=np.uint8)
    L.orc_gen_reads(seed, first, n, read_len, stride, dist, genome_len, _ptr(out))
    return out


def fnv1a64(arr):
    L = lib()
    a = np.ascontiguousarray(arr)
    return int(L.orc_fnv1a64(_ptr(a), a.nbytes))


# ---- driver for the real reference (build container only) ----------------------------------
def have_ref():
    return os.path.exists(REF_TOOL) and os.path.exists(REF_NTCARD)


def ref_hash(seqs, k, h=1, tmpdir="/tmp"):
    """Run the reference's ntHashIterator over each sequence -> list of (pos[], hashes[n,h])"""
    import tempfile
    with tempfile.TemporaryDirectory(dir=tmpdir) as d:
        fin, fout = os.path.join(d, "in.txt"), os.path.join(d, "out.txt")
        with open(fin, "wb") as f:
            for s in seqs:
                f.write(s + b"\n")
        subprocess.check_call([REF_TOOL, "hash", str(k), str(h), fin, fout])
        return _parse_rows(fout, h)


def ref_sthash(seqs, k, gap, tmpdir="/tmp"):
    import tempfile
    with tempfile.TemporaryDirectory(dir=tmpdir) as d:
        fin, fout = os.path.join(d, "in.txt"), os.path.join(d, "out.txt")
        with open(fin, "wb") as f:
            for s in seqs:
                f.write(s + b"\n")
        subprocess.check_call([REF_TOOL, "sthash", str(k), str(gap), fin, fout])
        return _parse_rows(fout, 1)


def _parse_rows(path, h):
    res = []
    with open(path) as f:
        lines = f.read().split("\n")
    i = 0
    while i < len(lines):
        if not lines[i].startswith("R "):
            i += 1
            continue
        n = int(lines[i][2:])
        pos = np.zeros(n, dtype=np.uint32)
        hs = np.zeros((n, h), dtype=np.uint64)
        for j in range(n):
            parts = lines[i + 1 + j].split()
            pos[j] = int(parts[0])
            for t in range(h):
                hs[j, t] = int(parts[1 + t], 16)
        res.append((pos, hs))
        i += 1 + n
    return res


REF_HLL_TOOL = os.path.join(REF_DIR, "ref_hll_tool")
REF_NTHLL = os.path.join(REF_DIR, "nthll_ref")


def ref_hll(seqs, k, n_bits=16, tmpdir="/tmp"):
    """registers + printed estimate of the real nthll code (whitebox tool around nthll.cpp)"""
    import tempfile
    with tempfile.TemporaryDirectory(dir=tmpdir) as d:
        fin, fout = os.path.join(d, "in.txt"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            for s in seqs:
                f.write(s + b"\n")
        line = subprocess.check_output([REF_HLL_TOOL, str(k), str(n_bits), fin, fout])
        regs = np.fromfile(fout, dtype=np.uint8)
    return regs, line


def ref_sketch(seqs, klist, gap, r_bits, s_bits, tmpdir="/tmp"):
    import tempfile
    with tempfile.TemporaryDirectory(dir=tmpdir) as d:
        fin, fout = os.path.join(d, "in.txt"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            for s in seqs:
                f.write(s + b"\n")
        subprocess.check_call([REF_TOOL, "sketch", str(r_bits), str(s_bits), str(gap),
                               ",".join(map(str, klist)), fin, fout])
        raw = np.fromfile(fout, dtype=np.uint8)
    nk = len(klist)
    f1 = raw[: 8 * nk].view(np.uint64).copy()
    counters = raw[8 * nk:].view(np.uint16).reshape(nk, 2, 1 << r_bits).copy()
    return counters, f1


def ref_est(counters_k, r_bits, s_bits, tmpdir="/tmp"):
    import tempfile
    with tempfile.TemporaryDirectory(dir=tmpdir) as d:
        fin, fout = os.path.join(d, "c.bin"), os.path.join(d, "out.bin")
        np.ascontiguousarray(counters_k, dtype=np.uint16).tofile(fin)
        subprocess.check_call([REF_TOOL, "est", str(r_bits), str(s_bits), fin, fout])
        raw = np.fromfile(fout, dtype=np.float64)
    return float(raw[0]), raw[1:].copy()
