"""k1h_model.py — CPU-side harness of the K1h kernel (ntcard_amd/csrc/gen_k1h.py): runs the GENERATED instruction list on the wave
emulator (k1h_asm.Emu) and models K1f, the fix-up kernel, in plain Python.  Test infrastructure only.

K1h's contract (gen_k1h.py): per wave it appends one hit-log key per sampled window it is sure about, adds valid-reads x windows to
F1, and leaves two bit arrays behind — dirty[tile][chunk][lane] (bit m: the 16-byte piece of read 64 m + lane holds a non-ACGTU
byte) and tie[tile][block][lane] (bit m: some window of that block has both strands flagged).  K1f owns every (read, block) whose
chunks b-2 .. b hold a dirty piece (all windows ending in the block: validity, F1 correction, hits) and, in the other blocks, the
windows both strands flag (hit through the canonical strand, nthash.hpp:275-279).
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ntcard_amd", "csrc"))

import gen_k1h  # noqa: E402
from k1h_asm import Emu  # noqa: E402
import orc  # noqa: E402

CODE2_OF = {ord("A"): 0, ord("C"): 1, ord("T"): 2, ord("G"): 3, ord("U"): 2, ord("a"): 0, ord("c"): 1, ord("t"): 2, ord("g"): 3, ord("u"): 2}
BASE_OF_CODE2 = "ACTG"


def build_table(k, r_bits, s_bits, gap=0):
    """[2][ng][64] uint32: 3 bases per entry (code2, base t of the group in bits 2t+1:2t); word = low r_bits bits of the strand's
    closed-form term (nthash.hpp:220-239) | its bit 62 << r_bits (s_bits == 7: tells sample 1 from sample 0)"""
    L = orc.lib()
    ng = gen_k1h.n_groups(k)
    out = np.zeros((2, ng, 64), dtype=np.uint32)
    for g in range(ng):
        for val in range(64):
            f = r = 0
            for t in range(3):
                i = 3 * g + t
                if i >= k:
                    break
                if gap and (k - gap) // 2 <= i < (k - gap) // 2 + gap:
                    continue  # a don't-care position of the spaced seed
                base = BASE_OF_CODE2[(val >> (2 * t)) & 3]
                f ^= L.orc_srol(L.orc_seed(ord(base)), k - 1 - i)
                r ^= L.orc_srol(L.orc_seed_comp(ord(base)), i)
            for st, x in ((0, f), (1, r)):
                ext = (x >> (63 - s_bits)) & ((1 << (s_bits - 7)) - 1)  # s_bits >= 8: the hash bits between the top 8 and ntComp's last pattern bit
                assert r_bits + 1 + (s_bits - 7) <= 32
                out[st, g, val] = (x & ((1 << r_bits) - 1)) | (((x >> 62) & 1) << r_bits) | (ext << (r_bits + 1))
    return out


class Layout:
    """flat device memory for the emulator"""

    def __init__(self, nbytes):
        self.mem = np.zeros(nbytes, dtype=np.uint8)
        self.top = 256

    def alloc(self, nbytes, align=256):
        self.top = (self.top + align - 1) // align * align
        a = self.top
        self.top += nbytes
        assert self.top <= self.mem.size
        return a


def run_k1h(tiles, n_reads, read_len, k, r_bits=16, s_bits=7, n_waves=2, log_regions=None, log_region_cap=4096, use_log=True, sus_cap=4096, gap=0, tails=None, sk_dirty_word=True):
    """-> dict(keys=uint32[], f1=int, dirty=[n_tiles][C][64], tie=[n_tiles][NB][64], insts=executed per wave)"""
    Cn = (read_len + 15) // 16
    n_tiles = (n_reads + 2047) // 2048
    phi = (k - 1) % 16
    NB = ((read_len - 1 + 16 - phi) >> 4) + 1
    if log_regions is None:
        log_regions = n_waves * 4
    lay = Layout(tiles.size + (1 << 22) + log_regions * log_region_cap * 4 + (4 << (r_bits + 1)) + n_waves * sus_cap * 16)
    a_karg = lay.alloc(256)
    a_tiles = lay.alloc(tiles.size)
    a_log = lay.alloc(log_regions * log_region_cap * 4)
    a_fill = lay.alloc(log_regions * 4)
    a_sk = lay.alloc(4 << (r_bits + 1))
    a_f1 = lay.alloc(64)   # (a timing build, K1H_EXP=timers, leaves its section clocks in f1[1 .. 4])
    a_dirty = lay.alloc(n_tiles * Cn * 256)
    a_tie = lay.alloc(n_tiles * NB * 256)
    a_sus = lay.alloc(n_waves * sus_cap * 16)
    a_susn = lay.alloc(n_waves * 4)
    a_skd = lay.alloc(64)  # the engine's "direct atomics happened" word (round 6)
    a_tails = lay.alloc(n_tiles * 64) if tails is not None else 0  # ragged batch: uint32 [n_tiles][16], reads of the tile with more than d bases in their last piece
    mem = lay.mem
    mem[a_tiles:a_tiles + tiles.size] = tiles
    m32, m64 = mem.view(np.uint32), mem.view(np.uint64)
    K = gen_k1h.KARG
    for name, val in (("tiles", a_tiles), ("log", a_log), ("log_fill", a_fill), ("sketch0", a_sk), ("f1", a_f1), ("dirty", a_dirty), ("tie", a_tie), ("sus", a_sus),
                      ("sus_count", a_susn)):
        m64[(a_karg + K[name]) // 8] = val
    nv_last = n_reads - (n_tiles - 1) * 2048
    for name, val in (("n_tiles", n_tiles), ("n_chunks", Cn), ("read_len", read_len), ("nv_last", nv_last), ("key_base", 0),
                      ("rmask2", (2 << r_bits) - 1), ("log_regions", log_regions if use_log else 0), ("log_region_cap", log_region_cap),
                      ("s_bits", s_bits)):
        m32[(a_karg + K[name]) // 4] = val
    total = n_tiles * NB
    bpw = (total + n_waves - 1) // n_waves
    m32[(a_karg + K["blocks_per_wave"]) // 4] = bpw
    m32[(a_karg + K["nb_magic"]) // 4] = (1 << 32) // NB
    m32[(a_karg + K["sus_cap"]) // 4] = sus_cap
    m64[(a_karg + K["tails"]) // 8] = a_tails
    m64[(a_karg + K["sk_dirty"]) // 8] = a_skd if sk_dirty_word else 0
    if tails is not None:
        m32[a_tails // 4: a_tails // 4 + n_tiles * 16] = np.asarray(tails, dtype=np.uint32).reshape(-1)
    lds = np.zeros(gen_k1h.LDS_BYTES, dtype=np.uint8)
    tab = build_table(k, r_bits, s_bits, gap)
    lds.view(np.uint32)[gen_k1h.TABLE_OFF // 4: gen_k1h.TABLE_OFF // 4 + tab.size] = tab.reshape(-1)
    prog = gen_k1h.Gen(k, 7 if s_bits == 7 else 8, gap).build(emu=True)
    insts = []
    for w in range(n_waves):
        e = Emu(prog, mem, lds)
        rng = np.random.default_rng(w)
        e.V[:] = rng.integers(0, 1 << 32, size=e.V.shape, dtype=np.uint64).astype(np.uint32)  # registers start as garbage
        e.S[0], e.S[1] = a_karg & 0xFFFFFFFF, a_karg >> 32
        e.S[2], e.S[3], e.S[4] = w, n_waves, (w % gen_k1h.WAVES) * gen_k1h.WAREA
        # the blocks the wave owns: the kernel's prologue weighs a workgroup's waves by SIMD sharing — here every other wave gets 3 parts to its neighbour's 2
        wts = [2 + (i & 1) for i in range(n_waves)]
        e.S[5], e.S[6] = total * sum(wts[:w]) // sum(wts), total * sum(wts[:w + 1]) // sum(wts)
        e.run()
        insts.append(e.executed)
    fill = m32[a_fill // 4: a_fill // 4 + log_regions]
    keys = [m32[a_log // 4 + rgn * log_region_cap: a_log // 4 + rgn * log_region_cap + int(fill[rgn])] for rgn in range(log_regions)]
    keys = np.concatenate(keys) if keys else np.zeros(0, dtype=np.uint32)
    sk = m32[a_sk // 4: a_sk // 4 + (2 << r_bits)].copy()
    susn = m32[a_susn // 4: a_susn // 4 + n_waves].copy()
    sus = [m32[a_sus // 4 + w * sus_cap * 4: a_sus // 4 + (w * sus_cap + min(int(susn[w]), sus_cap)) * 4].reshape(-1, 4).copy() for w in range(n_waves)]
    return dict(sus=np.concatenate(sus) if sus else np.zeros((0, 4), dtype=np.uint32), sus_overflow=bool(np.any(susn == 0xFFFFFFFF)), keys=keys.copy(), sketch=sk, sk_dirty=int(m32[a_skd // 4]), f1=int(m64[a_f1 // 8]), f1_raw=m64[a_f1 // 8: a_f1 // 8 + 8].copy(), dirty=m32[a_dirty // 4: a_dirty // 4 + n_tiles * Cn * 64].reshape(n_tiles, Cn, 64).copy(),
                tie=m32[a_tie // 4: a_tie // 4 + n_tiles * NB * 64].reshape(n_tiles, NB, 64).copy(), insts=insts, NB=NB, C=Cn)


def flags_of(fh, rh, s_bits):
    """K1h's per-strand candidate flags from the top 8 bits of both strands (gen_k1h.Gen.strand_flags)"""
    f8, r8 = fh >> 56, rh >> 56
    if s_bits == 7:
        def fl(x):
            return (x >> 1) == 0x3f, (x >> 1) >= 0x3f, x == 1, x >= 1
        fa, fg, fb, fnz = fl(f8)
        ra, rg, rb, rnz = fl(r8)
        return (fa and rg) or (fb and rnz), (ra and fg) or (rb and fnz)
    fa, fg, fb = f8 == 0x7f, f8 >= 0x7f, f8 == 0
    ra, rg, rb = r8 == 0x7f, r8 >= 0x7f, r8 == 0
    return (fa and rg) or fb, (ra and fg) or rb


def window_hashes(win, k, gap):
    """-> (ok, fh, rh) of one window; spaced seed: the don't-care positions' terms XORed out (nthash.hpp:641-646)"""
    L = orc.lib()
    fh, rh, bad = C.c_uint64(), C.c_uint64(), C.c_uint()
    if not L.orc_window_hash(win, k, C.byref(fh), C.byref(rh), C.byref(bad)):
        return False, 0, 0
    f, r = fh.value, rh.value
    for i in range((k - gap) // 2, (k - gap) // 2 + gap):
        f ^= L.orc_srol(L.orc_seed(win[i]), k - 1 - i)
        r ^= L.orc_srol(L.orc_seed_comp(win[i]), i)
    return True, f, r


def k1f_model(reads, read_len, k, r_bits, s_bits, dirty, tie, sus=None, sus_overflow=False, gap=0):
    """-> (keys list, f1_sub) of the fix-up kernel.
    F1: every window with a non-ACGTU byte leaves it (they all lie in dirty-affected blocks).
    Hits: suspects (K1h resolved them, K1f checks their bytes) — or, when the suspect list overflowed, every window of every
    dirty-affected block; plus the windows of tie blocks both strands flag."""
    L = orc.lib()
    phi = (k - 1) % 16
    NB = ((read_len - 1 + 16 - phi) >> 4) + 1
    Cn = (read_len + 15) // 16
    keys, f1_sub = [], 0
    fh, rh, bad = C.c_uint64(), C.c_uint64(), C.c_uint()

    def key_of(h):
        if (h >> (63 - s_bits)) == 1:
            return h & ((1 << r_bits) - 1)
        if (h >> (64 - s_bits)) == (1 << (s_bits - 1)) - 1:
            return (1 << r_bits) + (h & ((1 << r_bits) - 1))
        return None

    for r, seq in enumerate(reads):
        t, lane, m = r // 2048, (r % 2048) % 64, (r % 2048) // 64
        for b in range(NB):
            aff = any(0 <= c < Cn and (int(dirty[t, c, lane]) >> m) & 1 for c in (b - 2, b - 1, b))
            tb = (int(tie[t, b, lane]) >> m) & 1
            for e in range(max(16 * b - 16 + phi, k - 1), min(16 * b + phi - 1, len(seq) - 1) + 1):  # (a ragged batch: the read's own length)
                win = seq[e - k + 1: e + 1]
                ok, fv, rv = window_hashes(win, k, gap)
                if not ok:
                    assert aff
                    f1_sub += 1
                    continue
                if not sus_overflow:
                    continue
                cf, cr = flags_of(fv, rv, s_bits)
                if aff or (tb and cf and cr):  # slow path: every window of a dirty-affected block, and the both-flag windows of tie blocks
                    kk = key_of(min(fv, rv))
                    if kk is not None:
                        keys.append(kk)
    if sus is not None and not sus_overflow:
        for x, t, rw, mark in sus:
            r, w = int(t) * 2048 + (int(rw) & 2047), int(rw) >> 11
            win = reads[r][w: w + k]
            ok, fv, rv = window_hashes(win, k, gap)
            if ok:
                kk = key_of(min(fv, rv))
                assert kk is not None or s_bits > 7, (r, w)  # (s_bits >= 8: the walk tests a prefix of the patterns, a suspect may turn out to be none)
                # K1f's fast path: K1h's own verdict stands for every suspect that is not marked as a tie (mark bit 2) — its counter index, and "no hit"
                # (mark bit 1) where the pattern fails below the 8-bit prefix; a tie is re-derived from the bytes (the entry may be of the wrong strand).
                # Mark bits 4 .. 6: the read's dirty bits in the chunks the window starts in and the two behind it — all K1f reads besides the entry.
                tl, lane, m = r // 2048, (r % 2048) % 64, (r % 2048) // 64
                blk = (w + k - 1 - phi) // 16 + 1
                c0 = w // 16
                for j in range(3):
                    if c0 + j < Cn and 16 * j < (w % 16) + k:
                        assert (int(mark) >> (4 + j)) & 1 == (int(dirty[tl, c0 + j, lane]) >> m) & 1, (r, w, j, int(mark))
                if not int(mark) & 4:
                    assert (int(mark) & 2 != 0) == (kk is None), (r, w, int(mark))
                    assert kk is None or int(x) == kk, (r, w, int(x), kk)
                else:
                    assert (int(tie[tl, blk, lane]) >> m) & 1, (r, w)
                if kk is not None:
                    keys.append(kk)
    return keys, f1_sub
