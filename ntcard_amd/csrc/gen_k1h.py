#!/usr/bin/env python3
"""gen_k1h.py — emits the K1h kernel body (ntc_sketch_k1h.hip): ONE wave per 2048-read tile does everything.

What it computes is ntRead + ntComp (ntcard.cpp:132-158) for one k of 12 .. 32 over a batch in the TILED slot layout
(include/ntcard_hip.h, ntc_submit_tiled_device): canonical ntHash of every window of k ACGTU bases (nthash.hpp:242-257,275-279),
ntComp's two sampling patterns on its top bits, one hit-log entry (= deferred ++t_Counter[...]) per sampled window, F1.

Why a second tiled kernel (round 4): K1c (ntc_sketch_ts.hip) splits a tile over four specialised waves; 36 % of its wave-cycles are
hand-shake waits and every role is a lone wave on its SIMD (one instruction per ~5 clk).  K1h removes the hand-shakes: a wave owns
a tile, streams it chunk by chunk (16 bases x 2048 reads), and nothing it does waits for another wave.  The price is a live set the
compiler cannot hold (62 state planes + 96 base planes + 32 registers of loads in flight), hence this generator: explicit physical
registers, the kernel body as one assembly string, and an emulator (k1h_asm.py) that runs the very same instruction list on the CPU.

Per chunk iteration (flat over the wave's tiles; block n of a tile = the 16 window ENDS e in [16 n - 16 + phi, 16 n + phi), phi =
(k - 1) mod 16, so that the 16 windows of a block all START in one chunk: the resolve pass then reads wave-uniform ring slots):
  * 16 walk steps, both strands bit-sliced (k1h_terms.py: one VGPR = one bit of the 31-bit hash half for 32 reads).  Per step: function
    planes of the incoming / outgoing base, 62 in-place state updates, then — only for steps that complete a window — the sample
    test.  With both strands in one wave the test is EXACT per strand: cf = this window is sampled and the forward strand is
    canonical, cr = ... reverse; cf & cr = the top bits tie (settled by K1f from the bytes).  Candidates go to an LDS queue as
    (32-read hit word, lane | strand | window offset).
  * interleaved with the walk: the NEXT chunk's raw bytes (8 x buffer_load_dwordx4 in flight) are packed to 2 bits per base into the
    registers of planes that have just left the window, validity (non-ACGTU bytes) per 16-byte piece -> one dirty bit per read.
  * resolve passes (subroutine, whenever the queue holds 64 items, and to empty it at the end of a block): lowest set bit of each
    item -> the read's packed words of three ring slots -> 64-bit window code -> 3 bases per look-up in a 32-bit closed-form table
    of the candidate's own strand (the canonical one): low rBits bits of the hash + the bit that tells the two samples apart ->
    counter index -> hit log (coalesced) or, without a log, the literal device atomic (ntcard.cpp:142-143).
  * end of block: packed words -> LDS ring, 32 x 32 bit transpose in place, register rotation by v_swap.
Reads with non-ACGTU bytes and top-bit ties are NOT resolved here: every candidate of a block whose three chunks hold a dirty piece
of that read is dropped, the dirty bits and the tie bits go to two arrays at fixed positions, and K1f (ntc_sketch_k1h.hip,
fixup_kernel) recomputes exactly those (read, block) pairs from the raw bytes with ntHashIterator's semantics
(ntHashIterator.hpp:59-86).  F1 here counts every window; K1f takes the invalid ones back.

Nothing is copied from the reference: the four seeds are its constants (nthash.hpp:25-28), everything else is derived (k1h_terms.py).
"""
import os
import sys

from k1h_terms import step_terms, ttbl, g_of, hseed, rol31, tt4, COMP, CODE2  # noqa: F401
from k1h_asm import Prog, v, s, vr, sr, schedule

WAVES = 8                    # waves per workgroup = tiles in flight per CU: two per SIMD (round 5; six in round 4, when a wave's ring was three whole chunks)
QSLOT = 2048                 # one QUARTER of a packed chunk: 4 bases x 2048 reads = [8 rows][64 lanes] dwords, byte t of row i = the read 64 (i + 8 t) + lane
RQ_MAX = 9                   # quarter-slots of the ring: 4 j + 5 are in use, j = (k - 1) div 16 (what 4 windows span, see quarter_enter)
RING_BYTES = RQ_MAX * QSLOT  # 18 KiB (round 4: three packed chunks, 24 KiB)
QCAP = 128                   # queue items, kept as three arrays — hit word, reverse-strand mask (dwords), meta (16 bits: that is what lets 128 items fit) — of QCAP + one
QSTRIDE = QCAP + 1           # dummy slot: a lane with nothing to queue writes there, so the writes need no exec mask (an exec write costs ~2 issue slots).
                             # 128 items, not 64 (round 5): a pass for want of room then always finds 64 items (it ran with 49 on average, the queue could never hold a full
                             # pass AND a step), 154 passes per two tiles instead of 175 (tools: the queue simulation behind DESIGN 5)
Q_META_OFF = QSTRIDE * 8     # byte offset of the meta array behind the two dword arrays
WAREA = (RING_BYTES + QSTRIDE * 10 + 15) // 16 * 16  # 19728 bytes per wave
TABLE_OFF = WAVES * WAREA    # 153728: [2 strands][NG][64] dwords
LDS_BYTES = 160 * 1024


def n_groups(k):
    return (k + 2) // 3


def table_bytes(k):
    return 2 * n_groups(k) * 256


# ---- register map ------------------------------------------------------------------------------------------
# VGPRs
V_LANE4, V_LANE16, V_QDUMMY, V_QBASE, V_WAVE4, V_ONE, V_EXP1, V_VMASK = range(0, 8)  # V_WAVE4: 4 x the wave's number in the launch
V_D0, V_D1, V_D2, V_DN, V_CMASK, V_TACC, V_CARRY0, V_CARRY1, V_SPARE1 = range(8, 17)
V_DMASK = 17       # reads of this block with a dirty piece in one of its three chunks (and valid): their candidates are SUSPECTS
V_WRAPF, V_WRAPR = V_SPARE1, 254  # second homes of the two state bits that wrap around in a walk step (walk_step)
V_F = 18           # F[31]   (register tuples — loads, 64-bit LDS items — must start at even registers on gfx90a+)
V_R = 49           # R[31]
V_H0 = 80          # chunk n-2 planes / P_next
V_H1 = 112
V_I = 144
V_RAW = 176        # 8 slots x 4
V_T = 208          # temps 208 .. 252
V_PX = V_T         # forward candidate item (x, y)
V_RX = V_T + 2     # reverse candidate item
V_SXF = V_T + 4    # the same for suspects
V_SXR = V_T + 6
V_TIEW = V_T + 1   # windows of the step whose strands tie on the top bits
V_T0 = V_T + 4     # scratch of the walk / test / pack: V_T0 .. V_T0 + 29 (the suspect pairs are written behind the test, whose scratch they share)
V_TP = V_T + 8     # scratch of the resolve pass: V_TP .. V_TP + 25; the transpose uses V_T .. V_T + 31
# constants that VOP3 instructions cannot take as literals (gfx9: one SGPR or inline constant per instruction, no 32-bit literal)
V_CMUL, V_CPERMLO, V_CPERMHI, V_CP16A, V_CP16B, V_CP8A, V_CP8B, V_CM4, V_CM2, V_CM1, V_CQMASK4, V_CBYTE = [V_T0 + 30 + i for i in range(12)]
assert V_CBYTE <= 254
VCONST = ((V_CMUL, 0x00820820), (V_CPERMLO, 0x0c0c0703), (V_CPERMHI, 0x07030c0c), (V_CP16A, 0x05040100), (V_CP16B, 0x07060302),
          (V_CP8A, 0x06020400), (V_CP8B, 0x07030501), (V_CM4, 0x0f0f0f0f), (V_CM2, 0x33333333), (V_CM1, 0x55555555), (V_CQMASK4, (QCAP - 1) * 4), (V_CBYTE, 0x703))
N_VGPRS = 255

# SGPRs: s0 .. S_BASE - 1 are left to the compiler (the asm statement's few inputs live there)
S_BASE = 26
_sn = [S_BASE]


def _salloc(n=1, align=1):
    while _sn[0] % align or (_sn[0] < 34 and _sn[0] + n > 32):  # s32 / s33 are the ABI's stack and frame pointer: reserved even in a kernel without a stack
        _sn[0] += 1
    r = _sn[0]
    _sn[0] += n
    return r


S_EXP0 = _salloc()
S_CHUNKB = _salloc()         # bytes of one tile's slots = C * 32768 (the product with the tile index is 64-bit)
S_DESC = _salloc(4, 4)       # tile being loaded
S_TILES = _salloc(2, 2)
S_SK = _salloc(2, 2)
S_SUS = _salloc(2, 2)        # this wave's region of the suspect list
S_DIRTY = _salloc(2, 2)
S_TIE = _salloc(2, 2)
S_LOGBASE = _salloc(2, 2)    # current log region
S_RET = _salloc(2, 2)
S_F1ACC = _salloc(2, 2)
S_TMP = _salloc(2, 2)        # 64-bit scratch
S_KARG = _salloc(2, 2)       # kept: the pointers needed once in a while (log, log_fill, f1) are re-read from the kernel arguments
S_F0, S_S0 = _salloc(), _salloc()  # first block this wave owns / walks
S_SUSOFF, S_SUSCAP = _salloc(), _salloc()  # bytes used / capacity of the suspect region
S_NTILES, S_C, S_L, S_NVLAST, S_KEYBASE, S_RMASK2, S_LOGREG, S_LOGCAP4 = [_salloc() for _ in range(8)]  # (loaded in this order)
S_NB, S_FEND = _salloc(), _salloc()
S_WF = _salloc()             # flat index (tile * NB + block) of the block being walked
S_LREG, S_LFILL4, S_USELOG = [_salloc() for _ in range(3)]
S_NWAVES = _salloc()
S_WT, S_WN = _salloc(), _salloc()      # block being walked
S_PT, S_PN, S_PREAL = _salloc(), _salloc(), _salloc()  # chunk being packed
S_QT, S_QN, S_QREAL = _salloc(), _salloc(), _salloc()  # chunk being loaded next
S_PSOFF, S_QSOFF = _salloc(), _salloc()
S_STEPMASK = _salloc()
S_RQ = [_salloc() for _ in range(RQ_MAX)]  # LDS addresses of the ring's quarter-slots in window order: S_RQ[i] holds byte i of the oldest window still to be resolved
S_QHEAD4, S_QTAIL4 = _salloc(), _salloc()
S_N, S_A, S_B, S_CC = _salloc(), _salloc(), _salloc(), _salloc()  # scalar scratch
S_SPARE = _salloc()
S_END = _sn[0]
assert S_END <= 100, S_END
# S_STEPMASK: bits 0 .. 15 the steps of this block that complete a window of every read; bits 16 .. 31 (ragged batches, K1hArgs.tails != NULL) the steps that
# end in the reads' last 16-base piece.  S_USELOG: bit 0 the hit log is in use, bit 8 the batch is ragged, bit 9 the queue is being emptied (read by the pass).
# (The compiler reserves s100 / s101 next to VCC, FLAT_SCRATCH and XNACK_MASK: nothing of the kernel may live there.)
F_USELOG, F_RAGGED, F_DRAIN = 0, 8, 9

S_TACC = (S_F1ACC, S_F1ACC + 1, S_SPARE, S_SUSOFF)  # timing build only (S_SUSCAP = the last time stamp): no F1, no suspects

# the asm statement's "s" operands, in order
INPUTS = ["karg_lo", "karg_hi", "wave_gid", "n_waves", "lds_wbase", "first_block", "end_block"]
# byte offsets in struct K1hArgs (ntc_kernels.hpp); the kernel reads them with scalar loads
KARG = dict(tiles=0, log=8, log_fill=16, sketch0=24, f1=32, dirty=40, tie=48, n_tiles=56, n_chunks=60, read_len=64, nv_last=68, key_base=72,
            rmask2=76, log_regions=80, log_region_cap=84, table=88, blocks_per_wave=104, nb_magic=108, sus=112, sus_count=120, sus_cap=128, s_bits=96, tails=144,
            sk_dirty=160)


class Gen:
    def __init__(self, k, sb_class=7, gap=0):
        assert 12 <= k <= 32 and 0 <= gap < k - 1
        self.k = k
        self.sb = sb_class
        self.gap = gap                      # spaced seed "1" x (k-g)/2 "0" x g "1" x rest (ntcard.cpp:407-413): the g middle positions do not enter the hash
        self.g0 = (k - gap) // 2            # first don't-care position
        self.g1 = self.g0 + gap - 1         # last one
        self.phi = (k - 1) % 16
        self.j = (k - 1) // 16      # window start chunk = n - 1 - j
        self.ng = n_groups(k)
        self.rq = 4 * self.j + 5    # quarter-slots of the ring in use: the 4 windows of a quarter start in one quarter of chunk n - 1 - j and end in chunk n
        self.nby = (k + 6) // 4     # bytes (quarters) of the ring a window can touch: k bases from base 0 .. 3 of its first quarter
        assert self.nby <= self.rq <= RQ_MAX
        self.p = Prog()
        self.uid = 0
        self.f_terms, self.r_terms = step_terms(k)
        # spaced seed (nthash.hpp:641-646: the don't-care positions' terms are XORed out of fh / rh): shifting the window by one base changes
        # two more terms per strand — the base that leaves the don't-care block at its low end comes in, the one that enters it at its high
        # end goes out:  F'[j] ^= S[j-(k-g0)](leave) ^ S[j-(k-1-g1)](enter),  R'[j] ^= Sc[j+1-g0](leave) ^ Sc[j-g1](enter)
        if gap:
            self.fg_terms = [(tt4(jj - (k - self.g0)), tt4(jj - (k - 1 - self.g1))) for jj in range(31)]
            self.rg_terms = [(tt4(jj + 1 - self.g0, True), tt4(jj - self.g1, True)) for jj in range(31)]
        self.cold = []                      # out-of-line fragments (emitted behind the chunk loop): the common path of a test falls through —
                                            # a taken branch costs a lone wave an instruction-buffer refill
        self.exp = set(x for x in os.environ.get("K1H_EXP", "").split(",") if x)  # timing experiments (tools/k1h_variant.sh): WRONG results

    def lbl(self, base):
        self.uid += 1
        return f"{base}{self.uid}"

    # ---- small helpers ----
    def bitop3(self, d, a, b, c, fn):
        self.p.i("v_bitop3_b32", d, a, b, c, mods=f"bitop3:0x{ttbl(fn):02x}")

    def call(self, target):
        """subroutine call: target returns with ret + 4"""
        self.p.i("s_getpc_b64", sr(S_RET, 2))
        self.p.i("s_branch", "@" + target)

    def ret(self):
        self.p.i("s_add_u32", s(S_RET), s(S_RET), 4)
        self.p.i("s_addc_u32", s(S_RET + 1), s(S_RET + 1), 0)
        self.p.i("s_setpc_b64", sr(S_RET, 2))

    def probe(self, sec):
        """timing build (K1H_EXP=timers): the clocks since the last probe go to section `sec` (0 walk + test + push, 1 pack, 2 resolve
        passes, 3 end of block); the four sums are added to f1[1 .. 4] at the end (bench.py --k1h-timers).  F1 itself is not kept."""
        if "timers" not in self.exp:
            return
        p = self.p
        p.i("s_memtime", sr(S_TMP, 2))
        p.i("s_waitcnt", "lgkmcnt(0)")
        p.i("s_sub_u32", s(S_N), s(S_TMP), s(S_SUSCAP))
        p.i("s_add_u32", s(S_TACC[sec]), s(S_TACC[sec]), s(S_N))
        p.i("s_mov_b32", s(S_SUSCAP), s(S_TMP))

    # ---- poly-A start state: the hash of k A s, the track every tile starts on ----
    def poly_a(self, strand):
        h = 0
        for t in range(self.k):
            if self.gap and self.g0 <= t <= self.g1:
                continue
            h ^= rol31(hseed("A"), self.k - 1 - t) if strand == "F" else rol31(hseed(COMP["A"]), t)
        return h

    # ---- one walk step -------------------------------------------------------------------------------------
    def plane_back(self, a, d):
        """register of plane bit 0 of the base d positions behind the one step a takes in (d = 0: the incoming base, d = k: the outgoing one)"""
        q = self.phi + a - d                 # position relative to the start of chunk n - 1
        if q >= 16:
            return V_I + 2 * (q - 16)
        if q >= 0:
            return V_H1 + 2 * q
        if q >= -16:
            return V_H0 + 2 * (q + 16)
        assert q == -17 and self.j == 1      # (n - 3, 15): saved before its registers took packed words
        return V_CARRY0

    def planes_of_step(self, a):
        """registers of the incoming / outgoing base planes of step a"""
        i0, o0 = self.plane_back(a, 0), self.plane_back(a, self.k)
        return i0, i0 + 1, o0, o0 + 1

    def fn_plane(self, cache, t4, b0, b1, tpool):
        """function plane of a 4-bit truth table over (b0, b1): -> (register or None, invert)"""
        if t4 == 0x0:
            return None, 0
        if t4 == 0xf:
            return None, 1
        if t4 == 0b1010:
            return b0, 0
        if t4 == 0b0101:
            return b0, 1
        if t4 == 0b1100:
            return b1, 0
        if t4 == 0b0011:
            return b1, 1
        inv = 0
        if t4 & 1:
            t4 ^= 0xf
            inv = 1
        if t4 not in cache:
            r = tpool.pop()
            if t4 == 0b1000:
                self.p.i("v_and_b32", v(r), v(b0), v(b1))
            elif t4 == 0b1110:
                self.p.i("v_or_b32", v(r), v(b0), v(b1))
            elif t4 == 0b0110:
                self.p.i("v_xor_b32", v(r), v(b0), v(b1))
            else:
                g = g_of(t4)
                self.bitop3(v(r), v(b0), v(b1), v(b1), lambda x, y, z: g(x, y))
            cache[t4] = r
        return cache[t4], inv

    def emit_update(self, dst, prev, x, y, inv):
        if x is None and y is None:
            if inv:
                self.p.i("v_not_b32", v(dst), v(prev))
            elif dst != prev:
                self.p.i("v_mov_b32", v(dst), v(prev))
        elif x is None or y is None:
            z = x if y is None else y
            if inv:
                self.bitop3(v(dst), v(prev), v(z), v(z), lambda a, b, c: 1 ^ a ^ b)
            else:
                self.p.i("v_xor_b32", v(dst), v(prev), v(z))
        else:
            self.bitop3(v(dst), v(prev), v(x), v(y), (lambda a, b, c: 1 ^ a ^ b ^ c) if inv else (lambda a, b, c: a ^ b ^ c))

    def emit_tail_step(self):
        """subroutine (S_B = 4 d).  Ragged batch: the step ends at base 16 (C - 1) + d of the reads.  n = tails[tile][d] reads of the tile are that long — a
        prefix of it (longest first): the candidate masks shrink to it (they only ever shrink until the tile ends: the steps in the last piece come in the
        order of d) and F1 takes n windows."""
        p = self.p
        T = V_T0
        p.label("tailsub")
        p.i("s_load_dwordx2", sr(S_TMP, 2), sr(S_KARG, 2), hex(KARG["tails"]))
        p.i("s_lshl_b32", s(S_A), s(S_WT), 6)
        p.i("s_add_u32", s(S_A), s(S_A), s(S_B))
        p.i("s_waitcnt", "lgkmcnt(0)")
        p.i("s_load_dword", s(S_A), sr(S_TMP, 2), s(S_A))
        p.i("s_waitcnt", "lgkmcnt(0)")
        if "timers" not in self.exp:
            p.i("s_add_u32", s(S_F1ACC), s(S_F1ACC), s(S_A))
            p.i("s_addc_u32", s(S_F1ACC + 1), s(S_F1ACC + 1), 0)
        # bit m of the prefix mask = (64 m + lane < n): cnt = clamp((n - lane + 63) >> 6, 0, 32) low bits (n <= 2048; n = 0: none)
        p.i("v_lshrrev_b32", v(T), 2, v(V_LANE4))
        p.i("v_sub_u32", v(T), s(S_A), v(T))                    # n - lane (may be negative: lane >= n)
        p.i("v_add_u32", v(T), 63, v(T))
        p.i("v_ashrrev_i32", v(T), 6, v(T))
        p.i("v_max_i32", v(T), 0, v(T))
        p.i("v_min_u32", v(T), 32, v(T))
        p.i("v_cmp_gt_u32_e32", "vcc", 32, v(T))
        p.i("v_lshlrev_b32", v(T + 1), v(T), v(V_ONE))
        p.i("v_add_u32", v(T + 1), -1, v(T + 1))
        p.i("v_cndmask_b32_e64", v(T + 1), -1, v(T + 1), "vcc")
        p.i("v_and_b32", v(V_VMASK), v(V_VMASK), v(T + 1))
        p.i("v_or3_b32", v(T), v(V_D0), v(V_D1), v(V_D2))
        self.bitop3(v(V_CMASK), v(T), v(V_VMASK), v(V_VMASK), lambda x, y, z: (1 ^ x) & y)
        p.i("v_and_b32", v(V_DMASK), v(T), v(V_VMASK))
        self.ret()

    def walk_step(self, a):
        i0, i1, o0, o1 = self.planes_of_step(a)
        tpool = [V_T0 + 1 + i for i in range(22)][::-1]  # function planes: V_T0 + 1 ..; V_T0 = the wrap-around copy
        cin, cout, clv, cen = {}, {}, {}, {}
        tmp = V_T0
        if self.gap:
            l0 = self.plane_back(a, self.k - self.g0)       # leaves the don't-care block (window index g0 -> g0 - 1)
            e0 = self.plane_back(a, self.k - 1 - self.g1)   # enters it (g1 + 1 -> g1)
        plan = {}
        for strand, terms, gterms in (("F", self.f_terms, getattr(self, "fg_terms", None)), ("R", self.r_terms, getattr(self, "rg_terms", None))):
            for jj in range(31):
                tin, tout = terms[jj]
                x, xi = self.fn_plane(cin, tin, i0, i1, tpool)
                y, yi = self.fn_plane(cout, tout, o0, o1, tpool)
                inv = xi ^ yi
                ops = [z for z in (x, y) if z is not None]
                if self.gap:
                    tl, te = gterms[jj]
                    u, ui = self.fn_plane(clv, tl, l0, l0 + 1, tpool)
                    w, wi = self.fn_plane(cen, te, e0, e0 + 1, tpool)
                    inv ^= ui ^ wi
                    ops += [z for z in (u, w) if z is not None]
                plan[strand, jj] = (ops, inv)

        def update(dst, prev, ops, inv):
            # dst = prev ^ ops... ^ inv, three inputs per instruction
            ops = list(ops)
            cur = prev
            first = True
            while first or ops:
                first = False
                take, ops = ops[:2], ops[2:]
                last = not ops
                self.emit_update(dst, cur, take[0] if take else None, take[1] if len(take) > 1 else None, inv if last else 0)
                cur = dst
        # The rotation is in place — F'[j] = F[j-1] ^ terms from the top, R'[j] = R[j+1] ^ terms from the bottom — but for the bit that wraps around (F's bit 0 takes
        # F's old bit 30, R's bit 30 R's old bit 0): that one alternates between its home register and a second one (V_WRAPF / V_WRAPR), written FIRST, while the
        # old value it needs is still there: no copy (round 5; two v_mov per step before).  Even steps find the bit at home and leave it in the second register,
        # odd steps the other way round; a block has 16 steps, so every block starts and ends at home.
        f0_old, f0_new = (V_F + 0, V_WRAPF) if a % 2 == 0 else (V_WRAPF, V_F + 0)
        update(f0_new, V_F + 30, *plan["F", 0])
        for jj in range(30, 1, -1):
            update(V_F + jj, V_F + jj - 1, *plan["F", jj])
        update(V_F + 1, f0_old, *plan["F", 1])
        r30_old, r30_new = (V_R + 30, V_WRAPR) if a % 2 == 0 else (V_WRAPR, V_R + 30)
        update(r30_new, V_R + 0, *plan["R", 30])
        for jj in range(0, 29):
            update(V_R + jj, V_R + jj + 1, *plan["R", jj])
        update(V_R + 29, r30_old, *plan["R", 29])

    # ---- sample test of one strand: planes a (top bits == sample-1 pattern), g (>=), b (== sample-0 pattern), nz ----
    def strand_flags(self, base, out, top=None):
        """base: first register of the strand's state (top: the register that holds its bit 30 right now, walk_step); out: dict of result registers a, g, b, nz;
        temps t1, t2"""
        t7, b6, b5, b4, b3, b2, b1, b0 = [v(base + 30 - i) for i in range(8)]
        if top is not None:
            t7 = v(top)
        ra, rg, rb, rnz, t1, t2 = [v(out[x]) for x in ("a", "g", "b", "nz", "t1", "t2")]
        if self.sb == 7:
            self.bitop3(t1, b6, b5, b4, lambda x, y, z: x & y & z)
            self.bitop3(t1, t1, b3, b2, lambda x, y, z: x & y & z)          # bits 6..2 all set
            self.bitop3(ra, t7, t1, b1, lambda x, y, z: (1 ^ x) & y & z)     # top 7 bits == 0111111
            self.bitop3(rg, t7, t1, b1, lambda x, y, z: x | (y & z))         # top 7 bits >= 0111111
            self.bitop3(t1, t7, b6, b5, lambda x, y, z: x | y | z)
            self.bitop3(t2, b4, b3, b2, lambda x, y, z: x | y | z)
            self.bitop3(t1, t1, t2, b1, lambda x, y, z: 1 ^ (x | y | z))     # top 7 bits == 0
            self.p.i("v_and_b32", rb, t1, b0)                                # top 8 bits == 00000001
            self.bitop3(rnz, t1, b0, b0, lambda x, y, z: (1 ^ x) | y)        # top 8 bits >= 1
        else:
            # s_bits >= 8: 8-bit prefixes of the patterns: 0x7f (sample 1) and 0x00 (sample 0); the pass checks the rest
            self.bitop3(t1, b6, b5, b4, lambda x, y, z: x & y & z)
            self.bitop3(t1, t1, b3, b2, lambda x, y, z: x & y & z)
            self.bitop3(t1, t1, b1, b0, lambda x, y, z: x & y & z)          # low 7 of the 8 all set
            self.bitop3(ra, t7, t1, t1, lambda x, y, z: (1 ^ x) & y)         # == 0x7f
            self.p.i("v_or_b32", rg, t7, t1)                                 # >= 0x7f
            self.bitop3(t1, t7, b6, b5, lambda x, y, z: x | y | z)
            self.bitop3(t2, b4, b3, b2, lambda x, y, z: x | y | z)
            self.bitop3(t1, t1, t2, b1, lambda x, y, z: x | y | z)
            self.bitop3(rb, t1, b0, b0, lambda x, y, z: 1 ^ (x | y))         # == 0x00
            # nz: any value is >= 0 — handled by the caller (cf = (a & g') | b)

    def flags_and_push(self, a):
        """candidate planes of step a (exact per strand), tie plane, then the two pushes"""
        p = self.p
        T = V_T0
        fo = dict(a=T + 1, g=T + 2, b=T + 3, nz=T + 4, t1=T + 9, t2=T + 10)
        ro = dict(a=T + 5, g=T + 6, b=T + 7, nz=T + 8, t1=T + 9, t2=T + 10)
        self.strand_flags(V_F, fo)
        self.strand_flags(V_R, ro, top=V_WRAPR if a % 2 == 0 else V_R + 30)  # (behind an even step R's bit 30 sits in its second register)
        cf, cr, tie = T + 11, T + 12, T + 13
        if self.sb == 7:
            p.i("v_and_b32", v(T + 9), v(fo["b"]), v(ro["nz"]))
            self.bitop3(v(cf), v(fo["a"]), v(ro["g"]), v(T + 9), lambda x, y, z: (x & y) | z)
            p.i("v_and_b32", v(T + 9), v(ro["b"]), v(fo["nz"]))
            self.bitop3(v(cr), v(ro["a"]), v(fo["g"]), v(T + 9), lambda x, y, z: (x & y) | z)
        else:
            self.bitop3(v(cf), v(fo["a"]), v(ro["g"]), v(fo["b"]), lambda x, y, z: (x & y) | z)
            self.bitop3(v(cr), v(ro["a"]), v(fo["g"]), v(ro["b"]), lambda x, y, z: (x & y) | z)
        p.i("v_and_b32", v(tie), v(cf), v(cr))
        self.bitop3(v(V_PX), v(cf), v(V_CMASK), v(tie), lambda x, y, z: x & y & (1 ^ z))
        self.bitop3(v(V_RX), v(cr), v(V_CMASK), v(tie), lambda x, y, z: x & y & (1 ^ z))
        sus = "timers" not in self.exp
        if sus:
            # suspects: candidates of reads with a dirty piece near by — resolved like the others, then parked for K1f, which knows the bytes
            # ... and windows whose strands tie on the top bits: items of their own, marked as ties (K1f re-derives a tie's hash from the bytes, so
            # the strand it is queued under does not matter; every other suspect keeps K1h's verdict and only has its dirty pieces looked at)
            p.i("v_and_b32", v(V_TIEW), v(tie), v(V_VMASK))   # (kept in a register a resolve pass does not touch)
            self.bitop3(v(V_SXF), v(cf), v(V_DMASK), v(tie), lambda x, y, z: x & y & (1 ^ z))
            self.bitop3(v(V_SXR), v(cr), v(V_DMASK), v(tie), lambda x, y, z: x & y & (1 ^ z))
            p.i("v_or_b32", v(V_TACC), v(V_TACC), v(V_TIEW))  # (the tie array serves K1f's slow path only)
        else:
            self.bitop3(v(V_TACC), v(V_TACC), v(tie), v(V_CMASK), lambda x, y, z: x | (y & z))
        self.push_items(a, V_PX, V_RX, 0)
        if sus:
            nosus, notdirty = self.lbl("nosus"), self.lbl("notdirty")
            p.i("v_or3_b32", v(T + 4), v(V_SXF), v(V_SXR), v(V_TIEW))
            p.i("v_cmp_ne_u32_e32", "vcc", 0, v(T + 4))
            p.i("s_cbranch_vccz", "@" + nosus)                 # (the common case: no suspect in this step)
            p.i("v_or_b32", v(T + 4), v(V_SXF), v(V_SXR))
            p.i("v_cmp_ne_u32_e32", "vcc", 0, v(T + 4))
            p.i("s_cbranch_vccz", "@" + notdirty)
            self.push_items(a, V_SXF, V_SXR, 1)
            p.label(notdirty)
            p.i("v_cmp_ne_u32_e32", "vcc", 0, v(V_TIEW))
            tiep = self.lbl("tiepush")
            p.i("s_cbranch_vccnz", "@" + tiep)                 # (rare: two windows per block — out of line)
            p.label(nosus)

            def cold_tie(tiep=tiep, nosus=nosus, a=a):
                p.label(tiep)
                self.push_items(a, V_TIEW, V_TIEW, 1, tie=1)
                p.i("s_branch", "@" + nosus)
            self.cold.append(cold_tie)

    def push_items(self, a, xf, xr, suspect, tie=0):
        """queue the hit words of step a (xf: forward-strand candidates, xr: reverse; a read is never in both — ties are masked out or ride in
        xf alone): ONE item per lane = (xf | xr, meta, xr); resolve passes first while the queue lacks room"""
        p = self.p
        T = V_TP
        chk, go = self.lbl("chk"), self.lbl("go")
        p.label(chk)
        p.i("v_or_b32", v(T + 2), v(xf), v(xr))
        p.i("v_cmp_ne_u32_e32", "vcc", 0, v(T + 2))
        p.i("s_bcnt1_i32_b64", s(S_A), "vcc")
        p.i("s_sub_u32", s(S_N), s(S_QTAIL4), s(S_QHEAD4))
        p.i("s_lshr_b32", s(S_N), s(S_N), 2)
        p.i("s_add_u32", s(S_N), s(S_N), s(S_A))
        p.i("s_cmp_le_u32", s(S_N), QCAP)
        full = self.lbl("qfull")
        p.i("s_cbranch_scc0", "@" + full)                        # (rare: about one step in three runs a pass first)

        def cold_pass(full=full, chk=chk):
            p.label(full)
            self.call("pass")                                    # (keeps V_T .. V_T + 7: the four hit words)
            p.i("s_branch", "@" + chk)
        self.cold.append(cold_pass)
        p.label(go)
        meta = ((2 * a) << 9) | (suspect << 15) | (tie << 8)     # bits 0 .. 7: 4 x lane (added below)
        p.i("v_mbcnt_lo_u32_b32", v(T + 1), "vcc_lo", 0)
        p.i("v_mbcnt_hi_u32_b32", v(T + 1), "vcc_hi", v(T + 1))
        p.i("v_lshl_add_u32", v(T + 1), v(T + 1), 2, s(S_QTAIL4))
        p.i("v_and_b32", v(T + 1), v(V_CQMASK4), v(T + 1))
        p.i("v_or_b32", v(T + 3), hex(meta), v(V_LANE4))
        p.i("v_cndmask_b32_e64", v(T + 1), v(V_QDUMMY), v(T + 1), "vcc")   # V_QDUMMY = 4 QCAP: the dummy slot
        p.i("v_lshrrev_b32", v(T + 4), 1, v(T + 1))                         # (the meta array is 16 bits wide)
        p.i("v_add_u32", v(T + 4), v(V_QBASE), v(T + 4))
        p.i("v_add_u32", v(T + 1), v(V_QBASE), v(T + 1))
        p.i("ds_write2_b32", v(T + 1), v(T + 2), v(xr), mods=f"offset0:0 offset1:{QSTRIDE}")
        p.i("ds_write_b16", v(T + 4), v(T + 3), mods=f"offset:{Q_META_OFF}")
        p.i("s_lshl2_add_u32", s(S_QTAIL4), s(S_A), s(S_QTAIL4))

    # ---- pack one group of 64 pieces: RAW slot i -> packed word in register dst; then reload the slot ----
    def pack_group(self, i, dst, reload):
        p = self.p
        r = V_RAW + 4 * i
        t = [V_T0 + 12 + 6 * (i % 3) + x for x in range(6)]  # three sets of scratch registers: the scheduler interleaves neighbouring groups
        # 2 x code of every byte: (ascii >> 1) & 3 = A 0, C 1, T / U 2, G 3 — the multiply-gather's input AND the index of the validity look-up:
        # the letter that code stands for (A, C, T, G), XORed with the byte, is zero or the case bit for exactly the bytes ACGTacgt.  (Round 4 looked
        # the letter up by byte & 7, which told U from T at the price of four more v_and per 16 bases; now a U is "dirty": its windows are K1f's,
        # which knows that U is a base — exact, slower for reads that are RNA.)
        x, e = t[4], t[5]
        for q in range(4):
            p.i("v_and_b32", v(t[q]), "0x06060606", v(r + q))
        for q in range(4):
            p.i("v_perm_b32", v(e), s(S_EXP0), v(V_EXP1), v(t[q]))
            if q == 0:
                p.i("v_xor_b32", v(x), v(e), v(r + q))
            else:
                self.bitop3(v(x), v(e), v(r + q), v(x), lambda a, b, c: (a ^ b) | c)
            if "nomul" in self.exp:
                p.i("v_lshlrev_b32", v(t[q]), 3, v(t[q]))
            else:
                p.i("v_mul_lo_u32", v(t[q]), v(t[q]), v(V_CMUL))
        p.i("v_perm_b32", v(t[0]), v(t[1]), v(t[0]), v(V_CPERMLO))
        p.i("v_perm_b32", v(t[2]), v(t[3]), v(t[2]), v(V_CPERMHI))
        p.i("v_or_b32", v(dst), v(t[0]), v(t[2]))
        p.i("v_and_b32", v(x), "0xdfdfdfdf", v(x))
        if "nocarry" in self.exp:
            p.i("v_or_b32", v(V_DN), v(V_DN), v(x))
        else:
            p.i("v_add_co_u32", v(t[5]), "vcc", -1, v(x))          # carry = (x != 0)
            p.i("v_addc_co_u32", v(V_DN), "vcc", v(V_DN), v(V_DN), "vcc")  # dirty bits, group m ends up at bit 31 - m
        if reload is not None and "noload" not in self.exp:
            soff, imm = reload
            p.i("buffer_load_dwordx4", vr(r, 4), v(V_LANE16), sr(S_DESC, 4), s(soff), mods=f"offen offset:{imm} nt")

    def issue_batch(self, b, soff_base):
        """loads of batch b (groups 8 b .. 8 b + 7) of the chunk at soffset soff_base: S_A / S_B as scratch soffsets"""
        p = self.p
        p.i("s_add_u32", s(S_A), s(soff_base), b * 8192)
        p.i("s_add_u32", s(S_B), s(soff_base), b * 8192 + 4096)
        for i in range(8):
            if "noload" not in self.exp:
                p.i("buffer_load_dwordx4", vr(V_RAW + 4 * i, 4), v(V_LANE16), sr(S_DESC, 4), s(S_A if i < 4 else S_B), mods=f"offen offset:{(i & 3) * 1024} nt")

    def pack_batch(self, b):
        """pack batch b of the chunk P into H0[8 b ..]; reload every slot with batch b + 1 of P, or — behind batch 3 — with batch 0 of
        the chunk Q (S_A / S_B hold the two soffsets: the caller of batch 3 has pointed S_DESC / S_CC at Q, or at any valid chunk)"""
        p = self.p
        skip, real = self.lbl("pk_skip"), self.lbl("pk_real")
        self.probe(0)
        if b < 3:
            p.i("s_add_u32", s(S_A), s(S_PSOFF), (b + 1) * 8192)
        else:
            p.i("s_mov_b32", s(S_A), s(S_CC))
        p.i("s_add_u32", s(S_B), s(S_A), 4096)
        p.i("s_cmp_eq_u32", s(S_PREAL), 1)
        notreal = self.lbl("pk_none")
        p.i("s_cbranch_scc0", "@" + notreal)

        def cold_none(notreal=notreal, skip=skip, b=b):
            p.label(notreal)
            for i in range(8):  # a chunk that does not exist: 'A's
                p.i("v_mov_b32", v(V_H0 + 8 * b + i), 0)
            if b == 3 and "noload" not in self.exp:          # the next chunk's first loads all the same
                for i in range(8):
                    p.i("buffer_load_dwordx4", vr(V_RAW + 4 * i, 4), v(V_LANE16), sr(S_DESC, 4), s(S_A if i < 4 else S_B), mods=f"offen offset:{(i & 3) * 1024} nt")
            p.i("s_branch", "@" + skip)
        self.cold.append(cold_none)
        p.label(real)
        # Eight loads are in flight, issued in slot order, and loads return in order: when at most four vector-memory operations are outstanding
        # the first four slots have arrived — and again, behind their four reloads, the other four.  vmcnt(0) would also wait for every STORE a
        # resolve pass has just issued (hit log, suspects: stores count in vmcnt on gfx9 and take a microsecond to be acknowledged).
        p.i("s_waitcnt", "vmcnt(4)")
        if "nopack" in self.exp:
            p.i("s_branch", "@" + skip)
        for i in range(8):
            if i == 4:
                p.i("s_waitcnt", "vmcnt(4)")
            self.pack_group(i, V_H0 + 8 * b + i, (S_A if i < 4 else S_B, (i & 3) * 1024))
        p.label(skip)
        self.probe(1)

    # ---- 32 x 32 bit transpose of the packed words in H0 into the planes I, in two parts (round 5) ----
    # The two byte-level stages (J = 16, 8: v_perm) run at the end of the block and leave I QUARTER-MAJOR: I[8 q + i] holds, in byte t, the packed
    # byte (4 bases) of quarter q of the read 64 (i + 8 t) + lane — exactly what one quarter-slot of the ring stores.  The three bit-level stages of
    # quarter q (J = 4, 2, 1 stay inside the 8 registers of a quarter, and inside a byte) run in quarter_enter(q), right behind the ring write and
    # before the walk takes in the first base of that quarter: the packed bytes of a chunk wait in the registers they will be planes in, so the ring
    # in LDS never holds more than the 4 j + 5 quarters the 4 windows of a quarter span (round 4: three whole chunks, 24 KiB -> 18 KiB per wave).
    def perm_rotate(self):
        """(H0, H1, I) <- (H1, I, quarter-major bytes of the packed words in H0): the first stage reads H0 into the temps, which frees H0 for H1's planes
        and H1 for I's; the second stage lands in I"""
        p = self.p
        A = [V_H0 + i for i in range(32)]
        B = [V_T + i for i in range(32)]
        O = [V_I + i for i in range(32)]
        for kk in range(32):  # J = 16: A -> B
            if kk & 16 == 0:
                p.i("v_perm_b32", v(B[kk]), v(A[kk + 16]), v(A[kk]), v(V_CP16A))
                p.i("v_perm_b32", v(B[kk + 16]), v(A[kk + 16]), v(A[kk]), v(V_CP16B))
        for i in range(32):
            p.i("v_mov_b32", v(V_H0 + i), v(V_H1 + i))
        for i in range(32):
            p.i("v_mov_b32", v(V_H1 + i), v(V_I + i))
        for kk in range(32):  # J = 8: B -> O
            if kk & 8 == 0:
                p.i("v_perm_b32", v(O[kk]), v(B[kk + 8]), v(B[kk]), v(V_CP8A))
                p.i("v_perm_b32", v(O[kk + 8]), v(B[kk + 8]), v(B[kk]), v(V_CP8B))

    def transpose_quarter(self, q):
        """I[8 q .. 8 q + 7]: quarter-major bytes -> the 8 planes of the bases 4 q .. 4 q + 3 (temps: V_T .. V_T + 7)"""
        p = self.p
        O = [V_I + 8 * q + i for i in range(8)]
        n = 0
        for J, msk in ((4, V_CM4), (2, V_CM2), (1, V_CM1)):  # in place: O[k] = bfi(m, x, y << J), O[k+J] = bfi(m, x >> J, y)
            for kk in range(8):
                if kk & J == 0:
                    x, y = O[kk], O[kk + J]
                    b0, b1 = V_T + 2 * (n % 4), V_T + 2 * (n % 4) + 1  # (four pairs of temps: the pairs of a stage are independent of each other)
                    n += 1
                    p.i("v_lshlrev_b32", v(b0), J, v(y))
                    p.i("v_lshrrev_b32", v(b1), J, v(x))
                    # (a bit-select as v_bitop3: two waves of a SIMD overlap those, v_bfi_b32 they take turns for)
                    self.bitop3(v(x), v(msk), v(x), v(b0), lambda m, a, b: (m & a) | ((1 ^ m) & b))
                    self.bitop3(v(y), v(msk), v(b1), v(y), lambda m, a, b: (m & a) | ((1 ^ m) & b))

    def drain(self):
        """resolve passes until the queue is empty"""
        p = self.p
        drain, drained = self.lbl("drain"), self.lbl("drained")
        p.i("s_bitset1_b32", s(S_USELOG), F_DRAIN)             # "the queue is being emptied", read by the pass
        p.label(drain)
        p.i("s_cmp_eq_u32", s(S_QTAIL4), s(S_QHEAD4))
        p.i("s_cbranch_scc1", "@" + drained)
        self.call("pass")
        p.i("s_branch", "@" + drain)
        p.label(drained)
        p.i("s_bitset0_b32", s(S_USELOG), F_DRAIN)

    def quarter_enter(self, q):
        """before step 4 q of block n.  The 4 windows of the coming quarter start in quarter q of chunk n - 1 - j and end, at the latest, in quarter q of
        chunk n: 4 j + 5 quarters.  Quarter q of chunk n takes the slot of quarter q - 1 of chunk n - 1 - j, which only the windows of the steps behind
        us needed — so the queue is emptied first (q = 0: the end of the block has done that).  A resolve pass therefore only ever sees candidates of
        ONE quarter, and S_RQ[0 .. 4 j + 4] are the slots of the bytes of their windows, in order."""
        p = self.p
        if q:
            self.drain()
            self.probe(0)
        p.i("s_mov_b32", s(S_A), s(S_RQ[0]))
        for i in range(self.rq - 1):
            p.i("s_mov_b32", s(S_RQ[i]), s(S_RQ[i + 1]))
        p.i("s_mov_b32", s(S_RQ[self.rq - 1]), s(S_A))
        p.i("v_add_u32", v(V_T0 + 2), s(S_A), v(V_LANE4))
        for i in range(0, 8, 2):  # (offsets in units of 64 dwords: the slot's rows are 256 bytes apart)
            p.i("ds_write2st64_b32", v(V_T0 + 2), v(V_I + 8 * q + i), v(V_I + 8 * q + i + 1), mods=f"offset0:{i} offset1:{i + 1}")
        self.transpose_quarter(q)
        self.probe(3)

    # ---- the resolve pass (subroutine) ----------------------------------------------------------------------
    def emit_pass(self):
        p = self.p
        T = V_TP
        item = T            # (x, y)
        m, rest, col, a0, a1, a2 = T + 2, T + 3, T + 4, T + 5, T + 6, T + 7
        lo, hi, mid, tb, key, key1, t1 = T + 8, T + 9, T + 10, T + 11, T + 12, T + 13, T + 14
        fld = [T + 15 + g for g in range(11)]
        p.label("pass")
        self.probe(0)
        # A resolving wave goes first on its SIMD: the pass is chains of LDS round trips and slow-class VALU instructions (v_bfe, v_lshl_add, v_perm
        # class: no overlap with the other wave's instructions, profiles/r05_ubench_op_classes.txt), the other wave of the SIMD is most likely walking
        # (full-rate v_bitop3 that fills any gap).  -4 % hash time (profiles/r05_k1h_priority_ab.txt); s_setprio around every run of slow-class
        # instructions gives the same, raising the WALK's priority nothing.
        if "noprio" not in self.exp:
            p.i("s_setprio", 3)
        if "nopass" in self.exp:
            p.i("s_mov_b32", s(S_QHEAD4), s(S_QTAIL4))
            self.ret()
        # Items -> lanes, in rounds: the first round takes min(64, count) items.  A word with more than one candidate goes back to the END of the queue
        # with its lowest bit cleared; when the round has taken everything the queue held, those words are the only items left and the lanes that
        # are still idle take them in the next round (and so on for third candidates).  Without this a queue that has to be EMPTIED — before every
        # quarter of a block, quarter_enter — would end in passes of a handful of second and third candidates: 13.9 passes per block measured
        # against 8.6 with whole-chunk ring slots; with it a quarter's ~110 items are two passes.
        x, zr = lo, hi                                        # the item's hit word and reverse-strand mask (until the requeue below: lo / hi are free until the window's bytes are in)
        y = item + 1                                          # its meta (16 bits)
        p.i("s_mov_b32", s(S_CC), 0)                          # lanes filled so far
        p.label("passload")
        p.i("s_sub_u32", s(S_N), s(S_QTAIL4), s(S_QHEAD4))
        p.i("s_lshr_b32", s(S_N), s(S_N), 2)
        p.i("s_sub_u32", s(S_B), 64, s(S_CC))
        p.i("s_min_u32", s(S_N), s(S_N), s(S_B))
        p.i("s_bfm_b64", "exec", s(S_N), s(S_CC))
        p.i("s_cmp_eq_u32", s(S_N), 64)
        p.i("s_cselect_b64", "exec", -1, "exec")
        p.i("s_lshl_b32", s(S_B), s(S_CC), 2)
        p.i("s_sub_u32", s(S_B), s(S_QHEAD4), s(S_B))        # lane l takes item l - filled
        p.i("v_add_u32", v(t1), s(S_B), v(V_LANE4))
        p.i("v_and_b32", v(t1), v(V_CQMASK4), v(t1))
        p.i("v_lshrrev_b32", v(m), 1, v(t1))
        p.i("v_add_u32", v(m), v(V_QBASE), v(m))
        p.i("v_add_u32", v(t1), v(V_QBASE), v(t1))
        p.i("ds_read2_b32", vr(x, 2), v(t1), mods=f"offset0:0 offset1:{QSTRIDE}")
        p.i("ds_read_u16", v(y), v(m), mods=f"offset:{Q_META_OFF}")
        p.i("s_lshl2_add_u32", s(S_QHEAD4), s(S_N), s(S_QHEAD4))
        p.i("s_add_u32", s(S_CC), s(S_CC), s(S_N))
        p.i("s_waitcnt", "lgkmcnt(0)")
        p.i("v_ffbl_b32", v(m), v(x))
        p.i("v_add_u32", v(rest), -1, v(x))
        p.i("v_and_b32", v(rest), v(rest), v(x))
        p.i("v_lshrrev_b32", v(tb), v(m), v(zr))
        p.i("v_and_b32", v(tb), 1, v(tb))                     # strand of the read being resolved
        # ---- what is left of each word goes to the back of the queue (exec = the items of this round) ----
        p.i("v_cmp_ne_u32_e32", "vcc", 0, v(rest))
        p.i("s_bcnt1_i32_b64", s(S_A), "vcc")
        p.i("v_mbcnt_lo_u32_b32", v(t1), "vcc_lo", 0)
        p.i("v_mbcnt_hi_u32_b32", v(t1), "vcc_hi", v(t1))
        p.i("v_lshl_add_u32", v(t1), v(t1), 2, s(S_QTAIL4))
        p.i("v_and_b32", v(t1), v(V_CQMASK4), v(t1))
        p.i("v_cndmask_b32_e64", v(t1), v(V_QDUMMY), v(t1), "vcc")   # (a spent word goes to the dummy slot)
        p.i("v_lshrrev_b32", v(col), 1, v(t1))
        p.i("v_add_u32", v(col), v(V_QBASE), v(col))
        p.i("v_add_u32", v(t1), v(V_QBASE), v(t1))
        p.i("ds_write2_b32", v(t1), v(rest), v(zr), mods=f"offset0:0 offset1:{QSTRIDE}")
        p.i("ds_write_b16", v(col), v(y), mods=f"offset:{Q_META_OFF}")
        p.i("s_lshl2_add_u32", s(S_QTAIL4), s(S_A), s(S_QTAIL4))
        p.i("s_bitcmp1_b32", s(S_USELOG), F_DRAIN)            # another round?  only when the queue is being emptied (drain: a pass run for want of room
        p.i("s_cbranch_scc0", "@passloaded")                  # leaves the words it puts back to the next one, which costs nothing) ...
        p.i("s_cmp_lt_u32", s(S_CC), 64)                      # ... with idle lanes ...
        p.i("s_cbranch_scc0", "@passloaded")
        p.i("s_cmp_lg_u32", s(S_QTAIL4), s(S_QHEAD4))         # ... and items (then the queue holds nothing but the words this round has put back)
        p.i("s_cbranch_scc1", "@passload")
        p.label("passloaded")
        p.i("s_bfm_b64", "exec", s(S_CC), 0)
        p.i("s_cmp_eq_u32", s(S_CC), 64)
        p.i("s_cselect_b64", "exec", -1, "exec")
        # the window's bytes: byte i lies in quarter-slot S_RQ[i], row m & 7, lane, byte m >> 3 (every byte is loaded into a register of its own,
        # zero-extended: a d16 load keeps or clears the other half of its register depending on the ECC mode of the part)
        nby = self.nby
        addr = fld[:nby]                                      # (address registers now, fields later)
        byt = [a0, a1, a2, lo, hi, key, key1, fld[9], fld[10]][:nby]
        p.i("v_and_b32", v(col), 0xff, v(y))                  # 4 x lane
        p.i("v_lshl_or_b32", v(a0), v(m), 11, v(m))
        p.i("v_lshrrev_b32", v(a0), 3, v(a0))
        p.i("v_and_or_b32", v(col), v(a0), v(V_CBYTE), v(col))  # (m & 7) << 8 | 4 lane | m >> 3
        for i in range(nby):
            p.i("v_add_u32", v(addr[i]), s(S_RQ[i]), v(col))
        for i in range(nby):
            p.i("ds_read_u8", v(byt[i]), v(addr[i]))
        p.i("v_bfe_u32", v(t1), v(y), 9, 3)                   # shift = 2 x (window start within its quarter)
        p.i("v_mul_u32_u24", v(tb), hex(self.ng * 256), v(tb))
        p.i("v_add_u32", v(tb), TABLE_OFF, v(tb))             # the strand's table
        p.i("s_waitcnt", "lgkmcnt(0)")

        def word(bs):
            """bs[0] = the bytes bs (registers, zero-extended) as one dword, lowest first"""
            if len(bs) >= 2:
                p.i("v_lshl_or_b32", v(bs[0]), v(bs[1]), 8, v(bs[0]))
            if len(bs) == 4:
                p.i("v_lshl_or_b32", v(bs[2]), v(bs[3]), 8, v(bs[2]))
            if len(bs) >= 3:
                p.i("v_lshl_or_b32", v(bs[0]), v(bs[2]), 16, v(bs[0]))
            return bs[0]
        w0 = word(byt[0:4])
        w1 = word(byt[4:8]) if nby > 4 else None
        w2 = byt[8] if nby > 8 else None
        if w1 is None:
            p.i("v_lshrrev_b32", v(lo), v(t1), v(w0))         # (k <= 13: the window lies in four bytes)
        else:
            p.i("v_alignbit_b32", v(lo), v(w1), v(w0), v(t1))  # bases 0 .. 15 of the window
            if self.k > 16:
                if w2 is None:
                    p.i("v_lshrrev_b32", v(hi), v(t1), v(w1))
                else:
                    p.i("v_alignbit_b32", v(hi), v(w2), v(w1), v(t1))  # bases 16 .. 31
        # 3 bases per look-up: field g = bits [6 g, 6 g + 6) of hi:lo
        for g in range(self.ng):
            bit = 6 * g
            nb = min(3, self.k - 3 * g) * 2
            if bit + nb <= 32:
                src, off = lo, bit
            elif bit >= 32:
                src, off = hi, bit - 32
            else:
                if bit == 30:
                    p.i("v_alignbit_b32", v(mid), v(hi), v(lo), 30)
                src, off = mid, bit - 30
            p.i("v_bfe_u32", v(fld[g]), v(src), off, nb)
            p.i("v_lshl_add_u32", v(fld[g]), v(fld[g]), 2, v(tb))
        for g in range(self.ng):
            p.i("ds_read_b32", v(fld[g]), v(fld[g]), mods=f"offset:{g * 256}")
        p.i("s_waitcnt", "lgkmcnt(0)")
        words = list(fld[: self.ng])
        while len(words) > 1:
            if len(words) >= 3:
                a, b, c = words[:3]
                self.bitop3(v(a), v(a), v(b), v(c), lambda q, r, t: q ^ r ^ t)
                words = words[3:] + [a]
            else:
                a, b = words
                p.i("v_xor_b32", v(a), v(a), v(b))
                words = [a]
        w = words[0]
        p.i("v_and_b32", v(key), s(S_RMASK2), v(w))          # low rBits bits + the sample bit right above them
        p.i("v_add_u32", v(key), s(S_KEYBASE), v(key))
        sflag = a0                                            # (the ring words are spent) 0: a hit to log, bit 0: suspect, bit 1: no hit
        p.i("v_bfe_u32", v(sflag), v(y), 15, 1)
        if self.sb != 7:
            # s_bits >= 8: the walk only saw the 8-bit prefixes of ntComp's patterns; the table word carries the s_bits - 7 hash bits below them
            # (bits 55 .. 63 - s_bits, above the sample bit).  Sample 1 (0 1..1) needs all but the lowest of them set, sample 0 (0..0 1)
            # exactly the lowest; S_SPARE = 2^(s_bits - 7) - 1
            smp, ext = a1, a2
            p.i("s_bcnt1_i32_b32", s(S_B), s(S_RMASK2))      # r_bits + 1
            p.i("v_lshrrev_b32", v(ext), s(S_B), v(w))
            p.i("s_sub_u32", s(S_B), s(S_B), 1)
            p.i("v_lshrrev_b32", v(smp), s(S_B), v(w))
            p.i("v_and_b32", v(smp), 1, v(smp))
            p.i("v_or_b32", v(mid), 1, v(ext))
            p.i("v_cmp_ne_u32_e64", sr(S_TMP, 2), s(S_SPARE), v(mid))   # sample 1 fails
            p.i("v_cmp_ne_u32_e32", "vcc", 1, v(ext))                   # sample 0 fails
            p.i("v_cndmask_b32_e64", v(mid), 0, 2, sr(S_TMP, 2))
            p.i("v_cndmask_b32_e64", v(ext), 0, 2, "vcc")
            p.i("v_cmp_ne_u32_e32", "vcc", 0, v(smp))
            p.i("v_cndmask_b32_e64", v(mid), v(ext), v(mid), "vcc")     # 2 where the candidate's own pattern fails
            p.i("v_or_b32", v(sflag), v(sflag), v(mid))
        # ---- clean hits: append to the hit log ----
        nolog, logged = self.lbl("nolog"), self.lbl("logged")
        p.label("logswitch_back")
        if self.sb == 7:
            p.i("v_cmp_ne_u32_e32", "vcc", 0, v(sflag))       # vcc = suspects
        else:
            p.i("v_and_b32", v(t1), 1, v(sflag))
            p.i("v_cmp_ne_u32_e32", "vcc", 0, v(t1))
        p.i("v_cmp_eq_u32_e64", sr(S_TMP, 2), 0, v(sflag))    # S_TMP = hits to log
        p.i("s_bcnt1_i32_b64", s(S_B), sr(S_TMP, 2))
        p.i("s_bitcmp1_b32", s(S_USELOG), F_USELOG)
        p.i("s_cbranch_scc0", "@" + nolog)
        p.i("s_lshl2_add_u32", s(S_A), s(S_B), s(S_LFILL4))
        p.i("s_cmp_le_u32", s(S_A), s(S_LOGCAP4))
        p.i("s_cbranch_scc0", "@logswitch")
        p.i("v_mbcnt_lo_u32_b32", v(t1), s(S_TMP), 0)
        p.i("v_mbcnt_hi_u32_b32", v(t1), s(S_TMP + 1), v(t1))
        p.i("v_lshl_add_u32", v(t1), v(t1), 2, s(S_LFILL4))
        p.i("s_mov_b64", "exec", sr(S_TMP, 2))
        p.i("global_store_dword", v(t1), v(key), sr(S_LOGBASE, 2))
        p.i("s_mov_b32", s(S_LFILL4), s(S_A))
        p.i("s_branch", "@" + logged)
        p.label(nolog)
        p.label("direct")
        p.i("s_mov_b64", "exec", sr(S_TMP, 2))
        p.i("v_mov_b32", v(key1), 0)
        p.i("v_lshl_add_u64", vr(lo, 2), vr(key, 2), 2, sr(S_SK, 2))
        p.i("global_atomic_add", vr(lo, 2), v(V_ONE), "off")
        # (round 6) the sketch is no longer what the last reset or apply left: the engine's "direct atomics happened" word — the first apply behind a reset
        # WRITES its counts instead of adding them (ntc_apply.hip, count_kernel) unless this word says otherwise.  Rare path: engines without a log
        # (their pointer is NULL) and waves that ran out of log regions.
        skipflag = self.lbl("skipflag")
        p.i("s_load_dwordx2", sr(S_TMP, 2), sr(S_KARG, 2), hex(KARG["sk_dirty"]))
        p.i("s_waitcnt", "lgkmcnt(0)")
        p.i("s_cmp_eq_u64", sr(S_TMP, 2), 0)
        p.i("s_cbranch_scc1", "@" + skipflag)
        p.i("v_mov_b32", v(fld[2]), 0)
        p.i("global_store_dword", v(fld[2]), v(V_ONE), sr(S_TMP, 2))
        p.label(skipflag)
        p.label(logged)
        # ---- suspects: (key, tile, read | window << 11) -> this wave's region of the suspect list ----
        nosus, susfull = self.lbl("nosus"), self.lbl("susfull")
        p.i("s_cbranch_vccz", "@" + nosus)
        p.i("s_mov_b64", "exec", "vcc")
        p.i("s_bcnt1_i32_b64", s(S_B), "vcc")
        p.i("s_lshl_b32", s(S_B), s(S_B), 4)
        p.i("s_add_u32", s(S_A), s(S_SUSOFF), s(S_B))
        p.i("s_cmp_le_u32", s(S_A), s(S_SUSCAP))
        p.i("s_cbranch_scc0", "@" + susfull)
        p.i("v_mbcnt_lo_u32_b32", v(t1), "vcc_lo", 0)
        p.i("v_mbcnt_hi_u32_b32", v(t1), "vcc_hi", v(t1))
        p.i("v_lshl_add_u32", v(t1), v(t1), 4, s(S_SUSOFF))
        sx = fld[1]                                           # four consecutive registers from an even one: the entry
        assert sx % 2 == 0
        p.i("v_mov_b32", v(sx), v(key))
        p.i("v_mov_b32", v(sx + 1), s(S_WT))
        p.i("s_sub_u32", s(S_B), s(S_WN), 1 + self.j)          # the block's windows start in chunk n - 1 - j
        p.i("s_lshl_b32", s(S_B), s(S_B), 4)
        p.i("v_bfe_u32", v(sx + 2), v(y), 10, 4)
        p.i("v_add_u32", v(sx + 2), s(S_B), v(sx + 2))
        p.i("v_bfe_u32", v(sx + 3), v(y), 2, 6)               # lane
        p.i("v_lshl_or_b32", v(sx + 3), v(m), 6, v(sx + 3))   # read of the tile = 64 m + lane
        p.i("v_lshl_or_b32", v(sx + 2), v(sx + 2), 11, v(sx + 3))
        # marks: 2 = the candidate's own pattern fails below the walk's 8-bit prefix (s_bits >= 8: no hit, whatever its bytes); 4 = a tie (K1f re-derives
        # both strands from the bytes); bits 4 .. 6 = the read's dirty bits in the block's three chunks n - 2 .. n (the dirty words live in the lane the
        # item came from: fetched across lanes) — with them K1f reads nothing but the entry and the window's dirty pieces
        dl, d0, d1, d2 = fld[5], fld[6], fld[7], fld[8]
        p.i("s_mov_b64", sr(S_TMP, 2), "exec")
        p.i("s_mov_b64", "exec", -1)                           # (a lane that is switched off would hand out nothing)
        p.i("v_and_b32", v(dl), 0xff, v(y))                   # 4 x the item's lane
        dsrc = (V_D0, V_D1, V_D2) if self.j == 1 else (V_D1, V_D2)   # the chunks the window starts in and the ones behind it
        dregs = (d0, d1, d2)[:len(dsrc)]
        for dj, src in zip(dregs, dsrc):
            p.i("ds_bpermute_b32", v(dj), v(dl), v(src))
        p.i("s_waitcnt", "lgkmcnt(0)")
        p.i("s_mov_b64", "exec", sr(S_TMP, 2))
        p.i("v_and_b32", v(sx + 3), 2, v(sflag))
        p.i("v_bfe_u32", v(dl), v(y), 8, 1)
        p.i("v_lshl_or_b32", v(sx + 3), v(dl), 2, v(sx + 3))
        for j, dj in enumerate(dregs):
            p.i("v_lshrrev_b32", v(dj), v(m), v(dj))
            p.i("v_and_b32", v(dj), 1, v(dj))
            p.i("v_lshl_or_b32", v(sx + 3), v(dj), 4 + j, v(sx + 3))
        p.i("global_store_dwordx4", v(t1), vr(sx, 4), sr(S_SUS, 2))
        p.i("s_nop", 1)                                       # (a 128-bit store reads its data a little after it issues)
        p.label(susfull)                                      # no room: the count runs past the capacity, which K1f reads as "walk everything"
        p.i("s_mov_b32", s(S_SUSOFF), s(S_A))
        p.label(nosus)
        p.i("s_mov_b64", "exec", -1)
        p.i("s_waitcnt", "lgkmcnt(0)")
        self.probe(2)
        if "noprio" not in self.exp:
            p.i("s_setprio", 0)
        self.ret()
        # ---- rare: the log region is full ----
        p.label("logswitch")
        p.i("s_mov_b64", "vcc", "exec")
        self.store_log_fill(fld[1])
        p.i("s_add_u32", s(S_LREG), s(S_LREG), s(S_NWAVES))
        p.i("s_cmp_lt_u32", s(S_LREG), s(S_LOGREG))
        ok = self.lbl("lsw_ok")
        p.i("s_cbranch_scc1", "@" + ok)
        p.i("s_bitset0_b32", s(S_USELOG), F_USELOG)           # out of regions: ntComp's increment as device atomics from here on
        p.i("s_mov_b64", "exec", "vcc")
        p.i("s_branch", "@logswitch_back")
        p.label(ok)
        self.load_log_region()
        p.i("s_mov_b64", "exec", "vcc")
        p.i("s_branch", "@logswitch_back")

    def store_log_fill(self, T=V_TP + 16):
        """log_fill[LREG] = LFILL (one lane); T, T + 1: scratch"""
        p = self.p
        p.i("s_mov_b64", "exec", 1)
        p.i("s_lshl_b32", s(S_B), s(S_LREG), 2)
        p.i("s_lshr_b32", s(S_CC), s(S_LFILL4), 2)
        p.i("v_mov_b32", v(T), s(S_B))
        p.i("v_mov_b32", v(T + 1), s(S_CC))
        p.i("s_load_dwordx2", sr(S_TMP, 2), sr(S_KARG, 2), hex(KARG["log_fill"]))
        p.i("s_waitcnt", "lgkmcnt(0)")
        p.i("global_store_dword", v(T), v(T + 1), sr(S_TMP, 2))
        p.i("s_mov_b64", "exec", -1)

    def load_log_region(self):
        """LFILL4 and the buffer descriptor of region LREG"""
        p = self.p
        p.i("s_lshl_b32", s(S_B), s(S_LREG), 2)
        p.i("s_load_dwordx2", sr(S_TMP, 2), sr(S_KARG, 2), hex(KARG["log_fill"]))
        p.i("s_load_dwordx2", sr(S_LOGBASE, 2), sr(S_KARG, 2), hex(KARG["log"]))
        p.i("s_waitcnt", "lgkmcnt(0)")
        p.i("s_load_dword", s(S_LFILL4), sr(S_TMP, 2), s(S_B))
        p.i("s_mul_i32", s(S_TMP), s(S_LREG), s(S_LOGCAP4))
        p.i("s_mul_hi_u32", s(S_TMP + 1), s(S_LREG), s(S_LOGCAP4))
        p.i("s_add_u32", s(S_LOGBASE), s(S_LOGBASE), s(S_TMP))
        p.i("s_addc_u32", s(S_LOGBASE + 1), s(S_LOGBASE + 1), s(S_TMP + 1))
        p.i("s_waitcnt", "lgkmcnt(0)")
        p.i("s_lshl_b32", s(S_LFILL4), s(S_LFILL4), 2)

    # ---- scalar helpers: chunk cursor ----------------------------------------------------------------------
    def cursor_next(self, t, n, real):
        """(t, n) -> the next chunk of the wave's flat sequence; real = it exists (n < C and t < n_tiles)"""
        p = self.p
        same = self.lbl("cur")
        p.i("s_add_u32", s(n), s(n), 1)
        p.i("s_cmp_lt_u32", s(n), s(S_NB))
        p.i("s_cbranch_scc1", "@" + same)
        p.i("s_mov_b32", s(n), 0)
        p.i("s_add_u32", s(t), s(t), 1)
        p.label(same)
        p.i("s_cmp_lt_u32", s(n), s(S_C))
        p.i("s_cselect_b32", s(real), 1, 0)
        p.i("s_cmp_lt_u32", s(t), s(S_NTILES))
        p.i("s_cselect_b32", s(real), s(real), 0)

    def set_desc(self, t):
        """S_DESC = slots of tile t"""
        p = self.p
        p.i("s_mul_i32", s(S_TMP), s(t), s(S_CHUNKB))
        p.i("s_mul_hi_u32", s(S_TMP + 1), s(t), s(S_CHUNKB))
        p.i("s_add_u32", s(S_DESC), s(S_TILES), s(S_TMP))
        p.i("s_addc_u32", s(S_DESC + 1), s(S_TILES + 1), s(S_TMP + 1))

    # ---- the whole kernel body ------------------------------------------------------------------------------
    def build(self, emu=False):
        p = self.p
        k, phi = self.k, self.phi
        # -- inputs -> fixed registers (operands %0 .. are the compiler's; the emulator finds them in s0 ..)
        inp = {name: (s(i) if emu else f"%{i}") for i, name in enumerate(INPUTS)}
        p.i("s_mov_b32", s(S_KARG), inp["karg_lo"])
        p.i("s_mov_b32", s(S_KARG + 1), inp["karg_hi"])
        p.i("s_mov_b32", s(S_WT), inp["wave_gid"])
        p.i("s_lshl_b32", s(S_A), s(S_WT), 2)
        p.i("v_mov_b32", v(V_WAVE4), s(S_A))
        p.i("s_mov_b32", s(S_NWAVES), inp["n_waves"])
        p.i("s_mov_b32", s(S_RQ[0]), inp["lds_wbase"])
        for reg, name in ((S_TILES, "tiles"), (S_SK, "sketch0"), (S_DIRTY, "dirty"), (S_TIE, "tie"), (S_SUS, "sus")):
            p.i("s_load_dwordx2", sr(reg, 2), sr(S_KARG, 2), hex(KARG[name]))
        for reg, name in ((S_NTILES, "n_tiles"), (S_C, "n_chunks"), (S_L, "read_len"), (S_NVLAST, "nv_last"), (S_KEYBASE, "key_base"), (S_RMASK2, "rmask2"),
                          (S_LOGREG, "log_regions"), (S_LOGCAP4, "log_region_cap"), (S_SUSCAP, "sus_cap")):
            p.i("s_load_dword", s(reg), sr(S_KARG, 2), hex(KARG[name]))
        p.i("s_waitcnt", "lgkmcnt(0)")
        p.i("s_load_dwordx2", sr(S_TMP, 2), sr(S_KARG, 2), hex(KARG["tails"]))
        p.i("s_waitcnt", "lgkmcnt(0)")
        p.i("s_cmp_lg_u64", sr(S_TMP, 2), 0)
        p.i("s_cselect_b32", s(S_USELOG), 1 << F_RAGGED, 0)   # (bit F_USELOG joins it below)
        p.i("s_lshl_b32", s(S_LOGCAP4), s(S_LOGCAP4), 2)
        p.i("s_lshl_b32", s(S_SUSCAP), s(S_SUSCAP), 4)         # bytes
        p.i("s_mul_i32", s(S_A), s(S_WT), s(S_SUSCAP))
        p.i("s_mul_hi_u32", s(S_B), s(S_WT), s(S_SUSCAP))
        p.i("s_add_u32", s(S_SUS), s(S_SUS), s(S_A))
        p.i("s_addc_u32", s(S_SUS + 1), s(S_SUS + 1), s(S_B))
        p.i("s_mov_b32", s(S_SUSOFF), 0)
        p.i("s_mov_b32", s(S_EXP0), "0xff47ff54")         # validity look-up, codes 2 and 3: T, G
        if self.sb != 7:
            p.i("s_load_dword", s(S_SPARE), sr(S_KARG, 2), hex(KARG["s_bits"]))
            p.i("s_waitcnt", "lgkmcnt(0)")
            p.i("s_sub_u32", s(S_SPARE), s(S_SPARE), 7)
            p.i("s_bfm_b32", s(S_SPARE), s(S_SPARE), 0)       # 2^(s_bits - 7) - 1
        for reg, val in VCONST:
            p.i("v_mov_b32", v(reg), hex(val))
        p.i("s_mov_b32", s(S_DESC + 2), "0x7fffffff")          # records: a tile's slots (C x 32 KiB) always fit
        p.i("s_mov_b32", s(S_DESC + 3), "0x00020000")
        p.i("s_lshl_b32", s(S_CHUNKB), s(S_C), 15)             # bytes of one tile
        # lane constants
        p.i("v_mbcnt_lo_u32_b32", v(V_LANE4), -1, 0)
        p.i("v_mbcnt_hi_u32_b32", v(V_LANE4), -1, v(V_LANE4))
        p.i("v_lshlrev_b32", v(V_LANE16), 4, v(V_LANE4))
        p.i("v_lshlrev_b32", v(V_LANE4), 2, v(V_LANE4))
        p.i("v_mov_b32", v(V_ONE), 1)
        p.i("v_mov_b32", v(V_EXP1), "0xff43ff41")         # codes 0 and 1: A, C
        p.i("s_add_u32", s(S_A), s(S_RQ[0]), RING_BYTES)
        p.i("v_mov_b32", v(V_QBASE), s(S_A))
        p.i("v_mov_b32", v(V_QDUMMY), QCAP * 4)
        for i in range(1, self.rq):
            p.i("s_add_u32", s(S_RQ[i]), s(S_RQ[0]), i * QSLOT)
        p.i("s_mov_b32", s(S_QHEAD4), 0)
        p.i("s_mov_b32", s(S_QTAIL4), 0)
        p.i("s_mov_b64", sr(S_F1ACC, 2), 0)
        if "timers" in self.exp:
            p.i("s_mov_b32", s(S_SPARE), 0)
        # geometry: W = L - k + 1, NB = ((L - 1 + 16 - phi) >> 4) + 1
        p.i("s_add_u32", s(S_NB), s(S_L), 15 - phi)
        p.i("s_lshr_b32", s(S_NB), s(S_NB), 4)
        p.i("s_add_u32", s(S_NB), s(S_NB), 1)
        # hit log: this wave's first region
        p.i("s_cmp_lg_u32", s(S_LOGREG), 0)
        p.i("s_cselect_b32", s(S_A), 1, 0)
        p.i("s_mov_b32", s(S_LREG), s(S_WT))
        p.i("s_mov_b32", s(S_LFILL4), 0)
        p.i("s_cmp_lt_u32", s(S_LREG), s(S_LOGREG))
        p.i("s_cselect_b32", s(S_A), s(S_A), 0)
        p.i("s_or_b32", s(S_USELOG), s(S_USELOG), s(S_A))
        nolog0 = self.lbl("nolog0")
        p.i("s_bitcmp1_b32", s(S_USELOG), F_USELOG)
        p.i("s_cbranch_scc0", "@" + nolog0)
        self.load_log_region()
        p.label(nolog0)
        # This wave's share: the blocks of all tiles form one sequence (tile * NB + block); the wave owns [first_block, end_block) of it (the
        # kernel's prologue shares a workgroup's blocks out by where its waves sit: a wave alone on its SIMD gets through more than one of a pair)
        # — tiles are split wherever a boundary falls (19 tiles per CU over 6 waves are 4 rounds of whole tiles but 3.2 of blocks).
        # A wave that starts inside a tile first walks up to two blocks it does not own, masked: they fill the window (k - 1 <= 31 bases).
        p.i("s_load_dword", s(S_B), sr(S_KARG, 2), hex(KARG["nb_magic"]))
        p.i("s_mul_i32", s(S_CC), s(S_NTILES), s(S_NB))          # blocks of the batch
        p.i("s_mov_b32", s(S_F0), inp["first_block"])
        p.i("s_min_u32", s(S_FEND), inp["end_block"], s(S_CC))
        p.i("s_waitcnt", "lgkmcnt(0)")
        p.i("s_cmp_lt_u32", s(S_F0), s(S_FEND))
        p.i("s_cbranch_scc0", "@done")                           # nothing to walk: F1 += 0, no suspects
        p.i("s_mul_hi_u32", s(S_PT), s(S_F0), s(S_B))            # tile = F0 / NB with nb_magic = floor(2^32 / NB): never too large,
        p.i("s_mul_i32", s(S_A), s(S_PT), s(S_NB))
        p.i("s_sub_u32", s(S_PN), s(S_F0), s(S_A))               # block = F0 - tile NB ...
        fix, fixed = self.lbl("divfix"), self.lbl("divok")
        p.label(fix)
        p.i("s_cmp_lt_u32", s(S_PN), s(S_NB))
        p.i("s_cbranch_scc1", "@" + fixed)
        p.i("s_sub_u32", s(S_PN), s(S_PN), s(S_NB))              # ... corrected upwards
        p.i("s_add_u32", s(S_PT), s(S_PT), 1)
        p.i("s_branch", "@" + fix)
        p.label(fixed)
        p.i("s_min_u32", s(S_A), s(S_PN), 2)
        p.i("s_sub_u32", s(S_PN), s(S_PN), s(S_A))
        p.i("s_sub_u32", s(S_S0), s(S_F0), s(S_A))               # the first block walked
        p.i("s_sub_u32", s(S_WF), s(S_S0), 1)
        p.i("s_cmp_lt_u32", s(S_PN), s(S_C))
        p.i("s_cselect_b32", s(S_PREAL), 1, 0)
        p.i("s_mov_b32", s(S_QT), s(S_PT))
        p.i("s_mov_b32", s(S_QN), s(S_PN))
        self.cursor_next(S_QT, S_QN, S_QREAL)
        p.i("s_mov_b32", s(S_WN), -1)                           # the dummy iteration: nothing to walk
        p.i("s_mov_b32", s(S_STEPMASK), 0)
        self.set_desc(S_PT)
        p.i("s_lshl_b32", s(S_PSOFF), s(S_PN), 15)
        p.i("s_lshl_b32", s(S_QSOFF), s(S_QN), 15)
        p.i("v_mov_b32", v(V_DN), 0)
        p.i("v_mov_b32", v(V_TACC), 0)
        p.i("v_mov_b32", v(V_CMASK), 0)
        nofirst = self.lbl("nofirst")
        p.i("s_cmp_eq_u32", s(S_PREAL), 1)
        p.i("s_cbranch_scc0", "@" + nofirst)
        self.issue_batch(0, S_PSOFF)
        p.label(nofirst)

        if "timers" in self.exp:
            p.i("s_memtime", sr(S_TMP, 2))
            p.i("s_waitcnt", "lgkmcnt(0)")
            p.i("s_mov_b32", s(S_SUSCAP), s(S_TMP))
            p.i("s_mov_b64", sr(S_F1ACC, 2), 0)
        # ================================ the chunk loop ================================
        p.label("iter")
        for a in range(16):
            if a % 4 == 0:
                self.quarter_enter(a // 4)
            self.walk_step(a)
            skip, notstep, dostep, tailp = self.lbl("nostep"), self.lbl("notstep"), self.lbl("dostep"), self.lbl("tailstep")
            p.i("s_bitcmp1_b32", s(S_STEPMASK), a)
            p.i("s_cbranch_scc0", "@" + notstep)
            if "noflags" in self.exp:
                p.i("s_branch", "@" + skip)
            p.label(dostep)
            self.flags_and_push(a)
            p.label(skip)

            def cold_tail(a=a, skip=skip, notstep=notstep, dostep=dostep, tailp=tailp):
                # a step the mask leaves out (a filling block, the steps behind the read) — or, in a ragged batch, one that ends in the reads' last piece
                p.label(notstep)
                p.i("s_bitcmp1_b32", s(S_STEPMASK), 16 + a)
                p.i("s_cbranch_scc0", "@" + skip)
                p.i("s_mov_b32", s(S_B), 4 * ((self.phi + a) & 15))   # the step ends at base 16 (C - 1) + d of the reads: d x 4
                self.call("tailsub")
                p.i("s_branch", "@" + dostep)
            self.cold.append(cold_tail)
            if a in (4, 8, 12):
                self.pack_batch(a // 4 - 1)
            if a == 15:
                if self.j == 1:  # the plane pair that leaves the window at the first step of the next block
                    p.i("v_mov_b32", v(V_CARRY0), v(V_H0 + 30))
                    p.i("v_mov_b32", v(V_CARRY1), v(V_H0 + 31))
                # the loads behind batch 3 belong to chunk Q: its tile's descriptor (P's loads are all issued); a chunk that does not
                # exist is replaced by the first chunk of the tile the descriptor points at (loaded, never packed)
                noq = self.lbl("noq")
                p.i("s_mov_b32", s(S_CC), 0)
                p.i("s_cmp_eq_u32", s(S_QREAL), 1)
                p.i("s_cbranch_scc0", "@" + noq)
                self.set_desc(S_QT)
                p.i("s_mov_b32", s(S_CC), s(S_QSOFF))
                p.label(noq)
                self.pack_batch(3)
        # -- end of block: empty the queue (the next quarter_enter recycles the oldest quarter-slot of the ring)
        self.drain()
        self.probe(0)
        # -- tie bits of this block -> tie[(WT * NB + WN) * 64 + lane]
        notie = self.lbl("notie")
        p.i("s_add_u32", s(S_A), s(S_WF), 1)                    # (the dummy iteration of a wave that starts at block 0 has WF = -1)
        p.i("s_cmp_gt_u32", s(S_A), s(S_F0))
        p.i("s_cbranch_scc0", "@" + notie)
        p.i("s_lshl_b32", s(S_A), s(S_WF), 8)
        p.i("v_add_u32", v(V_T0), s(S_A), v(V_LANE4))
        p.i("global_store_dword", v(V_T0), v(V_TACC), sr(S_TIE, 2))
        p.label(notie)
        p.i("v_mov_b32", v(V_TACC), 0)
        # -- chunk P: dirty bits -> dirty[(PT * C + PN) * 64 + lane], packed words -> quarter-major bytes in I (ring and planes: quarter_enter)
        p.i("v_bfrev_b32", v(V_DN), v(V_DN))
        nopk = self.lbl("nopk")
        p.i("s_cmp_eq_u32", s(S_PREAL), 1)
        p.i("s_cbranch_scc0", "@" + nopk)
        p.i("s_mul_i32", s(S_A), s(S_PT), s(S_C))
        p.i("s_add_u32", s(S_A), s(S_A), s(S_PN))
        p.i("s_lshl_b32", s(S_A), s(S_A), 8)
        p.i("v_add_u32", v(V_T0), s(S_A), v(V_LANE4))
        p.i("global_store_dword", v(V_T0), v(V_DN), sr(S_DIRTY, 2))
        # -- rotate: planes (H0, H1, I) <- (H1, I, new), dirty words
        self.perm_rotate()
        rotated = self.lbl("rotated")
        p.i("s_branch", "@" + rotated)
        p.label(nopk)                                           # no chunk P (behind a tile's last one): its planes are all 'A'
        for i in range(32):
            p.i("v_mov_b32", v(V_H0 + i), v(V_H1 + i))
        for i in range(32):
            p.i("v_mov_b32", v(V_H1 + i), v(V_I + i))
        for i in range(32):
            p.i("v_mov_b32", v(V_I + i), 0)
        p.label(rotated)
        p.i("v_mov_b32", v(V_D0), v(V_D1))
        p.i("v_mov_b32", v(V_D1), v(V_D2))
        p.i("v_mov_b32", v(V_D2), v(V_DN))
        p.i("v_mov_b32", v(V_DN), 0)
        # -- advance: W <- P, P <- Q, Q <- next(Q)
        p.i("s_mov_b32", s(S_WT), s(S_PT))
        p.i("s_mov_b32", s(S_WN), s(S_PN))
        p.i("s_add_u32", s(S_WF), s(S_WF), 1)
        p.i("s_cmp_lt_u32", s(S_WF), s(S_FEND))
        p.i("s_cbranch_scc0", "@done")
        p.i("s_mov_b32", s(S_PT), s(S_QT))
        p.i("s_mov_b32", s(S_PN), s(S_QN))
        p.i("s_mov_b32", s(S_PREAL), s(S_QREAL))
        p.i("s_mov_b32", s(S_PSOFF), s(S_QSOFF))
        self.cursor_next(S_QT, S_QN, S_QREAL)
        p.i("s_lshl_b32", s(S_QSOFF), s(S_QN), 15)
        # -- a new tile?
        notile, newtile = self.lbl("notile"), self.lbl("newtile")
        p.i("s_cmp_eq_u32", s(S_WN), 0)
        p.i("s_cbranch_scc1", "@" + newtile)
        p.i("s_cmp_eq_u32", s(S_WF), s(S_S0))                    # (or the wave's first block, inside a tile)
        p.i("s_cbranch_scc0", "@" + notile)
        p.label(newtile)
        hf, hr = self.poly_a("F"), self.poly_a("R")
        for jj in range(31):
            p.i("v_mov_b32", v(V_F + jj), -1 if (hf >> jj) & 1 else 0)
            p.i("v_mov_b32", v(V_R + jj), -1 if (hr >> jj) & 1 else 0)
        for i in range(32):
            p.i("v_mov_b32", v(V_H0 + i), 0)
            p.i("v_mov_b32", v(V_H1 + i), 0)
        p.i("v_mov_b32", v(V_CARRY0), 0)
        p.i("v_mov_b32", v(V_CARRY1), 0)
        p.i("v_mov_b32", v(V_D0), 0)
        p.i("v_mov_b32", v(V_D1), 0)
        # valid reads of this tile: all 2048 but in a partial last tile
        p.i("s_add_u32", s(S_A), s(S_WT), 1)
        p.i("s_cmp_eq_u32", s(S_A), s(S_NTILES))
        p.i("s_cselect_b32", s(S_A), s(S_NVLAST), 2048)
        # VMASK bit m = (64 m + lane < valid): cnt = clamp((valid - lane + 63) >> 6, 0, 32) low bits set
        T = V_T0
        p.i("v_lshrrev_b32", v(T), 2, v(V_LANE4))
        p.i("v_sub_u32", v(T), s(S_A), v(T))                    # valid - lane
        p.i("v_add_u32", v(T), 63, v(T))
        p.i("v_lshrrev_b32", v(T), 6, v(T))                     # (valid >= 1 and lane <= 63: never negative) groups m with 64 m + lane < valid
        p.i("v_min_u32", v(T), 32, v(T))
        p.i("v_cmp_gt_u32_e32", "vcc", 32, v(T))
        p.i("v_lshlrev_b32", v(T + 1), v(T), v(V_ONE))
        p.i("v_add_u32", v(T + 1), -1, v(T + 1))
        p.i("v_cndmask_b32_e64", v(V_VMASK), -1, v(T + 1), "vcc")
        p.label(notile)
        # -- per block: candidates are dropped wherever one of the block's three chunks holds a dirty piece of the read
        p.i("v_or3_b32", v(V_T0), v(V_D0), v(V_D1), v(V_D2))
        self.bitop3(v(V_CMASK), v(V_T0), v(V_VMASK), v(V_VMASK), lambda x, y, z: (1 ^ x) & y)
        p.i("v_and_b32", v(V_DMASK), v(V_T0), v(V_VMASK))
        # steps of this block that complete a window: e = 16 (n - 1) + phi + a in [k - 1, L - 1]
        p.i("s_lshl_b32", s(S_A), s(S_WN), 4)
        p.i("s_add_i32", s(S_A), s(S_A), phi - 16)              # e0
        p.i("s_sub_i32", s(S_B), k - 1, s(S_A))
        p.i("s_max_i32", s(S_B), s(S_B), 0)                     # first valid step
        p.i("s_sub_i32", s(S_CC), s(S_L), s(S_A))
        p.i("s_sub_i32", s(S_CC), s(S_CC), 1)
        p.i("s_min_i32", s(S_CC), s(S_CC), 15)                  # last valid step
        p.i("s_sub_i32", s(S_CC), s(S_CC), s(S_B))
        p.i("s_add_i32", s(S_CC), s(S_CC), 1)                   # how many
        p.i("s_max_i32", s(S_CC), s(S_CC), 0)
        p.i("s_bfm_b32", s(S_STEPMASK), s(S_CC), s(S_B))
        p.i("s_cmp_ge_u32", s(S_WF), s(S_F0))                    # a block walked only to fill the window completes nothing
        p.i("s_cselect_b32", s(S_STEPMASK), s(S_STEPMASK), 0)
        p.i("s_cselect_b32", s(S_CC), s(S_CC), 0)
        # Ragged batch (round 5; ntRead takes any std::string, ntcard.cpp:173-189): the reads of a batch are 16 C - 15 .. 16 C bases long (read_len = 16 C) and
        # every tile's reads are sorted by length, longest first, so "the reads with a window ending at base 16 (C - 1) + d" are a PREFIX of the tile —
        # tails[tile][d] of them.  The steps that end in the last piece (e >= 16 (C - 1)) leave the step mask for the tail mask: their code (tail_step,
        # out of line) narrows the candidate masks to that prefix and counts tails[tile][d] windows into F1 instead of one per read.
        noragged = self.lbl("noragged")
        p.i("s_bitcmp1_b32", s(S_USELOG), F_RAGGED)
        p.i("s_cbranch_scc0", "@" + noragged)
        p.i("s_sub_u32", s(S_B), s(S_C), 1)
        p.i("s_lshl_b32", s(S_B), s(S_B), 4)
        p.i("s_sub_i32", s(S_B), s(S_B), s(S_A))                 # first step that ends in the last piece (e0 + a >= 16 (C - 1))
        p.i("s_max_i32", s(S_B), s(S_B), 0)
        p.i("s_min_i32", s(S_B), s(S_B), 16)
        p.i("s_bfm_b32", s(S_B), s(S_B), 0)                      # the steps before it
        p.i("s_andn2_b32", s(S_A), s(S_STEPMASK), s(S_B))        # (e0 is spent)
        p.i("s_and_b32", s(S_STEPMASK), s(S_STEPMASK), s(S_B))
        p.i("s_bcnt1_i32_b32", s(S_CC), s(S_STEPMASK))
        p.i("s_lshl_b32", s(S_A), s(S_A), 16)
        p.i("s_or_b32", s(S_STEPMASK), s(S_STEPMASK), s(S_A))
        p.label(noragged)
        # F1 (ntcard.cpp:154): every window of every valid read counts here (K1f takes the invalid ones back)
        p.i("s_add_u32", s(S_A), s(S_WT), 1)
        p.i("s_cmp_eq_u32", s(S_A), s(S_NTILES))
        p.i("s_cselect_b32", s(S_A), s(S_NVLAST), 2048)
        p.i("s_mul_i32", s(S_B), s(S_A), s(S_CC))
        if "timers" not in self.exp:
            p.i("s_add_u32", s(S_F1ACC), s(S_F1ACC), s(S_B))
            p.i("s_addc_u32", s(S_F1ACC + 1), s(S_F1ACC + 1), 0)
        self.probe(3)
        p.i("s_branch", "@iter")
        for frag in self.cold:                                   # out-of-line: the rare sides of the loop's tests
            frag()
        self.cold = []

        # ================================ epilogue ================================
        p.label("done")
        # F1 (ntcard.cpp:154)
        T = V_T0
        p.i("s_load_dwordx2", sr(S_TMP, 2), sr(S_KARG, 2), hex(KARG["f1"]))
        p.i("s_waitcnt", "lgkmcnt(0)")
        p.i("s_mov_b64", "exec", 1)
        p.i("v_mov_b32", v(T + 2), s(S_TMP))
        p.i("v_mov_b32", v(T + 3), s(S_TMP + 1))
        if "timers" in self.exp:
            p.i("v_mov_b32", v(T + 1), 0)
            for sec in range(4):
                p.i("v_mov_b32", v(T), s(S_TACC[sec]))
                p.i("global_atomic_add_x2", vr(T + 2, 2), vr(T, 2), "off", mods=f"offset:{8 * (1 + sec)}")
        else:
            p.i("v_mov_b32", v(T), s(S_F1ACC))
            p.i("v_mov_b32", v(T + 1), s(S_F1ACC + 1))
            p.i("global_atomic_add_x2", vr(T + 2, 2), vr(T, 2), "off")
            # suspects of this wave: their number (or all ones: the region overflowed, K1f falls back to walking every dirty block)
            p.i("s_load_dwordx2", sr(S_TMP, 2), sr(S_KARG, 2), hex(KARG["sus_count"]))
            p.i("s_lshr_b32", s(S_A), s(S_SUSOFF), 4)
            p.i("s_cmp_le_u32", s(S_SUSOFF), s(S_SUSCAP))
            p.i("s_cselect_b32", s(S_A), s(S_A), -1)
            p.i("s_waitcnt", "lgkmcnt(0)")
            p.i("v_mov_b32", v(T), s(S_A))
            p.i("global_store_dword", v(V_WAVE4), v(T), sr(S_TMP, 2))
        p.i("s_mov_b64", "exec", -1)
        nofill = self.lbl("nofill")
        p.i("s_bitcmp1_b32", s(S_USELOG), F_USELOG)
        p.i("s_cbranch_scc0", "@" + nofill)
        self.store_log_fill()
        p.label(nofill)
        p.label("exit")
        p.i("s_waitcnt", "vmcnt(0) lgkmcnt(0)")
        p.i("s_branch", "@end")
        self.emit_pass()
        self.emit_tail_step()
        p.label("end")
        if "nosched" not in self.exp:
            schedule(p)
        self.phase_fix(p)
        return p

    # VALU instructions that a second wave on the SIMD overlaps with completely (profiles/r05_ubench_op_classes.txt): every other VALU
    # instruction — and any of these with an SGPR source — occupies the pipe for a whole issue interval
    FAST_VALU = {"v_bitop3_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32",
                 "v_mov_b32"}

    @classmethod
    def is_slow_valu(cls, mnem, ops):
        if not mnem.startswith("v_"):
            return False
        if mnem not in cls.FAST_VALU:
            return True
        return any(o.startswith("s") or o.startswith("vcc") or o.startswith("exec") for o in ops[1:])

    def phase_fix(self, p):
        """K1H_EXP=nopslow | noprun | priorun | priowalk (results stay RIGHT; with priorun / priowalk add noprio, which takes the pass's own s_setprio out): scalar no-ops / priority changes next to the slow-class VALU
        instructions.  profiles/r05_ubench_sparse_slow.txt, r05_ubench_phase_fix.txt: one slow-class instruction drops a pair of waves into a
        persistent phase in which they do not overlap any more, and a scalar instruction next to it brings the overlap back."""
        if "priowalk" in self.exp:
            out, code = [], p.code
            i, n = 0, len(code)
            while i < n:
                c = code[i]
                if c[0] == "i" and c[1].startswith("v_") and not self.is_slow_valu(c[1], c[2]):
                    j = i
                    while j < n and code[j][0] == "i" and code[j][1].startswith("v_") and not self.is_slow_valu(code[j][1], code[j][2]):
                        j += 1
                    if j - i >= 12:
                        out.append(("i", "s_setprio", ["3"], ""))
                        out.extend(code[i:j])
                        out.append(("i", "s_setprio", ["0"], ""))
                    else:
                        out.extend(code[i:j])
                    i = j
                    continue
                out.append(c)
                i += 1
            p.code = out
            return
        mode = [m for m in ("nopslow", "noprun", "priorun") if m in self.exp]
        if not mode:
            return
        mode = mode[0]
        out = []
        pending = False   # a slow-class VALU instruction has been issued and neither a scalar instruction nor a fix since
        for c in p.code:
            if c[0] != "i":
                out.append(c)
                if c[0] == "l":
                    pending = False if mode != "priorun" else pending
                continue
            mnem, ops = c[1], c[2]
            if mnem.startswith("v_"):
                slow = self.is_slow_valu(mnem, ops)
                if slow:
                    if mode == "priorun" and not pending:
                        out.append(("i", "s_setprio", ["3"], ""))
                    out.append(c)
                    if mode == "nopslow":
                        out.append(("i", "s_nop", ["0"], ""))
                    else:
                        pending = True
                    continue
                if pending:
                    out.append(("i", "s_nop", ["0"], "") if mode == "noprun" else ("i", "s_setprio", ["0"], ""))
                    pending = False
                out.append(c)
                continue
            if mnem.startswith("s_") and mode != "priorun":
                pending = False   # a scalar instruction does what the no-op would
            if mode == "priorun" and pending and (mnem.startswith("s_cbranch") or mnem in ("s_branch", "s_setpc_b64", "s_endpgm")):
                out.append(("i", "s_setprio", ["0"], ""))
                pending = False
            out.append(c)
        p.code = out


def render_inc(k, sb, gap=0):
    g = Gen(k, sb, gap)
    prog = g.build()
    lines = prog.render(label_fmt=".Lk1h_{}_%=")
    body = "\n".join('\t"' + ln.replace('"', '\\"') + '\\n"' for ln in lines)
    return f"// GENERATED by gen_k1h.py (k = {k}, gap = {gap}, sBits class {sb}): {prog.n_insts()} instructions\n#define K1H_ASM_K{k}_G{gap}_S{sb} \\\n" + \
        "\n".join(ln + " \\" for ln in body.split("\n")) + "\n\n"


# (k, gap) the library is built with: every plain k of 12 .. 32, and ntcard's -g seed in the two forms SURVEY 8(d) names for BASELINE config 5
# (k = 12 / gap 2: round 4; k = 32 / gap 8: round 5)
VARIANTS = tuple((k, 0) for k in range(12, 33)) + ((12, 2), (32, 8))
PARTS = 4                      # the kernels are spread over this many objects (k % PARTS) so that `make -j` compiles them side by side


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else "ntc_k1h_gen.inc"
    defs = out.replace(".inc", "_defs.inc")
    with open(defs, "w") as f:
        f.write("// ntc_k1h_gen_defs.inc — GENERATED by gen_k1h.py (do not edit): geometry and the list of kernel variants\n")
        f.write(f"#define K1H_GEN_WAVES {WAVES}\n#define K1H_GEN_WAREA {WAREA} // LDS bytes per wave: ring + queue\n#define K1H_GEN_PARTS {PARTS}\n")
        f.write("#define K1H_VARIANTS_ALL(X) " + " ".join(f"X({k}, {g})" for k, g in VARIANTS) + "\n")
        for part in range(PARTS):
            f.write(f"#define K1H_VARIANTS_P{part}(X) " + " ".join(f"X({k}, {g})" for k, g in VARIANTS if k % PARTS == part) + "\n")
    with open(out, "w") as f:
        f.write("// ntc_k1h_gen.inc — GENERATED by gen_k1h.py (do not edit): one assembly string per (k, gap, sBits class)\n")
        for k, gap in VARIANTS:
            for sb in (7, 8):
                f.write(f"#if K1H_PART == {k % PARTS}\n")
                f.write(render_inc(k, sb, gap))
                f.write("#endif\n")
    print("wrote", out, "and", defs)
