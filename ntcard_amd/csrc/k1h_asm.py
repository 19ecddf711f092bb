"""k1h_asm.py — a small gfx950 program builder + wave emulator used by gen_k1h.py.

The K1h kernel (ntc_sketch_k1h.hip) is emitted as ONE assembly string with explicit physical registers: hipcc's register
allocator cannot hold its live set (62 hash-state planes + 96 base planes + 32 registers of loads in flight) inside 256 VGPRs,
and a single wave per 2048-read tile is only fast if nothing spills.  `Prog` collects instructions as (mnemonic, operands,
modifiers); `Prog.render()` prints them, `Emu` executes the same list on numpy lanes so the kernel's logic is checked on the
CPU against the oracle (tests/test_k1h_emulator.py) before it ever reaches a GPU.

Only the instructions the generator uses are modelled; an unknown mnemonic is an error, never a silent no-op.
"""
import re

import numpy as np

M32 = 0xFFFFFFFF
M64 = (1 << 64) - 1


class Prog:
    def __init__(self):
        self.code = []  # ('i', mnem, ops, mods) | ('l', label) | ('c', comment)
        self.count = {}

    def i(self, mnem, *ops, mods=""):
        self.code.append(("i", mnem, [str(o) for o in ops], mods))

    def label(self, name):
        self.code.append(("l", name))

    def comment(self, text):
        self.code.append(("c", text))

    def n_insts(self):
        return sum(1 for c in self.code if c[0] == "i")

    def render(self, label_fmt="{}"):
        """-> list of assembly lines; label_fmt decorates label names (inline asm: '.L{}_%=')"""
        out = []
        for c in self.code:
            if c[0] == "l":
                out.append(label_fmt.format(c[1]) + ":")
            elif c[0] == "c":
                out.append("; " + c[1])
            else:
                _, mnem, ops, mods = c
                ops2 = [label_fmt.format(o[1:]) if o.startswith("@") else o for o in ops]
                line = mnem + (" " + ", ".join(ops2) if ops2 else "") + (" " + mods if mods else "")
                out.append(line)
        return out


# ---- operand helpers -------------------------------------------------------------------------------------
def v(i):
    return f"v{i}"


def s(i):
    return f"s{i}"


def vr(i, n):
    return f"v[{i}:{i + n - 1}]"


def sr(i, n):
    return f"s[{i}:{i + n - 1}]"


_RE_V = re.compile(r"^v(\d+)$")
_RE_S = re.compile(r"^s(\d+)$")
_RE_VR = re.compile(r"^v\[(\d+):(\d+)\]$")
_RE_SR = re.compile(r"^s\[(\d+):(\d+)\]$")


def _parse(op):
    m = _RE_V.match(op)
    if m:
        return ("v", int(m.group(1)), 1)
    m = _RE_S.match(op)
    if m:
        return ("s", int(m.group(1)), 1)
    m = _RE_VR.match(op)
    if m:
        a, b = int(m.group(1)), int(m.group(2))
        return ("v", a, b - a + 1)
    m = _RE_SR.match(op)
    if m:
        a, b = int(m.group(1)), int(m.group(2))
        return ("s", a, b - a + 1)
    if op in ("vcc", "exec", "vcc_lo", "vcc_hi", "exec_lo", "exec_hi", "scc", "off", "m0"):
        return (op, 0, 0)
    if op.startswith("@"):
        return ("label", op[1:], 0)
    try:
        return ("imm", int(op, 0) & M64 if int(op, 0) >= 0 else int(op, 0), 0)
    except ValueError:
        return ("raw", op, 0)  # (s_waitcnt counters and the like: never read as a value)


def _mods(mods):
    d = {}
    for tok in mods.split():
        if ":" in tok:
            kk, vv = tok.split(":", 1)
            d[kk] = int(vv, 0)
        else:
            d[tok] = True
    return d


class Halt(Exception):
    pass


class Emu:
    """one wave64.  Global memory: a flat numpy byte array `mem` (addresses are offsets into it); LDS: `lds` bytes."""

    def __init__(self, prog, mem, lds, max_steps=50_000_000):
        self.V = np.zeros((256, 64), dtype=np.uint32)
        self.S = [0] * 128
        self.vcc = 0
        self.exec = M64
        self.scc = 0
        self.mem = mem
        self.mem32 = mem.view(np.uint32)
        self.lds = lds
        self.lds32 = lds.view(np.uint32)
        self.max_steps = max_steps
        self.lane_ids = np.arange(64, dtype=np.uint32)
        self.executed = 0
        self.hist = {}
        # decode
        self.labels = {}
        self.insts = []
        for c in prog.code:
            if c[0] == "l":
                self.labels[c[1]] = len(self.insts)
            elif c[0] == "i":
                self.insts.append((c[1], [_parse(o) for o in c[2]], _mods(c[3])))
        self.pc_of_addr = None

    # -- register access --
    def mask_arr(self, m=None):
        m = self.exec if m is None else m
        return ((m >> self.lane_ids.astype(np.uint64)) & np.uint64(1)).astype(bool)

    def rd_s(self, o, n=1):
        kind, a, _ = o
        if kind == "s":
            if n == 1:
                return self.S[a]
            return self.S[a] | (self.S[a + 1] << 32)
        if kind == "imm":
            val = a
            return val & (M64 if n == 2 else M32)
        if kind == "vcc":
            return self.vcc if n == 2 else self.vcc & M32
        if kind == "vcc_lo":
            return self.vcc & M32
        if kind == "vcc_hi":
            return self.vcc >> 32
        if kind == "exec":
            return self.exec if n == 2 else self.exec & M32
        if kind == "exec_lo":
            return self.exec & M32
        if kind == "exec_hi":
            return self.exec >> 32
        if kind == "scc":
            return self.scc
        raise ValueError(f"not a scalar operand: {o}")

    def wr_s(self, o, val, n=1):
        kind, a, _ = o
        if kind == "s":
            if n == 1:
                self.S[a] = val & M32
            else:
                self.S[a] = val & M32
                self.S[a + 1] = (val >> 32) & M32
        elif kind == "vcc":
            self.vcc = val & M64
        elif kind == "exec":
            self.exec = val & M64
        elif kind == "vcc_lo":
            self.vcc = (self.vcc & ~M32) | (val & M32)
        elif kind == "vcc_hi":
            self.vcc = (self.vcc & M32) | ((val & M32) << 32)
        else:
            raise ValueError(f"cannot write scalar {o}")

    def rd_v(self, o):
        """32-bit per-lane value of a VALU source (VGPR, SGPR or constant)"""
        kind, a, _ = o
        if kind == "v":
            return self.V[a]
        return np.full(64, self.rd_s(o) & M32, dtype=np.uint32)

    def wr_v(self, o, val, mask=None):
        kind, a, _ = o
        assert kind == "v", o
        m = self.mask_arr() if mask is None else mask
        self.V[a][m] = val.astype(np.uint32)[m]

    # -- execution --
    def run(self, entry=0):
        pc = entry
        n = len(self.insts)
        steps = 0
        while pc < n:
            mnem, ops, mods = self.insts[pc]
            steps += 1
            if steps > self.max_steps:
                raise RuntimeError("emulator: step limit")
            self.hist[mnem] = self.hist.get(mnem, 0) + 1
            npc = self.step(pc, mnem, ops, mods)
            pc = pc + 1 if npc is None else npc
        self.executed = steps
        return steps

    def step(self, pc, mnem, ops, mods):
        f = getattr(self, "op_" + mnem, None)
        if f is None:
            raise NotImplementedError(f"emulator: {mnem}")
        return f(pc, ops, mods)

    # ---- SALU ----
    def op_s_mov_b32(self, pc, o, m):
        self.wr_s(o[0], self.rd_s(o[1]))

    def op_s_mov_b64(self, pc, o, m):
        self.wr_s(o[0], self.rd_s(o[1], 2), 2)

    def _sbin(self, o, fn, n=1, scc=None):
        a, b = self.rd_s(o[1], n), self.rd_s(o[2], n)
        r = fn(a, b)
        msk = M64 if n == 2 else M32
        self.wr_s(o[0], r & msk, n)
        if scc is not None:
            self.scc = 1 if scc(r, a, b) else 0

    def op_s_add_u32(self, pc, o, m):
        self._sbin(o, lambda a, b: a + b, 1, lambda r, a, b: r > M32)

    def op_s_addc_u32(self, pc, o, m):
        c = self.scc
        self._sbin(o, lambda a, b: a + b + c, 1, lambda r, a, b: r > M32)

    def op_s_sub_u32(self, pc, o, m):
        self._sbin(o, lambda a, b: a - b, 1, lambda r, a, b: b > a)

    def op_s_sub_i32(self, pc, o, m):
        self._sbin(o, lambda a, b: a - b, 1, lambda r, a, b: False)

    def op_s_add_i32(self, pc, o, m):
        self._sbin(o, lambda a, b: a + b, 1, lambda r, a, b: False)

    def op_s_mul_i32(self, pc, o, m):
        self._sbin(o, lambda a, b: a * b)

    def op_s_mul_hi_u32(self, pc, o, m):
        self._sbin(o, lambda a, b: (a * b) >> 32)

    def op_s_and_b32(self, pc, o, m):
        self._sbin(o, lambda a, b: a & b, 1, lambda r, a, b: (r & M32) != 0)

    def op_s_or_b32(self, pc, o, m):
        self._sbin(o, lambda a, b: a | b, 1, lambda r, a, b: (r & M32) != 0)

    def op_s_xor_b32(self, pc, o, m):
        self._sbin(o, lambda a, b: a ^ b, 1, lambda r, a, b: (r & M32) != 0)

    def op_s_andn2_b32(self, pc, o, m):
        self._sbin(o, lambda a, b: a & ~b, 1, lambda r, a, b: (r & M32) != 0)

    def op_s_and_b64(self, pc, o, m):
        self._sbin(o, lambda a, b: a & b, 2, lambda r, a, b: (r & M64) != 0)

    def op_s_or_b64(self, pc, o, m):
        self._sbin(o, lambda a, b: a | b, 2, lambda r, a, b: (r & M64) != 0)

    def op_s_andn2_b64(self, pc, o, m):
        self._sbin(o, lambda a, b: a & ~b, 2, lambda r, a, b: (r & M64) != 0)

    def op_s_lshl_b32(self, pc, o, m):
        self._sbin(o, lambda a, b: a << (b & 31), 1, lambda r, a, b: (r & M32) != 0)

    def op_s_lshr_b32(self, pc, o, m):
        self._sbin(o, lambda a, b: a >> (b & 31), 1, lambda r, a, b: (r & M32) != 0)

    def op_s_lshl_b64(self, pc, o, m):
        a, b = self.rd_s(o[1], 2), self.rd_s(o[2])
        r = (a << (b & 63)) & M64
        self.wr_s(o[0], r, 2)
        self.scc = 1 if r else 0

    def op_s_lshr_b64(self, pc, o, m):
        a, b = self.rd_s(o[1], 2), self.rd_s(o[2])
        r = a >> (b & 63)
        self.wr_s(o[0], r, 2)
        self.scc = 1 if r else 0

    def op_s_lshl3_add_u32(self, pc, o, m):
        self._sbin(o, lambda a, b: (a << 3) + b, 1, lambda r, a, b: r > M32)

    def op_s_lshl2_add_u32(self, pc, o, m):
        self._sbin(o, lambda a, b: (a << 2) + b, 1, lambda r, a, b: r > M32)

    def op_s_min_u32(self, pc, o, m):
        self._sbin(o, lambda a, b: min(a, b), 1, lambda r, a, b: a <= b)

    def op_s_max_u32(self, pc, o, m):
        self._sbin(o, lambda a, b: max(a, b), 1, lambda r, a, b: a >= b)

    @staticmethod
    def _i32(x):
        x &= M32
        return x - (1 << 32) if x & 0x80000000 else x

    def op_s_min_i32(self, pc, o, m):
        self._sbin(o, lambda a, b: min(self._i32(a), self._i32(b)), 1, lambda r, a, b: self._i32(a) <= self._i32(b))

    def op_s_max_i32(self, pc, o, m):
        self._sbin(o, lambda a, b: max(self._i32(a), self._i32(b)), 1, lambda r, a, b: self._i32(a) >= self._i32(b))

    def op_s_bfm_b32(self, pc, o, m):
        self._sbin(o, lambda a, b: ((1 << (a & 31)) - 1) << (b & 31))

    def op_s_bfm_b64(self, pc, o, m):
        a, b = self.rd_s(o[1]), self.rd_s(o[2])
        self.wr_s(o[0], (((1 << (a & 63)) - 1) << (b & 63)) & M64, 2)

    def op_s_cselect_b32(self, pc, o, m):
        self.wr_s(o[0], self.rd_s(o[1]) if self.scc else self.rd_s(o[2]))

    def op_s_cselect_b64(self, pc, o, m):
        self.wr_s(o[0], self.rd_s(o[1], 2) if self.scc else self.rd_s(o[2], 2), 2)

    def op_s_bcnt1_i32_b64(self, pc, o, m):
        r = bin(self.rd_s(o[1], 2)).count("1")
        self.wr_s(o[0], r)
        self.scc = 1 if r else 0

    def op_s_bcnt1_i32_b32(self, pc, o, m):
        r = bin(self.rd_s(o[1])).count("1")
        self.wr_s(o[0], r)
        self.scc = 1 if r else 0

    def op_s_bitset1_b32(self, pc, o, m):
        self.wr_s(o[0], self.rd_s(o[0]) | (1 << (self.rd_s(o[1]) & 31)))

    def op_s_bitset0_b32(self, pc, o, m):
        self.wr_s(o[0], self.rd_s(o[0]) & ~(1 << (self.rd_s(o[1]) & 31)))

    def op_s_bitcmp1_b32(self, pc, o, m):
        self.scc = (self.rd_s(o[0]) >> (self.rd_s(o[1]) & 31)) & 1

    def op_s_bitcmp0_b32(self, pc, o, m):
        self.scc = 1 - ((self.rd_s(o[0]) >> (self.rd_s(o[1]) & 31)) & 1)

    def _scmp(self, o, fn, signed=False):
        a, b = self.rd_s(o[0]), self.rd_s(o[1])
        if signed:
            a, b = self._i32(a), self._i32(b)
        self.scc = 1 if fn(a, b) else 0

    def op_s_cmp_eq_u32(self, pc, o, m):
        self._scmp(o, lambda a, b: a == b)

    def op_s_cmp_lg_u32(self, pc, o, m):
        self._scmp(o, lambda a, b: a != b)

    def op_s_cmp_lt_u32(self, pc, o, m):
        self._scmp(o, lambda a, b: a < b)

    def op_s_cmp_le_u32(self, pc, o, m):
        self._scmp(o, lambda a, b: a <= b)

    def op_s_cmp_gt_u32(self, pc, o, m):
        self._scmp(o, lambda a, b: a > b)

    def op_s_cmp_ge_u32(self, pc, o, m):
        self._scmp(o, lambda a, b: a >= b)

    def op_s_cmp_lt_i32(self, pc, o, m):
        self._scmp(o, lambda a, b: a < b, True)

    def op_s_cmp_le_i32(self, pc, o, m):
        self._scmp(o, lambda a, b: a <= b, True)

    def op_s_cmp_gt_i32(self, pc, o, m):
        self._scmp(o, lambda a, b: a > b, True)

    def op_s_cmp_ge_i32(self, pc, o, m):
        self._scmp(o, lambda a, b: a >= b, True)

    def op_s_cmp_eq_u64(self, pc, o, m):
        self.scc = 1 if self.rd_s(o[0], 2) == self.rd_s(o[1], 2) else 0

    def op_s_cmp_lg_u64(self, pc, o, m):
        self.scc = 1 if self.rd_s(o[0], 2) != self.rd_s(o[1], 2) else 0

    def op_s_cmp_eq_u64(self, pc, o, m):
        self.scc = 1 if self.rd_s(o[0], 2) == self.rd_s(o[1], 2) else 0

    def op_s_branch(self, pc, o, m):
        return self.labels[o[0][1]]

    def op_s_cbranch_scc0(self, pc, o, m):
        return self.labels[o[0][1]] if not self.scc else None

    def op_s_cbranch_scc1(self, pc, o, m):
        return self.labels[o[0][1]] if self.scc else None

    def op_s_cbranch_vccz(self, pc, o, m):
        return self.labels[o[0][1]] if self.vcc == 0 else None

    def op_s_cbranch_vccnz(self, pc, o, m):
        return self.labels[o[0][1]] if self.vcc != 0 else None

    def op_s_cbranch_execz(self, pc, o, m):
        return self.labels[o[0][1]] if self.exec == 0 else None

    def op_s_getpc_b64(self, pc, o, m):
        # "address" = 4 x instruction index: every emulated instruction counts as 4 bytes (the generator only ever adds 4 to a
        # getpc value, to step over one s_branch, which IS 4 bytes on the hardware)
        self.wr_s(o[0], (pc + 1) * 4, 2)

    def op_s_setpc_b64(self, pc, o, m):
        return self.rd_s(o[0], 2) // 4

    def op_s_waitcnt(self, pc, o, m):
        return None

    def op_s_nop(self, pc, o, m):
        return None

    def op_s_setprio(self, pc, o, m):
        return None

    def op_s_sleep(self, pc, o, m):
        return None

    def op_s_endpgm(self, pc, o, m):
        return len(self.insts)

    def op_s_memtime(self, pc, o, m):
        self.wr_s(o[0], self.hist_total() * 4, 2)

    def hist_total(self):
        return sum(self.hist.values())

    def op_s_load_dword(self, pc, o, m):
        addr = self.rd_s(o[1], 2) + self.rd_s(o[2])
        self.wr_s(o[0], int(self.mem32[addr // 4]))

    def op_s_load_dwordx2(self, pc, o, m):
        addr = self.rd_s(o[1], 2) + self.rd_s(o[2])
        self.wr_s(o[0], int(self.mem32[addr // 4]) | (int(self.mem32[addr // 4 + 1]) << 32), 2)

    # ---- VALU ----
    def _vbin(self, o, fn):
        a, b = self.rd_v(o[1]), self.rd_v(o[2])
        self.wr_v(o[0], fn(a, b))

    def op_v_mov_b32(self, pc, o, m):
        self.wr_v(o[0], self.rd_v(o[1]))

    def op_v_not_b32(self, pc, o, m):
        self.wr_v(o[0], ~self.rd_v(o[1]))

    def op_v_and_b32(self, pc, o, m):
        self._vbin(o, lambda a, b: a & b)

    def op_v_or_b32(self, pc, o, m):
        self._vbin(o, lambda a, b: a | b)

    def op_v_xor_b32(self, pc, o, m):
        self._vbin(o, lambda a, b: a ^ b)

    def op_v_add_u32(self, pc, o, m):
        self._vbin(o, lambda a, b: a + b)

    def op_v_sub_u32(self, pc, o, m):
        self._vbin(o, lambda a, b: a - b)

    def op_v_subrev_u32(self, pc, o, m):
        self._vbin(o, lambda a, b: b - a)

    def op_v_lshlrev_b32(self, pc, o, m):
        self._vbin(o, lambda a, b: b << (a & np.uint32(31)))

    def op_v_lshrrev_b32(self, pc, o, m):
        self._vbin(o, lambda a, b: b >> (a & np.uint32(31)))

    def op_v_ashrrev_i32(self, pc, o, m):
        self._vbin(o, lambda a, b: (b.astype(np.int32) >> (a & np.uint32(31)).astype(np.int32)).astype(np.uint32))

    def op_v_max_i32(self, pc, o, m):
        self._vbin(o, lambda a, b: np.maximum(a.astype(np.int32), b.astype(np.int32)).astype(np.uint32))

    def op_v_mul_lo_u32(self, pc, o, m):
        self._vbin(o, lambda a, b: (a.astype(np.uint64) * b.astype(np.uint64)).astype(np.uint32))

    def op_v_mul_u32_u24(self, pc, o, m):
        self._vbin(o, lambda a, b: ((a & np.uint32(0xFFFFFF)).astype(np.uint64) * (b & np.uint32(0xFFFFFF)).astype(np.uint64)).astype(np.uint32))

    def op_v_min_u32(self, pc, o, m):
        self._vbin(o, np.minimum)

    def op_v_max_u32(self, pc, o, m):
        self._vbin(o, np.maximum)

    def op_v_add_co_u32(self, pc, o, m):
        # v_add_co_u32 vdst, vcc|sdst, src0, src1
        a, b = self.rd_v(o[2]).astype(np.uint64), self.rd_v(o[3]).astype(np.uint64)
        r = a + b
        self._wr_lane_mask(o[1], r > np.uint64(M32))
        self.wr_v(o[0], (r & np.uint64(M32)).astype(np.uint32))

    def op_v_addc_co_u32(self, pc, o, m):
        # v_addc_co_u32 vdst, vcc, src0, src1, vcc
        cin = self.mask_arr(self.rd_s(o[4], 2)).astype(np.uint64)
        a, b = self.rd_v(o[2]).astype(np.uint64), self.rd_v(o[3]).astype(np.uint64)
        r = a + b + cin
        self._wr_lane_mask(o[1], r > np.uint64(M32))
        self.wr_v(o[0], (r & np.uint64(M32)).astype(np.uint32))

    def _wr_lane_mask(self, o, cond):
        """write a per-lane condition into an SGPR pair / vcc: inactive lanes give 0"""
        em = self.mask_arr()
        bits = 0
        for i in np.nonzero(cond & em)[0]:
            bits |= 1 << int(i)
        self.wr_s(o, bits, 2)

    def op_v_perm_b32(self, pc, o, m):
        a, b, sel = self.rd_v(o[1]), self.rd_v(o[2]), self.rd_v(o[3])
        comb = (a.astype(np.uint64) << np.uint64(32)) | b.astype(np.uint64)  # bytes 0-3 = src1, 4-7 = src0
        out = np.zeros(64, dtype=np.uint32)
        for i in range(4):
            sb = (sel >> np.uint32(8 * i)) & np.uint32(0xFF)
            byte = ((comb >> (np.minimum(sb, 7).astype(np.uint64) * np.uint64(8))) & np.uint64(0xFF)).astype(np.uint32)
            byte = np.where(sb == 12, np.uint32(0), byte)
            byte = np.where(sb >= 13, np.uint32(0xFF), byte)
            if np.any((sb >= 8) & (sb <= 11)):
                raise NotImplementedError("v_perm sign-replicating selectors")
            out |= byte << np.uint32(8 * i)
        self.wr_v(o[0], out)

    def op_v_bitop3_b32(self, pc, o, m):
        a, b, c = self.rd_v(o[1]), self.rd_v(o[2]), self.rd_v(o[3])
        t = m["bitop3"]
        r = np.zeros(64, dtype=np.uint32)
        for idx in range(8):
            if (t >> idx) & 1:
                r |= (a if idx & 4 else ~a) & (b if idx & 2 else ~b) & (c if idx & 1 else ~c)
        self.wr_v(o[0], r)

    def op_v_bfi_b32(self, pc, o, m):
        msk, x, y = self.rd_v(o[1]), self.rd_v(o[2]), self.rd_v(o[3])
        self.wr_v(o[0], (msk & x) | (~msk & y))

    def op_v_bfe_u32(self, pc, o, m):
        x, off, w = self.rd_v(o[1]), self.rd_v(o[2]) & np.uint32(31), self.rd_v(o[3]) & np.uint32(31)
        self.wr_v(o[0], (x >> off) & ((np.uint32(1) << w) - np.uint32(1)))

    def op_v_alignbit_b32(self, pc, o, m):
        hi, lo, sh = self.rd_v(o[1]), self.rd_v(o[2]), self.rd_v(o[3]) & np.uint32(31)
        comb = (hi.astype(np.uint64) << np.uint64(32)) | lo.astype(np.uint64)
        self.wr_v(o[0], ((comb >> sh.astype(np.uint64)) & np.uint64(M32)).astype(np.uint32))

    def op_v_lshl_add_u32(self, pc, o, m):
        a, sh, c = self.rd_v(o[1]), self.rd_v(o[2]) & np.uint32(31), self.rd_v(o[3])
        self.wr_v(o[0], (a << sh) + c)

    def op_v_lshl_or_b32(self, pc, o, m):
        a, sh, c = self.rd_v(o[1]), self.rd_v(o[2]) & np.uint32(31), self.rd_v(o[3])
        self.wr_v(o[0], (a << sh) | c)

    def op_v_and_or_b32(self, pc, o, m):
        a, b, c = self.rd_v(o[1]), self.rd_v(o[2]), self.rd_v(o[3])
        self.wr_v(o[0], (a & b) | c)

    def op_v_or3_b32(self, pc, o, m):
        a, b, c = self.rd_v(o[1]), self.rd_v(o[2]), self.rd_v(o[3])
        self.wr_v(o[0], a | b | c)

    def op_v_add3_u32(self, pc, o, m):
        a, b, c = self.rd_v(o[1]), self.rd_v(o[2]), self.rd_v(o[3])
        self.wr_v(o[0], a + b + c)

    def op_v_mad_u32_u24(self, pc, o, m):
        a, b, c = self.rd_v(o[1]) & np.uint32(0xFFFFFF), self.rd_v(o[2]) & np.uint32(0xFFFFFF), self.rd_v(o[3])
        self.wr_v(o[0], ((a.astype(np.uint64) * b.astype(np.uint64)) & np.uint64(M32)).astype(np.uint32) + c)

    def op_v_lshl_add_u64(self, pc, o, m):
        kind, a, n = o[1]
        x = self.V[a].astype(np.uint64) | (self.V[a + 1].astype(np.uint64) << np.uint64(32))
        r = (x << np.uint64(self.rd_s(o[2]) & 63)) + np.uint64(self.rd_s(o[3], 2))
        kd, d, nd = o[0]
        lo, hi = (r & np.uint64(M32)).astype(np.uint32), (r >> np.uint64(32)).astype(np.uint32)
        self.wr_v(("v", d, 1), lo)
        self.wr_v(("v", d + 1, 1), hi)

    def op_v_ffbl_b32(self, pc, o, m):
        x = self.rd_v(o[1])
        out = np.full(64, M32, dtype=np.uint32)
        for i in range(64):
            xv = int(x[i])
            if xv:
                out[i] = (xv & -xv).bit_length() - 1
        self.wr_v(o[0], out)

    def op_v_bfrev_b32(self, pc, o, m):
        x = self.rd_v(o[1])
        out = np.zeros(64, dtype=np.uint32)
        for i in range(64):
            out[i] = int(f"{int(x[i]):032b}"[::-1], 2)
        self.wr_v(o[0], out)

    def op_v_bcnt_u32_b32(self, pc, o, m):
        x, acc = self.rd_v(o[1]), self.rd_v(o[2])
        cnt = np.array([bin(int(t)).count("1") for t in x], dtype=np.uint32)
        self.wr_v(o[0], cnt + acc)

    def op_v_swap_b32(self, pc, o, m):
        a, b = self.rd_v(o[0]).copy(), self.rd_v(o[1]).copy()
        self.wr_v(o[0], b)
        self.wr_v(o[1], a)

    def op_v_cndmask_b32(self, pc, o, m):
        sel = self.mask_arr(self.rd_s(o[3], 2))
        self.wr_v(o[0], np.where(sel, self.rd_v(o[2]), self.rd_v(o[1])))

    def op_v_mbcnt_lo_u32_b32(self, pc, o, m):
        msk = self.rd_s(o[1]) & M32
        out = np.array([bin(msk & ((1 << min(i, 32)) - 1)).count("1") for i in range(64)], dtype=np.uint32)
        self.wr_v(o[0], out + self.rd_v(o[2]))

    def op_v_mbcnt_hi_u32_b32(self, pc, o, m):
        msk = self.rd_s(o[1]) & M32
        out = np.array([bin(msk & ((1 << max(i - 32, 0)) - 1)).count("1") if i > 32 else 0 for i in range(64)], dtype=np.uint32)
        self.wr_v(o[0], out + self.rd_v(o[2]))

    def op_v_readfirstlane_b32(self, pc, o, m):
        if self.exec == 0:
            lane = 0
        else:
            lane = (self.exec & -self.exec).bit_length() - 1
        self.wr_s(o[0], int(self.rd_v(o[1])[lane]))

    def op_v_readlane_b32(self, pc, o, m):
        self.wr_s(o[0], int(self.rd_v(o[1])[self.rd_s(o[2]) & 63]))

    def _vcmp(self, o, fn):
        # v_cmp_xx vcc|s[a:b], src0, src1
        a, b = self.rd_v(o[1]), self.rd_v(o[2])
        self._wr_lane_mask(o[0], fn(a, b))

    def op_v_cmp_ne_u32(self, pc, o, m):
        self._vcmp(o, lambda a, b: a != b)

    def op_v_cmp_eq_u32(self, pc, o, m):
        self._vcmp(o, lambda a, b: a == b)

    def op_v_cmp_lt_u32(self, pc, o, m):
        self._vcmp(o, lambda a, b: a < b)

    def op_v_cmp_le_u32(self, pc, o, m):
        self._vcmp(o, lambda a, b: a <= b)

    def op_v_cmp_gt_u32(self, pc, o, m):
        self._vcmp(o, lambda a, b: a > b)

    def op_v_cmp_ge_u32(self, pc, o, m):
        self._vcmp(o, lambda a, b: a >= b)

    op_v_cmp_ne_u32_e32 = op_v_cmp_ne_u32
    op_v_cmp_eq_u32_e32 = op_v_cmp_eq_u32
    op_v_cmp_lt_u32_e32 = op_v_cmp_lt_u32
    op_v_cmp_gt_u32_e32 = op_v_cmp_gt_u32
    op_v_cmp_ge_u32_e32 = op_v_cmp_ge_u32
    op_v_cmp_le_u32_e32 = op_v_cmp_le_u32
    op_v_cmp_ne_u32_e64 = op_v_cmp_ne_u32
    op_v_cmp_eq_u32_e64 = op_v_cmp_eq_u32
    op_v_cmp_lt_u32_e64 = op_v_cmp_lt_u32
    op_v_cmp_gt_u32_e64 = op_v_cmp_gt_u32
    op_v_cmp_ge_u32_e64 = op_v_cmp_ge_u32
    op_v_cmp_le_u32_e64 = op_v_cmp_le_u32
    op_v_cndmask_b32_e64 = op_v_cndmask_b32

    # ---- LDS ----
    def _lds_addr(self, o, m):
        return self.rd_v(o).astype(np.int64) + m.get("offset", 0)

    def op_ds_read_b32(self, pc, o, m):
        em = self.mask_arr()
        addr = self._lds_addr(o[1], m)
        assert np.all(addr[em] % 4 == 0) and np.all(addr[em] + 4 <= self.lds.size), "ds_read_b32: bad address"
        val = np.zeros(64, dtype=np.uint32)
        val[em] = self.lds32[addr[em] // 4]
        self.wr_v(o[0], val)

    def op_ds_bpermute_b32(self, pc, o, m):
        """dst[lane] = src[(addr[lane] / 4) mod 64]; a source lane that is switched off hands out 0"""
        em = self.mask_arr()
        lane = (self.rd_v(o[1]).astype(np.int64) + m.get("offset", 0)) // 4 % 64
        src = self.rd_v(o[2])
        val = np.where(em[lane], src[lane], np.uint32(0)).astype(np.uint32)
        self.wr_v(o[0], val)

    def op_ds_read_u8(self, pc, o, m):
        em = self.mask_arr()
        addr = self._lds_addr(o[1], m)
        assert np.all(addr[em] >= 0) and np.all(addr[em] < self.lds.size), "ds_read_u8: bad address"
        val = np.zeros(64, dtype=np.uint32)
        val[em] = self.lds[addr[em]]
        self.wr_v(o[0], val)

    def op_ds_read_u16(self, pc, o, m):
        em = self.mask_arr()
        addr = self._lds_addr(o[1], m)
        assert np.all(addr[em] % 2 == 0) and np.all(addr[em] + 2 <= self.lds.size), "ds_read_u16: bad address"
        val = np.zeros(64, dtype=np.uint32)
        val[em] = self.lds.view(np.uint16)[addr[em] // 2]
        self.wr_v(o[0], val)

    def op_ds_write_b16(self, pc, o, m):
        em = self.mask_arr()
        addr = self._lds_addr(o[0], m)
        assert np.all(addr[em] % 2 == 0) and np.all(addr[em] + 2 <= self.lds.size), "ds_write_b16: bad address"
        self.lds.view(np.uint16)[addr[em] // 2] = (self.rd_v(o[1])[em] & np.uint32(0xFFFF)).astype(np.uint16)

    def op_ds_read_b64(self, pc, o, m):
        em = self.mask_arr()
        addr = self._lds_addr(o[1], m)
        assert np.all(addr[em] % 8 == 0) and np.all(addr[em] + 8 <= self.lds.size), "ds_read_b64: bad address"
        kind, a, n = o[0]
        for j in range(2):
            val = np.zeros(64, dtype=np.uint32)
            val[em] = self.lds32[addr[em] // 4 + j]
            self.wr_v(("v", a + j, 1), val)

    def op_ds_read2_b32(self, pc, o, m):
        em = self.mask_arr()
        base = self.rd_v(o[1]).astype(np.int64)
        kind, a, n = o[0]
        for j, key in enumerate(("offset0", "offset1")):
            addr = base + 4 * m.get(key, 0)
            assert np.all(addr[em] % 4 == 0) and np.all(addr[em] + 4 <= self.lds.size), "ds_read2_b32: bad address"
            val = np.zeros(64, dtype=np.uint32)
            val[em] = self.lds32[addr[em] // 4]
            self.wr_v(("v", a + j, 1), val)

    def op_ds_write2_b32(self, pc, o, m):
        em = self.mask_arr()
        base = self.rd_v(o[0]).astype(np.int64)
        for j, key in enumerate(("offset0", "offset1")):
            addr = base + 4 * m.get(key, 0)
            assert np.all(addr[em] % 4 == 0) and np.all(addr[em] + 4 <= self.lds.size), "ds_write2_b32: bad address"
            self.lds32[addr[em] // 4] = self.rd_v(o[1 + j])[em]

    def op_ds_write2st64_b32(self, pc, o, m):
        em = self.mask_arr()
        base = self.rd_v(o[0]).astype(np.int64)
        for j, key in enumerate(("offset0", "offset1")):
            addr = base + 256 * m.get(key, 0)
            assert np.all(addr[em] % 4 == 0) and np.all(addr[em] + 4 <= self.lds.size), "ds_write2st64_b32: bad address"
            self.lds32[addr[em] // 4] = self.rd_v(o[1 + j])[em]

    def op_ds_write_b32(self, pc, o, m):
        em = self.mask_arr()
        addr = self._lds_addr(o[0], m)
        assert np.all(addr[em] % 4 == 0) and np.all(addr[em] + 4 <= self.lds.size), "ds_write_b32: bad address"
        self.lds32[addr[em] // 4] = self.rd_v(o[1])[em]

    def op_ds_write_b64(self, pc, o, m):
        em = self.mask_arr()
        addr = self._lds_addr(o[0], m)
        assert np.all(addr[em] % 8 == 0) and np.all(addr[em] + 8 <= self.lds.size), "ds_write_b64: bad address"
        kind, a, n = o[1]
        for j in range(2):
            self.lds32[addr[em] // 4 + j] = self.V[a + j][em]

    # ---- buffer / global ----
    def _buf_addr(self, o, m):
        # op vdata, vaddr|off, s[rsrc], soffset [offen] [offset:imm]
        rs = o[2][1]
        base = self.S[rs] | ((self.S[rs + 1] & 0xFFFF) << 32)
        num = self.S[rs + 2]
        off = np.full(64, m.get("offset", 0), dtype=np.int64)
        if m.get("offen"):
            off = off + self.rd_v(o[1]).astype(np.int64)
        inb = off < num  # raw buffer range check (offset against num_records)
        soff = self.rd_s(o[3]) if o[3][0] != "off" else 0
        return base + soff + off, inb

    def op_buffer_load_dwordx4(self, pc, o, m):
        em = self.mask_arr()
        addr, inb = self._buf_addr(o, m)
        kind, a, n = o[0]
        for j in range(4):
            val = np.zeros(64, dtype=np.uint32)
            ok = em & inb
            assert np.all(addr[ok] % 4 == 0)
            val[ok] = self.mem32[(addr[ok] // 4) + j]
            self.wr_v(("v", a + j, 1), val)

    def op_buffer_load_dword(self, pc, o, m):
        em = self.mask_arr()
        addr, inb = self._buf_addr(o, m)
        val = np.zeros(64, dtype=np.uint32)
        ok = em & inb
        val[ok] = self.mem32[addr[ok] // 4]
        self.wr_v(o[0], val)

    def op_buffer_store_dword(self, pc, o, m):
        em = self.mask_arr()
        addr, inb = self._buf_addr(o, m)
        ok = em & inb
        assert np.all(addr[ok] % 4 == 0)
        self.mem32[addr[ok] // 4] = self.rd_v(o[0])[ok]

    def _gaddr(self, o_addr, o_s, m):
        kind, a, n = o_addr
        if o_s[0] == "off":
            addr = self.V[a].astype(np.int64) | (self.V[a + 1].astype(np.int64) << 32)
        else:  # saddr form: 64-bit scalar base + 32-bit vector offset
            addr = self.rd_s(o_s, 2) + self.V[a].astype(np.int64)
        return addr + m.get("offset", 0)

    def op_global_store_dword(self, pc, o, m):
        em = self.mask_arr()
        addr = self._gaddr(o[0], o[2], m)
        self.mem32[addr[em] // 4] = self.rd_v(o[1])[em]

    def op_global_store_dwordx4(self, pc, o, m):
        em = self.mask_arr()
        addr = self._gaddr(o[0], o[2], m)
        kind, a, n = o[1]
        assert np.all(addr[em] % 4 == 0)
        for j in range(4):
            self.mem32[addr[em] // 4 + j] = self.V[a + j][em]

    def op_global_load_dword(self, pc, o, m):
        em = self.mask_arr()
        addr = self._gaddr(o[1], o[2], m)
        val = np.zeros(64, dtype=np.uint32)
        val[em] = self.mem32[addr[em] // 4]
        self.wr_v(o[0], val)

    def op_global_atomic_add(self, pc, o, m):
        em = self.mask_arr()
        addr = self._gaddr(o[0], o[2], m)
        d = self.rd_v(o[1])
        for i in np.nonzero(em)[0]:
            self.mem32[addr[i] // 4] += d[i]

    def op_global_atomic_add_x2(self, pc, o, m):
        em = self.mask_arr()
        addr = self._gaddr(o[0], o[2], m)
        kind, a, n = o[1]
        m64 = self.mem.view(np.uint64)
        for i in np.nonzero(em)[0]:
            m64[addr[i] // 8] += np.uint64(int(self.V[a][i]) | (int(self.V[a + 1][i]) << 32))


# ---- list scheduler ------------------------------------------------------------------------------------------------------------
# A lone wave issues an instruction every ~4 clk, but an instruction that reads the result of the one right before it waits for
# the whole pipeline (measured on the first K1h: the pack's dependent chains ran at 14 clk per instruction).  Registers are
# assigned by hand, so reordering cannot spill: within a straight-line region the scheduler only has to respect the true, anti
# and output dependences on registers (vcc / exec / scc included) and the order of memory operations, and it picks, among the
# ready instructions, the one whose operands were produced longest ago.
_BOUNDARY = {"s_memtime", "s_branch", "s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_vccz", "s_cbranch_vccnz", "s_cbranch_execz", "s_getpc_b64", "s_setpc_b64",
             "s_waitcnt", "s_endpgm", "s_nop", "s_sleep"}
_SALU_NO_SCC = {"s_bitset0_b32", "s_bitset1_b32", "s_mov_b32", "s_mov_b64", "s_mul_i32", "s_mul_hi_u32", "s_bfm_b32", "s_bfm_b64", "s_load_dword", "s_load_dwordx2", "s_cselect_b32", "s_cselect_b64"}


def _regs(op):
    """operand string -> list of resource names"""
    kind, a, n = _parse(op)
    if kind in ("v", "s"):
        return [f"{kind}{a + i}" for i in range(max(n, 1))]
    if kind in ("vcc", "vcc_lo", "vcc_hi"):
        return ["vcc"]
    if kind in ("exec", "exec_lo", "exec_hi"):
        return ["exec"]
    if kind == "scc":
        return ["scc"]
    return []


def defs_uses(mnem, ops):
    """-> (defs, uses, is_mem) as lists of resource names"""
    R = [_regs(o) for o in ops]
    flat = lambda idx: [r for i in idx for r in R[i]]
    n = len(ops)
    if mnem.startswith("ds_write") or mnem.startswith("buffer_store") or mnem.startswith("global_store") or mnem.startswith("global_atomic"):
        return [], flat(range(n)) + ["exec"], True
    if mnem.startswith("ds_read") or mnem.startswith("ds_bpermute") or mnem.startswith("buffer_load") or mnem.startswith("global_load"):
        return R[0], flat(range(1, n)) + ["exec"], True
    if mnem.startswith("s_load"):
        return R[0], flat(range(1, n)), True
    if mnem.startswith("v_"):
        if mnem == "v_swap_b32":
            return R[0] + R[1], R[0] + R[1] + ["exec"], False
        if mnem.startswith("v_add_co") or mnem.startswith("v_addc_co"):
            return R[0] + R[1], flat(range(2, n)) + ["exec"], False
        if mnem.startswith("v_cmp"):
            return R[0], flat(range(1, n)) + ["exec"], False
        if mnem == "v_readfirstlane_b32" or mnem == "v_readlane_b32":
            return R[0], flat(range(1, n)) + ["exec"], False
        return R[0], flat(range(1, n)) + ["exec"], False
    if mnem.startswith("s_cmp") or mnem.startswith("s_bitcmp"):
        return ["scc"], flat(range(n)), False
    if mnem.startswith("s_"):
        d = list(R[0])
        u = flat(range(1, n))
        if mnem not in _SALU_NO_SCC:
            d.append("scc")
        if mnem in ("s_addc_u32", "s_cselect_b32", "s_cselect_b64"):
            u.append("scc")
        if mnem in ("s_bitset0_b32", "s_bitset1_b32"):  # read-modify-write of the destination
            u += d
        return d, u, False
    raise NotImplementedError(f"scheduler: {mnem}")


def schedule(prog):
    """reorder the instructions of every straight-line region of prog (in place); returns the number of moved instructions"""
    out = []
    region = []
    moved = [0]

    def flush():
        if len(region) > 2:
            order = _schedule_region(region)
            moved[0] += sum(1 for i, j in enumerate(order) if i != j)
            out.extend(region[j] for j in order)
        else:
            out.extend(region)
        region.clear()

    for c in prog.code:
        if c[0] == "i" and c[1] not in _BOUNDARY:
            region.append(c)
        else:
            flush()
            out.append(c)
    flush()
    prog.code = out
    return moved[0]


def _schedule_region(region):
    n = len(region)
    preds = [set() for _ in range(n)]   # every dependence
    raw = [set() for _ in range(n)]     # true dependences only (the ones that cost latency)
    last_def, last_uses = {}, {}
    last_mem = None
    for i, (_, mnem, ops, mods) in enumerate(region):
        d, u, mem = defs_uses(mnem, ops)
        for r in u:
            if r in last_def:
                preds[i].add(last_def[r])
                raw[i].add(last_def[r])
        for r in d:
            if r in last_def:
                preds[i].add(last_def[r])
            for j in last_uses.get(r, ()):
                if j != i:
                    preds[i].add(j)
        if mem:
            if last_mem is not None:
                preds[i].add(last_mem)
            last_mem = i
        for r in u:
            last_uses.setdefault(r, []).append(i)
        for r in d:
            last_def[r] = i
            last_uses[r] = []
    succs = [[] for _ in range(n)]
    npred = [len(p) for p in preds]
    for i in range(n):
        for j in preds[i]:
            succs[j].append(i)
    ready = [i for i in range(n) if npred[i] == 0]
    pos = {}
    order = []
    while ready:
        # the ready instruction whose newest operand is oldest; ties: program order
        best = min(ready, key=lambda i: (max((pos[j] for j in raw[i]), default=-1), i))
        ready.remove(best)
        pos[best] = len(order)
        order.append(best)
        for k in succs[best]:
            npred[k] -= 1
            if npred[k] == 0:
                ready.append(k)
    assert len(order) == n
    return order
