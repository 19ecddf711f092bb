#!/usr/bin/env python3
"""gen_bs.py — emits ntc_bs_gen.inc: the straight-line step bodies of the bit-sliced filter walk (K1b).

Bit-sliced layout: one VGPR holds ONE bit of the 31-bit rotating half H of the hash (nthash.hpp:186-217) for 32
different reads (bit i of lane l <-> read 64*i + l of the tile), so a wave carries 2048 reads.  A rotate is a
renaming of registers; the seed terms of the incoming / outgoing base are Boolean functions of the base's two
code bits (b0, b1), evaluated once per step for all 31 hash bits that share them ("function planes", one
v_bitop3_b32 each) and folded in with one three-input XOR per hash bit:

    forward  NTF64 (nthash.hpp:242-248):  F'[j] = F[j-1] ^ S[j](in)      ^ S[j-k](out)
    reverse  NTR64 (nthash.hpp:251-257):  R'[j] = R[j+1] ^ Sc[j+1-k](in) ^ Sc[j+1](out)

with S[m](c) = bit m of the H half of seed(c) (nthash.hpp:25-28) and Sc[m](c) = S[m](complement(c)); indices mod 31.
Only ntComp's sampling test (ntcard.cpp:135-138) is evaluated here, on the top bits of min(F, R); the sampled
windows are re-derived exactly from the bases by the resolve stage (ntc_sketch_bs.hip).

Base code used throughout K1b: code2 = (ascii >> 1) & 3  ->  A=0 C=1 T/U=2 G=3 (either case); b0 = bit 0, b1 = bit 1;
the complement flips b1.

Nothing here is copied from the reference: the four 64-bit seeds are its constants (nthash.hpp:25-28), everything
else is derived.  `python gen_bs.py --selftest` checks the model against a plain 64-bit rolling hash.
"""
import sys

SEED = {"A": 0x3c8bfbb395c60474, "C": 0x3193c18562a02b4c, "G": 0x20323ed082572324, "T": 0x295549f54be24456}
CODE2 = {"A": 0, "C": 1, "T": 2, "G": 3}
BASE_OF = {v: k for k, v in CODE2.items()}
COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}
M31 = (1 << 31) - 1


def hseed(base):
    return SEED[base] >> 33


def tt4(m, comp=False):
    """4-bit truth table (index = code2) of bit m of the H half of seed(c) (or of seed(complement(c)))"""
    t = 0
    for c2 in range(4):
        b = BASE_OF[c2]
        if comp:
            b = COMP[b]
        t |= ((hseed(b) >> (m % 31)) & 1) << c2
    return t


def step_terms(k):
    """per hash bit j: (tt4 of the in-term, tt4 of the out-term) for both strands"""
    f = [(tt4(j), tt4(j - k)) for j in range(31)]
    r = [(tt4(j + 1 - k, True), tt4(j + 1, True)) for j in range(31)]
    return f, r


# ---- truth tables for v_bitop3_b32: result bit = ttbl[(s0 << 2) | (s1 << 1) | s2] -------------------------
def ttbl(fn):
    t = 0
    for i in range(8):
        if fn((i >> 2) & 1, (i >> 1) & 1, i & 1) & 1:
            t |= 1 << i
    return t


def g_of(t4):
    return lambda b0, b1: (t4 >> (b0 | (b1 << 1))) & 1


class Emit:
    def __init__(self):
        self.lines = []
        self.n = 0

    def tmp(self):
        self.n += 1
        return f"t{self.n}"

    def op(self, a, b, c, fn, comment=""):
        v = self.tmp()
        self.lines.append(f"\tconst uint32_t {v} = __builtin_amdgcn_bitop3_b32({a}, {b}, {c}, 0x{ttbl(fn):02x});" + (f" // {comment}" if comment else ""))
        return v


def plane(e, cache, t4, b0, b1):
    """function plane g(b0, b1) for a 4-bit table: returns (expr or None, invert flag)"""
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
    if t4 & 1:  # keep g(0,0) = 0 so that a plane and its complement share one register
        t4 ^= 0xf
        inv = 1
    if t4 not in cache:
        g = g_of(t4)
        cache[t4] = e.op(b0, b1, b1, lambda s0, s1, s2: g(s0, s1), f"plane {t4:04b}")
    return cache[t4], inv


def emit_step(name, k, main):
    f_terms, r_terms = step_terms(k)
    e = Emit()
    args = "uint32_t (&F)[31], uint32_t (&R)[31], const uint32_t i0, const uint32_t i1" + (", const uint32_t o0, const uint32_t o1" if main else "")
    head = f"__device__ __forceinline__ void {name}({args})\n{{"
    cin, cout = {}, {}
    newF, newR = [], []
    for j in range(31):
        for strand, terms, prev in (("F", f_terms, f"F[{(j - 1) % 31}]"), ("R", r_terms, f"R[{(j + 1) % 31}]")):
            tin, tout = terms[j]
            if main:
                x, xi = plane(e, cin, tin, "i0", "i1")
                y, yi = plane(e, cout, tout, "o0", "o1")
                inv = xi ^ yi
                if x is None and y is None:
                    v = e.op(prev, prev, prev, lambda a, b, c: a ^ inv)
                elif y is None:
                    v = e.op(prev, x, x, lambda a, b, c: a ^ b ^ inv)
                elif x is None:
                    v = e.op(prev, y, y, lambda a, b, c: a ^ b ^ inv)
                else:
                    v = e.op(prev, x, y, lambda a, b, c: a ^ b ^ c ^ inv)
            else:  # window filling: no outgoing base yet, the in-term folds straight into the XOR
                g = g_of(tin)
                v = e.op(prev, "i0", "i1", lambda a, b, c: a ^ g(b, c))
            (newF if strand == "F" else newR).append(v)
    body = e.lines + [f"\tF[{j}] = {newF[j]};" for j in range(31)] + [f"\tR[{j}] = {newR[j]};" for j in range(31)]
    nops = len(e.lines)
    return head + "\n" + "\n".join(body) + "\n}\n", nops


def emit_test(name, s_bits):
    """ntComp's sampling test (ntcard.cpp:135-138) on the top bits of min(F, R); a[7]..a[0] = F[30]..F[23].
    s_bits <= 7: exact on the top s_bits+1 bits.  s_bits >= 8: a superset on the top 8 bits (patterns 0x00 and
    0x7f: the prefixes of 0..01 and 01..1), narrowed to the exact set by the resolve stage."""
    e = Emit()
    n = min(s_bits + 1, 8)
    a = [f"F[{30 - (n - 1) + i}]" for i in range(n)]  # a[i] = bit i of the n-bit prefix, a[n-1] = top bit
    b = [f"R[{30 - (n - 1) + i}]" for i in range(n)]

    def reduce3(vals, fn2, fn3):
        vals = list(vals)
        while len(vals) > 1:
            if len(vals) >= 3:
                x, y, z = vals[:3]
                vals = vals[3:] + [e.op(x, y, z, fn3)]
            else:
                x, y = vals
                vals = [e.op(x, y, y, fn2)]
        return vals[0]

    OR2, OR3 = (lambda p, q, r: p | q), (lambda p, q, r: p | q | r)
    AND2, AND3 = (lambda p, q, r: p & q), (lambda p, q, r: p & q & r)
    if s_bits <= 7:
        # sample 0: min == 0..01  <=>  (a == 1 and b >= 1) or (b == 1 and a >= 1)
        nza = reduce3(a[1:], OR2, OR3) if n > 1 else None  # some bit above bit 0 set
        nzb = reduce3(b[1:], OR2, OR3) if n > 1 else None
        a1 = e.op(nza, a[0], a[0], lambda p, q, r: (~p) & q, "a == 1")
        b1 = e.op(nzb, b[0], b[0], lambda p, q, r: (~p) & q, "b == 1")
        age = e.op(nza, a[0], a[0], lambda p, q, r: p | q, "a >= 1")
        bge = e.op(nzb, b[0], b[0], lambda p, q, r: p | q, "b >= 1")
        s0a = e.op(a1, bge, bge, AND2)
        s0 = e.op(s0a, b1, age, lambda p, q, r: p | (q & r), "sample 0")
        # sample 1: top s_bits bits of min == 01..1: A = a >> 1 (s_bits bits): (A == 01..1 and B >= 01..1) or (B == .. and A >= ..)
        if s_bits >= 2:
            pa = reduce3(a[1:n - 1], AND2, AND3) if n > 2 else None  # all bits below the top one (of A) set
            pb = reduce3(b[1:n - 1], AND2, AND3) if n > 2 else None
            ta, tb = a[n - 1], b[n - 1]
            if pa is None:  # s_bits == 1 never happens (engine requires s_bits >= 2)
                raise ValueError("s_bits too small")
            aeq = e.op(ta, pa, pa, lambda p, q, r: (~p) & q, "A == 01..1")
            beq = e.op(tb, pb, pb, lambda p, q, r: (~p) & q)
            ageq = e.op(ta, pa, pa, lambda p, q, r: p | q, "A >= 01..1")
            bgeq = e.op(tb, pb, pb, lambda p, q, r: p | q)
            s1a = e.op(aeq, bgeq, s0, lambda p, q, r: (p & q) | r)
            hit = e.op(s1a, beq, ageq, lambda p, q, r: p | (q & r), "sample 0 or 1")
    else:
        # candidates: top 8 bits of min == 0x00 (a == 0 or b == 0) or == 0x7f
        nza = reduce3(a, OR2, OR3)
        nzb = reduce3(b, OR2, OR3)
        pa = reduce3(a[:7], AND2, AND3)
        pb = reduce3(b[:7], AND2, AND3)
        aeq = e.op(a[7], pa, pa, lambda p, q, r: (~p) & q, "a == 0x7f")
        beq = e.op(b[7], pb, pb, lambda p, q, r: (~p) & q)
        ageq = e.op(a[7], pa, pa, lambda p, q, r: p | q, "a >= 0x7f")
        bgeq = e.op(b[7], pb, pb, lambda p, q, r: p | q)
        z = e.op(nza, nzb, nzb, lambda p, q, r: (~p) | (~q), "min == 0")
        s1a = e.op(aeq, bgeq, z, lambda p, q, r: (p & q) | r)
        hit = e.op(s1a, beq, ageq, lambda p, q, r: p | (q & r))
    head = f"__device__ __forceinline__ uint32_t {name}(const uint32_t (&F)[31], const uint32_t (&R)[31])\n{{"
    return head + "\n" + "\n".join(e.lines) + f"\n\treturn {hit};\n}}\n", len(e.lines)


def generate(ks=(32,)):
    out = ["// ntc_bs_gen.inc — GENERATED by gen_bs.py (do not edit): step bodies of the bit-sliced filter walk.",
           "// See gen_bs.py for the derivation; tables follow from the four seeds of nthash.hpp:25-28.", ""]
    stats = {}
    for k in ks:
        s, n1 = emit_step(f"bs_step_warm_k{k}", k, False)
        out.append(s)
        s, n2 = emit_step(f"bs_step_main_k{k}", k, True)
        out.append(s)
        stats[k] = (n1, n2)
    for sb in (2, 3, 4, 5, 6, 7, 8):
        s, n = emit_test(f"bs_test_s{sb}", sb)
        out.append(s)
        stats[f"s{sb}"] = n
    return "\n".join(out), stats


# ---- model (self test) ------------------------------------------------------------------------------------
def rol31(x, n):
    n %= 31
    return ((x << n) | (x >> (31 - n))) & M31 if n else x


def selftest():
    import random
    rng = random.Random(7)
    for k in (16, 32, 48, 20, 33):
        f_terms, r_terms = step_terms(k)
        nreads, L = 32, 100
        reads = ["".join(rng.choice("ACGT") for _ in range(L)) for _ in range(nreads)]
        # reference: plain rolling H halves per read
        ref = []
        for s in reads:
            hs = []
            for p in range(L - k + 1):
                fh = rh = 0
                for i in range(k):
                    fh ^= rol31(hseed(s[p + i]), k - 1 - i)
                    rh ^= rol31(hseed(COMP[s[p + i]]), i)
                hs.append((fh, rh))
            ref.append(hs)
        # bit-sliced model
        F, R = [0] * 31, [0] * 31

        def planes(pos):
            b0 = b1 = 0
            for i, s in enumerate(reads):
                c = CODE2[s[pos]]
                b0 |= (c & 1) << i
                b1 |= (c >> 1) << i
            return b0, b1
        full = (1 << nreads) - 1
        for j in range(L):
            i0, i1 = planes(j)
            if j >= k:
                o0, o1 = planes(j - k)
            nF, nR = [0] * 31, [0] * 31
            for b in range(31):
                for st, terms, prev, dst in (("F", f_terms, F[(b - 1) % 31], nF), ("R", r_terms, R[(b + 1) % 31], nR)):
                    tin, tout = terms[b]
                    v = prev
                    for i in range(nreads):
                        cin = ((i0 >> i) & 1) | (((i1 >> i) & 1) << 1)
                        v ^= ((tin >> cin) & 1) << i
                        if j >= k:
                            co = ((o0 >> i) & 1) | (((o1 >> i) & 1) << 1)
                            v ^= ((tout >> co) & 1) << i
                    dst[b] = v & full
            F, R = nF, nR
            if j >= k - 1:
                p = j - k + 1
                for i in range(nreads):
                    fh = sum(((F[b] >> i) & 1) << b for b in range(31))
                    rh = sum(((R[b] >> i) & 1) << b for b in range(31))
                    assert (fh, rh) == ref[i][p], (k, j, i)
    # the generated tests against the definition
    for sb in (2, 3, 5, 7):
        n = sb + 1
        for a in range(1 << n):
            for b in range(1 << n):
                m = min(a, b)
                want = (m == 1) or ((m >> 1) == (1 << (sb - 1)) - 1)
                got = eval_test(sb, a, b)
                assert got == want, (sb, a, b)
    for a in range(256):
        for b in range(256):
            m = min(a, b)
            assert eval_test(8, a, b) == (m == 0 or m == 0x7f), (a, b)
    print("gen_bs selftest ok")


def eval_test(sb, a, b):
    """run the emitted test function symbolically on one pair of prefixes"""
    src, _ = emit_test("t", sb)
    n = min(sb + 1, 8)
    env = {}
    F = [0] * 31
    R = [0] * 31
    for i in range(n):
        F[30 - (n - 1) + i] = (a >> i) & 1
        R[30 - (n - 1) + i] = (b >> i) & 1
    for line in src.splitlines():
        line = line.strip()
        if line.startswith("const uint32_t"):
            name = line.split()[2]
            inner = line[line.index("bitop3_b32(") + 11: line.index(");")]
            x, y, z, t = [s.strip() for s in inner.split(",")]
            val = lambda s: env[s] if s in env else (F[int(s[2:-1])] if s[0] == "F" else R[int(s[2:-1])])
            idx = (val(x) << 2) | (val(y) << 1) | val(z)
            env[name] = (int(t, 16) >> idx) & 1
        elif line.startswith("return"):
            return bool(env[line.split()[1].rstrip(";")])
    raise RuntimeError("no return")


if __name__ == "__main__":
    if "--selftest" in sys.argv:
        selftest()
        sys.exit(0)
    text, stats = generate()
    path = sys.argv[1] if len(sys.argv) > 1 else "ntc_bs_gen.inc"
    with open(path, "w") as f:
        f.write(text)
    print("wrote", path, stats)
