"""k1h_terms.py — the Boolean terms of the bit-sliced ntHash walk, derived from the four seeds (used by gen_k1h.py).

Bit-sliced layout: one VGPR holds ONE bit of the 31-bit rotating half H of the hash (nthash.hpp:186-217) for 32
different reads (bit i of lane l <-> read 64*i + l of the tile), so a wave carries 2048 reads.  A rotate is a
renaming of registers; the seed terms of the incoming / outgoing base are Boolean functions of the base's two
code bits (b0, b1), evaluated once per step for all 31 hash bits that share them ("function planes", one
v_bitop3_b32 each) and folded in with one three-input XOR per hash bit:

    forward  NTF64 (nthash.hpp:242-248):  F'[j] = F[j-1] ^ S[j](in)      ^ S[j-k](out)
    reverse  NTR64 (nthash.hpp:251-257):  R'[j] = R[j+1] ^ Sc[j+1-k](in) ^ Sc[j+1](out)

with S[m](c) = bit m of the H half of seed(c) (nthash.hpp:25-28) and Sc[m](c) = S[m](complement(c)); indices mod 31.

Base code: code2 = (ascii >> 1) & 3  ->  A=0 C=1 T/U=2 G=3 (either case); b0 = bit 0, b1 = bit 1; the complement flips b1.

Nothing here is copied from the reference: the four 64-bit seeds are its constants (nthash.hpp:25-28), everything
else is derived.  `python k1h_terms.py` checks the terms against a plain 64-bit rolling hash.
(Rounds 2 and 3 generated two more kernels, K1b and K1c, from these tables; round 5 retired both: K1h is the one bit-sliced kernel.)
"""
import sys  # noqa: F401

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


def rol31(x, n):
    n %= 31
    return ((x << n) | (x >> (31 - n))) & M31 if n else x


def selftest():
    """the per-bit terms reproduce the rolling 64-bit hash's H half (nthash.hpp:242-257) on random sequences, for several k"""
    import random
    rnd = random.Random(5)
    M64 = (1 << 64) - 1

    def srol(x, n=1):  # split rotate: 33-bit low field and 31-bit high field rotate separately (nthash.hpp:186-217)
        for _ in range(n):
            m = ((x & 0x8000000000000000) >> 30) | ((x & 0x100000000) >> 32)
            x = ((x << 1) & 0xFFFFFFFDFFFFFFFF) | m
        return x & M64
    for k in (12, 20, 32):
        f_terms, r_terms = step_terms(k)
        seq = [rnd.choice("ACGT") for _ in range(k + 40)]
        fh = rh = 0
        for i in range(k):
            fh ^= srol(SEED[seq[i]], k - 1 - i)
            rh ^= srol(SEED[COMP[seq[i]]], i)
        F, R = fh >> 33, rh >> 33
        for w in range(40):
            out_b, in_b = seq[w], seq[w + k]
            fh = srol(fh) ^ srol(SEED[out_b], k) ^ SEED[in_b]
            m = rh ^ SEED[COMP[out_b]] ^ srol(SEED[COMP[in_b]], k)
            # inverse split rotate by one
            lo = ((m & 1) << 32) | ((m & 0x1FFFFFFFF) >> 1)
            hi = m >> 33
            hi = ((hi & 1) << 30) | (hi >> 1)
            rh = (hi << 33) | lo
            ci, co = CODE2[in_b], CODE2[out_b]
            Fn = [((F >> ((j - 1) % 31)) & 1) ^ ((f_terms[j][0] >> ci) & 1) ^ ((f_terms[j][1] >> co) & 1) for j in range(31)]
            Rn = [((R >> ((j + 1) % 31)) & 1) ^ ((r_terms[j][0] >> ci) & 1) ^ ((r_terms[j][1] >> co) & 1) for j in range(31)]
            F = sum(b << j for j, b in enumerate(Fn))
            R = sum(b << j for j, b in enumerate(Rn))
            assert F == fh >> 33 and R == rh >> 33, (k, w)
    print("k1h_terms selftest ok")


if __name__ == "__main__":
    selftest()
