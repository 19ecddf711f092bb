#!/usr/bin/env python3
"""tools/k1h_mix.py [k] [sb] [gap] — static instruction mix of one K1h variant (gen_k1h.py), per part of the chunk iteration and per
instruction class.  "bit-op VALU" are the instructions two waves of a SIMD overlap completely (profiles/r04_ubench_issue.txt:
v_bitop3 / v_and / v_or / v_xor / v_not / v_mov); every other VALU instruction makes the waves of a pair take turns."""
import collections
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ntcard_amd", "csrc"))
import gen_k1h  # noqa: E402

BITOPS = {"v_bitop3_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32", "v_or3_b32"}


def classify(mn):
    if mn.startswith("v_"):
        return "VALU bit-op" if mn in BITOPS else "VALU other"
    if mn.startswith("ds_"):
        return "LDS"
    if mn.startswith("buffer_") or mn.startswith("global_"):
        return "VMEM"
    if mn.startswith("s_load"):
        return "SMEM"
    return "SALU/branch"


def count(code):
    c = collections.Counter()
    for x in code:
        if x[0] == "i":
            c[classify(x[1])] += 1
    return c


def part(g, fn):
    n0 = len(g.p.code)
    fn()
    return count(g.p.code[n0:])


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    sb = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    gap = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    g = gen_k1h.Gen(k, sb, gap)
    rows = []
    walk = sum((part(g, lambda a=a: g.walk_step(a)) for a in range(16)), collections.Counter())
    rows.append(("walk, 16 steps", walk))
    flags = sum((part(g, lambda a=a: g.flags_and_push(a)) for a in range(16)), collections.Counter())
    rows.append(("sample test + queue push, 16 steps (a suspect push included per step: rarely run)", flags))
    g2 = gen_k1h.Gen(k, sb, gap)
    pack = sum((part(g2, lambda b=b: g2.pack_batch(b)) for b in range(4)), collections.Counter())
    rows.append(("pack, 4 batches of 8 read groups (both branches: real chunk / chunk that does not exist)", pack))
    g3 = gen_k1h.Gen(k, sb, gap)
    rows.append(("transpose + plane rotation", part(g3, g3.transpose_rotate)))
    g4 = gen_k1h.Gen(k, sb, gap)
    rows.append(("one resolve pass (subroutine; log switch and suspect store included: rarely run)", part(g4, g4.emit_pass)))
    full = gen_k1h.Gen(k, sb, gap).build()
    rows.append(("whole kernel body as emitted (prologue, chunk loop, tile switch, epilogue, pass)", count(full.code)))
    classes = ["VALU bit-op", "VALU other", "SALU/branch", "LDS", "VMEM", "SMEM"]
    print(f"K1h k = {k}, sBits class {sb}, gap {gap}: static instruction counts (gen_k1h.py)")
    print("%-100s %s   total" % ("part", "  ".join("%11s" % c for c in classes)))
    for name, c in rows:
        print("%-100s %s   %5d" % (name, "  ".join("%11d" % c[x] for x in classes), sum(c.values())))
    print("\nPer block at run time (PMC, profiles/r04_driver/summary.txt: per wave and block of the driver command): 3 290 VALU, 390 SALU, 190 LDS, 70 VMEM;")
    print("~7 resolve passes per block at sBits = 7 (24 sampled windows per step and tile).")


if __name__ == "__main__":
    main()
