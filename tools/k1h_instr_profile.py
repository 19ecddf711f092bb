"""tools/k1h_instr_profile.py — dynamic instruction profile of the generated K1h body on the CPU wave emulator: one wave over
whole tiles of random 150 bp reads, executed instructions per label range and per mnemonic.  (Development aid: where do the
instructions of a tile go?)   python tools/k1h_instr_profile.py [k] [read_len] [n_tiles]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import k1h_model  # noqa: E402
import k1h_asm  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 2
n = 2048 * nt
rng = np.random.default_rng(5)
reads = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(n, L))
C = (L + 15) // 16
slots = np.zeros((n, C * 16), dtype=np.uint8)
slots[:, :L] = reads
# tile layout: [tile][chunk][m][lane][16]  (read = 64 m + lane)
tiles = slots.reshape(nt, 32, 64, C, 16).transpose(0, 3, 1, 2, 4).copy().reshape(-1)

per_pc = {}
orig_run = k1h_asm.Emu.run


def run(self, entry=0):
    pc = entry
    nn = len(self.insts)
    cnt = np.zeros(nn + 1, dtype=np.int64)
    while pc < nn:
        mnem, ops, mods = self.insts[pc]
        cnt[pc] += 1
        self.hist[mnem] = self.hist.get(mnem, 0) + 1
        npc = self.step(pc, mnem, ops, mods)
        pc = pc + 1 if npc is None else npc
    self.executed = int(cnt.sum())
    per_pc["cnt"] = cnt
    per_pc["labels"] = dict(self.labels)
    per_pc["hist"] = dict(self.hist)
    per_pc["insts"] = self.insts
    return self.executed


k1h_asm.Emu.run = run
out = k1h_model.run_k1h(tiles, n, L, k, r_bits=16, s_bits=7, n_waves=1, log_regions=64, log_region_cap=1 << 16)
cnt = per_pc["cnt"]
labels = sorted(per_pc["labels"].items(), key=lambda kv: kv[1])
total = int(cnt.sum())
print("k=%d L=%d tiles=%d: %d instructions executed = %.0f per tile, %d keys (%.0f per tile)" % (k, L, nt, total, total / nt, out["keys"].size, out["keys"].size / nt))
print("%-24s %8s %10s %6s" % ("label range", "static", "executed", "%"))
for i, (name, pc) in enumerate(labels):
    end = labels[i + 1][1] if i + 1 < len(labels) else len(cnt) - 1
    ex = int(cnt[pc:end].sum())
    if ex:
        print("%-24s %8d %10d %6.2f" % (name, end - pc, ex, 100.0 * ex / total))
print("by mnemonic:")
for m, c in sorted(per_pc["hist"].items(), key=lambda kv: -kv[1])[:40]:
    print("  %-28s %10d %6.2f" % (m, c, 100.0 * c / total))
