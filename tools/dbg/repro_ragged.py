import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import orc
import ntcard_amd as nt
rng = np.random.default_rng(7)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
lens = 100 + (np.arange(n) * 7919 + 3) % 51
p_bad = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0005
reads = []
for l in lens:
    row = alpha[rng.integers(0, 4, size=l)]
    if p_bad:
        row = np.where(rng.random(l) < p_bad, ord("N"), row).astype(np.uint8)
    reads.append(row.tobytes())
oc, of1 = orc.sketch_reads(reads, [32], 0, 20, 7)
for flags in (0, nt.FLAG_REQUIRE_TILED, nt.FLAG_LANE_KERNEL):
    with nt.Engine([32], r_bits=20, s_bits=7, flags=flags) as e:
        e.submit_reads(reads)
        tc, ph, f1 = e.finish(counters=True)
    print("flags", flags, "f1", int(f1[0]), int(of1[0]), "counters equal", np.array_equal(tc, oc), "diff", int((tc != oc).sum()))
