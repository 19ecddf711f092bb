import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import orc
import ntcard_amd as nt
rng = np.random.default_rng(7)
alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
def mk(n, p_bad=0.0005):
    lens = 100 + (np.arange(n) * 7919 + 3) % 51
    out = []
    for l in lens:
        row = alpha[rng.integers(0, 4, size=l)]
        row = np.where(rng.random(l) < p_bad, ord("N"), row).astype(np.uint8)
        out.append(row.tobytes())
    return out
batches = [mk(n) for n in [int(x) for x in os.environ.get("SIZES", "133000,133000,120000,140000,133000").split(",")]]
allr = sum(batches, [])
oc, of1 = orc.sketch_reads(allr, [32], 0, 20, 7)
for flags in (0, nt.FLAG_REQUIRE_TILED):
    with nt.Engine([32], r_bits=20, s_bits=7, flags=flags) as e:
        for b in batches:
            e.submit_reads(b)
        tc, ph, f1 = e.finish(counters=True)
    print("flags", flags, "f1", int(f1[0]), int(of1[0]), "counters equal", np.array_equal(tc, oc), "diff", int((tc != oc).sum()))
    # one by one against the oracle of the prefix
with nt.Engine([32], r_bits=20, s_bits=7, flags=0) as e:
    done = []
    for i, b in enumerate(batches):
        e.submit_reads(b)
        done += b
        tc, ph, f1 = e.finish(counters=True)
        oc2, of2 = orc.sketch_reads(done, [32], 0, 20, 7)
        print("after batch", i, "equal", np.array_equal(tc, oc2), "diff", int((tc != oc2).sum()))
