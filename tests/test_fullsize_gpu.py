"""Full-size parity (BASELINE.json configs 2, 4, 5 — both spaced seeds of SURVEY 8(d): -k 12 -g 2 and -k 32 -g 8 —, config 3's sBits=11 sampling: 100 M
synthetic 150 bp reads; rBits = 24 on 20 M) inside
the driver-run GPU suite.

The goldens under tests/golden/fullsize/ were produced in the build container by the REAL reference
(tools/make_fullsize_digests.py -> oracle/_ref/ref_tool fullsize: /root/reference/ntcard.cpp's ntRead / stRead /
outDefault over the same synthetic stream, whose device generator K0 is bit-identical to the oracle's, see
test_generator_matches_oracle).  Here the reads are regenerated on the device, hashed and sketched by the HIP path
through the C ABI, and compared: F1, sha1 of the raw uint16 t_Counter planes, and the .hist bytes.
"""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize")


def _meta():
    with open(os.path.join(GOLD, "digests.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name,flags", [("cfg2", 0), ("cfg2u", 0), ("cfg4", 0), ("cfg5", 0), ("cfg2", 8), ("cfg2u", 2),
                                        ("cfg3s", 0), ("cfg3s", 8), ("cfg4", 8), ("cfg5b", 0), ("cfg2r24", 0)])  # row slots: K1, adaptive / 8 = log forced / 2 = direct atomics
def test_fullsize_matches_reference_goldens(name, flags, tmp_path):
    assert torch.cuda.is_available(), "GPU tests need a HIP device (run on the MI355X box)"
    import ntcard_amd as nt
    meta = _meta()
    cfg = meta["configs"][name]
    n, L, rb, sb, cov = cfg.get("n_reads", meta["n_reads"]), meta["read_len"], cfg.get("r_bits", meta["r_bits"]), cfg.get("s_bits", meta["s_bits"]), meta["cov_max"]
    stride, R = 152, 10_000_000
    buf = torch.empty(R * stride + 16, dtype=torch.uint8, device="cuda")
    with nt.Engine(cfg["klist"], gap=cfg["gap"], r_bits=rb, s_bits=sb, flags=flags) as e:
        for first in range(0, n, R):
            m = min(R, n - first)
            nt.gen_reads_device(buf.data_ptr(), meta["seed"], first, m, L, stride, cfg["dist"], genome_len=100_000_000)
            e.submit_device(buf.data_ptr(), m, L, stride)
            e.sync()  # the batch buffer is regenerated in place
        tc, ph, f1 = e.finish(counters=True)
    for ki, pl in enumerate(cfg["planes"]):
        assert int(f1[ki]) == pl["f1"], (name, pl["k"])
        assert hashlib.sha1(np.ascontiguousarray(tc[ki]).data).hexdigest() == pl["t_counter_sha1"], (name, pl["k"])
        F0, f = nt.estimate(ph[ki], rb, sb, cov)
        out = tmp_path / f"{name}_k{pl['k']}.hist"
        nt.write_hist(out, f1[ki], F0, f, cov)
        got = out.read_bytes()
        assert hashlib.sha1(got).hexdigest() == pl["hist_sha1"], (name, pl["k"])
        assert got == open(os.path.join(GOLD, pl["hist_file"]), "rb").read()
