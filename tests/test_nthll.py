"""nthll: oracle vs the reference's registers/estimate (CPU), HIP path vs oracle and the
reference CLI's stdout (GPU)."""
import gzip
import json
import os
import random
import subprocess

import numpy as np
import pytest

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
NTHLL = os.path.join(ROOT, "ntcard_amd", "bin", "nthll")


def small_reads():
    with gzip.open(os.path.join(GOLD, "reads_small.fq.gz"), "rb") as f:
        lines = f.read().split(b"\n")
    return [lines[i] for i in range(1, len(lines) - 1, 4)], b"\n".join(lines)


def goldens():
    with open(os.path.join(GOLD, "nthll_goldens.json")) as f:
        return json.load(f)


def test_oracle_matches_reference_goldens():
    reads, _ = small_reads()
    for c in goldens()["cases"]:
        regs, est = orc.hll_reads(reads, c["k"], c["n_bits"])
        assert "%016x" % orc.fnv1a64(regs) == c["fnv1a64"] and int(regs.max()) == c["max"]
        assert "F0, Exp# of distnt kmers(k=%d): %d\n" % (c["k"], int(est)) == c["line"]


def test_host_estimate_matches_oracle():
    import ntcard_amd as nt
    rng = np.random.default_rng(5)
    for nb in (8, 12, 16):
        regs = rng.integers(0, 40, size=1 << nb, dtype=np.uint8)
        assert nt.hll_estimate(regs, nb) == orc.lib().orc_hll_estimate(regs.ctypes.data, nb)


@pytest.mark.skipif(not orc.have_ref(), reason="real reference build only exists in the build container")
def test_oracle_live_against_reference():
    rng = random.Random(8)
    seqs = ["".join(rng.choice("ACGTNacgu") for _ in range(rng.randint(0, 200))).encode() for _ in range(500)]
    for k, nb in ((16, 9), (33, 16)):
        rr, _ = orc.ref_hll(seqs, k, nb)
        oregs, _ = orc.hll_reads(seqs, k, nb)
        assert np.array_equal(rr, oregs)


@pytest.mark.gpu
def test_gpu_registers_match_oracle_and_goldens():
    import ntcard_amd as nt
    reads, _ = small_reads()
    for c in goldens()["cases"]:
        with nt.HllEngine(c["k"], c["n_bits"]) as e:
            e.submit_reads(reads)
            regs, f1 = e.finish()
        assert "%016x" % orc.fnv1a64(regs) == c["fnv1a64"]
        assert "F0, Exp# of distnt kmers(k=%d): %d\n" % (c["k"], int(nt.hll_estimate(regs, c["n_bits"]))) == c["line"]
    # dirty, ragged input in several submits (thresholds get refreshed between sub-batches)
    rng = random.Random(21)
    seqs = ["".join(rng.choice("ACGT" * 20 + "Nacgu") for _ in range(rng.choice([20, 100, 150, 151, 300, 4000]))).encode()
            for _ in range(30000)]
    with nt.HllEngine(25, 14) as e:
        for i in range(0, len(seqs), 7000):
            e.submit_reads(seqs[i:i + 7000])
        regs, f1 = e.finish()
    oregs, _ = orc.hll_reads(seqs, 25, 14)
    assert np.array_equal(regs, oregs)
    assert f1 == sum(len(orc.hash_read(s, 25)[0]) for s in seqs)


@pytest.mark.gpu
def test_gpu_large_device_batch_matches_oracle():
    """2 M reads from the device generator: the warm-up sub-batching and the moving threshold stay exact"""
    import torch
    import ntcard_amd as nt
    n, L, stride = 2_000_000, 150, 152
    d = torch.empty(n * stride + 16, dtype=torch.uint8, device="cuda")
    nt.gen_reads_device(d.data_ptr(), 3, 0, n, L, stride, 1, genome_len=50_000_000)
    with nt.HllEngine(32, 16) as e:
        e.submit_device(d.data_ptr(), n, L, stride)
        regs, f1 = e.finish()
    host = d[: n * stride].cpu().numpy().reshape(n, stride)[:, :L]
    bases = np.ascontiguousarray(host).reshape(-1)
    offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    oregs = np.zeros(1 << 16, dtype=np.uint8)
    orc.lib().orc_hll_update(oregs.ctypes.data, 16, bases.ctypes.data, offs.ctypes.data, n, 32, 0)
    assert np.array_equal(regs, oregs)


@pytest.mark.gpu
def test_gpu_fullsize_matches_the_reference():
    """BASELINE config 5 names nthll too: 100 M synthetic 150 bp reads (the device generator's stream) against the register file and the printed line of the
    REAL reference (oracle/_ref/ref_hll_tool fullsize = nthll.cpp's ntRead over the same stream: tools/make_fullsize_digests.py hll)"""
    import hashlib
    import json
    import torch
    import ntcard_amd as nt
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize")
    with open(os.path.join(gold, "digests.json")) as f:
        meta = json.load(f)
    c = meta["configs"]["hll"]
    n, L, stride, R = meta["n_reads"], meta["read_len"], 152, 10_000_000
    d = torch.empty(R * stride + 16, dtype=torch.uint8, device="cuda")
    with nt.HllEngine(c["k"], c["n_bits"]) as e:
        for first in range(0, n, R):
            m = min(R, n - first)
            nt.gen_reads_device(d.data_ptr(), meta["seed"], first, m, L, stride, c["dist"], genome_len=100_000_000)
            e.submit_device(d.data_ptr(), m, L, stride)
        regs, f1 = e.finish()
    want = np.fromfile(os.path.join(gold, c["regs_file"]), dtype=np.uint8)
    assert hashlib.sha1(want.tobytes()).hexdigest() == c["regs_sha1"]
    assert np.array_equal(regs, want)
    assert "F0, Exp# of distnt kmers(k=%d): %d\n" % (c["k"], int(nt.hll_estimate(regs, c["n_bits"]))) == c["line"]


@pytest.mark.gpu
def test_nthll_cli_prints_the_reference_line(tmp_path):
    _, fq = small_reads()
    (tmp_path / "reads.fq").write_bytes(fq)
    g = goldens()
    r = subprocess.run([NTHLL, "-k", "32", "reads.fq"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0 and r.stdout.decode() == g["cli_k32"], r.stderr
    r = subprocess.run([NTHLL, "-k", "20", "-b", "12", "reads.fq"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0 and r.stdout.decode() == g["cli_k20_b12"], r.stderr


def test_nthll_cli_argument_errors():
    r = subprocess.run([NTHLL], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"nthll: missing arguments" in r.stderr
    r = subprocess.run([NTHLL, "--version"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and b"nthll" in r.stderr


REF_NTHLL = os.path.join(ROOT, "oracle", "_ref", "nthll_ref")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_NTHLL), reason="the real reference binary (oracle/_ref) was not built")
def test_nthll_live_differential_against_reference_binary(tmp_path):
    """both nthll command lines on a multi-line FASTA with long, dirty contigs and on ragged FASTQ reads"""
    rng = random.Random(4)
    fa = []
    for i, n in enumerate([0, 10, 64, 5000, 90_000]):
        s = "".join(rng.choice("ACGT" * 12 + "Nacgt") for _ in range(n))
        fa.append(">s%d\n" % i + "\n".join(s[j:j + 70] for j in range(0, len(s), 70)) + ("\n" if s else ""))
    (tmp_path / "c.fa").write_text("".join(fa))
    fq = []
    for i in range(8000):
        s = "".join(rng.choice("ACGT" * 20 + "N") for _ in range(rng.choice([30, 100, 150, 151])))
        fq.append("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
    (tmp_path / "r.fq").write_text("".join(fq))
    for args in (["-k", "32", "c.fa"], ["-k", "21", "-b", "14", "r.fq"], ["-k", "48", "c.fa", "r.fq"]):
        ours = subprocess.run([NTHLL] + args, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        ref = subprocess.run([REF_NTHLL] + args, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert ours.returncode == 0 and ref.returncode == 0, (ours.stderr, ref.stderr)
        assert ours.stdout == ref.stdout, (args, ours.stdout, ref.stdout)
