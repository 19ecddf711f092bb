"""The drop-in `ntcard` front end (ntcard_amd/bin/ntcard): option handling on CPU, and — on the
GPU box — byte-for-byte reproduction of the reference CLI's own outputs (tests/golden/ref_*), for
every input rendering the reference's `make check` exercises (Makefile.am:47-83: DNA FASTQ .gz,
RNA, FASTA, each with and without -g 2) plus SAM, multi-k, -c, -o, -r and @list/-t."""
import gzip
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "ntcard_amd", "bin", "ntcard")
GOLD = os.path.join(ROOT, "tests", "golden")


def run(args, cwd=None):
    return subprocess.run([BIN] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)


def test_help_and_version_exit_zero():
    for flag, word in (("--help", b"Usage: ntCard"), ("--version", b"ntCard 1.2.2")):
        r = run([flag])
        assert r.returncode == 0 and word in r.stderr  # the reference prints both on stderr (ntcard.cpp:365-370)


def test_argument_errors_match_reference():
    r = run(["-k", "12"])
    assert r.returncode == 1
    assert b"ntCard: missing arguments\n" in r.stderr and b"ntCard: missing argument -p/-o ... \n" in r.stderr
    assert r.stderr.endswith(b"Try `ntCard --help' for more information.\n")
    r = run(["-p", "x", "reads.fq"])
    assert r.returncode == 1 and b"ntCard: missing argument -k ... \n" in r.stderr
    r = run(["-k", "12", "-g", "3", "-p", "x", "reads.fq"])
    assert r.returncode == 1 and b"Gap size and kmer must have the same modulus" in r.stderr
    r = run(["-k", "12,14", "-g", "2", "-p", "x", "reads.fq"])
    assert r.returncode == 1 and b"-g does not support multiple k currently." in r.stderr
    r = run(["-k", "12", "-c", "10x", "-p", "x", "reads.fq"])
    assert r.returncode == 1 and b"ntCard: invalid option: `-c10x'" in r.stderr
    r = run(["-k", "12", "-l", "5", "-p", "x", "reads.fq"])  # -l/-f: in the option string, never handled
    assert r.returncode == 1 and b"invalid option: `-l5'" in r.stderr


@pytest.fixture(scope="module")
def inputs(tmp_path_factory):
    d = tmp_path_factory.mktemp("cli")
    gz = os.path.join(GOLD, "reads_small.fq.gz")
    with gzip.open(gz, "rb") as f:
        fq = f.read()
    lines = fq.split(b"\n")
    names = [lines[i][1:] for i in range(0, len(lines) - 1, 4)]
    seqs = [lines[i] for i in range(1, len(lines) - 1, 4)]
    (d / "reads.fq.gz").write_bytes(open(gz, "rb").read())
    (d / "reads.fq").write_bytes(fq)
    (d / "rna.fq").write_bytes(b"".join(b"@%s\n%s\n+\n%s\n" % (n, s.replace(b"T", b"U"), b"I" * len(s)) for n, s in zip(names, seqs)))
    # multi-line FASTA: sequences wrapped at 40 columns (k-mers span the line breaks, ntcard.cpp:198-201)
    (d / "reads.fa").write_bytes(b"".join(b">%s\n%s\n%s\n" % (n, s[:40], s[40:]) for n, s in zip(names, seqs)))
    (d / "reads.sam").write_bytes(b"@HD\tVN:1.0\n@SQ\tSN:chr1\tLN:40000\n" + b"".join(
        b"%s\t0\tchr1\t1\t60\t70M\t*\t0\t0\t%s\t%s\n" % (n, s, b"I" * len(s)) for n, s in zip(names, seqs)))
    (d / "nohdr.sam").write_bytes(b"".join(
        b"%s\t0\tchr1\t1\t60\t70M\t*\t0\t0\t%s\t%s\n" % (n, s, b"I" * len(s)) for n, s in zip(names, seqs)))
    half = len(seqs) // 2
    for tag, sl in (("a", slice(0, half)), ("b", slice(half, None))):
        (d / f"part_{tag}.fq").write_bytes(b"".join(b"@%s\n%s\n+\n%s\n" % (n, s, b"I" * len(s)) for n, s in zip(names[sl], seqs[sl])))
    (d / "parts.txt").write_text("part_a.fq\npart_b.fq\n")
    return d


def gold(name):
    return open(os.path.join(GOLD, name), "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("src", ["reads.fq.gz", "reads.fq", "rna.fq", "reads.fa", "reads.sam", "nohdr.sam"])
def test_every_input_rendering_gives_the_reference_hist(inputs, src):
    r = run(["-k", "12", "-p", "out_" + src.replace(".", "_"), src], cwd=inputs)
    assert r.returncode == 0, r.stderr
    assert b"Runtime(sec): " in r.stderr
    assert (inputs / ("out_" + src.replace(".", "_") + "_k12.hist")).read_bytes() == gold("ref_k12__out_k12.hist")
    r = run(["-k", "12", "-g", "2", "-p", "gap_" + src.replace(".", "_"), src], cwd=inputs)
    assert r.returncode == 0, r.stderr
    assert (inputs / ("gap_" + src.replace(".", "_") + "_k12.hist")).read_bytes() == gold("ref_k12_g2__out_k12.hist")


@pytest.mark.gpu
def test_multi_k_cov_rbits_compact_and_lists(inputs):
    r = run(["-k", "16,24,32,48", "-p", "m", "reads.fq"], cwd=inputs)
    assert r.returncode == 0, r.stderr
    for k in (16, 24, 32, 48):
        assert (inputs / f"m_k{k}.hist").read_bytes() == gold(f"ref_multi__out_k{k}.hist")
    r = run(["-k", "20", "-c", "50", "-p", "c", "reads.fq"], cwd=inputs)
    assert r.returncode == 0 and (inputs / "c_k20.hist").read_bytes() == gold("ref_k20_c50__out_k20.hist")
    r = run(["-k", "24", "-r", "22", "-p", "r", "reads.fq"], cwd=inputs)
    assert r.returncode == 0 and (inputs / "r_k24.hist").read_bytes() == gold("ref_k24_s11_r22__out_k24.hist")
    r = run(["-k", "12,20", "-c", "20", "-o", "compact.tsv", "reads.fq"], cwd=inputs)
    assert r.returncode == 0 and (inputs / "compact.tsv").read_bytes() == gold("ref_compact__k12_20_c20.tsv")
    assert b"k=12\tF1\t" in r.stderr and b"k=20\tF0\t" in r.stderr
    # @list + two parser threads feeding one engine (ntcard.cpp:415-425,445-446)
    r = run(["-t", "2", "-k", "32", "-p", "l", "@parts.txt"], cwd=inputs)
    assert r.returncode == 0, r.stderr
    assert (inputs / "l_k32.hist").read_bytes() == gold("ref_k32__out_k32.hist")


@pytest.mark.gpu
def test_files_spread_over_several_engines_merge_exactly(inputs):
    """NTCARD_DEVICES: one engine (private sketch) per listed device, files dealt round-robin, sketches merged at the
    end.  Listing device 0 twice exercises the whole path on a one-GPU box."""
    env = dict(os.environ, NTCARD_DEVICES="0,0")
    r = subprocess.run([BIN, "-t", "2", "-k", "16,24,32,48", "-p", "md", "part_a.fq", "part_b.fq"], cwd=inputs, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr
    for k in (16, 24, 32, 48):
        assert (inputs / f"md_k{k}.hist").read_bytes() == gold(f"ref_multi__out_k{k}.hist")


@pytest.mark.gpu
def test_unreadable_input_fails_like_the_reference(inputs):
    r = run(["-k", "12", "-p", "x", "does_not_exist.fq"], cwd=inputs)
    assert r.returncode == 1 and b"Error in reading file: does_not_exist.fq" in r.stderr


REF_NTCARD = os.path.join(ROOT, "oracle", "_ref", "ntcard_ref")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_NTCARD), reason="the real reference binary (oracle/_ref) was not built")
def test_live_differential_against_reference_binary(tmp_path):
    """adversarial inputs through BOTH command lines (the reference binary built from its own sources travels in
    oracle/_ref): multi-line FASTA with long contigs, N runs, lower case, IUPAC codes, empty and short records,
    a gzip'ed FASTQ with ragged read lengths, and a file list with two parser threads"""
    import random
    rng = random.Random(99)

    def seq(n, pn):
        out = []
        i = 0
        while i < n:
            r = rng.random()
            if r < pn:                      # a run of N / IUPAC
                m = rng.choice([1, 1, 2, 5, 40])
                out.append("".join(rng.choice("NnRYKM") for _ in range(m)))
                i += m
            else:
                m = rng.randint(1, 60)
                out.append("".join(rng.choice("ACGTacgt") for _ in range(m)))
                i += m
        return "".join(out)[:n]

    fa = []
    for i, n in enumerate([0, 5, 31, 32, 33, 200, 1000, 30_000, 150_000, 70]):
        s = seq(n, 0.01)
        fa.append(">c%d some description\n" % i + "\n".join(s[j:j + 60] for j in range(0, len(s), 60)) + ("\n" if s else ""))
    (tmp_path / "contigs.fa").write_text("".join(fa))
    recs = []
    for i in range(20_000):
        s = seq(rng.choice([20, 36, 75, 100, 150, 151, 250]), 0.002)
        recs.append("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
    with gzip.open(tmp_path / "ragged.fq.gz", "wt") as f:
        f.write("".join(recs))
    (tmp_path / "both.txt").write_text("contigs.fa\nragged.fq.gz\n")
    cases = [(["-k", "25"], ["contigs.fa"]), (["-k", "12,33,64"], ["ragged.fq.gz"]), (["-k", "31", "-g", "5"], ["contigs.fa", "ragged.fq.gz"]),
             (["-t", "2", "-k", "32", "-c", "200"], ["@both.txt"])]
    env = dict(os.environ, NTC_BIN_MIN="1024", NTC_CLI_BLOCK_BYTES="400000")  # length bins from 1024 reads on take the tiled kernels (the default, 32 Ki, would send this small file to K1), in several batches
    for n, (args, files) in enumerate(cases):
        ours = subprocess.run([BIN] + args + ["-p", "gpu%d" % n] + files, cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        ref = subprocess.run([REF_NTCARD] + args + ["-p", "ref%d" % n] + files, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert ours.returncode == 0 and ref.returncode == 0, (ours.stderr, ref.stderr)
        ks = args[args.index("-k") + 1].split(",")
        for k in ks:
            a = (tmp_path / ("gpu%d_k%s.hist" % (n, k))).read_bytes()
            b = (tmp_path / ("ref%d_k%s.hist" % (n, k))).read_bytes()
            assert a == b, (args, files, k)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_NTCARD), reason="the real reference binary (oracle/_ref) was not built")
@pytest.mark.parametrize("threads", [None, "5"])
@pytest.mark.parametrize("block", [None, "64", "4096"])
def test_block_splitters_against_reference_binary(tmp_path, block, threads):
    """the block-based FASTA and SAM splitters (cli_common.hpp) against the reference's line-based ones (ntcard.cpp:191-235), quirks included:
    one-line and wrapped FASTA records mixed, blank lines, CR line ends, no final newline; SAM lines with fewer than ten fields and blank lines
    (the reference counts the PREVIOUS sequence again), spaces as separators, header-only files with and without a final newline.  With
    NTC_CLI_BLOCK_BYTES at 64 and 4096 every record straddles a block boundary and the grow path (a record longer than the block) runs.
    threads = 5 (round 6): a file's blocks are read and scanned for line ends by helper threads (`-t` threads beyond the number of files), FASTQ included."""
    import random
    rng = random.Random(7)

    def seq(n):
        return "".join(rng.choice("ACGTACGTACGTacgtN") for _ in range(n))

    fa = []
    for i in range(3000):
        s = seq(rng.choice([0, 10, 40, 100, 150, 151]))
        kind = rng.random()
        if kind < 0.8:
            fa.append(">r%d\n%s\n" % (i, s))
        elif kind < 0.9:
            fa.append(">r%d\n%s\n\n%s\n" % (i, s[:50], s[50:]))
        elif kind < 0.95:
            fa.append(">r%d\r\n%s\r\n" % (i, s))
        else:
            fa.append(">r%d\n" % i)
    fa.append(">last\n" + seq(100))  # no final newline
    (tmp_path / "reads1.fa").write_text("".join(fa), newline="")
    (tmp_path / "long1.fa").write_text(">c\n" + seq(20000) + "\n>d\n" + seq(300) + "\n", newline="")

    def samrec(i, s, sep="\t"):
        return sep.join(["q%d" % i, "0", "chr1", str(1 + i), "60", "%dM" % max(1, len(s)), "*", "0", "0", s, "I" * len(s), "NM:i:0"]) + "\n"

    sam = ["@HD\tVN:1.6\tSO:unsorted\n", "@SQ\tSN:chr1\tLN:100000\n", "@CO\ta b c d e f g h ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT k\n"]
    for i in range(3000):
        r = rng.random()
        if r < 0.9:
            sam.append(samrec(i, seq(rng.choice([36, 100, 150])), "\t" if rng.random() < 0.9 else " "))
        elif r < 0.95:
            sam.append("q%d\t4\t*\n" % i)        # short line: the previous sequence counts again
        else:
            sam.append("\n")
    sam.append("\n")
    (tmp_path / "quirky.sam").write_text("".join(sam), newline="")
    (tmp_path / "nohdr2.sam").write_text("".join(s for s in sam[3:] if s != "\n")[:-1], newline="")  # starts with a record, no final newline
    hdr = "@HD\tVN:1.6\n@CO\ta b c d e f g h ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTTTGACCA k"
    (tmp_path / "hdr_nl.sam").write_text(hdr + "\n", newline="")
    (tmp_path / "hdr_nonl.sam").write_text(hdr, newline="")
    env = dict(os.environ)
    if block:
        env["NTC_CLI_BLOCK_BYTES"] = block
    fq = []
    for i in range(4000):
        s2 = seq(rng.choice([0, 20, 100, 150, 151]))
        fq.append("@r%d\n%s%s\n+\n%s\n" % (i, s2, "\r" if rng.random() < 0.05 else "", "I" * len(s2)))
    fq.append("@last\n" + seq(150) + "\n+\n" + "I" * 150)  # the quality line ends the file without a newline: the record counts
    (tmp_path / "quirky.fq").write_text("".join(fq), newline="")
    cases = [(["-k", "21"], ["reads1.fa"]), (["-k", "32"], ["long1.fa"]), (["-k", "25"], ["quirky.sam"]), (["-k", "25", "-g", "3"], ["nohdr2.sam"]),
             (["-k", "12"], ["hdr_nl.sam", "reads1.fa"]), (["-k", "12"], ["hdr_nonl.sam", "reads1.fa"]), (["-k", "17"], ["hdr_nonl.sam"]),
             (["-k", "20"], ["quirky.fq"])]
    for n, (args, files) in enumerate(cases):
        ours = subprocess.run([BIN] + args + (["-t", threads] if threads else []) + ["-p", "gpu%d" % n] + files, cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        ref = subprocess.run([REF_NTCARD] + args + ["-p", "ref%d" % n] + files, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert ours.returncode == ref.returncode, (args, files, ours.stderr, ref.stderr)
        if ref.returncode != 0:
            continue
        k = args[1]
        a = (tmp_path / ("gpu%d_k%s.hist" % (n, k))).read_bytes()
        b = (tmp_path / ("ref%d_k%s.hist" % (n, k))).read_bytes()
        assert a == b, (args, files)
