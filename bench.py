#!/usr/bin/env python3
"""bench.py — k-mers/s hashed+sketched at k=32 on 150 bp reads (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--reads-per-step R] [--dist g|u]

A "step" is one pass of the hot path (ntc_submit_device / ntc_submit_tiled_device: ntHash -> sample -> count into
the device-resident t_Counter sketch) over one batch of synthetic reads that is already resident in HBM.
Default workload = BASELINE.json configs[1]: 100 M synthetic 150 bp reads, k=32, rBits=27, sBits=7,
as 10 steps x 10 M reads.  For N > 1 every rank (one per GPU; `--gpus N` re-launches itself under
torch.distributed.run when it is not already running under it) processes its own read-index range of the
same size (weak scaling), then the per-GPU sketches are merged inside the timed region: RCCL all-to-all of
the 16-bit counter slices, wrapping local sums, per-rank value histograms of the summed slices, histograms to rank 0.

One JSON line on rank 0: value = total k-mers (sum of F1 over ranks) / max-over-ranks wall time.
Extra objects: "roofline" (ALL kernels of a step — hash kernels and the deferred sketch update —, HIP-event timed,
algorithmic bytes; "roofline_hash" prices the hash kernels alone against the bytes they move) and "cpu_baseline" (the
reference's own ntRead compiled from /root/reference by oracle/Makefile, on a bounded sample, rank 0 / N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def klist_of(args):
    return [int(x) for x in args.klist.split(",")] if args.klist else [args.k]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads-per-step", type=int, default=10_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--klist", type=str, default="", help="comma separated k list (config 4: 32,64,96,128); overrides --k")
    ap.add_argument("--gap", type=int, default=0, help="gapped seed (config 5), single k only")
    ap.add_argument("--r-bits", type=int, default=27)
    ap.add_argument("--s-bits", type=int, default=7)
    ap.add_argument("--dist", choices=["g", "u"], default="g",
                    help="g: reads from a 100 Mbp random genome, 1%% subs, 0.05%% N; u: i.i.d. uniform ACGT")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="BASELINE.json config: 2 (default) 100 M reads k=32 sBits=7; 3: 1 B reads in total, sBits=11, read-index ranges split "
                         "over the ranks (strong scaling); 4: k=32,64,96,128 in one run; 5: spaced seed k=12 g=2")
    ap.add_argument("--repeats", type=int, default=9, help="the timed region (K steps + flush + merge) is run this many times in-process; ms_per_step / value "
                                                            "are the MEDIAN repeat's, ms_per_step_min / _max give the spread (the clocks differ from lease to lease and ramp)")
    ap.add_argument("--no-nodefer", action="store_true", help="skip the extra run without NTC_FLAG_DEFER_REDO (\"roofline_nodefer\": the default buffer contract of ntc_submit*_device)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not re-run this command under rocprofv3 --pmc for roofline.traffic / valu "
                                                               "(the committed table profiles/traffic_pmc.json is looked up instead)")
    ap.add_argument("--direct-atomics", action="store_true", help="A/B: one device atomic per sampled k-mer instead of the hit log")
    ap.add_argument("--always-log", action="store_true", help="A/B: never switch from the hit log to direct atomics")
    ap.add_argument("--lane-kernel", action="store_true", help="A/B: the lane-per-read kernel K1 takes every batch (tiled ones re-laid out as row slots)")
    ap.add_argument("--k1h-timers", action="store_true", help="print the section clocks a K1H_EXP=timers build of K1h left behind F1 (stderr)")
    ap.add_argument("--k1h-wave-clocks", action="store_true", help="print the spread of the per-wave first / last clocks a -DK1H_WAVE_CLOCKS build of K1h left behind (stderr; tools/k1h_variant.sh)")
    ap.add_argument("--layout", choices=["auto", "rows", "tiled"], default="auto",
                    help="slot layout of the resident batches: rows = one slot per read (ntc_submit_device: K1), tiled = the tiled layout "
                         "(ntc_submit_tiled_device: K1h + K1f); auto = tiled where K1h is built "
                         "for the configuration (every k of the list within 12..32, no gap, sBits >= 7), rows otherwise (a tiled batch would only be re-laid out)")
    ap.add_argument("--log-entries", type=int, default=0, help="capacity of the hit log in entries (0 = the engine's default: one per counter)")
    ap.add_argument("--merge", choices=["auto", "slices", "owner"], default="auto",
                    help="multi-GPU merge: slices = all-to-all of 16-bit counter slices (parallel.merge_to_value_histograms); owner = ship the sampled k-mers "
                         "to the rank that owns their counter range (parallel.merge_owner_engine; falls back to slices when a sketch update ran during the "
                         "steps); auto = owner for sBits >= 11 (fewer hits than counter bytes: BASELINE config 3), slices otherwise")
    ap.add_argument("--lib", type=str, default="", help="A/B: load this build of libntcard_hip.so instead of the in-tree one")
    ap.add_argument("--cpu-sample-reads", type=int, default=0, help="0 = auto (about 10-20 s of CPU work)")
    return ap.parse_args()


def host_cpu():
    """model string, sockets and PHYSICAL cores of the host from /proc/cpuinfo (`cores` in cpu_baseline is the number of threads used = logical CPUs)"""
    model, phys = None, set()
    cur = {}
    try:
        with open("/proc/cpuinfo") as f:
            for line in f.read().splitlines() + [""]:
                if not line.strip():
                    if cur:
                        phys.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor", "0"))))
                        model = model or cur.get("model name")
                    cur = {}
                    continue
                k, _, v = line.partition(":")
                cur[k.strip()] = v.strip()
    except OSError:
        pass
    return {"model": model, "physical_cores": len(phys) or None, "sockets": len({p for p, _ in phys}) or None}


def cpu_baseline(args, nt_stride):
    """The reference's own ntRead/stRead on this box's host cores (kind "reference": oracle/_ref/ref_tool, the real
    ntcard.cpp compiled where it lies, built in the build container and shipped with the snapshot), one thread per
    shard of the batch like the reference's one thread per file, on a bounded sample of the same read generator.
    If the reference build is absent the oracle's own port is timed instead (kind "port"; measured 1.40 x the
    reference's speed per thread in the build container).  Reported, never the target."""
    import subprocess
    cores = os.cpu_count() or 1
    host = host_cpu()
    n = args.cpu_sample_reads or min(4_000_000, 250_000 * cores)
    dist = 1 if args.dist == "g" else 0
    kl = klist_of(args)
    tool = os.path.join(ROOT, "oracle", "_ref", "ref_tool")
    if os.access(tool, os.X_OK):
        cmd = [tool, "bench", str(args.seed), "0", str(n), str(args.read_len), str(dist), ",".join(map(str, kl)), str(args.gap),
               str(args.r_bits), str(args.s_bits), str(cores), "10"]
        j = json.loads(subprocess.run(cmd, stdout=subprocess.PIPE, check=True, timeout=600).stdout.decode().strip().splitlines()[-1])
        return {"value": j["kmers"] / j["seconds"], "unit": "k-mers/s", "cores": cores, "cpu_model": host["model"], "physical_cores": host["physical_cores"],
                "sockets": host["sockets"], "kind": "reference",
                "sample": f"{j['chunks']} x {n} reads x {args.read_len} bp (same generator, dist={args.dist}), k={kl}, gap={args.gap}, "
                          f"the reference's ntRead/stRead (ntcard.cpp compiled from /root/reference by oracle/Makefile), one thread per "
                          f"shard, shared t_Counter with omp atomic, {j['seconds']:.2f} s timed"}
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import orc
    counters = np.zeros((len(kl), 2, 1 << args.r_bits), dtype=np.uint16)
    offs = (np.arange(n + 1, dtype=np.uint64) * np.uint64(args.read_len))
    total_f1, dt, chunks = 0, 0.0, 0
    while dt < 10.0 and chunks < 12:
        slots = orc.gen_reads(args.seed, chunks * n, n, args.read_len, nt_stride, dist, genome_len=100_000_000)
        bases = np.ascontiguousarray(slots.reshape(n, nt_stride)[:, : args.read_len]).reshape(-1)

        def run(b, o):
            if args.gap:
                return orc.sketch_update(counters, b, o, kl, args.gap, args.r_bits, args.s_bits, threads=cores)
            return orc.sketch_update_sharded(counters, b, o, kl, args.r_bits, args.s_bits, threads=cores)
        if chunks == 0:  # thread pool + page-fault warm-up on 2 % of the first chunk, then start from zero
            run(bases, offs[: max(1, n // 50) + 1])
            counters[:] = 0
        t0 = time.perf_counter()
        f1 = run(bases, offs)
        dt += time.perf_counter() - t0
        total_f1 += int(sum(int(x) for x in f1))
        chunks += 1
    return {"value": float(total_f1) / dt, "unit": "k-mers/s", "cores": cores, "cpu_model": host["model"], "physical_cores": host["physical_cores"],
            "sockets": host["sockets"], "kind": "port",
            "sample": f"{chunks} x {n} reads x {args.read_len} bp (same generator, dist={args.dist}), k={kl}, gap={args.gap}, "
                      f"oracle port of ntRead+ntComp (per-k tables, one thread per shard; 1.40 x the reference's per-thread speed in "
                      f"the build container), {dt:.2f} s timed"}


def under_profiler():
    """true when this process is itself being profiled (rocprofv3 preloads its tool library): no nested profiler runs"""
    return any("rocprof" in k.lower() or "rocprof" in v.lower() for k, v in os.environ.items() if k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES")
               or k.startswith("ROCPROF"))


def live_pmc(argv_inner, n_steps):
    """HBM traffic and VALU instruction counts of a step's kernels, measured NOW: this same command is re-run under
    `rocprofv3 --pmc <counter>` once per counter (separate passes, as MI355X_MICROARCH.md prescribes for FETCH_SIZE /
    WRITE_SIZE; no tracing domains), the counter is summed over every dispatch of the hash kernels (sketch_*_kernel) AND
    of the deferred sketch update (split / count / log_* kernels) and divided by the number of bench steps the command
    ran; "<counter>_hash" is the share of the hash kernels alone.  Returns {"FETCH_SIZE": KB, "WRITE_SIZE": KB,
    "SQ_INSTS_VALU": n, ...} per step, or None when rocprofv3 is missing or a pass fails (the committed table is the
    fallback then)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.access(rp, os.X_OK):
        return None
    res = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
        d = tempfile.mkdtemp(prefix="ntc_pmc_", dir="/tmp")
        try:
            r = subprocess.run([rp, "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__)] + argv_inner,
                               cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=180)
            files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
            if r.returncode != 0 or not files:
                return None
            tot, tot_hash, seen = 0.0, 0.0, 0
            for f in files:
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row["Counter_Name"] != ctr:
                            continue
                        kn = row["Kernel_Name"]
                        is_hash = any(x in kn for x in ("sketch_hf_kernel", "sketch_bs_kernel", "sketch_ts_kernel", "sketch_k1h_kernel", "k1h_fix_kernel", "k1h_slow_kernel", "append_slots"))
                        is_apply = any(x in kn for x in ("split_kernel", "split_packed_kernel", "count_kernel", "log_atomics", "log_total", "log_probe", "log_decide"))
                        if is_hash or is_apply:
                            tot += float(row["Counter_Value"])
                            seen += 1
                            if is_hash:
                                tot_hash += float(row["Counter_Value"])
            if seen == 0:
                return None
            res[ctr] = tot / n_steps
            res[ctr + "_hash"] = tot_hash / n_steps
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return res


def traffic_key(args, reads_per_launch):
    k = ",".join(map(str, klist_of(args)))
    return (f"dist={args.dist},L={args.read_len},k={k},gap={args.gap},r={args.r_bits},s={args.s_bits},reads={reads_per_launch}"
            + (",tiled" if args.layout == "tiled" else "") + (",lane-kernel" if args.lane_kernel else "") + (",direct-atomics" if args.direct_atomics else "")
            + (",always-log" if args.always_log else ""))


def pmc_valu(args, reads_per_launch):
    """VALU wave-instructions per launch of the hash kernel(s) from the same committed PMC passes (SQ_INSTS_VALU), or None"""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_pmc.json")) as f:
            return json.load(f)[traffic_key(args, reads_per_launch)].get("valu_insts_per_launch")
    except (OSError, KeyError, ValueError):
        return None


def pmc_traffic(args, reads_per_launch):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic_pmc.json); counters cannot
    be read from inside the process, so this is the measured value for the matching workload or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_pmc.json")) as f:
            table = json.load(f)
        return table[traffic_key(args, reads_per_launch)]["traffic_bytes"]
    except (OSError, KeyError, ValueError):
        return None


def read_sclk_mhz(device_index=0):
    """current shader clock of the device as rocm-smi reports it (MHz), or None; a reading taken right after a timed region shows what the
    run was clocked at a moment earlier — the lease-to-lease spread of the headline is mostly this number"""
    import re
    import shutil
    import subprocess
    smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.access(smi, os.X_OK):
        return None
    try:
        r = subprocess.run([smi, "-d", str(device_index), "--showclocks"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=20)
        m = re.search(r"sclk clock level:?\s*\d*:?\s*\(?(\d+)\s*Mhz", r.stdout.decode(errors="replace"), re.I)
        return int(m.group(1)) if m else None
    except Exception:
        return None


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` with N > 1 outside torch.distributed.run: become `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N ... bench.py <same flags>` (one rank per GPU over RCCL).  Fails loudly when the node has fewer GPUs."""
    import socket
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but this node has {have} HIP device(s)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execvpe(sys.executable, cmd, env)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        relaunch_under_torchrun(args)  # does not return
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" in os.environ and args.gpus != world_env and os.environ.get("RANK", "0") == "0":
        print(f"bench.py: --gpus {args.gpus} ignored, running with WORLD_SIZE={world_env} ranks", file=sys.stderr)
    strong_total = None
    if args.config == 3:  # 1 B reads in total over the ranks, sBits = 11 (the >= 50 GB branch of ntcard.cpp:427-431)
        args.s_bits = 11
        strong_total = 1_000_000_000
        args.steps = max(1, strong_total // (world_env * args.reads_per_step))
    elif args.config == 4:
        args.klist = "32,64,96,128"
    elif args.config == 5:
        args.k, args.gap = 12, 2
    if args.layout == "auto":
        kl = klist_of(args)
        k1h_gap = len(kl) == 1 and (kl[0], args.gap) in ((12, 2), (32, 8))  # K1h's spaced-seed variants (config 5 in both forms of SURVEY 8(d))
        # (a list of which only a part is K1h's — config 4: 32,64,96,128 — is tiled too: K1h + K1f take their k from the tiles, K1 stages the same tiles for the rest)
        args.layout = "tiled" if (any(12 <= k <= 32 for k in kl) and (args.gap == 0 or k1h_gap) and args.s_bits >= 7 and not args.lane_kernel) else "rows"
    import torch
    import torch.distributed as dist
    import ntcard_amd as nt
    from ntcard_amd import parallel
    if args.lib:
        nt._abi.LIB_PATH = os.path.abspath(args.lib)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)  # any torch.distributed.run launch
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if torch.cuda.device_count() <= local_rank:
            sys.exit(f"bench.py: rank {rank} wants device {local_rank} but this node has {torch.cuda.device_count()} HIP device(s)")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a HIP device; there is no CPU fallback"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    L, k = args.read_len, args.k
    klist = klist_of(args)
    nk = len(klist)
    tiled = args.layout == "tiled"
    stride = (L + 3) & ~3
    if stride == L:
        stride += 4  # keep at least one separator byte between slots
    R = args.reads_per_step
    K, W = args.steps, args.warmup
    dist_id = 1 if args.dist == "g" else 0
    stream = torch.cuda.current_stream().cuda_stream
    batch_bytes = nt.tiled_bytes(R, L) if tiled else R * stride + 16

    def gen(buf, seed, first):
        if tiled:
            nt.gen_reads_tiled_device(buf.data_ptr(), seed, first, R, L, dist_id, 100_000_000, device=local_rank, stream=stream)
        else:
            nt.gen_reads_device(buf.data_ptr(), seed, first, R, L, stride, dist_id, 100_000_000, device=local_rank, stream=stream)

    # ---- resident inputs: K step batches (+1 warmup batch), generated on the device (K0) ----
    reads_per_rank = R * K
    first, _ = parallel.read_range(rank, world, reads_per_rank)
    if strong_total is not None:  # config 3: contiguous read-index ranges of ONE 1 B-read stream (parallel.split_reads)
        first, reads_per_rank = parallel.split_reads(K * R * world, world)[rank]
    # at most `nb` distinct batches stay resident (all K when they fit in half of the free HBM); a larger K cycles over them
    free_b, _ = torch.cuda.mem_get_info(dev)
    nb = max(1, min(K, int(free_b * 0.5) // batch_bytes))
    if K > 24:
        nb = min(nb, 8)  # a long run (config 3: 100 steps per GPU) cycles over eight resident batches — one deferred launch's worth — instead of keeping 90 x 1.6 GB
    batches = []
    for s in range(nb):
        b = torch.empty(batch_bytes, dtype=torch.uint8, device=dev)
        gen(b, args.seed, first + s * R)
        batches.append(b)
    wb = torch.empty(batch_bytes, dtype=torch.uint8, device=dev)
    gen(wb, args.seed ^ 0x5eed, 0)

    # the sketch is a torch tensor so that torch.distributed (RCCL) can reduce it in place
    sketch = torch.zeros(nk * (2 << args.r_bits), dtype=torch.int32, device=dev)
    f1_big = torch.zeros(nk + 8, dtype=torch.int64, device=dev)  # (a K1h timing build adds its section clocks behind F1: tools/k1h_variant.sh)
    f1_dev = f1_big[:nk]
    # the resident batches stay untouched until the end of the run: the engine may share K1f, the second pass behind K1h, between
    # batches (NTC_FLAG_DEFER_REDO)
    base_flags = ((nt.FLAG_DIRECT_ATOMICS if args.direct_atomics else 0)
                  | (nt.FLAG_LANE_KERNEL if args.lane_kernel else 0) | (nt.FLAG_ALWAYS_LOG if args.always_log else 0)
                  | (nt.FLAG_REQUIRE_TILED if tiled and nk == 1 and klist[0] == 32 and not args.gap and args.s_bits >= 7 and not args.lane_kernel else 0))

    # a run whose sampled k-mers all fit a small log (sBits >= 11: 14.5 M per 125 M reads) does not need the default 4 GiB log + 8.9 GiB of partition scratch:
    # 2^27 entries hold a rank's share of config 3 nine times over (512 MiB + 1.2 GiB); with the eight resident batches, the sketch and the hand-over arrays a
    # rank of `--gpus 8 --config 3` stays under 20 GiB
    if args.log_entries == 0 and args.s_bits >= 11 and K > 24:
        args.log_entries = 1 << (27 if world > 1 else 28)  # (one GPU alone logs all 114 M keys of the 1 B reads)

    def make_engine(flags):
        return nt.Engine(klist, gap=args.gap, r_bits=args.r_bits, s_bits=args.s_bits, device=local_rank, stream=stream, ext_sketch=sketch, ext_f1=f1_dev,
                         log_entries=args.log_entries, flags=flags)
    eng = make_engine(base_flags | nt.FLAG_DEFER_REDO)  # (tiled batches too: the fix-up kernels K1f then take up to 8 batches per launch)

    def submit_to(e, buf):
        if tiled:
            e.submit_tiled_device(buf.data_ptr(), R, L)
        else:
            e.submit_device(buf.data_ptr(), R, L, stride)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def warm(e):
        for _ in range(W):
            submit_to(e, wb)
        if W > 0:
            e.flush()  # warm the deferred sketch update too (allocates its partition scratch)
        if use_dist and W > 0:  # warm the RCCL path too (same collectives as the timed merge)
            parallel.merge_to_value_histograms(sketch, f1_dev, nk, args.r_bits, dst=0)
        e.sync()

    def timed_region(e):
        """EXACTLY K steps + the flush (+ the multi-GPU merge), bracketed by barrier + synchronize on both sides; max over ranks"""
        e.reset()
        e.set_profiling(True)
        barrier()
        t0 = time.perf_counter()
        for s in range(K):
            submit_to(e, batches[s % nb])
        merged, merge_t = None, {}
        if use_dist and merge_mode == "owner":
            # the merge that ships hits (round 6): the pending log split by counter-range owner, one variable-size all-to-all, every rank counts the keys of
            # ITS range with the engine's own sketch update, histograms to rank 0.  None: some rank's sketch already holds counts -> counters are merged
            res = parallel.merge_owner_engine(e, sketch, f1_dev, nk, args.r_bits, dst=0, timings=merge_t)
            if res is not None:
                merged = res[0]
        e.flush()  # the sketch is final on the device inside the timed region: pending hit log -> t_Counter (owner mode: done above, nothing pending)
        if use_dist and merged is None:
            # the path's one exchange step: all-to-all of the 16-bit counter slices, wrapping local sums, per-rank value
            # histograms of the summed slices, histograms to rank 0 (what compEst consumes) — parallel.merge_to_value_histograms; narrowing,
            # sums and histograms are the library's kernels (ntc_narrow_u16_device / ntc_sum_slices_u16_device / ntc_value_hist_u16_device)
            merged, _ = parallel.merge_to_value_histograms(sketch, f1_dev, nk, args.r_bits, dst=0, timings=merge_t)
            merge_t["mode"] = "slices"
        barrier()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        ker, launches = e.kernel_time()
        app, applies = e.apply_time()
        return {"dt": float(tmax.item()), "ker_ms": ker, "launches": launches, "apply_ms": app, "applies": applies, "fix_ms": e.fixup_time(),
                "merged": merged, "merge_t": merge_t}

    merge_mode = args.merge if args.merge != "auto" else ("owner" if args.s_bits >= 11 else "slices")
    sclk_before = read_sclk_mhz(local_rank) if rank == 0 else None
    warm(eng)
    n_rep = max(1, args.repeats)
    runs = [timed_region(eng) for _ in range(n_rep)]
    sclk_after = read_sclk_mhz(local_rank) if rank == 0 else None
    order = sorted(range(n_rep), key=lambda i: runs[i]["dt"])
    med = runs[order[(n_rep - 1) // 2]]  # the median repeat (the lower one of an even count): every per-step figure below is THIS repeat's
    dt_max = med["dt"]
    ker_ms, launches, apply_ms, applies, fix_ms, merge_t = med["ker_ms"], med["launches"], med["apply_ms"], med["applies"], med["fix_ms"], med["merge_t"]
    ph_merged = runs[-1]["merged"]  # (the counters on the device are the last repeat's; every repeat counts the same reads)
    update_mode = eng.update_mode()
    if ph_merged is not None:
        eng.sync()
        ph, f1 = ph_merged.cpu().numpy().astype("uint32"), f1_dev.cpu().numpy().astype("uint64")
    else:
        _, ph, f1 = eng.finish(counters=False, p_hist=True)
    total_kmers = int(sum(int(x) for x in f1))  # after the reduce rank 0 holds the sum over ranks (and over the k list)
    if args.k1h_timers:
        tm = [int(x) for x in f1_big.cpu().numpy()[nk:nk + 4]]
        tot = max(sum(tm), 1)
        print("k1h timers (clk summed over waves and launches): walk+test+push %d (%.1f%%), pack %d (%.1f%%), passes %d (%.1f%%), block end %d (%.1f%%)"
              % (tm[0], 100 * tm[0] / tot, tm[1], 100 * tm[1] / tot, tm[2], 100 * tm[2] / tot, tm[3], 100 * tm[3] / tot), file=sys.stderr)

    if os.environ.get("NTC_SPLIT_CLOCKS_OUT"):
        import ctypes
        import numpy as np
        sc = np.zeros(3 * 1024, dtype=np.uint64)
        fn = nt._abi.lib().ntc_dbg_split_clocks
        fn.argtypes = [ctypes.c_void_p]
        assert fn(sc.ctypes.data) == 0
        np.save(os.environ["NTC_SPLIT_CLOCKS_OUT"], sc)
    if os.environ.get("NTC_HF_CLOCKS_OUT"):
        import ctypes
        import numpy as np
        hc = np.zeros(2 * 8192, dtype=np.uint64)
        fn = nt._abi.lib().ntc_dbg_hf_clocks
        fn.argtypes = [ctypes.c_void_p]
        assert fn(hc.ctypes.data) == 0
        np.save(os.environ["NTC_HF_CLOCKS_OUT"], hc)
    if args.k1h_wave_clocks:
        import ctypes
        import numpy as np
        wc = np.zeros(2 * 4096, dtype=np.uint64)
        fn = getattr(nt._abi.lib(), "ntc_dbg_k1h_wave_clocks_p%d" % (klist[0] % 4))
        fn.argtypes = [ctypes.c_void_p]
        assert fn(wc.ctypes.data) == 0
        if os.environ.get("K1H_WAVE_CLOCKS_OUT"):
            np.save(os.environ["K1H_WAVE_CLOCKS_OUT"], wc)
        t0s, t1s = wc[0::2].astype(np.int64), wc[1::2].astype(np.int64)
        live = t1s > 0
        t0s, t1s = t0s[live], t1s[live]
        base, end = t0s.min(), t1s.max()
        life = (t1s - t0s) / 100.0  # us (100 MHz)
        fin = (t1s - base) / 100.0
        q = lambda a, f: float(np.quantile(a, f))
        print("k1h wave clocks of the last launch: %d waves, launch %.1f us from first start to last end; start spread %.1f us; wave life mean %.1f min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f us; "
              "finish time p1 %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f us; mean life / launch = %.3f"
              % (len(life), (end - base) / 100.0, (t0s.max() - base) / 100.0, life.mean(), life.min(), q(life, .1), q(life, .5), q(life, .9), life.max(),
                 q(fin, .01), q(fin, .1), q(fin, .5), q(fin, .9), q(fin, .99), fin.max(), life.mean() / ((end - base) / 100.0)), file=sys.stderr)
        # by workgroup position: XCD = workgroup % 8 (round-robin dispatch), SIMD pair = wave % 4
        wg = np.nonzero(live)[0] // 8
        for x in range(8):
            m = (wg % 8) == x
            print("  xcd %d: mean finish %.1f us, mean life %.1f us" % (x, fin[m].mean(), life[m].mean()), file=sys.stderr)

    if rank == 0:
        import numpy as np
        hits = int(sum((ph[ki].astype(np.uint64) * np.arange(65536, dtype=np.uint64)).sum() for ki in range(nk)))
        # --- roofline, per step, this rank.  A step's kernels = the hash kernels (K1h + K1f, or K1) AND its share of the
        # deferred sketch update (partition + count passes), both HIP-event timed on the engine's stream; the algorithmic bytes
        #  are the bases + 4 B per read, read once, and 2 B read + 2 B written per sampled increment — the increments
        # are carried out by the update kernels, so numerator and denominator cover the same work.  "roofline_hash" prices the hash
        # kernels alone against the read stream alone.
        per_step_kmers = total_kmers / max(world, 1) / max(K, 1)
        per_step_hits = hits / max(world, 1) / max(K, 1)
        read_bytes = R * (L + 4)
        alg_bytes = read_bytes + 4.0 * per_step_hits
        hash_ms = ker_ms / max(K, 1)
        step_ms = (ker_ms + apply_ms + fix_ms) / max(K, 1)  # (fix_ms: the deferred K1f launches, one per up to 8 batches, on the engine's stream like everything else)
        achieved = alg_bytes / (step_ms * 1e-3) / 1e9 if step_ms > 0 else 0.0
        achieved_hash = read_bytes / (hash_ms * 1e-3) / 1e9 if hash_ms > 0 else 0.0
        if (tiled and ((all(12 <= k <= 32 for k in klist) and not args.gap) or (nk == 1 and (klist[0], args.gap) in ((12, 2), (32, 8)))) and args.s_bits >= 7
                and args.r_bits + 1 + args.s_bits - 7 <= 32 and not args.lane_kernel):
            kern = ("sketch_k1h_kernel (K1h: one wave per tile, eight waves per CU; with NTC_FLAG_DEFER_REDO up to 8 resident batches = bench steps share ONE launch, "
                    "hash_ms is the launches' HIP-event time divided by the steps) + k1h_fix_kernel / k1h_slow_kernel (K1f: one launch per up to 8 batches)")
        else:
            kern = "sketch_hf_kernel (K1: lane per read)"
        peak_valu = 256 * 4 * 2.4e9 / 2  # MI355X_MICROARCH.md: a wave64 VALU instruction occupies a SIMD-32 for 2 clk
        out = {
            "metric": "k-mers/s hashed+sketched (whole node) at k=32, 150 bp reads",
            "value": total_kmers / dt_max,
            "unit": "k-mers/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": dt_max * 1e3 / K,
            # the timed region was run `repeats` times in this process: every figure of this line is the MEDIAN repeat's; min / max / all give the spread
            "repeats": n_rep,
            "ms_per_step_min": runs[order[0]]["dt"] * 1e3 / K,
            "ms_per_step_max": runs[order[-1]]["dt"] * 1e3 / K,
            "ms_per_step_all": [r["dt"] * 1e3 / K for r in runs],
            "kernel_ms_per_step_all": [{"hash": r["ker_ms"] / K, "fixup": r["fix_ms"] / K, "apply": r["apply_ms"] / K} for r in runs],
            "sclk_mhz": {"before": sclk_before, "after": sclk_after, "source": "rocm-smi --showclocks (current sclk level), read before the warm-up and right behind the last repeat"},
            "higher_is_better": True,
            "scaling": "strong" if strong_total is not None else "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"{world}x{reads_per_rank} synthetic {L} bp reads (dist={args.dist}, seed={args.seed}), "
                                   f"k={','.join(map(str, klist))}{', gap=%d' % args.gap if args.gap else ''}, rBits={args.r_bits}, sBits={args.s_bits}, "
                                   f"{K} steps x {R} reads per GPU, {'tiled' if tiled else 'row-major'} slots"
                                   + (f" ({nb} distinct resident batches, cycled)" if nb < K else "")
                                   + ((", RCCL all-to-all of the sampled k-mers to their counter-range owners + value histograms to rank 0 inside the timed region" if merge_mode == "owner" else
                                       ", RCCL all-to-all of 16-bit counter slices + value histograms to rank 0 inside the timed region") if world > 1 else ""),
                       "traffic_key": traffic_key(args, R), "k": klist[0] if nk == 1 else klist, "gap": args.gap, "read_len": L, "reads_per_gpu": reads_per_rank,
                       "r_bits": args.r_bits, "s_bits": args.s_bits, "layout": args.layout, "parallelism": f"read-sharded x{world}",
                       "rccl_ranks": dist.get_world_size() if use_dist else 0},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args, R),
                         "kernel": kern + " + split_kernel / split_packed_kernel / count_kernel (deferred sketch update)",
                         "avg_launch_ms": step_ms, "hash_ms": hash_ms, "apply_ms": apply_ms / max(K, 1), "fixup_ms": fix_ms / max(K, 1), "launches": launches,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "kmers_per_launch": per_step_kmers},
            # the hash kernels alone against the read stream alone (what the sketch update costs is in "roofline" above)
            "roofline_hash": {"achieved": achieved_hash, "frac": achieved_hash / HBM_PEAK_GBS, "unit": "GB/s", "avg_launch_ms": hash_ms,
                              "algorithmic_bytes_per_launch": read_bytes, "traffic": None},
            # whole job: algorithmic bytes of a step / wall time of a step (includes launch gaps and the final flush)
            "roofline_step": {"achieved": alg_bytes / (dt_max / K) / 1e9, "frac": alg_bytes / (dt_max / K) / 1e9 / HBM_PEAK_GBS, "unit": "GB/s"},
            # deferred sketch update (ntc_apply.hip), HIP-event timed like the hash kernels; inside the timed region
            "sketch_apply": {"mode_at_end": "direct atomics" if update_mode else "hit log", "applies": applies, "total_ms": apply_ms, "ns_per_increment": apply_ms * 1e6 / max(hits, 1)},
            # the path is VALU-issue bound, so it is also priced against the vector ALUs: wave-instructions of ALL of a
            # step's kernels (SQ_INSTS_VALU), per 64-lane base step; peak = 256 CUs x 4 SIMDs x 2.4 GHz / 2 clk per wave64 instruction
            "valu": (lambda v: {"wave_insts_per_launch": v, "per_wave_step": (v / (R * L / 64.0)) if v else None,
                                "wave_insts_per_s": (v / (step_ms * 1e-3)) if v and step_ms > 0 else None,
                                "peak_wave_insts_per_s": peak_valu,
                                "frac": (v / (step_ms * 1e-3) / peak_valu) if v and step_ms > 0 else None})(pmc_valu(args, R)),
            # N > 1: this rank's share of the one exchange step (inside the timed region): narrow to 16 bits / all-to-all of the slices / wrapping
            # sums / value histograms of the summed slice / histograms + F1 to rank 0
            "merge": (merge_t if use_dist else None),
            "device_memory_used_gib": round((torch.cuda.mem_get_info(dev)[1] - torch.cuda.mem_get_info(dev)[0]) / 2**30, 2),  # this rank, behind the last repeat (everything resident)
            "f1_total": total_kmers,
            "sampled_increments": hits,
        }
        # roofline.traffic / valu: measured live (this command again under rocprofv3 --pmc, one counter per pass) at N = 1;
        # the committed table of the same passes (profiles/traffic_pmc.json, tools/prof.sh) is the fallback
        out["roofline"]["traffic_source"] = "profiles/traffic_pmc.json" if out["roofline"]["traffic"] is not None else None
        # the same workload with the DEFAULT buffer contract of ntc_submit*_device (no NTC_FLAG_DEFER_REDO: the fix-up kernels follow every
        # launch in stream order, the caller may refill a buffer as soon as its own stream-ordered work allows)
        if world == 1 and not use_dist and not args.no_nodefer:
            eng.close()
            e2 = make_engine(base_flags)
            warm(e2)
            r2 = [timed_region(e2) for _ in range(min(3, n_rep))]
            m2 = sorted(r2, key=lambda r: r["dt"])[(len(r2) - 1) // 2]
            step2 = (m2["ker_ms"] + m2["apply_ms"] + m2["fix_ms"]) / max(K, 1)
            ach2 = alg_bytes / (step2 * 1e-3) / 1e9 if step2 > 0 else 0.0
            out["roofline_nodefer"] = {"achieved": ach2, "frac": ach2 / HBM_PEAK_GBS, "unit": "GB/s", "ms_per_step": m2["dt"] * 1e3 / K, "value": total_kmers / m2["dt"],
                                       "avg_launch_ms": step2, "hash_ms": m2["ker_ms"] / max(K, 1), "apply_ms": m2["apply_ms"] / max(K, 1),
                                       "fixup_ms": m2["fix_ms"] / max(K, 1), "repeats": len(r2),
                                       "note": "engine without NTC_FLAG_DEFER_REDO: one K1f launch behind every K1h launch, in stream order (fixup_ms), instead of one per 8 batches"}
            e2.close()
        if world == 1 and not use_dist and not args.no_live_pmc and not under_profiler():
            eng.close()
            del batches, wb
            torch.cuda.empty_cache()
            inner = [a for a in sys.argv[1:] if a not in ("--no-cpu-baseline", "--no-live-pmc", "--no-nodefer")] + ["--no-cpu-baseline", "--no-live-pmc", "--no-nodefer", "--repeats", "1"]
            live = live_pmc(inner, K + W)
            if live is not None:
                # MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE are in KB; wide streaming reads are
                # under-counted by half on gfx950 -> traffic = 2 * FETCH_SIZE + WRITE_SIZE (an upper bound: applied to all of FETCH)
                out["roofline"]["traffic"] = int((2.0 * live["FETCH_SIZE"] + live["WRITE_SIZE"]) * 1024)
                out["roofline"]["traffic_source"] = "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, hash + sketch-update kernels"
                out["roofline_hash"]["traffic"] = int((2.0 * live["FETCH_SIZE_hash"] + live["WRITE_SIZE_hash"]) * 1024)
                v = live["SQ_INSTS_VALU"]
                out["valu"].update({"wave_insts_per_launch": v, "per_wave_step": v / (R * L / 64.0),
                                    "wave_insts_per_s": (v / (step_ms * 1e-3)) if step_ms > 0 else None,
                                    "frac": (v / (step_ms * 1e-3) / peak_valu) if step_ms > 0 else None,
                                    "hash_kernels_only": live["SQ_INSTS_VALU_hash"], "source": "live: rocprofv3 --pmc SQ_INSTS_VALU"})
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args, stride)
            except Exception as ex:  # the checker is optional for the measurement itself
                out["cpu_baseline"] = {"value": None, "unit": "k-mers/s", "cores": os.cpu_count(), "kind": "reference",
                                       "sample": f"failed: {ex}"}
        print(json.dumps(out))
    eng.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
