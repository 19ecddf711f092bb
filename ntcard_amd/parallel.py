"""Multi-GPU plumbing for the one exchange step the path has: reads are sharded by
index range, every rank owns a private full sketch, and the per-rank sketches are merged once at the end
(RCCL over xGMI when the tensors live on GPUs; gloo in the CPU tests): an all-to-all of 16-bit counter slices +
local wrapping sums for the estimator (merge_to_value_histograms), or a plain uint32 SUM reduce (reduce_sketch).

The reference's analogue is the shared t_Counter all OpenMP threads increment atomically
(ntcard.cpp:142-143,445) and the atomic F1 merge (ntcard.cpp:464-466): both are commutative sums,
so summing private copies is exact.  Counters are uint32 on the device, stored in int32 tensors
(two's-complement addition is the same ring); the reference's uint16 wrap is applied afterwards.
"""
import torch
import torch.distributed as dist


def read_range(rank, world, reads_per_rank):
    """contiguous read-index range [first, first+n) owned by `rank` (weak scaling: fixed n per rank)"""
    return rank * reads_per_rank, reads_per_rank


def split_reads(n_reads, world):
    """strong-scaling split of n_reads into `world` contiguous ranges -> list of (first, n)"""
    base, rem = divmod(n_reads, world)
    out, first = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((first, n))
        first += n
    return out


def reduce_sketch(sketch, f1, dst=0):
    """in-place SUM reduce of the sketch (int32 view of uint32 counters) and F1 (int64) to rank dst"""
    if dist.is_available() and dist.is_initialized():
        dist.reduce(sketch, dst=dst, op=dist.ReduceOp.SUM)
        dist.reduce(f1, dst=dst, op=dist.ReduceOp.SUM)
    return sketch, f1


class _Phases:
    """per-phase clock of the merge: HIP events on the current stream for device tensors, perf_counter for CPU tensors (gloo tests)"""

    def __init__(self, device, enabled):
        self.cuda = enabled and device.type == "cuda"
        self.enabled = enabled
        self.marks = []

    def mark(self, name):
        if not self.enabled:
            return
        if self.cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.marks.append((name, ev))
        else:
            import time
            self.marks.append((name, time.perf_counter()))

    def result(self):
        out = {}
        if not self.enabled or len(self.marks) < 2:
            return out
        if self.cuda:
            self.marks[-1][1].synchronize()
        for (_, a), (name, b) in zip(self.marks, self.marks[1:]):
            out[name + "_ms"] = a.elapsed_time(b) if self.cuda else (b - a) * 1e3
        out["total_ms"] = sum(out.values())
        return out


def _dev_args(t):
    """(device index, current stream handle) of a CUDA tensor, for the library's device entry points"""
    return t.device.index or 0, torch.cuda.current_stream(t.device).cuda_stream


def exchange_and_sum_u16(sketch, shard, phases=None):
    """Slice `rank` of the SUM over ranks of the counters modulo 2^16, as an int16 tensor (the uint16 bit patterns).

    t_Counter is uint16 with wrap-around (ntcard.cpp:142-143,439), so only the low 16 bits of every per-rank counter
    matter for the merged sketch: (sum_r c_r) mod 2^16 == (sum_r (c_r mod 2^16)) mod 2^16.  Every rank therefore narrows
    its counters to 16 bits, sends slice j to rank j (ONE all-to-all: 512 MiB·(N-1)/N per rank and k instead of the 1 GiB
    a uint32 reduce-scatter moves, and every one of a rank's point-to-point xGMI links carries exactly one slice at the
    same time, where a ring reduce-scatter makes N-1 dependent hops) and adds up the N slices it received with a
    wrapping 16-bit add.  RCCL has no 16-bit integer SUM; none is needed.

    Device tensors: the narrowing and the sums are the LIBRARY's kernels (ntc_narrow_u16_device / ntc_sum_slices_u16_device, the ones
    ntc_merge_devices runs between its peer copies — one merge implementation, round 5); only the all-to-all is torch.distributed's.
    CPU tensors (the gloo tests of this module's slicing and wrap arithmetic): the same steps as torch expressions, the exchange as
    batched isend/irecv (gloo has no all-to-all for this)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    on_gpu = sketch.is_cuda
    if on_gpu:
        from . import engine as _eng
        dev, st = _dev_args(sketch)
        low = torch.empty(sketch.numel(), dtype=torch.int16, device=sketch.device)
        _eng.narrow_u16_device(sketch.data_ptr(), sketch.numel(), low.data_ptr(), device=dev, stream=st)
    else:
        low = sketch.to(torch.int16)  # int32 -> int16 keeps the low 16 bits (two's complement)
    recv = torch.empty(world * shard, dtype=torch.int16, device=sketch.device)
    if phases:
        phases.mark("narrow")
    if dist.get_backend() == "gloo":
        ops = []
        for peer in range(world):
            if peer == rank:
                recv[rank * shard:(rank + 1) * shard] = low[rank * shard:(rank + 1) * shard]
            else:
                ops.append(dist.P2POp(dist.isend, low[peer * shard:(peer + 1) * shard].contiguous(), peer))
                ops.append(dist.P2POp(dist.irecv, recv[peer * shard:(peer + 1) * shard], peer))
        for req in (dist.batch_isend_irecv(ops) if ops else []):
            req.wait()
    else:
        dist.all_to_all_single(recv.view(torch.uint8), low.view(torch.uint8))  # bytes: RCCL has no 16-bit integer type, and none is needed to move them
    if phases:
        phases.mark("exchange")
    if on_gpu:
        _eng.sum_slices_u16_device(recv.data_ptr(), shard, world, shard, device=dev, stream=st)  # slice 0 += the others (wrapping)
        out = recv[:shard]
    else:
        parts = recv.view(world, shard)
        out = parts[0].clone()
        for r in range(1, world):
            out += parts[r]  # int16 addition wraps: exactly the uint16 counter arithmetic
    if phases:
        phases.mark("sum")
    return out


def merge_to_value_histograms(sketch, f1, n_k, r_bits, value_hist=None, dst=0, timings=None):
    """Multi-GPU merge for the estimator (compEst only needs the value histogram p[2][65536] of the SUMMED counters,
    ntcard.cpp:240-247): every rank ends up with one slice of the summed uint16 counters (exchange_and_sum_u16),
    histograms that slice locally, and only the histograms (256 KiB per plane) and F1 go to rank `dst`.

    sketch: int32 tensor [n_k * 2 * 2^r_bits] (uint32 counters), f1: int64 [n_k]; f1 is reduced in place.
    value_hist(counters_u16_slice, hist_slice): accumulates the histogram of the uint16 counters (an int16 tensor of their bit patterns)
    into an int32[65536] view; None: the library's ntc_value_hist_u16_device for device tensors, torch.bincount for CPU tensors.
    Returns (p_hist int32 [n_k, 2, 65536], f1) — meaningful on rank dst.  timings: a dict that receives narrow_ms / exchange_ms / sum_ms /
    histogram_ms / reduce_ms / total_ms of this rank (bench.py reports them so that a scaling run separates hashing from the merge)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    ph = _Phases(sketch.device, timings is not None)
    ph.mark("start")
    plane = 1 << r_bits
    n = sketch.numel()
    assert n == n_k * 2 * plane and n % world == 0 and (n // world) % 8 == 0
    shard = n // world
    if value_hist is None:
        if sketch.is_cuda:
            from . import engine as _eng
            dev, st = _dev_args(sketch)

            def value_hist(c, h):
                _eng.value_hist_u16_device(c.data_ptr(), c.numel(), h.data_ptr(), device=dev, stream=st)
        else:
            def value_hist(c, h):
                h += torch.bincount(c.to(torch.int32) & 0xFFFF, minlength=65536).to(torch.int32)
    mine = exchange_and_sum_u16(sketch, shard, ph)
    hist = torch.zeros(n_k * 2 * 65536, dtype=torch.int32, device=sketch.device)
    pos, end = rank * shard, (rank + 1) * shard
    while pos < end:  # a slice may cover several (k, sample) planes, or a fraction of one
        pl = pos // plane
        stop = min((pl + 1) * plane, end)
        value_hist(mine[pos - rank * shard:stop - rank * shard], hist[pl * 65536:(pl + 1) * 65536])
        pos = stop
    ph.mark("histogram")
    dist.reduce(hist, dst=dst, op=dist.ReduceOp.SUM)
    dist.reduce(f1, dst=dst, op=dist.ReduceOp.SUM)
    ph.mark("reduce")
    if timings is not None:
        timings.update(ph.result())
    return hist.view(n_k, 2, 65536), f1


def exchange_keys(send, send_counts):
    """One variable-size all-to-all of hit-log keys: `send` (int32, the counter indices of this rank's sampled k-mers grouped by owner: owner p's
    send_counts[p] keys follow owner p - 1's) -> (recv, recv_counts): what every rank holds for THIS rank's counter range, rank by rank.
    RCCL: all_gather of the counts + all_to_all_single with split sizes; gloo (CPU tests): the same as batched isend / irecv."""
    world, rank = dist.get_world_size(), dist.get_rank()
    sc = torch.tensor([int(c) for c in send_counts], dtype=torch.int64, device=send.device)
    table = [torch.empty_like(sc) for _ in range(world)]
    dist.all_gather(table, sc)
    recv_counts = [int(t[rank]) for t in table]  # (one small device-to-host copy per peer: the sizes of the receive buffer)
    recv = torch.empty(sum(recv_counts), dtype=send.dtype, device=send.device)
    soff = [0]
    for c in send_counts:
        soff.append(soff[-1] + int(c))
    roff = [0]
    for c in recv_counts:
        roff.append(roff[-1] + c)
    if dist.get_backend() == "gloo":
        ops = []
        for peer in range(world):
            if peer == rank:
                recv[roff[rank]:roff[rank + 1]] = send[soff[rank]:soff[rank + 1]]
                continue
            if soff[peer + 1] > soff[peer]:
                ops.append(dist.P2POp(dist.isend, send[soff[peer]:soff[peer + 1]].contiguous(), peer))
            if recv_counts[peer]:
                ops.append(dist.P2POp(dist.irecv, recv[roff[peer]:roff[peer + 1]], peer))
        for req in (dist.batch_isend_irecv(ops) if ops else []):
            req.wait()
    else:
        dist.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=[int(c) for c in send_counts])
    return recv, recv_counts


def merge_owner(send, send_counts, count_keys, f1, n_k, r_bits, value_hist=None, dst=0, timings=None, ph=None):
    """The merge that ships HITS instead of counters ("owner" mode, round 6): every rank owns the counter range
    [rank * C / N, (rank + 1) * C / N) of the ONE merged sketch, receives every rank's sampled k-mers of that range (exchange_keys), counts them
    (count_keys(recv) -> int32 tensor of the range's C / N counters; device: the engine's own sketch update over ntc_log_replace_device, CPU: bincount),
    histograms its range and only the 256 KiB histograms and F1 travel to rank dst — as in merge_to_value_histograms, whose counter slices are what this
    mode does not send.  Counting is a commutative sum (ntcard.cpp:142-143), so it is exact whichever rank counts a k-mer.  Pays when a run's hits are
    fewer than its counters' bytes: the reference's sBits = 11 branch (BASELINE config 3: 58 MB of keys per rank against 448 MiB of 16-bit slices)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    ph = ph or _Phases(send.device, timings is not None)
    if not ph.marks:
        ph.mark("start")
    plane = 1 << r_bits
    n = n_k * 2 * plane
    assert n % world == 0
    shard = n // world
    recv, _ = exchange_keys(send, send_counts)
    ph.mark("exchange")
    mine = count_keys(recv)
    assert mine.numel() == shard
    ph.mark("count")
    if value_hist is None:
        if mine.is_cuda:
            from . import engine as _eng
            dev, st = _dev_args(mine)

            def value_hist(c, h):
                _eng.value_hist_device(c.data_ptr(), c.numel(), h.data_ptr(), device=dev, stream=st)
        else:
            def value_hist(c, h):
                h += torch.bincount(c.to(torch.int64) & 0xFFFF, minlength=65536).to(torch.int32)
    hist = torch.zeros(n_k * 2 * 65536, dtype=torch.int32, device=send.device)
    pos, end = rank * shard, (rank + 1) * shard
    while pos < end:  # a range may cover several (k, sample) planes, or a fraction of one
        pl = pos // plane
        stop = min((pl + 1) * plane, end)
        value_hist(mine[pos - rank * shard:stop - rank * shard], hist[pl * 65536:(pl + 1) * 65536])
        pos = stop
    ph.mark("histogram")
    dist.reduce(hist, dst=dst, op=dist.ReduceOp.SUM)
    dist.reduce(f1, dst=dst, op=dist.ReduceOp.SUM)
    ph.mark("reduce")
    if timings is not None:
        timings.update(ph.result())
        timings["mode"] = "owner"
        timings["keys_sent"] = int(sum(int(c) for c in send_counts))
    return hist.view(n_k, 2, 65536), f1


def merge_owner_engine(engine, sketch, f1, n_k, r_bits, dst=0, timings=None):
    """merge_owner for an engine on this rank's GPU: the pending hit log is split by owner (ntc_log_export_device), exchanged, and the keys this rank receives
    become its engine's pending log (ntc_log_replace_device) — the engine's own sketch update then counts them into `sketch` (the engine's ext_sketch).
    -> (p_hist, f1), or None when some rank's sketch already holds counts (an update ran during the run, or direct atomics): every rank then returns None
    and the caller merges counters (merge_to_value_histograms), which is always possible."""
    from . import _abi
    world, rank = dist.get_world_size(), dist.get_rank()
    ph = _Phases(sketch.device, timings is not None)
    ph.mark("start")
    try:
        counts = engine.log_export(world)
        ok = 1
    except _abi.NtcError:
        counts, ok = [0] * world, 0
    flag = torch.tensor([ok], dtype=torch.int32, device=sketch.device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag[0]) == 0:
        return None
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    send = torch.empty(max(offs[-1], 1), dtype=torch.int32, device=sketch.device)
    engine.log_export(world, send.data_ptr(), offs[:-1])
    send = send[:offs[-1]]
    ph.mark("export")
    shard = sketch.numel() // world

    def count_keys(recv):
        engine.log_replace(recv.data_ptr(), recv.numel())
        engine.flush()
        count_keys.keep = recv  # (until the stream has passed the copy)
        return sketch[rank * shard:(rank + 1) * shard]
    return merge_owner(send, counts, count_keys, f1, n_k, r_bits, dst=dst, timings=timings, ph=ph)


def reduce_hll(regs, f1, dst=0):
    """in-place MAX reduce of nthll's register file (any integer tensor) and SUM of F1 to rank dst.
    The reference merges its per-thread register files the same way (nthll.cpp:240-245)."""
    if dist.is_available() and dist.is_initialized():
        dist.reduce(regs, dst=dst, op=dist.ReduceOp.MAX)
        dist.reduce(f1, dst=dst, op=dist.ReduceOp.SUM)
    return regs, f1


def to_uint16_counters(sketch_i32):
    """the reference's t_Counter view of a merged sketch: wrap to 16 bits (ntcard.cpp:142-143,439)"""
    return (sketch_i32 & 0xFFFF).to(torch.int32)
