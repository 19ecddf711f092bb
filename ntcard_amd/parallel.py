"""Multi-GPU plumbing for the one exchange step the path has: reads are sharded by
index range, every rank owns a private full sketch, and the per-rank sketches are merged with ONE
integer SUM reduce (RCCL over xGMI when the tensors live on GPUs; gloo in the CPU tests).

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


def merge_to_value_histograms(sketch, f1, n_k, r_bits, value_hist, dst=0):
    """Multi-GPU merge for the estimator (compEst only needs the value histogram p[2][65536] of the SUMMED counters,
    ntcard.cpp:240-247): reduce-scatter the per-rank sketches so that every rank holds one slice of the sum, histogram
    that slice locally, and send only the histograms (256 KiB per plane) to rank `dst`.  On a fully connected xGMI node a
    reduce-scatter keeps every link busy, whereas a reduce to one rank funnels 1 GiB per k into it.

    sketch: int32 tensor [n_k * 2 * 2^r_bits] (uint32 counters), f1: int64 [n_k]; both are consumed.
    value_hist(counters_slice, hist_slice): accumulates the histogram of (counter & 0xffff) into an int32[65536] view.
    Returns (p_hist int32 [n_k, 2, 65536], f1) — meaningful on rank dst."""
    world, rank = dist.get_world_size(), dist.get_rank()
    plane = 1 << r_bits
    n = sketch.numel()
    assert n == n_k * 2 * plane and n % world == 0 and (n // world) % 4 == 0
    shard = n // world
    if dist.get_backend() == "gloo":  # gloo has no reduce_scatter: same result through an all-reduce (CPU tests only)
        dist.all_reduce(sketch, op=dist.ReduceOp.SUM)
        mine = sketch[rank * shard:(rank + 1) * shard]
    else:
        mine = torch.empty(shard, dtype=sketch.dtype, device=sketch.device)
        dist.reduce_scatter_tensor(mine, sketch, op=dist.ReduceOp.SUM)
    hist = torch.zeros(n_k * 2 * 65536, dtype=torch.int32, device=sketch.device)
    pos, end = rank * shard, (rank + 1) * shard
    while pos < end:  # a slice may cover several (k, sample) planes, or a fraction of one
        pl = pos // plane
        stop = min((pl + 1) * plane, end)
        value_hist(mine[pos - rank * shard:stop - rank * shard], hist[pl * 65536:(pl + 1) * 65536])
        pos = stop
    dist.reduce(hist, dst=dst, op=dist.ReduceOp.SUM)
    dist.reduce(f1, dst=dst, op=dist.ReduceOp.SUM)
    return hist.view(n_k, 2, 65536), f1


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
