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
