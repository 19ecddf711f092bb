"""world_size-2 gloo test of the multi-GPU merge path (CPU tensors): per-rank private sketches,
read-index sharding, one SUM reduce, uint16 wrap afterwards == the single-process sketch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import orc
from ntcard_amd import parallel

K, RB, SB, N, L = 25, 14, 3, 3000, 100


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, n = parallel.split_reads(N, world)[rank]
    slots = orc.gen_reads(3, first, n, L, L + 4, 1, genome_len=20_000)
    reads = [slots[i * (L + 4): i * (L + 4) + L].tobytes() for i in range(n)]
    # every rank pre-loads one counter so that the merged value wraps past 65535 (uint16 semantics)
    counters, f1 = orc.sketch_reads(reads, [K], 0, RB, SB)
    sk = torch.from_numpy(counters.astype(np.int64).reshape(-1)).to(torch.int32)
    sk[7] += 40000
    f1t = torch.from_numpy(f1.astype(np.int64))
    parallel.reduce_sketch(sk, f1t, dst=0)
    if rank == 0:
        np.save(os.path.join(out_dir, "sk.npy"), parallel.to_uint16_counters(sk).numpy())
        np.save(os.path.join(out_dir, "f1.npy"), f1t.numpy())
    dist.destroy_process_group()


def test_two_rank_merge_equals_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sk = np.load(tmp_path / "sk.npy")
    f1 = np.load(tmp_path / "f1.npy")
    slots = orc.gen_reads(3, 0, N, L, L + 4, 1, genome_len=20_000)
    reads = [slots[i * (L + 4): i * (L + 4) + L].tobytes() for i in range(N)]
    oc, of1 = orc.sketch_reads(reads, [K], 0, RB, SB)
    expect = oc.astype(np.int64).reshape(-1)
    expect[7] = (expect[7] + 80000) & 0xFFFF
    assert int(f1[0]) == int(of1[0])
    assert np.array_equal(sk.astype(np.int64), expect)


def _hll_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, n = parallel.split_reads(N, world)[rank]
    slots = orc.gen_reads(3, first, n, L, L + 4, 1, genome_len=20_000)
    reads = [slots[i * (L + 4): i * (L + 4) + L].tobytes() for i in range(n)]
    regs, _ = orc.hll_reads(reads, K, 10)
    rt = torch.from_numpy(regs.astype(np.int32))
    f1t = torch.tensor([sum(len(orc.hash_read(r, K)[0]) for r in reads)], dtype=torch.int64)
    parallel.reduce_hll(rt, f1t, dst=0)
    if rank == 0:
        np.save(os.path.join(out_dir, "regs.npy"), rt.numpy())
        np.save(os.path.join(out_dir, "f1.npy"), f1t.numpy())
    dist.destroy_process_group()


def test_two_rank_hll_merge_equals_single_process(tmp_path):
    world = 2
    mp.spawn(_hll_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    slots = orc.gen_reads(3, 0, N, L, L + 4, 1, genome_len=20_000)
    reads = [slots[i * (L + 4): i * (L + 4) + L].tobytes() for i in range(N)]
    regs, _ = orc.hll_reads(reads, K, 10)
    assert np.array_equal(np.load(tmp_path / "regs.npy"), regs.astype(np.int32))
    assert int(np.load(tmp_path / "f1.npy")[0]) == sum(len(orc.hash_read(r, K)[0]) for r in reads)


def _hist_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    klist = [21, 40]
    first, n = parallel.split_reads(N, world)[rank]
    slots = orc.gen_reads(3, first, n, L, L + 4, 1, genome_len=20_000)
    reads = [slots[i * (L + 4): i * (L + 4) + L].tobytes() for i in range(n)]
    counters, f1 = orc.sketch_reads(reads, klist, 0, RB, SB)
    sk = torch.from_numpy(counters.astype(np.int64).reshape(-1)).to(torch.int32)
    sk[5] += 50000  # the merged counter wraps past 65535
    sk[9] += 70000 + rank  # a per-rank uint32 counter that is itself past 65535 (only its low 16 bits may count)

    ph, f1t = parallel.merge_to_value_histograms(sk, torch.from_numpy(f1.astype(np.int64)), len(klist), RB, dst=0)  # (CPU tensors: torch arithmetic)
    if rank == 0:
        np.save(os.path.join(out_dir, "ph.npy"), ph.numpy())
        np.save(os.path.join(out_dir, "f1.npy"), f1t.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_value_histogram_merge(tmp_path, world):
    """16-bit slice exchange + wrapping local sums + per-rank histograms == histogram of the single-process sketch"""
    klist = [21, 40]
    mp.spawn(_hist_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    slots = orc.gen_reads(3, 0, N, L, L + 4, 1, genome_len=20_000)
    reads = [slots[i * (L + 4): i * (L + 4) + L].tobytes() for i in range(N)]
    oc, of1 = orc.sketch_reads(reads, klist, 0, RB, SB)
    oc = oc.astype(np.int64)
    oc[0, 0, 5] = (oc[0, 0, 5] + world * 50000) & 0xFFFF
    oc[0, 0, 9] = (oc[0, 0, 9] + world * 70000 + world * (world - 1) // 2) & 0xFFFF
    ph = np.load(tmp_path / "ph.npy")
    for ki in range(len(klist)):
        assert np.array_equal(ph[ki].astype(np.uint32), orc.value_hist(oc[ki].astype(np.uint16), RB))
    assert np.array_equal(np.load(tmp_path / "f1.npy").astype(np.uint64), of1)


def _owner_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    klist = [21, 40]
    first, n = parallel.split_reads(N, world)[rank]
    slots = orc.gen_reads(3, first, n, L, L + 4, 1, genome_len=20_000)
    reads = [slots[i * (L + 4): i * (L + 4) + L].tobytes() for i in range(n)]
    counters, f1 = orc.sketch_reads(reads, klist, 0, RB, SB)
    # this rank's hit log: every increment of its private sketch as one key (+ one counter hit 30000 times per rank: the merged value wraps past 65535;
    # rank 1 has no hits at all for the last owner's range: an empty message)
    c = counters.astype(np.int64).reshape(-1)
    keys = np.repeat(np.arange(c.size, dtype=np.int64), c)
    keys = np.concatenate([keys, np.full(30000, 5, dtype=np.int64)])
    ncnt = c.size
    shard = ncnt // world
    if rank == 1:
        keys = keys[keys < (world - 1) * shard]
    rng = np.random.default_rng(rank)
    rng.shuffle(keys)
    owner = keys // shard
    order = np.argsort(owner, kind="stable")
    send = torch.from_numpy(keys[order].astype(np.int32))
    send_counts = np.bincount(owner, minlength=world).tolist()

    def count_keys(recv):
        assert bool(((recv.to(torch.int64) // shard) == rank).all())
        return torch.bincount(recv.to(torch.int64) - rank * shard, minlength=shard).to(torch.int32)
    t = {}
    ph, f1t = parallel.merge_owner(send, send_counts, count_keys, torch.from_numpy(f1.astype(np.int64)), len(klist), RB, dst=0, timings=t)
    assert t["mode"] == "owner" and t["keys_sent"] == len(keys) and "exchange_ms" in t and "count_ms" in t
    if rank == 0:
        np.save(os.path.join(out_dir, "ph.npy"), ph.numpy())
        np.save(os.path.join(out_dir, "f1.npy"), f1t.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_owner_mode_merge(tmp_path, world):
    """the merge that ships hits (round 6): keys by counter-range owner in one variable-size exchange, counted by the owner, histograms to rank 0 ==
    the value histogram of the single-process sketch (incl. a counter that wraps past 65535 and an empty message)"""
    klist = [21, 40]
    mp.spawn(_owner_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    oc = np.zeros((len(klist), 2, 1 << RB), dtype=np.int64)
    ncnt = oc.size
    shard = ncnt // world
    for rank in range(world):  # the sum of what the ranks logged (rank 1 dropped its hits of the last range)
        first, n = parallel.split_reads(N, world)[rank]
        slots = orc.gen_reads(3, first, n, L, L + 4, 1, genome_len=20_000)
        reads = [slots[i * (L + 4): i * (L + 4) + L].tobytes() for i in range(n)]
        c, _ = orc.sketch_reads(reads, klist, 0, RB, SB)
        c = c.astype(np.int64).reshape(-1)
        if rank == 1:
            c[(world - 1) * shard:] = 0
        oc += c.reshape(oc.shape)
    oc.reshape(-1)[5] += world * 30000
    oc &= 0xFFFF
    ph = np.load(tmp_path / "ph.npy")
    for ki in range(len(klist)):
        assert np.array_equal(ph[ki].astype(np.uint32), orc.value_hist(oc[ki].astype(np.uint16), RB))
    slots = orc.gen_reads(3, 0, N, L, L + 4, 1, genome_len=20_000)
    reads = [slots[i * (L + 4): i * (L + 4) + L].tobytes() for i in range(N)]
    _, of1 = orc.sketch_reads(reads, klist, 0, RB, SB)
    assert np.array_equal(np.load(tmp_path / "f1.npy").astype(np.uint64), of1)


def test_split_reads_covers_everything():
    for n, w in ((10, 3), (100_000_000, 8), (7, 8)):
        parts = parallel.split_reads(n, w)
        assert sum(c for _, c in parts) == n
        assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(w - 1))
    assert parallel.read_range(3, 8, 1000) == (3000, 1000)
