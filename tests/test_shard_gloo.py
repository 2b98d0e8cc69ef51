"""CPU, world_size 2 over gloo: the N>1 path of the engine is "units shard by rank, sizes and
status gathered on the host" (SURVEY.md 8e).  Checks the partition is contiguous, disjoint,
covering and byte-balanced, and that the gather reproduces unit order and first-error-wins."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from decompress_amd import shard


class U:
    def __init__(self, in_off, in_limit, out_cap):
        self.in_off, self.in_limit, self.out_cap = in_off, in_limit, out_cap


def make_units(n=1000, seed=3):
    import random
    rng = random.Random(seed)
    units, off = [], 0
    for _ in range(n):
        sz = rng.choice([100, 1000, 5000, 90000])
        units.append(U(off, off + sz, sz * 3))
        off += sz
    return units


def test_partition_properties():
    units = make_units()
    for world in (1, 2, 3, 4, 8):
        ranges = [shard.shard_units(units, r, world) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == len(units)
        for (a, b), (c, d) in zip(ranges, ranges[1:]):
            assert b == c and a <= b
        sizes = [sum(u.in_limit - u.in_off for u in units[a:b]) for a, b in ranges]
        assert max(sizes) - min(sizes) <= 2 * 90000
    assert shard.partition([], 4) == [(0, 0)] * 4
    assert shard.partition([5], 4)[0] == (0, 1)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    units = make_units()
    lo, hi = shard.shard_units(units, rank, world)
    # stand-in for the device call: status 0 everywhere except one unit owned by the last rank
    local = [(0 if i != 777 else 45, units[i].out_cap) for i in range(lo, hi)]
    flat, first_bad = shard.gather_outcomes(local, dist)
    t = torch.tensor([float(sum(n for _, n in local))], dtype=torch.float64)
    dist.all_reduce(t)                                        # what bench.py does for its timing/size totals
    q.put((rank, lo, hi, len(flat), first_bad, float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    units = make_units()
    (r0, lo0, hi0, n0, bad0, tot0), (r1, lo1, hi1, n1, bad1, tot1) = got
    assert (lo0, hi1) == (0, len(units)) and hi0 == lo1
    assert n0 == n1 == len(units) and bad0 == bad1 == 777
    assert tot0 == tot1 == float(sum(u.out_cap for u in units))
