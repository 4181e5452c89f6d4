"""CPU, world_size 2 over gloo: the ensemble sharding / all-gather logic (prediff_amd.ensemble) is invariant to the
world size and returns members in order on every rank.  The per-member "sampler" is a cheap deterministic function of
the member's own noise stream (the HIP engine itself cannot run here)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from prediff_amd.ensemble import all_gather_members, member_noise_fn, sample_ensemble, shard_members

LAT = (2, 4, 4, 3)


def _toy_sample_fn(base_seed):
    def fn(cond, members):
        noise = member_noise_fn(LAT, members, base_seed, "cpu")
        z = noise(0)
        for s in range(1, 4):
            z = 0.9 * z + 0.1 * noise(s) + cond["y"].mean()
        return z
    return fn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, M, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        y = torch.full((1, 3, 8, 8, 1), 0.25)
        out = sample_ensemble(None, {"y": y}, M, base_seed=1000, sample_fn=_toy_sample_fn(1000))
        q.put((rank, out.clone()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("M", [4, 5])
def test_world2_matches_world1(M):
    y = torch.full((1, 3, 8, 8, 1), 0.25)
    ref = sample_ensemble(None, {"y": y}, M, base_seed=1000, sample_fn=_toy_sample_fn(1000))      # world 1
    assert ref.shape == (M,) + LAT
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, M, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        assert torch.equal(got[r], ref), f"rank {r}: members differ from the single-process ensemble"


def test_sharding_helpers():
    assert shard_members(8, 1, 4) == [1, 5] and shard_members(5, 1, 2) == [1, 3]
    assert sum(len(shard_members(32, r, 8)) for r in range(8)) == 32
    a = member_noise_fn(LAT, [3], 7, "cpu")
    b = member_noise_fn(LAT, [1, 3], 7, "cpu")
    assert torch.equal(a(0)[0], b(0)[1]) and torch.equal(a(1)[0], b(1)[1])      # a member's stream ignores its batch-mates
    x = torch.randn(3, 2)
    assert all_gather_members(x, 3, 0, 1) is x
