"""CPU, world_size 2, gloo: the N > 1 path of the batch sharder (image-index round-robin, no data-path collective,
torch.distributed only for the optional gather of decoded outputs).  The per-rank 'decode' is the CPU oracle here;
on the GPU box the same shard.py code runs over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from gamut_amd import shard


def test_shard_indices_cover_batch_exactly_once():
    for n in (0, 1, 7, 8, 8192):
        for world in (1, 2, 3, 8):
            seen = sorted(i for r in range(world) for i in shard.shard_indices(n, r, world))
            assert seen == list(range(n))
            assert all(shard.owner_of(i, world) == r for r in range(world) for i in shard.shard_indices(n, r, world))
            sizes = [len(shard.shard_indices(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _batch(n):
    """mixed batch: image i is a random-coefficient JPEG frame of varying size / sampling mode (sizes differ per image)"""
    items = []
    for i in range(n):
        rng = np.random.default_rng(100 + i)
        st = [4, 1, 0, 2][i % 4]
        w, h = 24 + 8 * (i % 5), 16 + 8 * (i % 3)
        mw, mh = {0: (8, 8), 1: (8, 8), 2: (16, 8), 4: (16, 16)}[st]
        nb = {0: 1, 1: 3, 2: 4, 4: 6}[st]
        co = rng.integers(-300, 300, (((w + mw - 1) // mw) * ((h + mh - 1) // mh) * nb, 64)).astype(np.int16)
        items.append((w, h, st, co))
    return items


def _decode(item):
    w, h, st, co = item
    return torch.from_numpy(O.jpeg_reconstruct(w, h, 1 if st == 0 else 3, st, co, None, 4).reshape(-1).copy())


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    items = _batch(n)
    mine = shard.shard_indices(n, rank, world)
    outs = [_decode(items[i]) for i in mine]
    allout = shard.gather_outputs(outs, n, rank, world)
    ok = all(torch.equal(allout[i], _decode(items[i])) for i in range(n))
    # throughput accounting as in bench.py: MAX of per-rank times over ranks
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, ok, len(mine), float(t.item())))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [9, 2, 1])
def test_two_rank_round_robin_and_gather(n):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert sum(r[2] for r in res) == n
    assert all(r[3] == 2.0 for r in res)


def test_a_rank_without_images_still_has_a_device_for_the_collective():
    """n_items < world: the empty rank's buffers must live where the other ranks' do (an RCCL all_gather with one CPU tensor fails).
    gather_outputs takes the device from its argument, the outputs, or -- backend nccl -- the process's GPU; never 'cpu' by accident."""
    t = torch.zeros(3, dtype=torch.uint8)
    assert shard._collective_device([t], None, None) == t.device
    assert shard._collective_device([], "cpu", None) == torch.device("cpu")
    assert shard._collective_device([], torch.device("meta"), None) == torch.device("meta")
    assert shard._collective_device([], None, None) == torch.device("cpu")          # no process group: nothing to match
