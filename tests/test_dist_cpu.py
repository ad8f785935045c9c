"""world_size-2 gloo rehearsal of the N>1 path of bench.py: ranks take disjoint shards, no data-path
collective; only the timing reduction (MAX of seconds, SUM of frames) crosses ranks."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tacotron2_amd.sharding import make_batches, shard_batches
    lens = np.random.default_rng(3).integers(60, 141, 256)
    batches = make_batches(lens, 64)
    mine = shard_batches(len(batches), world, rank)
    frames = float(sum(len(batches[b]) * 800 for b in mine))
    tot = torch.tensor([0.5 + rank, frames], dtype=torch.float64)
    tmax = tot.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    q.put((rank, mine, float(tmax[0]), float(tot[1])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_reduction():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2] and res[1][1] == [1, 3]
    assert res[0][2] == res[1][2] == 1.5 and res[0][3] == res[1][3] == 256 * 800.0
