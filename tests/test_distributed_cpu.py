"""CPU (-m "not gpu"): the N>1 plumbing with the gloo backend, world_size 2."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from radarslampy_amd.distributed import broadcast_keyframe, shard_sequences
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    kf = dict(pose=np.array([1.5, -2.0, 0.3]), velocity=np.array([4.0, 0.1, -0.02]),
              features=rng.random((137, 2)).astype(np.float32), peaks=rng.integers(0, 2025, (5100, 2)).astype(np.int32))
    out = []
    for src in range(world):
        got = broadcast_keyframe(kf if rank == src else None, src, dist)
        out.append(all(np.array_equal(got[k], kf[k]) for k in kf))
    # empty payload
    e = dict(pose=np.zeros(3), velocity=np.zeros(3), features=np.zeros((0, 2), np.float32), peaks=np.zeros((0, 2), np.int32))
    got = broadcast_keyframe(e if rank == 0 else None, 0, dist)
    out.append(got["features"].shape == (0, 2) and got["peaks"].shape == (0, 2))
    mine = shard_sequences(8, rank, world)
    import torch
    t = torch.tensor([float(len(mine))])
    dist.all_reduce(t)                      # every sequence owned exactly once
    out.append(int(t.item()) == 8 and all(s % world == rank for s in mine))
    tmax = torch.tensor([0.1 * (rank + 1)])
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)   # the bench's max-over-ranks timing
    out.append(abs(float(tmax.item()) - 0.1 * world) < 1e-6)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out))


def test_keyframe_broadcast_and_sharding_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in res:
        assert all(out), (rank, out)
