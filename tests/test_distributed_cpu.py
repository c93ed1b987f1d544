"""CPU (-m "not gpu"): the N>1 plumbing at world_size 2 - sequence sharding, the rendezvous directory, the file-backed
stand-in of the communicator (same interface as the RCCL one) - cross-checked against torch.distributed's gloo backend
in the same two processes."""
import os
import socket

import numpy as np


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Eng:
    def __init__(self, rank):
        self.rank = rank

    def live_keyframe(self, lane):
        rng = np.random.default_rng(5 + self.rank)
        return dict(pose=np.array([1.5, -2.0, 0.3]) + self.rank, velocity=np.array([4.0, 0.1, -0.02]),
                    prunedUndistortedLocals=rng.random((137 + self.rank, 2)), peaks=rng.integers(0, 2025, (5100, 2)).astype(np.int32),
                    scan=3, lane=lane)


def _worker(rank, world, port, rdv_dir, q):
    import torch
    import torch.distributed as dist
    from radarslampy_amd.distributed import FileComm, FileRendezvous, shard_sequences
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = FileComm(FileRendezvous(rdv_dir, rank, world, timeout=60))
    out = [comm.info() == (rank, world)]
    for src in range(world):
        got = comm.bcast_keyframe(_Eng(rank), src, 2)
        want = _Eng(src).live_keyframe(2)
        out.append(all(np.array_equal(got[k], want[k]) for k in ("pose", "velocity", "prunedUndistortedLocals", "peaks")) and got["lane"] == 2)
    mine = shard_sequences(8, rank, world)
    t = torch.tensor([float(len(mine))])
    dist.all_reduce(t)                                   # every sequence owned exactly once
    out.append(int(t.item()) == 8 and all(s % world == rank for s in mine))
    tmax = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)          # the bench's max-over-ranks time: file comm == gloo
    out.append(comm.allreduce_max(0.1 * (rank + 1)) == float(tmax.item()))
    comm.barrier()
    dist.barrier()
    comm.close()
    dist.destroy_process_group()
    q.put((rank, out))


def test_sharding_rendezvous_and_file_comm_world2(tmp_path):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path / "rdv"), q)) for r in range(2)]
    os.makedirs(tmp_path / "rdv")
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in res:
        assert all(out), (rank, out)


def test_rendezvous_dir_is_shared_by_the_ranks_of_one_launch():
    from radarslampy_amd.distributed import rank_env, rendezvous_dir
    env = dict(RANK="3", LOCAL_RANK="1", WORLD_SIZE="4", MASTER_PORT="1234", TORCHELASTIC_RUN_ID="abc", TMPDIR="/tmp")
    assert rank_env(env) == (3, 1, 4)
    d = rendezvous_dir(env)
    assert d == rendezvous_dir(dict(env, RANK="0", LOCAL_RANK="0")) and "1234" in d and "abc" in d
    os.rmdir(d)
    assert rendezvous_dir(dict(ROAM_RDV_DIR=d)) == d
    os.rmdir(d)
