"""CPU (-m "not gpu"): bench.py's multi-rank plumbing with gloo and a stub engine (no GPU): the
barrier / max-over-ranks reduction / keyframe broadcast code path that the driver launches with
torch.distributed.run is exercised at world_size 2."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_distributed_plumbing_world2(tmp_path):
    stub = tmp_path / "stub_bench.py"
    stub.write_text(textwrap.dedent(f"""
        import os, sys, time, json
        sys.path.insert(0, {ROOT!r})
        import numpy as np
        import torch, torch.distributed as dist
        from radarslampy_amd.distributed import broadcast_keyframe, shard_sequences
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        dist.barrier()
        t0 = time.perf_counter(); time.sleep(0.05 * (rank + 1)); dist.barrier()
        dt = time.perf_counter() - t0
        tt = torch.tensor([dt]); dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        mine = dict(pose=np.arange(3.0) + rank, velocity=np.zeros(3), features=np.full((5 + rank, 2), rank, np.float32),
                    peaks=np.full((7, 2), rank, np.int32))
        ok = True
        for src in range(world):
            got = broadcast_keyframe(mine if rank == src else None, src, dist)
            ok &= got["features"].shape == (5 + src, 2) and float(got["pose"][0]) == float(src)
        if rank == 0:
            print(json.dumps({{"n_gpus": world, "max_dt": float(tt.item()), "ok": bool(ok), "mine": shard_sequences(8, rank, world)}}))
        dist.barrier(); dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", str(stub)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["ok"] and d["max_dt"] >= 0.1 and d["mine"] == [0, 2, 4, 6]
