"""CPU (-m "not gpu"): bench.py's REAL multi-rank code path - its own `--gpus N` launcher and the torch.distributed.run
launch the driver uses - with `--dry-engine` (a stub instead of the GPU engine; rendezvous, rank environment handling,
barrier, max-over-ranks time, keyframe broadcast bookkeeping and the one-JSON-line contract are the real code)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
ARGS = ["--dry-engine", "--steps", "4", "--warmup", "1", "--lanes", "8"]


def _line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout                      # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def _check(d, world, launcher):
    assert d["n_gpus"] == world and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["lanes_per_gpu"] == 8 and d["config"]["launcher"] == launcher
    # max over ranks: the stub's rank r sleeps 2 ms * (r + 1) per step
    assert d["ms_per_step"] >= 2.0 * world * 0.9
    assert abs(d["value"] - 8 * 4 * world / (d["ms_per_step"] * 4e-3)) / d["value"] < 1e-3
    if world > 1:
        assert d["config"]["collective_backend"] == "file" and d["config"]["comm_rank_world_seen"] == [0, world]
        assert d["config"]["keyframe_broadcast_ms"] is not None


def test_bench_single_process_dry():
    r = subprocess.run([sys.executable, BENCH] + ARGS, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    _check(_line(r.stdout), 1, "single process")


def test_bench_own_launcher_world2():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + ARGS, capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    _check(_line(r.stdout), 2, "bench.py --gpus")


def test_bench_under_torch_distributed_run_world2():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", BENCH, "--gpus", "2"] + ARGS, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    _check(_line(r.stdout), 2, "torch.distributed.run env")
