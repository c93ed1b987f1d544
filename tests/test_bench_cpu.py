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
        # the consumer of the broadcast: every rank's global map holds the warm-up keyframe + one from every rank
        assert d["config"]["global_map_keyframes_and_senders"] == [world + 1, list(range(world))]


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


def test_bench_under_torch_distributed_run_world8():
    """the driver's 8-GPU command line, as it will be issued on an 8-GPU node (non-config5 default path), on the stub engine:
    one JSON line from rank 0 with n_gpus 8, every rank seen by the collective, every rank's keyframe in every rank's map"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", "29651", BENCH, "--gpus", "8"] + ARGS, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    _check(_line(r.stdout), 8, "torch.distributed.run env")


def test_render_pool_is_shared_between_the_ranks_of_a_node(monkeypatch):
    """eight ranks render their sequences at the same time: the default pool of a rank is cores / (2 * world), never cores / 2"""
    sys.path.insert(0, ROOT)
    import bench
    seen = []

    class FakePool:
        def __init__(self, n):
            seen.append(n)

        def map(self, f, jobs):
            return [None for _ in jobs]

        def close(self):
            pass

        def join(self):
            pass

    class FakeCtx:
        Pool = FakePool

    import multiprocessing as mp
    monkeypatch.setattr(mp, "get_context", lambda kind: FakeCtx)
    monkeypatch.setattr(os, "cpu_count", lambda: 128)
    for world, want in ((1, 16), (8, 8)):
        monkeypatch.setenv("WORLD_SIZE", str(world))
        bench.render_sequences(list(range(16)), 7, True, {}, 0)
        assert seen[-1] == want, (world, seen)


def test_launcher_ends_quickly_and_nonzero_when_a_rank_dies():
    """rank 1 raises in its second step while rank 0 is on its way into a barrier: the launcher notices the exit code, tells the
    survivor through the rendezvous directory, and returns non-zero within seconds - nobody waits out a 300 s rendezvous timeout
    (on a GPU box: nobody stays blocked in an RCCL collective holding its GPU)"""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["ROAM_DRY_FAIL_RANK"] = "1"
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + ARGS, capture_output=True, text=True, timeout=60, env=env)
    dt = time.monotonic() - t0
    assert r.returncode != 0 and dt < 10.0, (r.returncode, dt)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]          # no result line from a broken run
    assert "rank(s) [1] exited" in r.stderr and "injected failure" in r.stderr, r.stderr[-1500:]


def test_launcher_refuses_more_ranks_than_gpus(monkeypatch):
    """the GPU count comes from the KFD topology in sysfs - the launcher itself never initialises HIP"""
    sys.path.insert(0, ROOT)
    import bench
    n = bench.visible_gpu_count()
    assert n is None or n >= 0
    src = open(BENCH).read()
    launcher = src[src.index("def visible_gpu_count"):src.index("# ------------------------------------------------------------------------------------------------ engines")]
    assert "_ffi" not in launcher and "Context(" not in launcher and "load_library" not in launcher


def _check_config5(d, world, frames):
    assert d["n_gpus"] == world and d["steps"] == frames - 1 and d["scaling"] == "weak"
    c = d["config"]
    assert c["backend"] == "file" and c["frames"] == frames
    # every keyframe of every rank arrived on rank 0 exactly once: the map holds the sum of what the ranks sent
    assert len(c["keyframes_sent_per_rank"]) == world and c["map_received"] == sum(c["keyframes_sent_per_rank"]) > 0
    assert c["senders"] == [r for r in range(world) if c["keyframes_sent_per_rank"][r]]


def test_config5_exchange_schedule_world2_own_launcher():
    """BASELINE config 5's loop shape - one sequence per rank, the keyframe exchange after EVERY step, keyframes falling on
    different steps on different ranks - on the stub engine: the fixed schedule (one collective per step, empty records
    when a rank has no new keyframe) delivers every keyframe to every rank"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--config5", "--dry-engine", "--c5-frames", "41"], capture_output=True, text=True,
                       timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    _check_config5(_line(r.stdout), 2, 41)


def test_config5_exchange_schedule_world8_torch_distributed_run():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", "29641", BENCH, "--gpus", "8", "--config5", "--dry-engine", "--c5-frames", "21"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    _check_config5(_line(r.stdout), 8, 21)


def test_png_prefetch_pool_keeps_the_order_and_reports_a_bad_frame_at_its_turn(tmp_path):
    """parseData.prefetchRadarRecords (8f-f2: PNG inflate off the thread that feeds the pinned ring): records come back in path
    order whatever the pool's timing, equal to the one-at-a-time decode; a file that does not decode raises when its turn comes"""
    import numpy as np
    import pytest
    from PIL import Image
    sys.path.insert(0, ROOT)
    from radarslampy_amd.parseData import prefetchRadarRecords, readRadarRecord
    rng = np.random.default_rng(5)
    paths, want = [], []
    for i in range(23):
        r = rng.integers(0, 256, (40, 379), dtype=np.uint8)
        r[0, 0] = i
        p = tmp_path / f"{i:04d}.png"
        Image.fromarray(r).save(p)
        paths.append(str(p)); want.append(r)
    for workers, depth in ((1, 0), (4, 0), (6, 2)):
        got = list(prefetchRadarRecords(paths, workers, depth))
        assert len(got) == 23 and all(np.array_equal(g, w) for g, w in zip(got, want)), workers
    assert np.array_equal(readRadarRecord(paths[7]), want[7])
    bad = paths[:5] + [str(tmp_path / "missing.png")] + paths[5:8]
    it = prefetchRadarRecords(bad, 4)
    for i in range(5):
        assert np.array_equal(next(it), want[i])
    with pytest.raises(FileNotFoundError):
        next(it)


def test_png_decode_process_pool_keeps_the_order(tmp_path):
    """parseData.RecordDecodePool: the same contract as the thread pool on a pool of processes with a shared-memory ring - frames
    in path order (views valid until the next one is asked for), a missing file raises at its turn, the pool can be reused"""
    import numpy as np
    import pytest
    from PIL import Image
    sys.path.insert(0, ROOT)
    from radarslampy_amd.parseData import RecordDecodePool
    rng = np.random.default_rng(6)
    paths, want = [], []
    for i in range(19):
        r = rng.integers(0, 256, (40, 379), dtype=np.uint8)
        p = tmp_path / f"{i:04d}.png"
        Image.fromarray(r).save(p)
        paths.append(str(p)); want.append(r)
    with RecordDecodePool(workers=3, depth=4, rec_bytes=40 * 379) as pool:
        for _ in range(2):
            got = [g.copy() for g in pool.records(paths)]
            assert len(got) == 19 and all(np.array_equal(g, w) for g, w in zip(got, want))
        it = pool.records(paths[:3] + [str(tmp_path / "missing.png")] + paths[3:6])
        for i in range(3):
            assert np.array_equal(next(it), want[i])
        with pytest.raises(FileNotFoundError):
            next(it)
        got = [g.copy() for g in pool.records(paths[5:9])]                 # still usable afterwards
        assert all(np.array_equal(g, w) for g, w in zip(got, want[5:9]))


def test_png_decode_process_pool_reports_dead_workers(tmp_path):
    """a pool whose decode processes cannot start (spawned from a __main__ that is no importable file: `python -` from stdin) raises
    instead of waiting for frames for ever"""
    from PIL import Image
    import numpy as np
    p = tmp_path / "a.png"
    Image.fromarray(np.zeros((4, 16), np.uint8)).save(p)
    code = (f"import sys; sys.path.insert(0, {ROOT!r})\n"
            "from radarslampy_amd.parseData import RecordDecodePool\n"
            "try:\n"
            "    pool = RecordDecodePool(workers=2, depth=2, rec_bytes=64)\n"
            f"    list(pool.records([{str(p)!r}]))\n"
            "    pool.close()\n"
            "except RuntimeError as e:\n"
            "    print('RAISED', e)\n")
    out = subprocess.run([sys.executable, "-"], input=code, capture_output=True, text=True, timeout=120, cwd=str(tmp_path))
    assert "RAISED RecordDecodePool: 2 of 2 decode processes died" in out.stdout, out.stdout + out.stderr[-2000:]
