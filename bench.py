#!/usr/bin/env python3
"""bench.py — scan-pairs/s of the MI355X-native radar-odometry front end.

One "step" = every resident lane (independent sequence) advances by ONE scan pair through the
whole hot path (ingest+peaks -> warp+quantise -> pyramid -> KLT -> outlier rejection ->
Kabsch -> motion-distortion LM -> keyframe bookkeeping), inputs already resident in HBM.
value = lanes * steps * n_gpus / max-over-ranks wall time.  Multi-GPU = one process per GPU,
sequences sharded by rank, no data-path collective ("weak" scaling).

Prints ONE JSON line (rank 0) with the `roofline` and `cpu_baseline` objects described in
DESIGN.md.  The cpu_baseline leg times the oracle (CPU restatement, 1 core) on a bounded
sample of the same synthetic workload - it is a reported baseline, not the optimisation target."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--lanes", type=int, default=None, help="independent sequences resident per GPU (default 4096; 1024 with --h2d)")
    ap.add_argument("--frames", type=int, default=7, help="frames per synthetic sequence (played ping-pong)")
    ap.add_argument("--distinct", type=int, default=4, help="distinct synthetic sequences generated per rank")
    ap.add_argument("--cpu-pairs", type=int, default=120, help="scan pairs timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-md", action="store_true", help="motionDistortion OFF (Kabsch dead reckoning)")
    ap.add_argument("--kernel-reps", type=int, default=10)
    ap.add_argument("--h2d", action="store_true", help="stream every scan from pinned host memory over PCIe (double-buffered pool); reports the PCIe-inclusive rate")
    ap.add_argument("--engines", type=int, default=1, help="independent engine instances (contexts/streams) per GPU; lanes are split between them")
    args = ap.parse_args()
    if args.lanes is None:
        args.lanes = 1024 if args.h2d else 4096

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist_mod
        torch.cuda.set_device(local_rank)
        try:
            dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL on ROCm
        except Exception as e:                                                                  # plumbing only: fall back to gloo
            sys.stderr.write(f"[bench] nccl init failed ({e}); using gloo for the barrier/reduction\n")
            dist_mod.init_process_group("gloo")
        dist = dist_mod

    from radarslampy_amd import _ffi, synth
    from radarslampy_amd.engine import Engine

    ctx = _ffi.Context(local_rank)
    info = ctx.device_info()
    E = max(1, args.engines)
    B, T, D = args.lanes, args.frames, max(1, min(args.distinct, args.lanes))
    assert B % E == 0
    BE = B // E
    seqs = [synth.make_sequence(1000 * rank + 17 * d + 5, T, n_static=460, n_movers=24, distortion=not args.no_md) for d in range(D)]
    # every lane owns private copies of its T records (device-to-device replicas of the D distinct
    # sequences): identical content, distinct HBM addresses -> input reads are real HBM traffic
    ctxs = [ctx] + [_ffi.Context(local_rank) for _ in range(E - 1)]
    engs = []
    for e in range(E):
        en = Engine(BE, BE * T, ctx=ctxs[e], motion_distortion=not args.no_md)
        Dn = min(D, BE)
        for d in range(Dn):
            for t in range(T):
                en.upload_scan(d * T + t, seqs[d][0][t])      # lanes 0..D-1 hold the originals
        for b in range(Dn, BE):
            for t in range(T):
                en.copy_scan(b * T + t, (b % Dn) * T + t)
        for b in range(BE):
            d = b % Dn
            en.init_lane(b, b * T, seqs[d][2], seqs[d][1][0])
        engs.append(en)
    eng = engs[0]

    # ping-pong frame schedule 1,2,..,T-1,T-2,..,0,1,..
    cyc = list(range(1, T)) + list(range(T - 2, -1, -1))

    def idx(step):
        t = cyc[step % len(cyc)]
        return np.array([b * T + t for b in range(BE)], np.int32)

    def step_all(i):
        ix = idx(i)
        for en in engs:
            en.step(ix)

    def barrier():
        for en in engs:
            en.synchronize()
        if dist is not None:
            dist.barrier()
            for en in engs:
                en.synchronize()

    s = 0
    if args.h2d:
        # PCIe-inclusive mode (f2): one engine, pool = two halves of B slots; lanes of one sequence are contiguous so
        # that D replicated uploads (host stride 0) feed all lanes; upload(i+1) overlaps step(i) on the copy stream
        assert E == 1
        for en in engs:
            en.close()
        eng = Engine(B, 2 * B, ctx=ctx, motion_distortion=not args.no_md)
        engs = [eng]
        per = B // D
        pinned = ctx.host_alloc((D * T, 400, 3779))
        for d in range(D):
            for t in range(T):
                pinned[d * T + t] = seqs[d][0][t]

        def upload(step, half):
            t = 0 if step < 0 else cyc[step % len(cyc)]
            for d in range(D):
                n = per if d < D - 1 else B - per * (D - 1)
                eng.upload_scans_async(half * B + d * per, pinned[d * T + t], n=n, stride=0)

        upload(-1, 0)
        eng.synchronize()
        for b in range(B):
            d = min(b // per, D - 1)
            eng.init_lane(b, b, seqs[d][2], seqs[d][1][0])
        upload(0, 1)

        def step_all(i):                                   # noqa: F811
            half = (i + 1) % 2                             # scans of step i live in half (i+1)%2 (step 0 -> half 1)
            eng.fence()
            eng.step(np.arange(B, dtype=np.int32) + half * B)
            upload(i + 1, i % 2)

    for _ in range(args.warmup):
        step_all(s); s += 1
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_all(s); s += 1
    barrier()
    dt = time.perf_counter() - t0
    res = eng.results()
    stages = eng.stage_times()
    if dist is not None:
        import torch
        tt = torch.tensor([dt], device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # BASELINE config 5: the only exchange the path has - every rank broadcasts one keyframe payload
    # {pose, velocity, features, polar peaks} of its lane 0 over RCCL (untimed w.r.t. `value`, reported)
    kf_ms = None
    if dist is not None:
        try:
            from radarslampy_amd.distributed import broadcast_keyframe
            mine = dict(pose=res[0]["pose"], velocity=res[0]["velocity"], features=eng.lane_features(0), peaks=eng.lane_peaks(0))
            broadcast_keyframe(mine if rank == 0 else None, 0, dist)            # warm-up (communicator setup)
            k0 = time.perf_counter()
            for src in range(world):
                got = broadcast_keyframe(mine if rank == src else None, src, dist)
                assert got["features"].shape[1] == 2 and got["peaks"].shape[1] == 2
            kf_ms = (time.perf_counter() - k0) * 1e3 / world
        except Exception as e:                                                    # never lose the bench line over the extra
            sys.stderr.write(f"[bench] keyframe broadcast skipped: {e}\n")

    out = None
    if rank == 0:
        pairs = B * args.steps * world
        value = pairs / dt
        # ---- roofline of the dominant HBM-streaming kernel.  `avg_launch_ms` is the kernel's average launch duration
        # over the K timed steps, from HIP event pairs recorded on the stream it runs on (roam_engine_kernel_avg; no
        # synchronisation inside the timed region).  In the pipelined engine other kernels share the GPU during
        # those launches, so the same kernels are also re-launched alone after the timed region
        # (roam_engine_time_kernel) and reported as `isolated_*`.
        names = ("ingest_peaks", "warp_quantise", "pyramid")
        live = {k: eng.kernel_avg(k, args.steps)[0] for k in names}
        iso = {k: eng.time_kernel(k, args.kernel_reps) for k in names}
        dom_stream = max(iso, key=lambda k: iso[k][0])          # dominant by its own (isolated) cost: in the pipeline the
        # in-step durations of concurrent kernels stretch over each other and say little about which one costs most
        ms, algo_bytes = live[dom_stream], iso[dom_stream][1]
        achieved = algo_bytes / (ms * 1e-3) / 1e9
        # HBM traffic of that kernel from the committed PMC passes (profiles/pmc_run.sh), per launch
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            kname = {"warp_quantise": "warp_gather_kernel", "ingest_peaks": "peaks_rows_u8_wave_kernel", "pyramid": "pyr_down_rows_kernel"}[dom_stream]
            traffic = tj["kernels"][kname]["traffic_bytes_per_scan"] * B
        except Exception:
            pass
        roofline = {"bound": "hbm", "kernel": dom_stream, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "avg_launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": algo_bytes,
                    "isolated_achieved": round(algo_bytes / (iso[dom_stream][0] * 1e-3) / 1e9, 2),
                    "isolated_frac": round(algo_bytes / (iso[dom_stream][0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "in_step_kernel_ms": {k: round(v, 4) for k, v in live.items()},
                    "isolated_kernel_ms": {k: round(v[0], 4) for k, v in iso.items()},
                    "isolated_kernel_GBs": {k: round(v[1] / (v[0] * 1e-3) / 1e9, 1) for k, v in iso.items()}}
        # whole-path view: SURVEY 8d B_min = 13.07 MB per steady pair
        path_gbs = 13.07e6 * (value / world) / 1e9
        cpu = None
        if args.cpu_pairs > 0 and world == 1:          # the CPU leg is timed on rank 0 of a single-GPU run only
            import oracle
            recs, poses, feat = seqs[0]
            P = oracle.OdometryPipeline(recs[0], feat, poses[0], motion_distortion=not args.no_md)
            n = 0
            c0 = time.perf_counter()
            while n < args.cpu_pairs:
                P.step(recs[cyc[n % len(cyc)]]); n += 1
            cdt = time.perf_counter() - c0
            cpu = {"value": round(n / cdt, 3), "unit": "scan-pairs/s", "cores": 1, "kind": "port",
                   "sample": f"{n} consecutive scan pairs of synthetic sequence 0 (same workload, oracle C/numpy restatement, 1 thread)"}
        out = {
            "metric": "radar scan-pairs/sec (400x3768 polar)", "value": round(value, 2), "unit": "scan-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/f32/f64",
            "data": f"synthetic Oxford-format 400x3779 u8 records; {D} distinct seeded sequences x {T} frames per rank, replicated into {B} lane-private HBM copies, ping-pong replay",
            "config": {"workload": "steady-state scan pair, full hot path (ingest+peaks, warp, pyramid, KLT, max-clique outlier rejection, Kabsch, "
                                   + ("motion-distortion LM" if not args.no_md else "dead reckoning") + ")",
                       "lanes_per_gpu": B, "engines_per_gpu": E, "h2d_streaming": bool(args.h2d), "frames": T, "device": info["name"], "arch": info["arch"],
                       "initial_features": int(np.mean([len(q[2]) for q in seqs])),
                       "mean_tracked": round(float(np.mean([r["n_tracked"] for r in res])), 1),
                       "mean_inliers": round(float(np.mean([r["n_inliers"] for r in res])), 1),
                       "mean_lm_nfev": round(float(np.mean([r["lm_nfev"] for r in res])), 1),
                       "stage_ms_last_step": {k: round(v, 4) for k, v in stages.items()},
                       "whole_path_Bmin_GBs_per_gpu": round(path_gbs, 3),
                       "keyframe_broadcast_ms": None if kf_ms is None else round(kf_ms, 3)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
    for en in engs:
        en.close()
    for c in ctxs:
        c.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
