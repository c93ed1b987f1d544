#!/usr/bin/env python3
"""bench.py — scan-pairs/s of the MI355X-native radar-odometry front end.

One "step" = every resident lane (independent sequence) advances by ONE scan pair through the whole hot path
(ingest+peaks -> warp+quantise -> pyramid -> KLT -> outlier rejection -> Kabsch -> motion-distortion LM -> keyframe
bookkeeping, plus the feature re-detection of the lanes that ran out of features), inputs already resident in HBM.
value = lanes * steps * n_gpus / max-over-ranks wall time.

Multi-GPU = one process per GPU, sequences sharded by rank, no data-path collective ("weak" scaling).
Launch either way:
    python bench.py --gpus N                                  (this script starts the N rank processes itself)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N     (the driver's launcher; only its
                                                               RANK / LOCAL_RANK / WORLD_SIZE variables are used)
Neither path imports torch: barrier and max-over-ranks time go through RCCL (roam_comm_*), the ncclUniqueId through a
rendezvous directory.  Rank 0 prints ONE JSON line with the `roofline` and `cpu_baseline` objects of DESIGN.md §5.
`--dry-engine` swaps the GPU engine for a stub (CPU test of this launcher; never a measurement)."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec
SIMDS, CLOCK_HZ = 1024, 2.4e9


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--lanes", type=int, default=None, help="independent sequences resident per GPU (default 4096; 1024 with --h2d)")
    ap.add_argument("--frames", type=int, default=7, help="frames per synthetic sequence (played ping-pong)")
    ap.add_argument("--distinct", type=int, default=4, help="distinct synthetic sequences generated per rank")
    ap.add_argument("--cpu-pairs", type=int, default=120, help="scan pairs timed on the CPU oracle, 1 core (0 = skip)")
    ap.add_argument("--cpu-procs", type=int, default=-1, help="processes of the N-core CPU leg (-1 = half the logical cores, 0 = skip)")
    ap.add_argument("--no-md", action="store_true", help="motionDistortion OFF (Kabsch dead reckoning)")
    ap.add_argument("--kernel-reps", type=int, default=10)
    ap.add_argument("--h2d", action="store_true", help="stream every scan from pinned host memory over PCIe (double-buffered pool); reports the PCIe-inclusive rate")
    ap.add_argument("--engines", type=int, default=1, help="independent engine instances (contexts/streams) per GPU; lanes are split between them")
    ap.add_argument("--force-comm", action="store_true", help="create the RCCL communicator even at world size 1 (exercises the collective path on a 1-GPU box)")
    ap.add_argument("--dry-engine", action="store_true", help="CPU stub instead of the GPU engine: exercises launcher / rendezvous / reduction only")
    args = ap.parse_args(argv)
    if args.lanes is None:
        args.lanes = 1024 if args.h2d else 4096
    return args


# ------------------------------------------------------------------------------------------------ launcher
def spawn_ranks(args, argv):
    """--gpus N without a launcher: start N rank processes BEFORE anything touches HIP (fresh children, never an exec of
    a process that initialised the GPU), pass rank / device through the environment, relay rank 0's JSON line."""
    rdv = tempfile.mkdtemp(prefix="roam_rdv_")
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), ROAM_RDV_DIR=rdv,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out)
    sys.stdout.flush()
    return max(abs(c) for c in codes)


# ------------------------------------------------------------------------------------------------ engines
class DryEngine:
    """no GPU: keeps the launcher / comm / JSON code path testable on CPU (results are placeholders)"""

    class _Cfg:
        peaks_cap = 64

    def __init__(self, lanes, rank):
        self.lanes, self.rank, self.n, self.cfg = lanes, rank, 0, self._Cfg()

    def step(self, idx):
        assert len(idx) == self.lanes
        self.n += 1
        time.sleep(0.002 * (self.rank + 1))

    def synchronize(self):
        pass

    def results(self):
        return [dict(pose=np.array([float(self.n), float(self.rank), 0.0]), velocity=np.zeros(3), n_tracked=100, n_inliers=90, lm_nfev=8)] * self.lanes

    def live_keyframe(self, lane):
        return dict(pose=np.array([float(self.rank), float(lane), 0.5]), velocity=np.zeros(3),
                    prunedUndistortedLocals=np.full((5 + self.rank, 2), float(self.rank)), peaks=np.full((7, 2), self.rank, np.int32),
                    scan=self.n, lane=lane)

    def close(self):
        pass


def cpu_worker(job):
    """one independent oracle pipeline (N-core leg of the CPU baseline); returns (pairs, seconds)"""
    seed, frames, pairs, md = job
    import oracle
    from radarslampy_amd import synth
    recs, poses, feat = synth.make_sequence(seed, frames, n_static=460, n_movers=24, distortion=md)
    cyc = list(range(1, frames)) + list(range(frames - 2, -1, -1))
    P = oracle.OdometryPipeline(recs[0], feat, poses[0], motion_distortion=md)
    t0 = time.perf_counter()
    for n in range(pairs):
        P.step(recs[cyc[n % len(cyc)]])
    return pairs, time.perf_counter() - t0


def run_rank(args):
    from radarslampy_amd import distributed as D
    rank, local_rank, world = D.rank_env()
    if world != args.gpus:
        sys.stderr.write(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: using the launcher's world size\n")
    B, T, Dn_ = args.lanes, args.frames, max(1, min(args.distinct, args.lanes))
    cyc = list(range(1, T)) + list(range(T - 2, -1, -1))           # ping-pong frame schedule 1,2,..,T-1,T-2,..,0,1,..
    rdv = D.FileRendezvous(D.rendezvous_dir(), rank, world) if (world > 1 or args.force_comm) else None

    if args.dry_engine:
        eng = DryEngine(B, rank)
        engs, ctxs, info = [eng], [], dict(name="dry", arch="none")
        comm = D.FileComm(rdv) if rdv else None
        step_all = lambda i: eng.step(np.zeros(B, np.int32))       # noqa: E731
        seqs = None
    else:
        from radarslampy_amd import _ffi, synth
        from radarslampy_amd.engine import Engine
        ctx = _ffi.Context(local_rank)
        info = ctx.device_info()
        comm = D.RcclComm(ctx, rdv) if rdv else None
        E = max(1, args.engines)
        assert B % E == 0
        BE = B // E
        seqs = [synth.make_sequence(1000 * rank + 17 * d + 5, T, n_static=460, n_movers=24, distortion=not args.no_md) for d in range(Dn_)]
        # every lane owns private copies of its T records (device-to-device replicas of the D distinct
        # sequences): identical content, distinct HBM addresses -> input reads are real HBM traffic
        ctxs = [ctx] + [_ffi.Context(local_rank) for _ in range(E - 1)]
        engs = []
        for e in range(E):
            en = Engine(BE, BE * T, ctx=ctxs[e], motion_distortion=not args.no_md)
            Dn = min(Dn_, BE)
            for d in range(Dn):
                for t in range(T):
                    en.upload_scan(d * T + t, seqs[d][0][t])      # lanes 0..D-1 hold the originals
            for b in range(Dn, BE):
                for t in range(T):
                    en.copy_scan(b * T + t, (b % Dn) * T + t)
            en.synchronize()
            for b in range(BE):
                d = b % Dn
                en.init_lane(b, b * T, seqs[d][2], seqs[d][1][0])
            engs.append(en)
        eng = engs[0]

        def step_all(i):
            t = cyc[i % len(cyc)]
            ix = np.arange(BE, dtype=np.int32) * T + t
            for en in engs:
                en.step(ix)

        if args.h2d:
            # PCIe-inclusive mode (f2): one engine, pool = two halves of B slots; lanes of one sequence are contiguous so
            # that D replicated uploads (host stride 0) feed all lanes; upload(i+1) overlaps step(i) on the copy stream
            assert E == 1
            for en in engs:
                en.close()
            eng = Engine(B, 2 * B, ctx=ctx, motion_distortion=not args.no_md)
            engs = [eng]
            per = B // Dn_
            pinned = ctx.host_alloc((Dn_ * T, 400, 3779))
            for d in range(Dn_):
                for t in range(T):
                    pinned[d * T + t] = seqs[d][0][t]

            def upload(step, half):
                t = 0 if step < 0 else cyc[step % len(cyc)]
                for d in range(Dn_):
                    n = per if d < Dn_ - 1 else B - per * (Dn_ - 1)
                    eng.upload_scans_async(half * B + d * per, pinned[d * T + t], n=n, stride=0)

            upload(-1, 0)
            eng.synchronize()
            for b in range(B):
                d = min(b // per, Dn_ - 1)
                eng.init_lane(b, b, seqs[d][2], seqs[d][1][0])
            upload(0, 1)

            def step_all(i):                                   # noqa: F811
                half = (i + 1) % 2                             # scans of step i live in half (i+1)%2 (step 0 -> half 1)
                eng.fence()
                eng.step(np.arange(B, dtype=np.int32) + half * B)
                upload(i + 1, i % 2)

    def barrier():
        for en in engs:
            en.synchronize()
        if comm is not None:
            comm.barrier()

    s = 0
    for _ in range(args.warmup):
        step_all(s); s += 1
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_all(s); s += 1
    barrier()
    dt = time.perf_counter() - t0
    res = eng.results()
    if comm is not None:
        dt = comm.allreduce_max(dt)                            # max over ranks (RCCL all-reduce, no torch)

    # BASELINE config 5: the only exchange the path has - every rank in turn broadcasts the live keyframe of its lane 0
    # {pose, velocity, undistorted features, polar peaks} from HBM over RCCL (reported, not part of `value`)
    kf_ms, comm_seen = None, None
    if comm is not None:
        comm_seen = comm.info()
        comm.bcast_keyframe(eng, 0, 0)                                         # warm-up (channel setup)
        k0 = time.perf_counter()
        for src in range(world):
            got = comm.bcast_keyframe(eng, src, 0)
            assert got["prunedUndistortedLocals"].shape[1] == 2 and got["peaks"].shape[1] == 2 and got["lane"] == 0
        kf_ms = (time.perf_counter() - k0) * 1e3 / world

    out = None
    if rank == 0:
        pairs = B * args.steps * world
        value = pairs / dt
        out = {
            "metric": "radar scan-pairs/sec (400x3768 polar)", "value": round(value, 2), "unit": "scan-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/f32/f64",
            "data": f"synthetic Oxford-format 400x3779 u8 records; {Dn_} distinct seeded sequences x {T} frames per rank, replicated into {B} lane-private HBM copies, ping-pong replay",
            "config": {"workload": "steady-state scan pair, full hot path (ingest+peaks, warp, pyramid, KLT, max-clique outlier rejection, Kabsch, "
                                   + ("motion-distortion LM" if not args.no_md else "dead reckoning") + ")",
                       "lanes_per_gpu": B, "engines_per_gpu": len(engs), "h2d_streaming": bool(args.h2d), "frames": T,
                       "device": info["name"], "arch": info["arch"], "launcher": "torch.distributed.run env" if "TORCHELASTIC_RUN_ID" in os.environ else ("bench.py --gpus" if world > 1 else "single process"),
                       "collective_backend": None if comm is None else comm.backend,
                       "comm_rank_world_seen": comm_seen,
                       "mean_tracked": round(float(np.mean([r["n_tracked"] for r in res])), 1),
                       "mean_inliers": round(float(np.mean([r["n_inliers"] for r in res])), 1),
                       "mean_lm_nfev": round(float(np.mean([r["lm_nfev"] for r in res])), 1),
                       "keyframe_broadcast_ms": None if kf_ms is None else round(kf_ms, 3)},
            "roofline": None, "cpu_baseline": None,
        }
        if not args.dry_engine:
            out["config"]["initial_features"] = int(np.mean([len(q[2]) for q in seqs]))
            out["config"]["stage_ms_last_step"] = {k: round(v, 4) for k, v in eng.stage_times().items()}
            out["config"]["whole_path_Bmin_GBs_per_gpu"] = round(13.07e6 * (value / world) / 1e9, 3)   # SURVEY 8d B_min per steady pair
            out["roofline"] = roofline(eng, args, B)
            if world == 1:
                out["cpu_baseline"] = cpu_baseline(args, seqs, cyc)
    for en in engs:
        en.close()
    if comm is not None:
        comm.close()
    for c in ctxs:
        c.close()
    if out is not None:
        print(json.dumps(out), flush=True)


def roofline(eng, args, B):
    """roofline of the dominant HBM-streaming kernel.  `avg_launch_ms` is the kernel's average launch duration over the K
    timed steps, from HIP event pairs recorded on the stream it runs on (roam_engine_kernel_avg; no synchronisation inside the
    timed region).  In the pipelined engine other kernels share the GPU during those launches, so the same kernels are also
    re-launched alone after the timed region (roam_engine_time_kernel) and reported as `isolated_*`."""
    names = ("ingest_peaks", "warp_quantise", "pyramid")
    live = {k: eng.kernel_avg(k, args.steps)[0] for k in names}
    iso = {k: eng.time_kernel(k, args.kernel_reps) for k in names}
    dom = max(iso, key=lambda k: iso[k][0])          # dominant by its own (isolated) cost: in the pipeline the in-step
    # durations of concurrent kernels stretch over each other and say little about which one costs most
    ms, algo_bytes = live[dom], iso[dom][1]
    achieved = algo_bytes / (ms * 1e-3) / 1e9
    # HBM traffic and VALU instruction counts of that kernel from the PMC passes taken AT THIS LANE COUNT
    # (profiles/pmc_run.sh -> profiles/pmc_traffic.py); null when no pass at this lane count is committed
    traffic, valu_frac, src = None, None, None
    kname = {"warp_quantise": "warp_gather_kernel", "ingest_peaks": "peaks_rows_u8_wave_kernel", "pyramid": "pyr_down_wave_kernel"}[dom]
    for cand in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json")), reverse=True):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", cand)))
            if tj.get("lanes") == B and kname in tj["kernels"]:
                k = tj["kernels"][kname]
                traffic = k["traffic_bytes_per_scan"] * B
                if k.get("valu_wave_insts_per_launch"):
                    # issue-bound view: VALU wave-instructions x 4 cycles each over 1024 SIMDs, relative to the launch time alone
                    valu_frac = k["valu_wave_insts_per_launch"] * 4 / (SIMDS * CLOCK_HZ * iso[dom][0] * 1e-3)
                src = cand
                break
        except Exception:
            continue
    iso_frac = algo_bytes / (iso[dom][0] * 1e-3) / 1e9 / HBM_PEAK_GBS
    bound = "valu_issue" if (valu_frac is not None and valu_frac > iso_frac) else "hbm"
    return {"bound": bound, "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": src,
            "valu_issue_frac_isolated": None if valu_frac is None else round(valu_frac, 4),
            "avg_launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": algo_bytes,
            "isolated_achieved": round(iso_frac * HBM_PEAK_GBS, 2), "isolated_frac": round(iso_frac, 5),
            "in_step_kernel_ms": {k: round(v, 4) for k, v in live.items()},
            "isolated_kernel_ms": {k: round(v[0], 4) for k, v in iso.items()},
            "isolated_kernel_GBs": {k: round(v[1] / (v[0] * 1e-3) / 1e9, 1) for k, v in iso.items()}}


def cpu_baseline(args, seqs, cyc):
    """the oracle (CPU restatement, kind "port") on a bounded sample of the same workload: (i) one core - the reference is
    single-threaded Python, this is the like-for-like figure - and (ii) N independent sequences on N processes."""
    if args.cpu_pairs <= 0:
        return None
    import oracle
    recs, poses, feat = seqs[0]
    P = oracle.OdometryPipeline(recs[0], feat, poses[0], motion_distortion=not args.no_md)
    c0 = time.perf_counter()
    for n in range(args.cpu_pairs):
        P.step(recs[cyc[n % len(cyc)]])
    one = args.cpu_pairs / (time.perf_counter() - c0)
    cpu = {"value": round(one, 3), "unit": "scan-pairs/s", "cores": 1, "kind": "port",
           "sample": f"{args.cpu_pairs} consecutive scan pairs of synthetic sequence 0 (same workload, oracle C/numpy restatement, 1 thread)"}
    nproc = args.cpu_procs if args.cpu_procs >= 0 else max(1, (os.cpu_count() or 2) // 2)
    if nproc > 1:
        import multiprocessing as mp
        per = max(8, args.cpu_pairs // 4)
        jobs = [(5000 + i, args.frames, per, not args.no_md) for i in range(nproc)]
        with mp.get_context("spawn").Pool(nproc) as pool:
            w0 = time.perf_counter()
            done = pool.map(cpu_worker, jobs)
            wall = time.perf_counter() - w0
        # rate inside the timed loops (sequence rendering excluded): sum of pairs / longest loop
        cpu["all_cores"] = {"value": round(sum(p for p, _ in done) / max(t for _, t in done), 2), "unit": "scan-pairs/s", "cores": nproc,
                            "logical_cores_of_host": os.cpu_count(), "wall_s_incl_rendering": round(wall, 1),
                            "sample": f"{nproc} independent sequences x {per} pairs, one process each"}
    return cpu


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, argv))
    run_rank(args)


if __name__ == "__main__":
    main()
