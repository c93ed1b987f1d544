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
`--dry-engine` swaps the GPU engine for a stub (CPU test of this launcher; never a measurement).
`--stream` measures ONE sequence instead (BASELINE configs 3 / 4 as the reference runs them): engine-only scan pairs/s of a single
lane fed from a pinned ring, and its per-pair latency when every pose is awaited before the next frame is stepped."""
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
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic sequences generated per rank")
    ap.add_argument("--endless", action="store_true", help="round-2 workload: every lane plays ONE endless ping-pong sequence (no new sequences): the retrack rate then decays with the number of steps played")
    ap.add_argument("--preroll", type=int, default=16, help="untimed steps before the warm-up (not counted in it): every lane goes through a few retrack cycles of its own, so that the retrack rate of the timed steps is the stationary one")
    ap.add_argument("--render-procs", type=int, default=0, help="processes rendering the synthetic sequences (0 = min(distinct, cores / 2))")
    ap.add_argument("--launch-timeout", type=float, default=3600.0, help="--gpus N launcher: wall-clock limit for the whole run, seconds")
    ap.add_argument("--stream", action="store_true", help="single-sequence mode: one lane, pinned ring, result ring (configs 3 / 4)")
    ap.add_argument("--stream-frames", type=int, default=240, help="frames of the --stream sequence")
    ap.add_argument("--png", action="store_true", help="with --stream: also feed the sequence from PNG files (written first, untimed) - decode pool + H2D + step timed")
    ap.add_argument("--png-workers", type=int, default=0, help="host threads inflating PNGs ahead of the engine (0 = min(32, cores / 2))")
    ap.add_argument("--stream-start", type=int, default=280, help="first ground-truth motion of full_seq_1 used by --stream (the vehicle stands still for the first ~230 frames)")
    ap.add_argument("--cpu-pairs", type=int, default=120, help="scan pairs timed on the CPU oracle, 1 core (0 = skip)")
    ap.add_argument("--cpu-procs", type=int, default=-1, help="processes of the N-core CPU leg (-1 = half the logical cores, 0 = skip)")
    ap.add_argument("--no-md", action="store_true", help="motionDistortion OFF (Kabsch dead reckoning)")
    ap.add_argument("--no-retrack", action="store_true", help="round-1 workload: host-seeded features, no re-detection in the timed loop")
    ap.add_argument("--retrack-slots", type=int, default=0, help="lanes whose detection scratch is resident at once (0 = min(lanes, 2048))")
    ap.add_argument("--kernel-reps", type=int, default=10)
    ap.add_argument("--h2d", action="store_true", help="stream every scan from pinned host memory over PCIe (double-buffered pool); reports the PCIe-inclusive rate")
    ap.add_argument("--engines", type=int, default=1, help="independent engine instances (contexts/streams) per GPU; lanes are split between them")
    ap.add_argument("--force-comm", action="store_true", help="create the RCCL communicator even at world size 1 (exercises the collective path on a 1-GPU box)")
    ap.add_argument("--config5", action="store_true", help="BASELINE config 5: ONE sequence per rank (seed 10 + rank) through a 1-lane streaming engine, "
                    "the keyframe exchange (roam_keyframe_exchange: one RCCL all-gather of fixed-size records) after EVERY step inside the loop, every rank "
                    "appending what it receives to its device-resident global map")
    ap.add_argument("--c5-frames", type=int, default=1000, help="scans per sequence of --config5")
    ap.add_argument("--no-segments", action="store_true", help="skip the extra segments of the default run (endless ping-pong steps, single-sequence streaming)")
    ap.add_argument("--dry-engine", action="store_true", help="CPU stub instead of the GPU engine: exercises launcher / rendezvous / reduction only")
    args = ap.parse_args(argv)
    if args.lanes is None:
        args.lanes = 1024 if args.h2d else 4096
    return args


# ------------------------------------------------------------------------------------------------ launcher
def visible_gpu_count():
    """GPUs of this node WITHOUT any HIP call (the launcher must stay GPU-free: its children are fresh processes): KFD topology
    nodes that have SIMDs.  None when the topology is not readable (no KFD: a CPU box)."""
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for d in os.listdir(root):
            try:
                props = dict(l.split()[:2] for l in open(os.path.join(root, d, "properties")) if len(l.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        return n
    except OSError:
        return None


def spawn_ranks(args, argv):
    """--gpus N without a launcher: start N rank processes BEFORE anything touches HIP (fresh children, never an exec of a
    process that initialised the GPU), pass rank / device through the environment, watch ALL of them, relay rank 0's JSON line.
    A rank that exits non-zero (or the wall-clock limit) ends the run within seconds: the others are told through the rendezvous
    directory (their watchdog threads leave even a blocked RCCL collective), then terminated, and the exit code is non-zero."""
    from radarslampy_amd import distributed as D
    if not args.dry_engine:
        have = visible_gpu_count()
        if have is not None and have < args.gpus:
            sys.stderr.write(f"[bench] --gpus {args.gpus} but this node shows {have} GPU(s)\n")
            return 2
    rdv = tempfile.mkdtemp(prefix="roam_rdv_")
    procs, logs = [], []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), ROAM_RDV_DIR=rdv,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        out, err = open(os.path.join(rdv, f"out.{r}"), "w+"), open(os.path.join(rdv, f"err.{r}"), "w+")
        logs.append((out, err))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out, stderr=err, text=True))
    t0, bad = time.monotonic(), None
    while True:
        codes = [p.poll() for p in procs]
        failed = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if failed:
            bad = f"rank(s) {failed} exited with {[codes[r] for r in failed]}"
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() - t0 > args.launch_timeout:
            bad = f"wall-clock limit of {args.launch_timeout:.0f} s reached"
            failed = [r for r, c in enumerate(codes) if c is None]
            break
        time.sleep(0.05)
    if bad:
        for r in failed:
            D.FileRendezvous(rdv, r, args.gpus).mark_failed(bad)            # the survivors' watchdogs see it and leave
        t1 = time.monotonic()
        while any(p.poll() is None for p in procs) and time.monotonic() - t1 < 3.0:
            time.sleep(0.05)
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t1 = time.monotonic()
        while any(p.poll() is None for p in procs) and time.monotonic() - t1 < 3.0:
            time.sleep(0.05)
        for p in procs:
            if p.poll() is None:
                p.kill()
        sys.stderr.write(f"[bench] {bad}; rank logs kept in {rdv}\n")
        for r, (out, err) in enumerate(logs):
            err.seek(0)
            tail = err.read()[-1500:]
            if tail.strip():
                sys.stderr.write(f"---- rank {r} stderr (tail) ----\n{tail}\n")
        return 1
    for r, (out, err) in enumerate(logs):
        err.seek(0)
        sys.stderr.write(err.read())
    logs[0][0].seek(0)
    sys.stdout.write(logs[0][0].read())
    sys.stdout.flush()
    for out, err in logs:
        out.close(); err.close()
    try:                                                     # (rank 0 removes the directory itself when the communicator closes)
        for f in os.listdir(rdv):
            try:
                os.unlink(os.path.join(rdv, f))
            except OSError:
                pass
        os.rmdir(rdv)
    except OSError:
        pass
    return 0


# ------------------------------------------------------------------------------------------------ engines
class DryEngine:
    """no GPU: keeps the launcher / comm / JSON code path testable on CPU (results are placeholders)"""

    class _Cfg:
        peaks_cap = 64

    def __init__(self, lanes, rank):
        self.lanes, self.rank, self.n, self.cfg = lanes, rank, 0, self._Cfg()
        self.remote_map = []

    def remote_map_add(self, kf):
        self.remote_map.append(kf)

    def step(self, idx):
        assert len(idx) == self.lanes
        self.n += 1
        if os.environ.get("ROAM_DRY_FAIL_RANK") == str(self.rank) and self.n == 2:
            raise RuntimeError("injected failure (ROAM_DRY_FAIL_RANK)")        # CPU test of the launcher's failure handling
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


WORK_RETRACK = dict(n_static=460, n_movers=120, scintillation=0.6)
WORK_STEADY = dict(n_static=460, n_movers=24)


def cpu_worker(job):
    """one independent oracle pipeline (N-core leg of the CPU baseline); returns (pairs, seconds)"""
    seed, frames, pairs, md, retrack, new_seq = job
    import oracle
    from radarslampy_amd import synth
    recs, poses, feat = synth.make_sequence(seed, frames, distortion=md, **(WORK_RETRACK if retrack else WORK_STEADY))
    cyc = list(range(1, frames)) + list(range(frames - 2, -1, -1))
    det = (lambda cart: oracle.getFeatures(cart)[0]) if retrack else None
    if retrack:
        feat = oracle.append_dedupe(np.empty((0, 2)), det(oracle.convertPolarImageToCartesian(recs[0][:, 11:11 + 2025].astype(np.float32) / np.float32(255.))))
    P = oracle.OdometryPipeline(recs[0], feat, poses[0], motion_distortion=md, detect=det)
    t0 = time.perf_counter()
    done = 0
    for n in range(pairs):
        k = cyc[n % len(cyc)]
        if retrack and new_seq and k == 0:
            P.blobCoord = np.empty((0, 2), np.float32)         # a new sequence starts on frame 0 (like the GPU lanes): detection, no pair
        else:
            done += 1
        P.step(recs[k])
    return done, time.perf_counter() - t0


def _render_sequence(job):
    seed, frames, md, work = job
    from radarslampy_amd import synth
    return synth.make_sequence(seed, frames, distortion=md, **work)


def render_sequences(seeds, frames, md, work, procs):
    """the rank's distinct synthetic sequences, rendered on host cores in parallel (input generation, before the GPU is touched)"""
    jobs = [(s, frames, md, work) for s in seeds]
    # (default: half the logical cores, shared between the ranks of the node - eight ranks rendering at once must not oversubscribe it)
    world = max(1, int(os.environ.get("WORLD_SIZE", "1")))
    procs = procs if procs > 0 else max(1, min(len(jobs), (os.cpu_count() or 2) // (2 * world)))
    if procs == 1 or len(jobs) == 1:
        return [_render_sequence(j) for j in jobs]
    import multiprocessing as mp
    pool = mp.get_context("spawn").Pool(procs)
    try:
        out = pool.map(_render_sequence, jobs)
        pool.close()                                     # workers leave by themselves: no SIGTERM (a profiler's preloaded signal
        pool.join()                                      # handler in the workers turned Pool.terminate() into a hang)
        return out
    except BaseException:
        pool.terminate()
        raise


def run_rank(args):
    from radarslampy_amd import distributed as D
    rank, local_rank, world = D.rank_env()
    if world != args.gpus:
        sys.stderr.write(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: using the launcher's world size\n")
    rdv = D.FileRendezvous(D.rendezvous_dir(), rank, world) if (world > 1 or args.force_comm) else None
    if rdv is not None and world > 1:
        rdv.watchdog()                                   # another rank's failure ends this process too
    try:
        _run_rank(args, D, rank, local_rank, world, rdv)
    except BaseException as ex:                          # noqa: BLE001
        if rdv is not None:
            rdv.mark_failed(repr(ex))
        raise


def _run_rank(args, D, rank, local_rank, world, rdv):
    B, T, Dn_ = args.lanes, args.frames, max(1, min(args.distinct, args.lanes))
    cyc = list(range(1, T)) + list(range(T - 2, -1, -1))           # ping-pong frame schedule 1,2,..,T-1,T-2,..,0,1,..
    retracks_per_step = []
    stream_recs = None

    if args.dry_engine:
        eng = DryEngine(B, rank)
        engs, ctxs, info = [eng], [], dict(name="dry", arch="none")
        comm = D.FileComm(rdv) if rdv else None
        step_all = lambda i, sequences_end=True: eng.step(np.zeros(B, np.int32)) or 0       # noqa: E731
        seqs = None
    else:
        # workload: reflector world with 120 movers and scan-to-scan scintillation: ~26 % of the tracked correspondences are
        # rejected per pair (the paper reports 28 %) and features run out every 3-5 pairs (real `tiny` scans: every 2-3), so
        # the feature re-detection (DoH + ANMS) is part of the timed loop at the cadence the data dictates
        WORK = WORK_STEADY if args.no_retrack else WORK_RETRACK
        seqs = render_sequences([1000 * rank + 17 * d + 5 for d in range(Dn_)], T, not args.no_md, WORK, args.render_procs)
        if rank == 0 and world == 1 and not (args.no_segments or args.no_retrack or args.h2d or args.no_md):
            stream_recs = stream_render(args, True)                # the --stream segment's records, before the GPU is touched
        from radarslampy_amd import _ffi
        from radarslampy_amd.engine import Engine
        stream_cfg = {}
        if stream_recs is not None:
            # BASELINE configs 3 / 4 on the same clock: ONE sequence through a 1-lane engine (records rendered before the timed region).
            # Measured BEFORE the 4 096-lane engine exists, i.e. as `python bench.py --stream` measures it: a process that has created
            # and closed such an engine (67 GB of detection scratch) enqueues ~8 % slower afterwards (profiles/pageable_effect.py), and
            # a single sequence's rate is its enqueue rate
            sctx = _ffi.Context(local_rank)
            for md_ in (True, False):
                _, cfg_ = stream_measure(stream_recs[0], stream_recs[1], md_, sctx)
                tag = "md_on" if md_ else "md_off"
                stream_cfg[f"stream_pairs_per_s_{tag}"] = cfg_["pipelined_pairs_per_s"]
                stream_cfg[f"stream_pairs_per_s_{tag}_loop_only"] = cfg_["pipelined_pairs_per_s_loop_only"]     # without the once-per-sequence set-up
                stream_cfg[f"stream_ms_per_pair_awaited_{tag}"] = cfg_["latency_ms_per_pair"]["median"]
                stream_cfg[f"stream_ms_retrack_pair_awaited_{tag}"] = cfg_["latency_ms_per_pair"]["median_retrack_pair"]
            stream_cfg["stream_segment"] = (f"{len(stream_recs[0])} frames along full_seq_1's ground-truth motions, 1 lane, pinned ring + result ring, "
                                            "measured before the batch engine is created; = python bench.py --stream [--no-md]")
            sctx.close()
        ctx = _ffi.Context(local_rank)
        info = ctx.device_info()
        comm = None
        if rdv:
            # native RCCL communicator (csrc/comm.hip).  The ranks first agree - through the rendezvous directory, before anyone
            # enters the collective ncclCommInitRank, which has no timeout - that librccl can be bound everywhere; then that the
            # communicator came up everywhere.  Otherwise all of them aggregate the timing through files instead, so that a line is
            # still printed; "collective_backend" says which one ran
            votes = rdv.gather("rccl_available", b"1" if D.RcclComm.available(ctx) else b"0")
            if all(v == b"1" for v in votes):
                try:
                    comm = D.RcclComm(ctx, rdv)
                except Exception as ex:                                     # noqa: BLE001
                    sys.stderr.write(f"[bench] rank {rank}: RCCL communicator failed ({ex}); falling back to the file communicator\n")
                votes = rdv.gather("rccl_ok", b"1" if comm is not None else b"0")
            if any(v != b"1" for v in votes):
                if comm is not None:
                    comm.close_native()
                comm = D.FileComm(rdv)
        E = max(1, args.engines)
        assert B % E == 0
        BE = B // E
        # every lane owns private copies of its T records (device-to-device replicas of the D distinct
        # sequences): identical content, distinct HBM addresses -> input reads are real HBM traffic
        ctxs = [ctx] + [_ffi.Context(local_rank) for _ in range(E - 1)]
        engs = []
        period = 2 * T - 2
        cyc_full = list(range(T)) + list(range(T - 2, 0, -1))          # ping-pong 0,1,..,T-1,T-2,..,1

        for e in range(E):
            en = Engine(BE, BE * T, ctx=ctxs[e], motion_distortion=not args.no_md, retrack_on_device=not args.no_retrack,
                        retrack_slots=args.retrack_slots)
            Dn = min(Dn_, BE)
            for d in range(Dn):
                for t in range(T):
                    en.upload_scan(d * T + t, seqs[d][0][t])      # lanes 0..D-1 hold the originals
            for b in range(Dn, BE):
                for t in range(T):
                    en.copy_scan(b * T + t, (b % Dn) * T + t)
            en.synchronize()
            # replicas of one sequence start at different frames of the cycle
            en.phase = np.array([0 if (args.no_retrack or args.h2d) else (b // Dn) % period for b in range(BE)])
            t0s = [cyc_full[en.phase[b]] for b in range(BE)]
            if args.no_retrack:
                for b in range(BE):
                    en.init_lane(b, b * T, seqs[b % Dn][2], seqs[b % Dn][1][0])
            else:
                # first features detected on the device (DoH + ANMS), all lanes in one pass
                en.init_lanes_detect(0, [b * T + t0s[b] for b in range(BE)], np.array([seqs[b % Dn][1][t0s[b]] for b in range(BE)]))
            engs.append(en)
        eng = engs[0]
        cyc_arr = np.array(cyc_full)

        # A lane is a STREAM OF FINITE SEQUENCES: its T frames forward and backward (2T - 2 scans, one fewer pairs - the reference's
        # own data/tiny is 11 scans), then the next sequence begins on frame 0 with a first-frame detection and no pair
        # (ROAM_STEP_NEW_SEQUENCE).  Lanes are spread over the phases of that cycle, so in every step the same share of them is 0, 1,
        # .. 2T - 3 pairs into its sequence: feature ages, rejection rate and retrack rate are stationary, whatever steps are timed.
        # (A single endless ping-pong is not: the features that survive a few passes are the stable ones, and the retrack rate
        # decays for hundreds of steps.)
        new_seq = 0 if (args.no_retrack or args.endless) else _ffi.STEP_NEW_SEQUENCE

        def step_all(i, sequences_end=True):
            n_new = 0
            for en in engs:
                ph = (en.phase + i + 1) % period
                starts = (ph == 0) & (new_seq != 0) & sequences_end
                en.step(((np.arange(BE) * T + cyc_arr[ph]) | np.where(starts, new_seq, 0)).astype(np.int32))
                n_new += int(np.count_nonzero(starts))
            return n_new

        if args.h2d:
            # PCIe-inclusive mode (f2): one engine, pool = two halves of B slots; lanes of one sequence are contiguous so
            # that D replicated uploads (host stride 0) feed all lanes; upload(i+1) overlaps step(i) on the copy stream
            assert E == 1
            for en in engs:
                en.close()
            eng = Engine(B, 2 * B, ctx=ctx, motion_distortion=not args.no_md, retrack_on_device=not args.no_retrack, retrack_slots=args.retrack_slots)
            engs = [eng]
            per = max(1, B // Dn_)
            # every lane's record has its own pinned source (T x B records, 1.5 MB each): the copy kernel reads host memory
            # through the GPU's caches, so a replicated source would be served from cache instead of crossing PCIe
            pinned = ctx.host_alloc((T, B, 400, 3779))
            lane_seq = [min(b // per, Dn_ - 1) for b in range(B)]
            for t in range(T):
                for b in range(B):
                    pinned[t, b] = seqs[lane_seq[b]][0][t]
            cyc = cyc_full[1:] + cyc_full[:1]

            def upload(step, half):
                t = 0 if step < 0 else cyc[step % len(cyc)]
                eng.upload_scans_async(half * B, pinned[t], n=B)

            upload(-1, 0)
            eng.synchronize()
            if args.no_retrack:
                for b in range(B):
                    eng.init_lane(b, b, seqs[lane_seq[b]][2], seqs[lane_seq[b]][1][0])
            else:
                eng.init_lanes_detect(0, np.arange(B), np.array([seqs[lane_seq[b]][1][0] for b in range(B)]))
            upload(0, 1)

            def step_all(i, sequences_end=True):               # noqa: F811
                half = (i + 1) % 2                             # scans of step i live in half (i+1)%2 (step 0 -> half 1)
                eng.fence()
                eng.step(np.arange(B, dtype=np.int32) + half * B)
                upload(i + 1, i % 2)
                return 0

    def barrier():
        for en in engs:
            en.synchronize()
        if comm is not None:
            comm.barrier()

    # per-step records are consumed INSIDE the timed loop, two steps behind the enqueue front (result ring: waiting for step
    # s - 2 does not drain steps s - 1 and s): every pose of every lane is read, retracks are counted as they happen
    stat = dict(steps=0, retracks=0, tracked=0, good=0, inliers=0, overflow=0, unproven=0)

    def consume(step):
        if args.dry_engine or step < 0:
            return
        tot = 0
        for en in engs:
            r = en.results_array(step)
            stat["steps"] += 1
            nrt = int(np.count_nonzero(r["flags"] & 8))
            stat["retracks"] += nrt
            tot += nrt
            if en is engs[0]:
                stat.setdefault("per_step", {})[step] = nrt           # lanes of engine 0 that re-detected in this step
            stat["overflow"] += int(np.count_nonzero((r["flags"] >> 8) & 15))
            # result flag bit 0 = the maximum clique is PROVEN maximum (clique.hip: the search stayed below its node limit; an empty
            # problem counts as proven): anything else would be less work than the reference does inside the timed region
            stat["unproven"] += int(np.count_nonzero((r["flags"] & 1) == 0))
            stat["tracked"] += int(r["n_tracked"].sum()); stat["good"] += int(r["n_good"].sum()); stat["inliers"] += int(r["n_inliers"].sum())
        retracks_per_step.append(tot)

    s = 0
    pre = 0 if (args.dry_engine or args.no_retrack) else max(0, args.preroll)
    for _ in range(pre + args.warmup):
        step_all(s); s += 1
    barrier()
    t0 = time.perf_counter()
    first_frames = 0                                           # lanes x steps that opened a new sequence (a scan, but no pair)
    for k in range(args.steps):
        first_frames += step_all(s); s += 1
        if k >= 2:
            consume(s - 3)
    barrier()
    dt = time.perf_counter() - t0
    for k in range(max(0, args.steps - 2), args.steps):
        consume(pre + args.warmup + k)
    res = eng.results()
    # in-step kernel durations of the K timed steps (HIP event pairs recorded by the engine on the kernels' own streams), taken NOW:
    # the steady / forced segments below enqueue more steps
    live = None
    if not args.dry_engine:
        knames = ["ingest_peaks", "warp_quantise", "pyramid"] + ([] if args.no_retrack else ["doh_integral", "doh_det_maxima"])
        live = {k: eng.kernel_avg(k, args.steps)[0] for k in knames[:3]}
        if not args.no_retrack:
            # detection kernels: EVERY busy chunk launch of the timed steps.  A step launches ceil(lanes / retrack_slots) chunks whatever
            # the number n of lanes that re-detect (only the device knows it); chunk c holds clamp(n - c * slots, 0, slots) detections,
            # and n is in the result records consumed above.  live[k] = mean duration of the busy launches, their detections on average
            # = doh_units_per_launch (the first chunk of a step overlaps the front end of later steps, the others mostly run alone)
            slots_ = eng.detect_chunk()                        # detections per launch: retrack_slots, or 1 024 in the two-stream form (engine)
            nst = min(args.steps, 64)
            n_step = [stat.get("per_step", {}).get(pre + args.warmup + k, 0) for k in range(args.steps - nst, args.steps)]
            # (chunks of fewer than 200 detections take the two-pass integral kernels - three times the traffic of the one-sweep kernel
            # whose algorithmic bytes the roofline uses - so they are left out of the live average of BOTH kernels)
            # Two-stream form (engines of >= 2 048 lanes: detect_chunk() < retrack_slots): the determinants of chunk c run on a second stream
            # BESIDE the integral images of chunk c + 1, so a launch's duration there is that of two kernels sharing the GPU.  The live
            # figure of a kernel is then taken from its launches that have no detection kernel beside them - the integral images of a
            # step's FIRST chunk, the determinants of its LAST busy chunk (as every launch was until late round 6) - and the average over
            # ALL busy launches is reported next to it (roofline.all_launches).
            two_stream = slots_ < min(B // len(engs), args.retrack_slots or 2048)
            units_tot = 0
            all_l = {}
            for k in knames[3:]:
                m = eng.kernel_chunk_ms(k, nst)
                tot, nb, units_tot = 0.0, 0, 0
                tot_a, nb_a, units_a = 0.0, 0, 0
                for srow, n_ in zip(m[-len(n_step):], n_step[-len(m):]):
                    busy = min(len(srow), -(-n_ // slots_))
                    for c_ in range(busy):
                        held = min(slots_, n_ - c_ * slots_)
                        if held >= 200:
                            tot_a += float(srow[c_]); nb_a += 1; units_a += held
                            alone = (not two_stream) or (c_ == 0 if k == "doh_integral" else c_ == busy - 1)
                            if alone:
                                tot += float(srow[c_]); nb += 1; units_tot += held
                live[k] = tot / nb if nb else 0.0
                live.setdefault("doh_units_per_launch", {})[k] = float(units_tot) / nb if nb else 0.0      # (per kernel: the two-stream form's "alone" launches differ)
                if k == "doh_integral":
                    live["doh_busy_launches"] = nb
                if two_stream and nb_a:
                    all_l[k] = {"avg_launch_ms": round(tot_a / nb_a, 4), "units_per_launch": round(units_a / nb_a, 1), "launches": nb_a}
            if all_l:
                live["doh_all_launches"] = all_l
    if comm is not None:
        dt = comm.allreduce_max(dt)                            # max over ranks (RCCL all-reduce, no torch)

    # BASELINE config 5: the only exchange the path has - every rank in turn broadcasts the live keyframe of its lane 0
    # {pose, velocity, undistorted features, polar peaks} from HBM over RCCL (reported, not part of `value`); every rank
    # appends what it receives to its device-resident global map (Mapping.Map.addKeyframe, roam_remote_map_*)
    kf_ms, comm_seen, map_seen = None, None, None
    if comm is not None:
        comm_seen = comm.info()
        if not args.dry_engine:
            eng.remote_map_reserve(max(8, 2 * world))
        comm.bcast_keyframe(eng, 0, 0)                                         # warm-up (channel setup)
        k0 = time.perf_counter()
        for src in range(world):
            got = comm.bcast_keyframe(eng, src, 0)
            assert got["prunedUndistortedLocals"].shape[1] == 2 and got["peaks"].shape[1] == 2 and got["lane"] == 0
        kf_ms = (time.perf_counter() - k0) * 1e3 / world
        if args.dry_engine:
            map_seen = [len(eng.remote_map), sorted({kf["root"] for kf in eng.remote_map})]
        else:
            n_rec, n_res = eng.remote_map_count()
            last = eng.remote_map_get(n_res - 1)
            assert last["root"] == world - 1 and np.array_equal(last["pose"], got["pose"]) and np.array_equal(last["peaks"], got["peaks"])
            map_seen = [n_rec, sorted({eng.remote_map_get(i)["root"] for i in range(n_res)})]

    # steady vs retrack throughput beside the mix (after the timed region; single numbers of this rank)
    extra = {}
    if not args.dry_engine and not args.no_retrack and not args.h2d:
        stage_mix = eng.stage_times()
        for en in engs:
            en.set_retrack(2)                                   # every lane re-detects: the cost of a retrack pair
        step_all(s, False); s += 1
        for en in engs:
            en.synchronize()
        st_forced = eng.stage_times()
        extra["retrack_stage_ms_all_lanes"] = round(st_forced["retrack"], 3)
        extra["retrack_us_per_lane"] = round(st_forced["retrack"] * 1e3 / (B // len(engs)), 2)
        # the steady pair: re-detection suspended after every lane has just re-detected; one untimed step lets the feature sets
        # decay to the mix's level (~300 -> ~170 per lane), the next two are timed (`steady_mean_tracked`: features per lane in them)
        for en in engs:
            en.set_retrack(0)
        for _ in range(1):
            step_all(s, False); s += 1
        barrier()
        for en in engs:
            en.results_array()                                  # latest records: the engine shrinks its launch width back to what the lanes hold
        k2, trk = 2, []
        t1 = time.perf_counter()
        for _ in range(k2):
            step_all(s, False); s += 1
        for en in engs:
            en.synchronize()
        extra["steady_pairs_per_s"] = round(B * k2 / (time.perf_counter() - t1), 1)
        for q in range(k2):
            trk += [en.results_array(en.steps_enqueued() - 1 - q)["n_tracked"].mean() for en in engs]
        extra["steady_mean_tracked"] = round(float(np.mean(trk)), 1)
        steady_ms = B / extra["steady_pairs_per_s"] * 1e3
        extra["retrack_pairs_per_s"] = round(B / ((steady_ms + st_forced["retrack"] * len(engs)) * 1e-3), 1)
        for en in engs:
            en.set_retrack(1)
        extra["stage_ms_last_mix_step"] = {k: round(v, 4) for k, v in stage_mix.items()}
        if not args.no_segments and not args.endless:
            # round 2's workload on the same clock: every lane keeps playing its ping-pong sequence, no new sequences (the retrack
            # rate then decays with the steps played: ENDLESS_STEPS steps after ENDLESS_WARM untimed ones from the mix's state)
            for _ in range(ENDLESS_WARM):
                step_all(s, False); s += 1
            barrier()
            t2 = time.perf_counter()
            for _ in range(ENDLESS_STEPS):
                step_all(s, False); s += 1
            barrier()
            dt2 = time.perf_counter() - t2
            nrt = 0
            for en in engs:
                for q in range(min(ENDLESS_STEPS, 8)):
                    nrt += int(np.count_nonzero(en.results_array(en.steps_enqueued() - 1 - q)["flags"] & 8))
            extra["endless_pairs_per_s"] = round(B * ENDLESS_STEPS / dt2, 1)
            extra["endless_retrack_fraction_last_steps"] = round(nrt / (B * min(ENDLESS_STEPS, 8)), 4)
            extra["endless_segment"] = f"{ENDLESS_STEPS} timed steps after {ENDLESS_WARM} untimed ones, no new sequences (round 2's workload shape), same lanes"

    out = None
    if rank == 0:
        # scan PAIRS: a lane's step that opens a new sequence consumes a scan (ingest, warp, pyramid, peaks, first-frame detection)
        # but yields no pair - it is not counted (the same share on every rank: phases are assigned alike)
        pairs = (B * args.steps - first_frames) * world
        value = pairs / dt
        rps = retracks_per_step[-args.steps:] if retracks_per_step else []
        out = {
            "metric": "radar scan-pairs/sec (400x3768 polar)", "value": round(value, 2), "unit": "scan-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/f32/f64",
            "data": f"synthetic Oxford-format 400x3779 u8 records; {Dn_} distinct seeded sequences x {T} frames per rank (460 static reflectors + "
                    + ("24 movers" if args.no_retrack else "120 movers, scan-to-scan scintillation 0.6") + f"), replicated into {B} lane-private HBM copies, ping-pong replay with staggered phases"
                    + ("" if (args.no_retrack or args.endless) else f"; every lane is a stream of finite sequences ({2 * T - 2} scans = {2 * T - 3} pairs each, first frame detected on the device), lanes spread evenly over the phases"),
            "config": {"workload": "scan pair, full hot path (ingest+peaks, warp, pyramid, KLT, max-clique outlier rejection, Kabsch, "
                                   + ("motion-distortion LM" if not args.no_md else "dead reckoning")
                                   + (", keyframe bookkeeping; features host-seeded, no re-detection)" if args.no_retrack else
                                      ", keyframe bookkeeping, DoH + ANMS re-detection on the device whenever a lane runs out of features; every lane's pose read back every step)"),
                       "workload_note": None if (args.no_retrack or args.endless or args.dry_engine) else
                                        "stationary by construction (streams of finite sequences): 46 % of the lanes re-detect in EVERY step and 29 % of the "
                                        "correspondences are rejected - harder than round 2's endless ping-pong, whose retrack rate decays with the steps played "
                                        "(0.28 in the window the driver timed).  That workload is still here: `--endless --preroll 0 --distinct 4` = 67.7 k scan-pairs/s "
                                        "on this build against round 2's 53.2 k (profiles/r03_bench_round2_workload_20_5.json)",
                       "lanes_per_gpu": B, "engines_per_gpu": len(engs), "preroll_steps": pre,
                       "scans_per_step": B, "first_frame_scans_in_timed_steps": first_frames, "pairs_counted": (B * args.steps - first_frames), "h2d_streaming": bool(args.h2d), "frames": T, "distinct_sequences": Dn_,
                       "device": info["name"], "arch": info["arch"], "launcher": "torch.distributed.run env" if "TORCHELASTIC_RUN_ID" in os.environ else ("bench.py --gpus" if world > 1 else "single process"),
                       "collective_backend": None if comm is None else comm.backend,
                       "comm_rank_world_seen": comm_seen, "global_map_keyframes_and_senders": map_seen,
                       "mean_tracked": round(stat["tracked"] / max(1, stat["steps"] * (B // max(1, len(engs)))), 1) if stat["steps"] else round(float(np.mean([r["n_tracked"] for r in res])), 1),
                       "mean_inliers": round(stat["inliers"] / max(1, stat["steps"] * (B // max(1, len(engs)))), 1) if stat["steps"] else round(float(np.mean([r["n_inliers"] for r in res])), 1),
                       "rejected_fraction": round(1.0 - stat["inliers"] / max(1, stat["good"]), 4) if stat["steps"] else None,
                       "retrack_fraction": round(stat["retracks"] / max(1, stat["steps"] * (B // max(1, len(engs)))), 4) if stat["steps"] else None,
                       "retracks_per_step": None if not rps else {"min": int(min(rps)), "mean": round(float(np.mean(rps)), 1), "max": int(max(rps))},
                       "detect_overflows": stat["overflow"],
                       "clique_unproven": stat["unproven"],
                       "mean_lm_nfev": round(float(np.mean([r["lm_nfev"] for r in res])), 1),
                       "keyframe_broadcast_ms": None if kf_ms is None else round(kf_ms, 3)},
            "roofline": None, "cpu_baseline": None,
        }
        if not args.dry_engine:
            out["config"].update(extra)
            if not extra:
                out["config"]["stage_ms_last_step"] = {k: round(v, 4) for k, v in eng.stage_times().items()}
            out["config"]["whole_path_Bmin_GBs_per_gpu"] = round(13.07e6 * (value / world) / 1e9, 3)   # SURVEY 8d B_min per steady pair ONLY
            # steady bytes of every scan + the strict bytes of every re-detection (polar payload + f64 integral image written, then read
            # once: 66.4 MB) over the step time: what the whole step moves algorithmically
            out["roofline"] = roofline(eng, args, B // max(1, len(engs)), out["config"].get("retrack_fraction") or 0.0, live)
            if rps and "algorithmic_bytes_per_detection" in out["roofline"]:
                # (round 6: the integral image counts with the part of it that exists - the tiles the determinant kernel reads, 87.7 % of a
                # 2024 x 2024 image - written once and read once; rounds 2-5 counted the whole image twice: 66.4 MB)
                det_bytes = out["roofline"]["algorithmic_bytes_per_detection"]
                out["config"]["whole_step_algorithmic_GBs"] = round((13.07e6 * B + det_bytes * float(np.mean(rps))) / (dt / args.steps) / 1e9, 1)
            if world == 1:
                out["cpu_baseline"] = cpu_baseline(args, seqs, cyc)
    for en in engs:
        en.close()
    if comm is not None:
        comm.close()
    for c in ctxs:
        c.close()
    if out is not None and stream_recs is not None:
        out["config"].update(stream_cfg)
    if out is not None:
        print(json.dumps(out), flush=True)


ENDLESS_WARM, ENDLESS_STEPS = 5, 20
HBM_ACHIEVABLE_GBS = 6300.0   # MI355X_MICROARCH.md: what streaming kernels reach


def _sha16(path):
    import hashlib
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def pmc_record(kname):
    """the committed PMC measurement of a kernel (profiles/r*_pmc_traffic.json, written by profiles/pmc_traffic.py on the GPU box) -
    used only while the kernel's SOURCE is the one that was measured (the file records the source's hash); else None"""
    for cand in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json")), reverse=True):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", cand)))
            k = tj["kernels"].get(kname)
            if not k:
                continue
            src = os.path.join(ROOT, "radarslampy_amd", "csrc", k.get("source_file", ""))
            if k.get("source_fingerprint"):                      # round 4: translation unit + headers + build flags, hashed when MEASURED
                from radarslampy_amd import build as _build
                fresh = os.path.isfile(src) and _build.fingerprint([k["source_file"]]) == k["source_fingerprint"]
            else:
                fresh = bool(k.get("source_sha16")) and os.path.isfile(src) and _sha16(src) == k["source_sha16"]
            if not fresh:
                return None, cand + " (stale: the kernel's source, a header or the build flags changed since the PMC passes)"
            return dict(k, units_per_launch=tj.get("units_per_launch", tj.get("lanes"))), cand
        except Exception:                                              # noqa: BLE001
            continue
    return None, None


def roofline(eng, args, B, retrack_fraction, live_all):
    """(B = the lanes of ONE engine - `eng`, engine 0: its isolated re-launches, its live event timings and its per-step detection
    counts all refer to that engine, also with --engines > 1)
    roofline of the dominant HBM-streaming kernel of a step.  Candidates: the three front-end kernels (once per lane and
    step) and the two image-scale kernels of the feature re-detection (once per RETRACKING lane: weighted by the observed
    retrack fraction).  `avg_launch_ms` is the kernel's average launch duration over the K timed steps from HIP event pairs the
    engine records on the stream the kernel runs on (roam_engine_kernel_avg / roam_engine_kernel_chunk_ms, read right after the timed
    region; nothing synchronises inside it) - for a detection kernel every busy chunk launch of those steps, whose algorithmic bytes
    are those of the detections they really held (`units_per_launch`, from the per-step result records).  In the pipelined engine other kernels
    share the GPU during those launches, so every candidate is also re-launched alone after the timed region
    (roam_engine_time_kernel, HIP events on its stream) = `isolated_*`.
    `traffic` = memory-side bytes per launch from the committed PMC passes of that kernel, gfx950-corrected as the microarchitecture
    guide prescribes (FETCH_SIZE tallies 128-byte requests at 64 bytes: x 2; cross-checked with TCC_MISS x 128 B), scaled to the
    detections per launch; null when the kernel's source changed after the passes were taken."""
    names = ["ingest_peaks", "warp_quantise", "pyramid"]
    live = {k: v for k, v in live_all.items() if k not in ("doh_units_per_launch", "doh_busy_launches", "doh_all_launches")}
    iso = {k: eng.time_kernel(k, args.kernel_reps) for k in names}
    per_step = {k: iso[k][0] for k in names}                       # ms of the kernel alone per step
    slots = None
    if not args.no_retrack:
        slots = min(B, args.retrack_slots or 2048)
        for k in ("doh_integral", "doh_det_maxima"):
            iso[k] = eng.time_kernel(k, max(1, args.kernel_reps // 3))      # one launch = `slots` detections
            per_step[k] = iso[k][0] / slots * retrack_fraction * B
            names.append(k)
    dom = max(per_step, key=per_step.get)
    # (the two detection kernels are within a few per cent of each other and trade places from box to box: the integral image is the
    # one reported, as in rounds 3-6, unless the determinants are clearly the larger)
    if dom == "doh_det_maxima" and per_step.get("doh_integral", 0.0) >= 0.9 * per_step[dom]:
        dom = "doh_integral"
    algo_bytes = iso[dom][1]
    units = slots if dom.startswith("doh") else B
    # in-step duration over the K timed steps.  Front-end kernels: one launch per step over all lanes.  Detection kernels: every
    # busy chunk launch of those steps - the algorithmic bytes of the live figure are those of the average number of detections per
    # such launch
    ms = live.get(dom, iso[dom][0])
    if dom.startswith("doh") and (live_all.get("doh_units_per_launch") or {}).get(dom, 0) > 0 and ms > 0:
        units = live_all["doh_units_per_launch"][dom]
        algo_bytes = iso[dom][1] / slots * units
    else:
        ms = live.get(dom, iso[dom][0]) if not dom.startswith("doh") else iso[dom][0]
    achieved = algo_bytes / (ms * 1e-3) / 1e9
    kname = {"warp_quantise": "warp_gather_kernel", "ingest_peaks": "peaks_rows_u8_wave_kernel", "pyramid": "pyr_down_wave_kernel",
             "doh_integral": "rt_integral_kernel", "doh_det_maxima": "rt_det_strip_kernel"}[dom]
    rec, src = pmc_record(kname)
    traffic = valu_frac = lds_frac = mem_frac = None
    detail = None
    if rec:
        scale = units / float(rec["units_per_launch"])
        traffic = int(rec["hbm_bytes_per_launch_corrected"] * scale)
        iso_s = iso[dom][0] * 1e-3
        mem_frac = rec["hbm_bytes_per_launch_corrected"] / iso_s / 1e9 / HBM_ACHIEVABLE_GBS
        if rec.get("valu_wave_insts_per_launch"):
            valu_frac = rec["valu_wave_insts_per_launch"] * 4 / (SIMDS * CLOCK_HZ * iso_s)       # 4 issue cycles per wave64 instruction
        if rec.get("lds_active_cycles_per_launch"):
            lds_frac = rec["lds_active_cycles_per_launch"] / (256 * CLOCK_HZ * iso_s)
        detail = rec.get("note")
    iso_frac = iso[dom][1] / (iso[dom][0] * 1e-3) / 1e9 / HBM_PEAK_GBS
    # the sampling-map words the integral kernel also reads (4 B per pixel, the same table for every detection, mostly L2 hits):
    # shown, never part of `achieved`
    W_ = 2 * (eng.cfg.clip // 2)
    map_bytes = int(units * W_ * W_ * 4) if dom == "doh_integral" else 0
    return {"bound": "hbm", "bound_detail": detail, "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": src,
            "traffic_over_algorithmic": None if not traffic else round(traffic / algo_bytes, 3),
            "algorithmic_bytes_strict": algo_bytes, "algorithmic_bytes_incl_shared_map": algo_bytes + map_bytes,
            "frac_incl_shared_map": round((algo_bytes + map_bytes) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "hbm_frac_measured": None if not traffic else round(traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "isolated_busy_fractions": {"hbm_of_6.3TBs_corrected_traffic": None if mem_frac is None else round(mem_frac, 3),
                                        "valu_issue": None if valu_frac is None else round(valu_frac, 3),
                                        "lds_array": None if lds_frac is None else round(lds_frac, 3)},
            "avg_launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": algo_bytes,
            **({"algorithmic_bytes_per_detection": round((iso["doh_integral"][1] + iso["doh_det_maxima"][1]) / slots, 1),
                "frac_whole_image_bytes_as_rounds_2_to_5": round(units * (eng.cfg.rows * eng.cfg.clip * (dom == "doh_integral") + 8.0 * W_ * W_) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
               if slots and dom.startswith("doh") else {}),
            "units_per_launch": round(units, 1),
            **({"all_launches": {"note": "two-stream form: the determinants of a chunk of 1 024 detections run beside the next chunk's integral images; "
                                         "achieved / frac above are this kernel's launches with no detection kernel beside them (a step's first chunk), here every busy launch",
                                 **{k: {**v, "frac": round(iso[k][1] / slots * v["units_per_launch"] / (v["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
                                    for k, v in live_all["doh_all_launches"].items()}}}
               if live_all.get("doh_all_launches") and slots else {}),
            "isolated_achieved": round(iso_frac * HBM_PEAK_GBS, 2), "isolated_frac": round(iso_frac, 5),
            "kernel_ms_per_step_alone": {k: round(v, 4) for k, v in per_step.items()},
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
    det = None if args.no_retrack else (lambda cart: oracle.getFeatures(cart)[0])
    if det is not None:                                            # like the GPU lanes: first features detected, not given
        feat = oracle.append_dedupe(np.empty((0, 2)), det(oracle.convertPolarImageToCartesian(recs[0][:, 11:11 + 2025].astype(np.float32) / np.float32(255.))))
    P = oracle.OdometryPipeline(recs[0], feat, poses[0], motion_distortion=not args.no_md, detect=det)
    c0 = time.perf_counter()
    done = 0
    for n in range(args.cpu_pairs):
        k = cyc[n % len(cyc)]
        if det is not None and not args.endless and k == 0:
            P.blobCoord = np.empty((0, 2), np.float32)         # a new sequence starts on frame 0 (like the GPU lanes): detection, no pair
        else:
            done += 1
        P.step(recs[k])
    one = done / (time.perf_counter() - c0)
    cpu = {"value": round(one, 3), "unit": "scan-pairs/s", "cores": 1, "kind": "port",
           "sample": f"{args.cpu_pairs} consecutive scans of synthetic sequence 0 as a stream of finite sequences ({done} pairs; same workload incl. first-frame detections and re-detections, oracle C/numpy restatement, 1 thread)"}
    nproc = args.cpu_procs if args.cpu_procs >= 0 else max(1, (os.cpu_count() or 2) // 2)
    if nproc > 1:
        import multiprocessing as mp
        per = max(8, args.cpu_pairs // 4)
        jobs = [(5000 + i, args.frames, per, not args.no_md, not args.no_retrack, not args.endless) for i in range(nproc)]
        pool = mp.get_context("spawn").Pool(nproc)
        try:
            w0 = time.perf_counter()
            done = pool.map(cpu_worker, jobs)
            wall = time.perf_counter() - w0
            pool.close()
            pool.join()
        except BaseException:
            pool.terminate()
            raise
        # rate inside the timed loops (sequence rendering excluded): sum of pairs / longest loop
        cpu["all_cores"] = {"value": round(sum(p for p, _ in done) / max(t for _, t in done), 2), "unit": "scan-pairs/s", "cores": nproc,
                            "logical_cores_of_host": os.cpu_count(), "wall_s_incl_rendering": round(wall, 1),
                            "sample": f"{nproc} independent sequences x {per} pairs, one process each"}
    return cpu


def stream_render(args, md):
    """records + ground-truth poses of the --stream sequence (rendered by a pool of host processes, before the GPU is touched)"""
    from radarslampy_amd import synth
    n = args.stream_frames
    gold = os.path.join(ROOT, "tests", "golden", "full_seq_1_gt_deltas.npz")
    deltas = np.load(gold)["deltas"][args.stream_start:args.stream_start + n - 1]
    poses = synth.poses_from_deltas(deltas)
    jobs = synth.stream_jobs(synth.StreamWorld(11, mover_fraction=0.15), poses, distortion=md, scintillation=0.4)
    import multiprocessing as mp
    pool = mp.get_context("spawn").Pool(max(1, min(48, (os.cpu_count() or 2) // 2)))
    try:
        recs = pool.map(synth._render_job, jobs, chunksize=4)
        pool.close()
        pool.join()
    except BaseException:
        pool.terminate()
        raise
    return recs, poses


def stream_measure(recs, poses, md, ctx):
    """one sequence through a 1-lane engine: (i) pipelined, (ii) every pose awaited before the next frame is stepped"""
    from radarslampy_amd.RawROAMSystem import stream_records
    n = len(recs)
    flags = {"rejectOutliers": True, "correctMotionDistortion": md}
    stream_records(iter(recs[:12]), 12, poses[0], flags, ctx)                   # warm-up (allocations, first launches)
    t0 = time.perf_counter()
    tm = {}
    est, log = stream_records(iter(recs), n, poses[0], flags, ctx, timing=tm)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    est2, log2, lat = stream_records(iter(recs), n, poses[0], flags, ctx, synchronous=True)
    dt2 = time.perf_counter() - t1
    assert est.tobytes() == est2.tobytes()
    lat = np.array(lat) * 1e3
    rt = np.array([e["retrack"] for e in log], bool)
    err = np.hypot(*(est[:, :2] - poses[1:, :2]).T)
    return dt, {"frames": n,
                "pipelined_pairs_per_s": round((n - 1) / dt, 2),
                # the same run without what a sequence of any length pays once (engine creation, first uploads, the first frame's detection,
                # tear-down): full_seq_1 has 8 866 frames, this segment 240
                "pipelined_pairs_per_s_loop_only": round((n - 1) / tm["loop_s"], 2), "setup_and_teardown_ms": round((dt - tm["loop_s"]) * 1e3, 2),
                "synchronous_pairs_per_s": round((n - 1) / dt2, 2),
                "latency_ms_per_pair": {"median": round(float(np.median(lat)), 3), "p95": round(float(np.percentile(lat, 95)), 3), "max": round(float(lat.max()), 3),
                                        "median_steady_pair": round(float(np.median(lat[~rt])), 3) if (~rt).any() else None,
                                        "median_retrack_pair": round(float(np.median(lat[rt])), 3) if rt.any() else None},
                "retrack_fraction": round(float(rt.mean()), 4), "keyframes": int(sum(e["new_keyframe"] for e in log)),
                "position_rmse_m": round(float(np.sqrt(np.mean(err ** 2))), 3), "distance_m": round(float(np.hypot(*np.diff(poses[:, :2], axis=0).T).sum()), 1)}


def stream_png_measure(recs, poses, md, ctx, workers):
    """8f-f2 on the clock: the same sequence from PNG FILES - zlib inflate + un-filter on the library's pool of host threads
    (parseData.NativeRecordReader -> roam_png_pool_*) straight into a ring of pinned slots, H2D from those slots, step.  The files are
    written first (untimed; tmpfs when there is one).  Beside it, for the record: round 5's pool of Pillow processes with its
    shared-memory ring, and one Pillow decode on the feeding thread (round 4)."""
    import shutil
    import tempfile
    from PIL import Image
    from radarslampy_amd.RawROAMSystem import RING, default_decode_workers, stream_records
    from radarslampy_amd.parseData import NativeRecordReader, RecordDecodePool, prefetchRadarRecords, readRadarRecord
    n = len(recs)
    flags = {"rejectOutliers": True, "correctMotionDistortion": md}
    w = workers if workers > 0 else default_decode_workers()
    d = tempfile.mkdtemp(prefix="roam_png_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        paths = []
        from radarslampy_amd.synth import png_bytes_gray8
        for i, r in enumerate(recs):                                 # files like the data set's: Sub filter, zlib level 1 / Z_RLE, 8 KB IDAT chunks
            paths.append(os.path.join(d, f"{i:06d}.png"))
            with open(paths[-1], "wb") as f:
                f.write(png_bytes_gray8(r))
        size = float(np.mean([os.path.getsize(p) for p in paths]))
        assert np.array_equal(readRadarRecord(paths[0]), recs[0]) and np.array_equal(np.array(Image.open(paths[1])), recs[1])
        t0 = time.perf_counter()
        for p in paths[:24]:
            readRadarRecord(p)
        one = (time.perf_counter() - t0) / 24
        with NativeRecordReader(w, ctx=ctx, hold=RING) as rd:
            for _ in rd.records(paths[:64]):
                pass
            t0 = time.perf_counter()
            k = sum(1 for _ in rd.records(paths))
            dec = time.perf_counter() - t0
            assert k == n
            stream_records(rd.records(paths[:12]), 12, poses[0], flags, ctx, records_pinned=True)       # warm-up
            rd.wait_s = 0.0
            t0 = time.perf_counter()
            est, _ = stream_records(rd.records(paths), n, poses[0], flags, ctx, records_pinned=True)
            dt = time.perf_counter() - t0
            blocked = rd.wait_s
        with RecordDecodePool(w) as pool:                                                             # (start-up of the processes: untimed)
            for _ in pool.records(paths[:64]):
                pass
            t0 = time.perf_counter()
            est_p, _ = stream_records(pool.records(paths), n, poses[0], flags, ctx)
            dtp = time.perf_counter() - t0
        assert est_p.tobytes() == est.tobytes()
        t0 = time.perf_counter()
        est1, _ = stream_records(prefetchRadarRecords(paths, 1), n, poses[0], flags, ctx)            # the decode on the feeding thread (round 4)
        dt1 = time.perf_counter() - t0
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return est, est1, {"png_inclusive_pairs_per_s": round((n - 1) / dt, 2), "png_inclusive_pairs_per_s_pillow_process_pool": round((n - 1) / dtp, 2),
                       "png_inclusive_pairs_per_s_one_decode_thread": round((n - 1) / dt1, 2),
                       "png_decode_only_frames_per_s": round(n / dec, 1), "png_consumer_blocked_on_decode_s": round(blocked, 4), "png_run_s": round(dt, 4),
                       "png_decode_ms_per_frame_one_thread": round(one * 1e3, 2),
                       "png_decode_threads": w, "png_mean_file_bytes": int(size),
                       "png_format": "8-bit grey, Sub filter on every line, zlib level 1 Z_RLE, 8 KB IDAT chunks (as the reference's data/tiny files)"}


def run_stream(args):
    """BASELINE configs 3 / 4 as the reference runs them: ONE sequence (full_seq_1-like motion, unbounded reflector world with
    movers, scintillation and, with motion distortion on, intra-scan distortion) through a 1-lane engine - frames from a pinned
    ring on the copy stream, poses through the result ring.  (i) pipelined: steps enqueued as fast as the rings allow = engine-only
    scan pairs/s of one sequence; (ii) latency: every pose awaited before the next frame is stepped."""
    from radarslampy_amd import _ffi
    n = args.stream_frames
    md = not args.no_md
    recs, poses = stream_render(args, md)
    ctx = _ffi.Context(0)
    info = ctx.device_info()
    dt, cfg = stream_measure(recs, poses, md, ctx)
    if args.png:
        est_png, est_png1, pc = stream_png_measure(recs, poses, md, ctx, args.png_workers)
        cfg.update(pc)
        cfg["png_within_of_in_memory_rate"] = round(pc["png_inclusive_pairs_per_s"] / cfg["pipelined_pairs_per_s"], 3)
    ctx.close()
    out = {"metric": "radar scan-pairs/sec (400x3768 polar), ONE sequence", "value": round((n - 1) / dt, 2), "unit": "scan-pairs/s",
           "n_gpus": 1, "steps": n - 1, "warmup": 11, "ms_per_step": round(dt / (n - 1) * 1e3, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "u8/f32/f64",
           "data": f"synthetic Oxford-format records along ground-truth motions {args.stream_start}..{args.stream_start + n - 2} of full_seq_1 (unbounded reflector world, 15 % movers, scintillation 0.4"
                   + (", intra-scan distortion)" if md else ")"),
           "config": dict({"workload": "single-sequence streaming (BASELINE config " + ("4: motionDistortion ON" if md else "3: motionDistortion OFF")
                                       + " + outlier rejection): 1 lane, frames uploaded from a pinned ring on the copy stream (host staging copy + PCIe included), every pose read back",
                           "device": info["name"], "arch": info["arch"]}, **cfg),
           "roofline": None, "cpu_baseline": None}
    print(json.dumps(out), flush=True)



# ------------------------------------------------------------------------------------------------ BASELINE config 5
class DryStreamEngine:
    """CPU stand-in of the 1-lane streaming engine + its keyframe exchange (launcher / schedule test only)"""

    def __init__(self, rank, comm):
        self.rank, self.comm, self.n, self.map, self.sent = rank, comm, 0, [], 0

    def step(self):
        self.n += 1
        time.sleep(0.0005)

    def keyframe_exchange(self, lane=0):
        mine = (self.n * 7 + self.rank * 3) % 5 == 0                      # "this step made a keyframe", rank- and step-dependent
        self.sent += mine
        got = self.comm.rdv.gather(f"kfx{self.n}", json.dumps(dict(valid=bool(mine), root=self.rank, step=self.n)).encode())
        self.map += [json.loads(g) for g in got if json.loads(g)["valid"]]


def run_config5(args):
    """one process per GPU, one sequence per rank; after EACH step every rank contributes its new keyframe (or an empty record) to
    one all-gather and appends what it receives to its remote map - inside the loop, without draining the step pipeline"""
    from radarslampy_amd import distributed as D
    rank, local_rank, world = D.rank_env()
    rdv = D.FileRendezvous(D.rendezvous_dir(), rank, world)
    if world > 1:
        rdv.watchdog()
    n = args.c5_frames
    md = not args.no_md
    try:
        if args.dry_engine:
            comm = D.FileComm(rdv)
            eng = DryStreamEngine(rank, comm)
            comm.barrier()
            t0 = time.perf_counter()
            for k in range(1, n):
                eng.step()
                eng.keyframe_exchange(0)
            comm.barrier()
            dt = comm.allreduce_max(time.perf_counter() - t0)
            sent = [int(x) for x in rdv.gather("sent", str(eng.sent).encode())]
            out = dict(map_received=len(eng.map), keyframes_sent_per_rank=sent, senders=sorted({m["root"] for m in eng.map}),
                       backend=comm.backend, us_per_exchange=None, pairs_per_s_without_exchange=None, device="dry", arch="none")
            assert len(eng.map) == sum(sent), (len(eng.map), sent)
            comm.close()
        else:
            from radarslampy_amd import _ffi, synth
            from radarslampy_amd.RawROAMSystem import stream_records
            gold = os.path.join(ROOT, "tests", "golden", "full_seq_1_gt_deltas.npz")
            start = args.stream_start + 97 * rank                        # every rank drives its own stretch of full_seq_1's motions
            deltas = np.load(gold)["deltas"][start:start + n - 1]
            poses = synth.poses_from_deltas(deltas)
            jobs = synth.stream_jobs(synth.StreamWorld(10 + rank, mover_fraction=0.15), poses, distortion=md, scintillation=0.4)
            import multiprocessing as mp
            pool = mp.get_context("spawn").Pool(max(1, min(48, (os.cpu_count() or 2) // (2 * world))))
            try:
                recs = pool.map(synth._render_job, jobs, chunksize=4)
                pool.close(); pool.join()
            except BaseException:
                pool.terminate()
                raise
            ctx = _ffi.Context(local_rank)
            info = ctx.device_info()
            votes = rdv.gather("rccl_available", b"1" if D.RcclComm.available(ctx) else b"0")
            if not all(v == b"1" for v in votes):
                raise RuntimeError("config 5 needs RCCL on every rank")
            comm = D.RcclComm(ctx, rdv)
            flags = {"rejectOutliers": True, "correctMotionDistortion": md}
            stream_records(iter(recs[:12]), 12, poses[0], flags, ctx)                       # warm-up (allocations, first launches)
            state = {}

            def on_engine(eng):
                eng.remote_map_reserve(4096)

            def after_step(eng, k):
                eng.keyframe_exchange(0)

            def before_close(eng):
                state["map"] = eng.remote_map_count()
                state["last"] = eng.remote_map_get(state["map"][1] - 1) if state["map"][1] else None
            # without the exchange first (= --stream), then with it, both between barriers
            comm.barrier()
            t0 = time.perf_counter()
            est0, log0 = stream_records(iter(recs), n, poses[0], flags, ctx)
            dt0 = comm.allreduce_max(time.perf_counter() - t0)
            comm.barrier()
            t0 = time.perf_counter()
            est, log = stream_records(iter(recs), n, poses[0], flags, ctx, on_engine=on_engine, after_step=after_step, before_close=before_close)
            dt = comm.allreduce_max(time.perf_counter() - t0)
            assert est.tobytes() == est0.tobytes()                                          # the exchange does not touch the odometry
            mine = int(sum(e["new_keyframe"] for e in log))
            sent = [int(x) for x in rdv.gather("sent", str(mine).encode())]
            n_rec, n_res = state["map"]
            assert n_rec == sum(sent), (n_rec, sent)                                        # every keyframe of every rank arrived, once
            last = state["last"]
            assert last is None or (last["prunedUndistortedLocals"].shape[1] == 2 and 0 <= last["root"] < world)
            out = dict(map_received=int(n_rec), keyframes_sent_per_rank=sent, senders=sorted({ctx_r for ctx_r in range(world) if sent[ctx_r]}),
                       backend=comm.backend, us_per_exchange=round((dt - dt0) / (n - 1) * 1e6, 2),
                       pairs_per_s_without_exchange=round(world * (n - 1) / dt0, 2), device=info["name"], arch=info["arch"],
                       comm_rank_world_seen=comm.info(), retrack_fraction=round(float(np.mean([e["retrack"] for e in log])), 4))
            comm.close()
            ctx.close()
    except BaseException as ex:                          # noqa: BLE001
        rdv.mark_failed(repr(ex))
        raise
    if rank == 0:
        line = {"metric": "radar scan-pairs/sec (400x3768 polar), one sequence per GPU + keyframe exchange", "value": round(world * (n - 1) / dt, 2),
                "unit": "scan-pairs/s", "n_gpus": world, "steps": n - 1, "warmup": 11, "ms_per_step": round(dt / (n - 1) * 1e3, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/f32/f64",
                "data": f"synthetic Oxford-format records, one sequence of {n} scans per rank (reflector world seed 10 + rank, 15 % movers, scintillation 0.4, "
                        "its own stretch of full_seq_1's ground-truth motions)",
                "config": dict({"workload": "BASELINE config 5: one sequence per GPU through a 1-lane streaming engine (pinned ring, result ring); after EVERY step "
                                            "each rank puts its new keyframe {pose, velocity, undistorted features, <= 32768 polar peaks} or an empty record into ONE "
                                            "ncclAllGather of fixed-size records on the exchange stream and appends every non-empty record to its device-resident "
                                            "global map (Mapping.Map.addKeyframe on every rank); no host synchronisation inside the loop",
                                "frames": n, "launcher": "torch.distributed.run env" if "TORCHELASTIC_RUN_ID" in os.environ else ("bench.py --gpus" if world > 1 else "single process")}, **out),
                "roofline": None, "cpu_baseline": None}
        print(json.dumps(line), flush=True)


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.stream:
        run_stream(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, argv))
    if args.config5:
        run_config5(args)
        return
    run_rank(args)


if __name__ == "__main__":
    main()
