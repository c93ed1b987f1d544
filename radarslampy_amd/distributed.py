"""Multi-GPU plumbing: one process per GPU, sequences sharded by rank, NO data-path collective.

The only exchange the path has is handing a keyframe to a global map (reference Mapping.Map.addKeyframe,
Mapping.py:118-147; BASELINE config 5): `RcclComm.bcast_keyframe` broadcasts the device-resident keyframe of one lane
of the owning rank with ncclBroadcast (RCCL over xGMI) through the C-ABI (roam_comm_* / roam_bcast_keyframe,
csrc/comm.hip) - no torch, no host bounce of the payload on the sending side.

Rendezvous (who am I, where is the ncclUniqueId) is a directory on the node's local file system: rank 0 writes the
128-byte id, the others read it.  `FileComm` offers the same interface on files only; it exists for `bench.py
--dry-engine`, the CPU test of the launcher, and as bench.py's agreed fallback when RCCL cannot be initialised on some rank
(the JSON line then says "collective_backend": "file")."""
import json
import os
import time

import numpy as np


def shard_sequences(n_sequences: int, rank: int, world: int):
    """Sequence s is owned by rank s % world."""
    return [s for s in range(n_sequences) if s % world == rank]


def rank_env(env=os.environ):
    """(rank, local_rank, world) as torch.distributed.run / bench.py's own launcher export them"""
    return int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", env.get("RANK", "0"))), int(env.get("WORLD_SIZE", "1"))


def rendezvous_dir(env=os.environ) -> str:
    """One directory per launch: ROAM_RDV_DIR if the launcher set it, otherwise derived from what all ranks of one
    torch.distributed.run launch share (the agent's pid and the master port)."""
    d = env.get("ROAM_RDV_DIR")
    if not d:
        d = os.path.join(env.get("TMPDIR", "/tmp"), f"roam_rdv_{os.getppid()}_{env.get('MASTER_PORT', '0')}_{env.get('TORCHELASTIC_RUN_ID', 'none')}")
    os.makedirs(d, exist_ok=True)
    return d


class FileRendezvous:
    """Tiny key/value exchange between the ranks of one node through a directory (atomic rename on write)."""

    def __init__(self, path: str, rank: int, world: int, timeout: float = 300.0):
        self.path, self.rank, self.world, self.timeout = path, rank, world, timeout
        self._seq = 0

    def put(self, key: str, data: bytes):
        tmp = os.path.join(self.path, f".{key}.{self.rank}.tmp")
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, os.path.join(self.path, f"{key}.{self.rank}"))

    def get(self, key: str, rank: int) -> bytes:
        p = os.path.join(self.path, f"{key}.{rank}")
        t0 = time.monotonic()
        n = 0
        while not os.path.exists(p):
            n += 1
            if n % 50 == 0:                                  # a rank that died says so: nobody waits out the timeout for it
                dead = self.failed_ranks()
                if dead:
                    raise RuntimeError(f"rendezvous: rank(s) {dead} failed while rank {self.rank} waited for {key}.{rank}")
            if time.monotonic() - t0 > self.timeout:
                raise TimeoutError(f"rendezvous: {p} did not appear within {self.timeout} s")
            time.sleep(0.002)
        with open(p, "rb") as f:
            return f.read()

    def mark_failed(self, why: str = ""):
        """called by a rank (or by the launcher on its behalf) that cannot go on"""
        try:
            self.put("failed", why.encode()[:2000])
        except OSError:
            pass

    def failed_ranks(self):
        try:
            return sorted(int(f.split(".")[1]) for f in os.listdir(self.path) if f.startswith("failed.") and f.split(".")[1].isdigit())
        except OSError:
            return []

    def watchdog(self, period: float = 0.25):
        """daemon thread: leave the process (exit code 3) as soon as ANOTHER rank reports failure - a rank blocked inside an
        RCCL collective cannot be interrupted any other way, and must not keep its GPU"""
        import threading

        def run():
            while True:
                time.sleep(period)
                dead = [r for r in self.failed_ranks() if r != self.rank]
                if dead:
                    import sys
                    sys.stderr.write(f"[rank {self.rank}] rank(s) {dead} failed - exiting\n")
                    sys.stderr.flush()
                    os._exit(3)

        t = threading.Thread(target=run, daemon=True)
        t.start()
        return t

    def gather(self, tag: str, data: bytes):
        """every rank contributes `data`; returns the list of all contributions (also a barrier)"""
        self._seq += 1
        key = f"{tag}_{self._seq}"
        self.put(key, data)
        return [self.get(key, r) for r in range(self.world)]

    def cleanup(self):
        """collective: every rank signs off; rank 0 removes the directory once nobody reads it any more"""
        self.put("done", b"1")
        if self.rank == 0:
            for r in range(self.world):
                self.get("done", r)
            for f in os.listdir(self.path):
                try:
                    os.unlink(os.path.join(self.path, f))
                except OSError:
                    pass
            try:
                os.rmdir(self.path)
            except OSError:
                pass


class FileComm:
    """CPU stand-in with RcclComm's interface (bench.py --dry-engine and the CPU tests only)."""
    backend = "file"

    def __init__(self, rdv: FileRendezvous):
        self.rdv, self.rank, self.world = rdv, rdv.rank, rdv.world

    def info(self):
        return self.rank, self.world

    def barrier(self):
        self.rdv.gather("barrier", b"1")

    def allreduce_max(self, value: float) -> float:
        return max(float(x) for x in self.rdv.gather("max", repr(float(value)).encode()))

    def bcast_keyframe(self, engine, root: int, lane: int) -> dict:
        self.rdv._seq += 1
        key = f"kf_{self.rdv._seq}"
        if self.rank == root:
            kf = engine.live_keyframe(lane)
            blob = json.dumps({k: np.asarray(v).tolist() for k, v in kf.items()}).encode()
            self.rdv.put(key, blob)
        d = json.loads(self.rdv.get(key, root))
        self.barrier()
        kf = dict(pose=np.array(d["pose"], np.float64), velocity=np.array(d["velocity"], np.float64),
                  prunedUndistortedLocals=np.array(d["prunedUndistortedLocals"], np.float64).reshape(-1, 2),
                  peaks=np.array(d["peaks"], np.int32).reshape(-1, 2), scan=int(d["scan"]), lane=int(d["lane"]))
        if hasattr(engine, "remote_map_add"):                # the consumer (Map.addKeyframe on every rank), stand-in side
            engine.remote_map_add(dict(kf, root=root))
        return kf

    def close(self):
        self.barrier()
        self.rdv.cleanup()


class _stdout_to_stderr:
    """librccl prints a version banner on file descriptor 1 while the communicator is created; a program whose stdout is a
    protocol (bench.py: ONE JSON line) parks fd 1 on stderr for that moment (and flushes C stdio before un-parking)"""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        import ctypes
        ctypes.CDLL(None).fflush(None)                 # the banner sits in C stdio's buffer: flush it while fd 1 is parked
        os.dup2(self._saved, 1)
        os.close(self._saved)


class RcclComm:
    """RCCL communicator of one context (roam_comm_*): ncclUniqueId of rank 0 travels through the rendezvous directory."""
    backend = "rccl"

    @staticmethod
    def available(ctx) -> bool:
        """can librccl.so be bound in this process?  (no communicator is created: safe to call before the ranks have agreed)"""
        return bool(ctx.lib.roam_comm_available())

    def __init__(self, ctx, rdv: FileRendezvous):
        import ctypes as C
        from . import _ffi
        self.ctx, self.rdv, self.rank, self.world = ctx, rdv, rdv.rank, rdv.world
        self._C, self._ffi = C, _ffi
        if self.rank == 0:
            ident = (C.c_uint8 * _ffi.COMM_ID_BYTES)()
            rc = ctx.lib.roam_comm_unique_id(ident)
            if rc != _ffi.ROAM_OK:
                rdv.put("rccl_id", b"")                          # the other ranks fail at once instead of waiting for an id
                raise _ffi.RoamError(rc, "roam_comm_unique_id failed (librccl.so missing?)")
            rdv.put("rccl_id", bytes(ident))
        raw = rdv.get("rccl_id", 0)
        if len(raw) != _ffi.COMM_ID_BYTES:
            raise _ffi.RoamError(_ffi.ROAM_E_STATE, "rank 0 could not create an RCCL unique id")
        ident = (C.c_uint8 * _ffi.COMM_ID_BYTES).from_buffer_copy(raw)
        with _stdout_to_stderr():
            ctx.check(ctx.lib.roam_comm_init(ctx.h, ident, self.rank, self.world))
            ctx.check(ctx.lib.roam_comm_barrier(ctx.h))          # first collective: channels are set up here

    def info(self):
        """(rank, world) as RCCL itself reports them"""
        r, w = self._C.c_int32(-1), self._C.c_int32(-1)
        self.ctx.check(self.ctx.lib.roam_comm_info(self.ctx.h, self._C.byref(r), self._C.byref(w)))
        return r.value, w.value

    def barrier(self):
        self.ctx.check(self.ctx.lib.roam_comm_barrier(self.ctx.h))

    def allreduce_max(self, value: float) -> float:
        v = (self._C.c_double * 1)(float(value))
        self.ctx.check(self.ctx.lib.roam_comm_allreduce_f64(self.ctx.h, v, 1, 0))
        return float(v[0])

    def bcast_keyframe(self, engine, root: int, lane: int) -> dict:
        """collective: the live keyframe of `lane` on rank `root`, received from HBM to HBM on every rank"""
        C, _ffi = self._C, self._ffi
        hdr = _ffi.KeyframeHdr()
        loc = np.empty((_ffi.MAX_FEATURES, 2), np.float64)
        cap = engine.cfg.peaks_cap
        pk = np.empty((cap, 2), np.int32)
        self.ctx.check(self.ctx.lib.roam_bcast_keyframe(self.ctx.h, int(root), int(lane), C.byref(hdr), _ffi._ptr(loc), loc.shape[0],
                                                        _ffi._ptr(pk), cap))
        return dict(pose=np.array(hdr.pose[:]), velocity=np.array(hdr.velocity[:]),
                    prunedUndistortedLocals=loc[:hdr.n_features].copy(), peaks=pk[:hdr.n_peaks].copy(), scan=hdr.scan, lane=hdr.lane)

    def close_native(self):
        """drop the communicator only (the rendezvous stays: bench.py's fallback to FileComm)"""
        self.ctx.lib.roam_comm_destroy(self.ctx.h)

    def close(self):
        self.barrier()
        self.ctx.check(self.ctx.lib.roam_comm_destroy(self.ctx.h))
        self.rdv.cleanup()
