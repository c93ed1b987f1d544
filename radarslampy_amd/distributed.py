"""Multi-GPU plumbing: one process per GPU, sequences sharded by rank, no data-path collective.

The only exchange the path has is the keyframe broadcast for a global map (BASELINE config 5,
SURVEY §8e): {pose, velocity, features (K,2) f32, polar peaks (P,2) i32} from the owning rank to
all ranks - one RCCL broadcast over xGMI (backend "nccl" on ROCm), ~45 KB, latency-bound.
torch.distributed is used as plumbing only (process group + broadcast); with the gloo backend the
same code runs on CPU tensors (tests)."""
import numpy as np


def shard_sequences(n_sequences: int, rank: int, world: int):
    """Sequence s is owned by rank s % world."""
    return [s for s in range(n_sequences) if s % world == rank]


def _device(dist):
    import torch
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def broadcast_keyframe(kf, src: int, dist=None):
    """kf = dict(pose (3,), velocity (3,), features (K,2) f32, peaks (P,2) i32) on rank `src`
    (ignored elsewhere); returns the same dict on every rank."""
    import torch
    if dist is None:
        import torch.distributed as dist
    dev = _device(dist)
    me = dist.get_rank()
    hdr = torch.zeros(2, dtype=torch.int64, device=dev)
    if me == src:
        feats = np.ascontiguousarray(kf["features"], np.float32).reshape(-1, 2)
        peaks = np.ascontiguousarray(kf["peaks"], np.int32).reshape(-1, 2)
        hdr[0], hdr[1] = feats.shape[0], peaks.shape[0]
    dist.broadcast(hdr, src)
    K, P = int(hdr[0].item()), int(hdr[1].item())
    nbytes = 48 + K * 8 + P * 8
    if me == src:
        buf = np.empty(nbytes, np.uint8)
        buf[:24] = np.ascontiguousarray(kf["pose"], np.float64).view(np.uint8)
        buf[24:48] = np.ascontiguousarray(kf["velocity"], np.float64).view(np.uint8)
        buf[48:48 + K * 8] = feats.view(np.uint8).reshape(-1)
        buf[48 + K * 8:] = peaks.view(np.uint8).reshape(-1)
        t = torch.from_numpy(buf).to(dev)
    else:
        t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dist.broadcast(t, src)
    b = t.cpu().numpy()
    return dict(pose=b[:24].view(np.float64).copy(), velocity=b[24:48].view(np.float64).copy(),
                features=b[48:48 + K * 8].view(np.float32).reshape(K, 2).copy(),
                peaks=b[48 + K * 8:].view(np.int32).reshape(P, 2).copy())
