"""Multi-GPU plumbing of the BA hot path.

One process per GPU.  Landmarks (with all their observations) are partitioned
across ranks by the rule of ba_partition_points (include/ba_hip.h): locality
order, contiguous chunks balanced by observation count.  Poses and cameras are
replicated.  Per LM iteration there is exactly one data exchange — the
sum-all-reduce of the partial reduced camera system S||rhs — plus one 4-double
all-reduce of the LM scalars; both go through torch.distributed (backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""
import ctypes as C

import numpy as np

from . import _lib


def partition_points(pr, world):
    """Owner rank of every point for a C-ABI level problem dict (host only)."""
    lib = _lib.load()
    n_pt = pr["pt_X"].shape[0]
    owner = np.zeros(n_pt, np.int32)
    pf = np.ascontiguousarray(pr["pose_fixed"], np.uint8)
    qf = np.ascontiguousarray(pr["pt_fixed"], np.uint8)
    op = np.ascontiguousarray(pr["obs_pose"], np.int32)
    oq = np.ascontiguousarray(pr["obs_pt"], np.int32)
    _lib.check(lib.ba_partition_points(
        pr["pose_T"].shape[0], pf.ctypes.data_as(_lib._U8), n_pt,
        qf.ctypes.data_as(_lib._U8), op.shape[0],
        op.ctypes.data_as(_lib._I32), oq.ctypes.data_as(_lib._I32), world,
        owner.ctypes.data_as(_lib._I32)), "ba_partition_points")
    return owner


def shard_observations(pr, owner, rank):
    """The sub-problem rank `rank` linearises: all cameras / poses / points,
    only the observations of the points it owns (insertion order kept)."""
    keep = owner[pr["obs_pt"]] == rank
    out = dict(pr)
    for k in ("obs_cam", "obs_pose", "obs_pt", "obs_uv"):
        out[k] = np.ascontiguousarray(pr[k][keep])
    return out


class TorchExchange:
    """Owns the two exchange buffers as torch tensors, binds them into a
    BaProblem and serves the all-reduce hook with torch.distributed."""

    def __init__(self, problem, dist, device, stage_host=False):
        import torch
        self.dist = dist
        # stage_host: all-reduce a host copy (a backend without device
        # collectives, e.g. gloo when several ranks rehearse on one card)
        self.stage_host = stage_host
        self.bufs = []
        for which in (0, 1):
            n = problem.reduce_buffer_size(which)
            t = torch.zeros(n, dtype=torch.float64, device=device)
            problem.bind_reduce_buffer(which, t.data_ptr(), n)
            self.bufs.append(t)
        problem.set_allreduce(self.hook)

    def hook(self, which, ptr, n, stream):
        # issued on torch's current stream == the stream the kernels run on
        if self.stage_host:
            h = self.bufs[which].cpu()
            self.dist.all_reduce(h)
            self.bufs[which].copy_(h)
        else:
            self.dist.all_reduce(self.bufs[which])
        return 0
