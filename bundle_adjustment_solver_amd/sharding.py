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
    BaProblem and serves the all-reduce hook with torch.distributed.

    The C ABI hands the hook the stream the problem's kernels run on
    (include/ba_hip.h: "ordered on hip_stream"): the collective is issued with
    THAT stream current (torch.cuda.ExternalStream), whatever stream the handle
    uses — its own (the default) or one given through ba_set_stream."""

    def __init__(self, problem, dist, device, stage_host=False):
        import torch
        self.torch = torch
        self.dist = dist
        self.device = torch.device(device)
        # stage_host: all-reduce a host copy (a backend without device
        # collectives, e.g. gloo when several ranks rehearse on one card)
        self.stage_host = stage_host
        self.bufs = []
        self._streams = {}
        for which in (0, 1):
            n = problem.reduce_buffer_size(which)
            t = torch.zeros(n, dtype=torch.float64, device=self.device)
            problem.bind_reduce_buffer(which, t.data_ptr(), n)
            self.bufs.append(t)
        problem.set_allreduce(self.hook)

    def _stream(self, ptr):
        ptr = int(ptr or 0)
        st = self._streams.get(ptr)
        if st is None:
            # 0 is HIP's NULL stream = torch's default stream; ExternalStream(0)
            # is NOT that stream on this PyTorch-ROCm build (observed: the collective
            # then overtakes the kernels enqueued on NULL)
            st = (self.torch.cuda.default_stream(self.device) if ptr == 0 else
                  self.torch.cuda.ExternalStream(ptr, device=self.device))
            self._streams[ptr] = st
        return st

    def hook(self, which, ptr, n, stream):
        buf = self.bufs[which]
        if int(ptr or 0) != buf.data_ptr() or n > buf.numel():
            return -1   # the library is not using the bound buffer: refuse
        with self.torch.cuda.stream(self._stream(stream)):
            if self.stage_host:
                h = buf.cpu()              # waits for the kernels on `stream`
                self.dist.all_reduce(h)
                buf.copy_(h)
            else:
                self.dist.all_reduce(buf)  # stream-ordered: RCCL waits on / is
                #                            waited for by the current stream
        return 0
