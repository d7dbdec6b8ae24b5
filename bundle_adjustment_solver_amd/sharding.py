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
        if problem is not None:
            self.attach(problem)

    def attach(self, problem):
        """Bind the three exchange buffers (reduced system, LM scalars, final
        point gather) of a finalized problem and register the hook."""
        torch = self.torch
        self.bufs = []
        for which in (0, 1, 2):
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


class RcclExchange:
    """The exchange over RCCL INSIDE the library (include/ba_hip.h, "RCCL
    exchange"; csrc/ba_rccl.cpp): the hook is a C function, so no Python runs
    between the kernels of an LM iteration.  torch.distributed (any backend) is
    used once, to hand rank 0's communicator id to the other ranks.

    Use torch's process group for barriers and host-side reductions; the
    per-iteration collectives go through the library's own communicator."""

    def __init__(self, problem, dist, rank, world, device_index):
        import torch
        self.lib = _lib.load()
        if not self.lib.ba_rccl_available():
            raise RuntimeError("RCCL exchange unavailable: %s" %
                               (self.lib.ba_last_error() or b"?").decode())
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            _lib.check(self.lib.ba_rccl_get_unique_id(ident), "ba_rccl_get_unique_id")
        if world > 1:
            on_gpu = dist.get_backend() == "nccl"
            t = torch.tensor(list(ident), dtype=torch.uint8,
                             device=torch.device("cuda", device_index) if on_gpu else "cpu")
            dist.broadcast(t, 0)
            ident = (C.c_uint8 * 128)(*t.cpu().tolist())
        comm = C.c_void_p()
        _lib.check(self.lib.ba_rccl_comm_create(C.byref(comm), rank, world, ident,
                                                device_index), "ba_rccl_comm_create")
        self.comm = comm
        self.rank, self.world = rank, world
        if problem is not None:
            self.attach(problem)

    def attach(self, problem):
        problem.set_allreduce_native(self.lib.ba_rccl_allreduce_hook, self.comm.value)

    def size(self):
        """Ranks the communicator itself reports (ncclCommCount)."""
        return int(_lib.check(self.lib.ba_rccl_comm_size(self.comm), "ba_rccl_comm_size"))

    def calls(self):
        return int(self.lib.ba_rccl_comm_calls(self.comm))

    def close(self):
        if self.comm:
            self.lib.ba_rccl_comm_destroy(self.comm)
            self.comm = None
