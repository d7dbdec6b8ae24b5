"""Python mirror of the reference's solver interface over the HIP C ABI.

Mirrors (names, argument meaning, error behaviour):
  * FullBundleAdjustmentSolver   reference core/full_bundle_adjustment_solver.h:127-146
  * PoseOnlyBundleAdjustmentSolver.Solve_Monocular_6Dof
                                 reference core/pose_only_bundle_adjustment_solver.h:25-67
  * Options / Summary / OptimizationInfo / IterationStatus / SolverType
                                 reference core/solver_option_and_summary.h:25-93
Everything numerical happens in libba_hip.so (include/ba_hip.h); this module
only does what the reference's facade does on the host: pointer(identity)->
index maps, the 0.01 scaling, pose inversion and the final write-back.
"""
import ctypes as C
import enum
import sys
import time

import numpy as np

from . import _lib
from ._lib import BaIterInfo, BaOptions, BaPoIter, check

SCALER = 0.01          # reference core/full_bundle_adjustment_solver.cpp:38
INVERSE_SCALER = 1.0 / SCALER


class SolverType(enum.IntEnum):
    UNDEFINED = -1
    GRADIENT_DESCENT = 0
    GAUSS_NEWTON = 1
    LEVENBERG_MARQUARDT = 2


class IterationStatus(enum.IntEnum):
    UNDEFINED = -1
    UPDATE = 0
    UPDATE_TRUST_MORE = 1
    SKIPPED = 2


class OptimizationInfo:
    def __init__(self):
        self.cost = -1.0
        self.cost_change = -1.0
        self.average_reprojection_error = -1.0
        self.abs_gradient = -1.0
        self.abs_step = -1.0
        self.damping_term = -1.0
        self.iter_time = -1.0
        self.iteration_status = IterationStatus.UNDEFINED
        # extras (not in the reference): trust-region internals
        self.rho = float("nan")
        self.model_change = float("nan")
        self.trial_cost = float("nan")


class _Handle:
    pass


class Options:
    """reference core/solver_option_and_summary.h:47-71 (float fields)."""

    def __init__(self):
        self.solver_type = SolverType.GAUSS_NEWTON  # ignored by full BA (Q10)
        self.convergence_handle = _Handle()
        self.convergence_handle.threshold_step_size = 1e-5
        self.convergence_handle.threshold_cost_change = 1e-5
        self.outlier_handle = _Handle()
        self.outlier_handle.threshold_huber_loss = 1.0
        self.outlier_handle.threshold_outlier_rejection = 2.0
        self.iteration_handle = _Handle()
        self.iteration_handle.max_num_iterations = 50
        self.trust_region_handle = _Handle()
        self.trust_region_handle.initial_lambda = 100.0
        self.trust_region_handle.decrease_ratio_lambda = 0.33
        self.trust_region_handle.increase_ratio_lambda = 3.0

    def to_c(self):
        o = BaOptions()
        o.threshold_step_size = self.convergence_handle.threshold_step_size
        o.threshold_cost_change = self.convergence_handle.threshold_cost_change
        o.threshold_huber_loss = self.outlier_handle.threshold_huber_loss
        o.threshold_outlier_rejection = \
            self.outlier_handle.threshold_outlier_rejection
        o.max_num_iterations = int(self.iteration_handle.max_num_iterations)
        o.initial_lambda = self.trust_region_handle.initial_lambda
        o.decrease_ratio_lambda = \
            self.trust_region_handle.decrease_ratio_lambda
        o.increase_ratio_lambda = \
            self.trust_region_handle.increase_ratio_lambda
        return o


def _yellow(s):
    return "\033[0;33m" + s + "\033[0m"


def _green(s):
    return "\033[0;32m" + s + "\033[0m"


class Summary:
    """reference core/solver_option_and_summary.h:74-93, .cpp:8-84."""

    def __init__(self):
        self.optimization_info_list_ = []
        self.max_iteration_ = 0
        self.total_time_in_millisecond_ = 0.0
        self.threshold_step_size_ = 0.0
        self.threshold_cost_change_ = 0.0
        self.convergence_status_ = False

    def GetTotalTimeInSecond(self):
        return self.total_time_in_millisecond_ * 0.001

    def BriefReport(self):
        lines = ["itr   total_cost   avg.reproj.  cost_change  |step|   "
                 "|gradient|  damp_term  itr_time[ms] itr_stat"]
        for it, info in enumerate(self.optimization_info_list_):
            row = "%3d  %.6e    %.2e    %.2e   %.2e   %.2e    %.2e   %.2e" % (
                it, info.cost, info.average_reprojection_error,
                info.cost_change, info.abs_step, info.abs_gradient,
                info.damping_term, info.iter_time)
            if info.iteration_status == IterationStatus.UPDATE:
                row += "     UPDATE"
            elif info.iteration_status == IterationStatus.SKIPPED:
                row += "     " + _yellow(" SKIP ")
            elif info.iteration_status == IterationStatus.UPDATE_TRUST_MORE:
                row += "     " + _green("UPDATE")
            else:
                row += "     "
            lines.append(row)
        n = len(self.optimization_info_list_)
        lines.append("Analytic Solver Report:")
        lines.append("  Iterations      : %d" % n)
        lines.append("  Total time      : %.5g [second]" %
                     (self.total_time_in_millisecond_ * 0.001))
        if n:  # the reference dereferences front()/back() unconditionally
            first, last = (self.optimization_info_list_[0],
                           self.optimization_info_list_[-1])
            lines.append("  Initial cost    : %.5g" % first.cost)
            lines.append("  Final cost      : %.5g" % last.cost)
            lines.append("  Initial reproj. : %.5g [pixel]" %
                         first.average_reprojection_error)
            lines.append("  Final reproj.   : %.5g [pixel]" %
                         last.average_reprojection_error)
        lines.append(", Termination     : " +
                     (_green("CONVERGENCE") if self.convergence_status_
                      else _yellow("NO_CONVERGENCE")))
        if self.max_iteration_ == n:
            lines.append(_yellow(" WARNIING: MAX ITERATION is reached ! The "
                                 "solution could be local minima."))
        return "\n".join(lines) + "\n"


class Camera:
    """_BA_Camera, reference core/full_bundle_adjustment_solver.h:92-107."""

    def __init__(self, fx=0.0, fy=0.0, cx=0.0, cy=0.0, pose_this_to_cam0=None):
        self.fx, self.fy, self.cx, self.cy = fx, fy, cx, cy
        self.pose_this_to_cam0 = (np.eye(4) if pose_this_to_cam0 is None
                                  else np.array(pose_this_to_cam0, float))


def rigid_inverse(T):
    """Isometry inverse (R^T, -R^T t) as Eigen's Transform<..., Isometry>
    (SURVEY Q11); works on (..., 4, 4)."""
    T = np.asarray(T, dtype=np.float64)
    R = T[..., :3, :3]
    t = T[..., :3, 3]
    out = np.zeros_like(T)
    Rt = np.swapaxes(R, -1, -2)
    out[..., :3, :3] = Rt
    out[..., :3, 3] = -np.einsum("...ij,...j->...i", Rt, t)
    out[..., 3, 3] = 1.0
    return out


def _T44_to_12(T):
    T = np.asarray(T, dtype=np.float64).reshape(-1, 4, 4)
    return np.concatenate([T[:, :3, :3].reshape(-1, 9), T[:, :3, 3]], axis=1)


def _T12_to_44(T12):
    T12 = np.asarray(T12, dtype=np.float64).reshape(-1, 12)
    out = np.zeros((T12.shape[0], 4, 4))
    out[:, :3, :3] = T12[:, :9].reshape(-1, 3, 3)
    out[:, :3, 3] = T12[:, 9:]
    out[:, 3, 3] = 1.0
    return out


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class BaProblem:
    """Thin numpy wrapper over one ba_handle, in the solver's SCALED units.

    This is the level the parity tests drive: set_* / finalize / stage_* /
    get_* map one-to-one onto the C ABI.
    """

    def __init__(self, device=0):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.ba_create(C.byref(h), device), "ba_create")
        self.h = h
        self._keep = []
        self.n_pose = self.n_pt = 0
        self.N = self.M = 0
        self.M_global = 0

    def close(self):
        if self.h:
            self.lib.ba_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- construction --
    def set_cameras(self, intr4, T_cj12):
        intr4 = np.ascontiguousarray(intr4, np.float64).reshape(-1, 4)
        T = np.ascontiguousarray(T_cj12, np.float64).reshape(-1, 12)
        check(self.lib.ba_set_cameras(self.h, intr4.shape[0], _dp(intr4),
                                      _dp(T)), "ba_set_cameras")

    def set_poses(self, T_jw12, fixed):
        T = np.ascontiguousarray(T_jw12, np.float64).reshape(-1, 12)
        f = np.ascontiguousarray(fixed, np.uint8)
        self.n_pose = T.shape[0]
        self.pose_fixed = f.copy()
        check(self.lib.ba_set_poses(self.h, T.shape[0], _dp(T), _up(f)),
              "ba_set_poses")

    def set_points(self, X3, fixed):
        X = np.ascontiguousarray(X3, np.float64).reshape(-1, 3)
        f = np.ascontiguousarray(fixed, np.uint8)
        self.n_pt = X.shape[0]
        self.pt_fixed = f.copy()
        check(self.lib.ba_set_points(self.h, X.shape[0], _dp(X), _up(f)),
              "ba_set_points")

    def set_observations(self, cam, pose, pt, uv):
        cam = np.ascontiguousarray(cam, np.int32)
        pose = np.ascontiguousarray(pose, np.int32)
        pt = np.ascontiguousarray(pt, np.int32)
        uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 2)
        check(self.lib.ba_set_observations(self.h, cam.shape[0], _ip(cam),
                                           _ip(pose), _ip(pt), _dp(uv)),
              "ba_set_observations")

    def set_shard(self, rank, world):
        check(self.lib.ba_set_shard(self.h, rank, world), "ba_set_shard")

    def set_stream(self, stream_ptr):
        check(self.lib.ba_set_stream(self.h, C.c_void_p(stream_ptr)),
              "ba_set_stream")

    def finalize(self):
        check(self.lib.ba_finalize(self.h), "ba_finalize")
        self.N = self.lib.ba_num_opt_poses(self.h)
        self.M = self.lib.ba_num_opt_points(self.h)
        self.M_global = int(np.sum(self.pt_fixed == 0))
        self.P = self.lib.ba_num_pairs(self.h)

    def update_values(self, T_jw12=None, X3=None):
        """New parameter values for the finalized structure (no re-planning)."""
        T = None if T_jw12 is None else np.ascontiguousarray(T_jw12, np.float64).reshape(-1, 12)
        X = None if X3 is None else np.ascontiguousarray(X3, np.float64).reshape(-1, 3)
        if T is not None and T.shape[0] != self.n_pose or X is not None and X.shape[0] != self.n_pt:
            raise ValueError("update_values: the structure is fixed (same number of poses / points)")
        check(self.lib.ba_update_values(self.h, None if T is None else _dp(T),
                                        None if X is None else _dp(X)), "ba_update_values")

    def set_allreduce(self, pyfunc):
        """pyfunc(which:int, dev_ptr:int, n_doubles:int, stream:int) -> int"""
        def _cb(user, which, ptr, n, stream):
            try:
                return int(pyfunc(which, ptr or 0, n, stream or 0) or 0)
            except Exception as e:  # never let an exception cross the ABI
                sys.stderr.write("all-reduce hook failed: %r\n" % (e,))
                return 1
        cb = _lib.ALLREDUCE_FN(_cb)
        self._keep.append(cb)
        check(self.lib.ba_set_allreduce(self.h, cb, None), "ba_set_allreduce")

    def set_allreduce_native(self, fn_ptr, user_ptr):
        """Register a C function (ba_allreduce_fn, e.g. ba_rccl_allreduce_hook)
        with its user pointer: no Python inside the LM loop."""
        fn = C.cast(fn_ptr, _lib.ALLREDUCE_FN)
        self._keep.append(fn)
        check(self.lib.ba_set_allreduce(self.h, fn, C.c_void_p(user_ptr)),
              "ba_set_allreduce")

    def gather_points(self):
        """Final exchange of a sharded Solve (reference :1018-1022 writes back
        every point): afterwards get_points() returns all points on every rank."""
        check(self.lib.ba_gather_points(self.h), "ba_gather_points")

    def reduce_buffer_size(self, which):
        return int(self.lib.ba_reduce_buffer_size(self.h, which))

    def bind_reduce_buffer(self, which, dev_ptr, n):
        check(self.lib.ba_bind_reduce_buffer(self.h, which,
                                             C.c_void_p(dev_ptr), n),
              "ba_bind_reduce_buffer")

    # -- LM loop --
    def solve(self, opt, cap=None):
        cap = cap or max(1, opt.max_num_iterations)
        rows = (BaIterInfo * cap)()
        n = C.c_int(0)
        conv = C.c_int(0)
        check(self.lib.ba_solve(self.h, C.byref(opt), rows, cap, C.byref(n),
                                C.byref(conv)), "ba_solve")
        return [rows[i] for i in range(min(n.value, cap))], bool(conv.value)

    def lm_begin(self, opt):
        check(self.lib.ba_lm_begin(self.h, C.byref(opt)), "ba_lm_begin")

    def lm_iterate(self, n):
        check(self.lib.ba_lm_iterate(self.h, n), "ba_lm_iterate")

    def lm_sync(self, cap=0):
        rows = (BaIterInfo * max(cap, 1))()
        n = C.c_int(0)
        conv = C.c_int(0)
        rc = check(self.lib.ba_lm_sync(self.h, rows, cap, C.byref(n),
                                       C.byref(conv)), "ba_lm_sync")
        return ([rows[i] for i in range(min(n.value, cap))], n.value,
                bool(conv.value), bool(rc))

    # -- stages --
    def stage_cost(self):
        v = C.c_double(0)
        check(self.lib.ba_stage_cost(self.h, C.byref(v)), "ba_stage_cost")
        return v.value

    def stage_linearize(self, lam, huber):
        check(self.lib.ba_stage_linearize(self.h, lam, huber),
              "ba_stage_linearize")

    def stage_schur(self):
        check(self.lib.ba_stage_schur(self.h), "ba_stage_schur")

    def stage_solve_reduced(self):
        check(self.lib.ba_stage_solve_reduced(self.h),
              "ba_stage_solve_reduced")

    def stage_backsub_update(self):
        check(self.lib.ba_stage_backsub_update(self.h),
              "ba_stage_backsub_update")

    def stage_scalars(self):
        a, b, c, d = (C.c_double(0) for _ in range(4))
        check(self.lib.ba_stage_scalars(self.h, C.byref(a), C.byref(b),
                                        C.byref(c), C.byref(d)),
              "ba_stage_scalars")
        return a.value, b.value, c.value, d.value

    def stage_commit(self, accept):
        check(self.lib.ba_stage_commit(self.h, int(bool(accept))),
              "ba_stage_commit")

    def enable_stage_timing(self, on=True):
        check(self.lib.ba_enable_stage_timing(self.h, int(on)),
              "ba_enable_stage_timing")

    def get_stage_ms(self, reset=True):
        out = np.zeros(8)
        check(self.lib.ba_get_stage_ms(self.h, _dp(out), int(reset)),
              "ba_get_stage_ms")
        return out

    # -- readers --
    def get_poses(self):
        out = np.zeros((self.n_pose, 12))
        check(self.lib.ba_get_poses(self.h, _dp(out)), "ba_get_poses")
        return out

    def get_points(self, into=None):
        out = np.zeros((self.n_pt, 3)) if into is None else into
        mask = np.zeros(self.n_pt, np.uint8)
        check(self.lib.ba_get_points(self.h, _dp(out), _up(mask)),
              "ba_get_points")
        return out, mask.astype(bool)

    def get_A(self):
        A = np.zeros((self.N, 6, 6))
        a = np.zeros((self.N, 6))
        check(self.lib.ba_get_A(self.h, _dp(A), _dp(a)), "ba_get_A")
        return A, a

    def get_C(self):
        Cm = np.zeros((self.M_global, 3, 3))
        b = np.zeros((self.M_global, 3))
        check(self.lib.ba_get_C(self.h, _dp(Cm), _dp(b)), "ba_get_C")
        return Cm, b

    def get_Cinv(self):
        Ci = np.zeros((self.M_global, 3, 3))
        cb = np.zeros((self.M_global, 3))
        check(self.lib.ba_get_Cinv(self.h, _dp(Ci), _dp(cb)), "ba_get_Cinv")
        return Ci, cb

    def get_pairs(self):
        P = int(self.P)
        pi = np.zeros(P, np.int32)
        pj = np.zeros(P, np.int32)
        W = np.zeros((P, 6, 3))
        check(self.lib.ba_get_pairs(self.h, _ip(pi), _ip(pj), _dp(W)),
              "ba_get_pairs")
        return pi, pj, W

    def get_S(self):
        n6 = 6 * self.N
        S = np.zeros((n6, n6))
        rhs = np.zeros(n6)
        check(self.lib.ba_get_S(self.h, _dp(S), _dp(rhs)), "ba_get_S")
        return S, rhs

    def get_xy(self):
        x = np.zeros((self.N, 6))
        y = np.zeros((self.M_global, 3))
        check(self.lib.ba_get_xy(self.h, _dp(x), _dp(y)), "ba_get_xy")
        return x, y

    def get_kernel_ms(self, reset=True):
        """{kernel name: (total ms, launches)} accumulated in timing mode."""
        n = self.lib.ba_kernel_count()
        ms = np.zeros(n)
        calls = np.zeros(n, np.int64)
        check(self.lib.ba_get_kernel_ms(
            self.h, _dp(ms), calls.ctypes.data_as(C.POINTER(C.c_int64)),
            int(reset)), "ba_get_kernel_ms")
        return {self.lib.ba_kernel_name(k).decode(): (float(ms[k]), int(calls[k]))
                for k in range(n)}

    def get_dense_info(self):
        out = np.zeros(4)
        check(self.lib.ba_get_dense_info(self.h, _dp(out)),
              "ba_get_dense_info")
        return dict(fill=out[0], flops=out[1], levels=int(out[2]),
                    npad=int(out[3]))

    def get_schur_info(self):
        """How the Schur complement is accumulated: covisibility-group workgroups
        (32- / 64-wide tiles), landmarks they cover, super-runs for the rest."""
        v = (C.c_int64 * 8)()
        check(self.lib.ba_get_schur_info(self.h, v), "ba_get_schur_info")
        return dict(groups32=int(v[0]), groups64=int(v[1]), grouped_landmarks=int(v[2]),
                    super_runs=int(v[3]), grouped_pairs=int(v[4]), grouped_triples=int(v[5]),
                    group_mfma=int(v[6]), list_triples=int(v[7]))

    def get_lin_info(self):
        """How the shard is linearised: pieces of the covisibility-group kernel
        (landmark and pose side in one pass), the observations they cover, chunks left
        to the chunk kernel, observations on the pose-major list."""
        v = (C.c_int64 * 4)()
        check(self.lib.ba_get_lin_info(self.h, v), "ba_get_lin_info")
        return dict(group_pieces=int(v[0]), group_observations=int(v[1]), chunks=int(v[2]),
                    pose_major_observations=int(v[3]))

    def get_mask_info(self):
        """Superset (masked) covisibility groups of this shard."""
        v = (C.c_int64 * 4)()
        check(self.lib.ba_get_mask_info(self.h, v), "ba_get_mask_info")
        return dict(masked_pieces=int(v[0]), masked_landmarks=int(v[1]),
                    padded_observation_slots=int(v[2]), padded_pairs=int(v[3]))

    def get_dropped_pivots(self, reset=False):
        """Non-positive pivots met by the reduced-system Cholesky since lm_begin
        (see include/ba_hip.h: where it differs from the reference's pivoted
        LDLT)."""
        v = C.c_int64(0)
        check(self.lib.ba_get_dropped_pivots(self.h, C.byref(v), int(reset)),
              "ba_get_dropped_pivots")
        return int(v.value)

    def dense_spd_solve(self, A, b):
        A = np.ascontiguousarray(A, np.float64)
        b = np.ascontiguousarray(b, np.float64)
        n = A.shape[0]
        x = np.zeros(n)
        ms = C.c_double(0)
        check(self.lib.ba_dense_spd_solve(self.h, n, _dp(A), _dp(b), _dp(x),
                                          C.byref(ms)), "ba_dense_spd_solve")
        return x, ms.value

    def pose_only_mono6(self, X3, uv2, fx, fy, cx, cy, T12, mask, opt,
                        cap=None, want_debug=False):
        X = np.ascontiguousarray(X3, np.float32).reshape(-1, 3)
        uv = np.ascontiguousarray(uv2, np.float32).reshape(-1, 2)
        n = X.shape[0]
        T = np.ascontiguousarray(T12, np.float32).reshape(12).copy()
        m = np.ascontiguousarray(mask, np.uint8).copy()
        cap = cap or max(1, opt.max_num_iterations)
        rows = (BaPoIter * cap)()
        n_it = C.c_int(0)
        conv = C.c_int(0)
        dbg = np.zeros((cap, 12), np.float32) if want_debug else None
        rc = check(self.lib.ba_pose_only_mono6(
            self.h, _fp(X), _fp(uv), n, fx, fy, cx, cy, _fp(T), _up(m),
            C.byref(opt), rows, cap, C.byref(n_it), C.byref(conv),
            _fp(dbg) if want_debug else None), "ba_pose_only_mono6")
        nrows = n_it.value - 1 if conv.value else n_it.value
        nrows = max(0, min(nrows, cap))
        return dict(T12=T, mask=m.astype(bool), n_iter=n_it.value,
                    converged=bool(conv.value), success=(rc == 0),
                    rows=[(rows[i].cost, rows[i].cost_change,
                           rows[i].abs_step) for i in range(nrows)],
                    debug=dbg[:min(n_it.value, cap)] if want_debug else None)


    def pose_only_stereo6(self, X3, uvl2, uvr2, intr_l, intr_r, T_lr12, T12,
                          mask_l, mask_r, opt, cap=None, want_debug=False):
        """reference core/pose_only_bundle_adjustment_solver.cpp:172-399."""
        X = np.ascontiguousarray(X3, np.float32).reshape(-1, 3)
        ul = np.ascontiguousarray(uvl2, np.float32).reshape(-1, 2)
        ur = np.ascontiguousarray(uvr2, np.float32).reshape(-1, 2)
        if ul.shape[0] != X.shape[0] or ur.shape[0] != X.shape[0]:
            raise RuntimeError(  # reference :203-214
                "In PoseOnlyBundleAdjustmentSolver::"
                "SolveStereoPoseOnlyBundleAdjustment6Dof(), "
                "world_position_list.size() != current_pixel_list.size()")
        n = X.shape[0]
        il = np.ascontiguousarray(intr_l, np.float32).reshape(4)
        ir = np.ascontiguousarray(intr_r, np.float32).reshape(4)
        Tlr = np.ascontiguousarray(T_lr12, np.float32).reshape(12)
        T = np.ascontiguousarray(T12, np.float32).reshape(12).copy()
        ml = np.ascontiguousarray(mask_l, np.uint8).copy()
        mr = np.ascontiguousarray(mask_r, np.uint8).copy()
        cap = cap or max(1, opt.max_num_iterations)
        rows = (BaPoIter * cap)()
        n_it = C.c_int(0)
        conv = C.c_int(0)
        dbg = np.zeros((cap, 12), np.float32) if want_debug else None
        rc = check(self.lib.ba_pose_only_stereo6(
            self.h, _fp(X), _fp(ul), _fp(ur), n, _fp(il), _fp(ir), _fp(Tlr),
            _fp(T), _up(ml), _up(mr), C.byref(opt), rows, cap, C.byref(n_it),
            C.byref(conv), _fp(dbg) if want_debug else None),
            "ba_pose_only_stereo6")
        nrows = n_it.value - 1 if conv.value else n_it.value
        nrows = max(0, min(nrows, cap))
        return dict(T12=T, mask_l=ml.astype(bool), mask_r=mr.astype(bool),
                    n_iter=n_it.value, converged=bool(conv.value),
                    success=(rc == 0),
                    rows=[(rows[i].cost, rows[i].cost_change,
                           rows[i].abs_step) for i in range(nrows)],
                    debug=dbg[:min(n_it.value, cap)] if want_debug else None)


class BaStream:
    """Observation streaming (include/ba_hip.h ba_stream_*): the
    same problem-construction calls and LM loop as BaProblem for a problem whose
    landmark-side data does not have to fit in device memory — n_chunks landmark
    chunks pass through two device arenas of arena_bytes each, twice per iteration."""

    def __init__(self, device=0, n_chunks=4, arena_bytes=1 << 30):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.ba_stream_create(C.byref(h), device, n_chunks, arena_bytes),
              "ba_stream_create")
        self.h = h
        self.n_pose = self.n_pt = 0

    def close(self):
        if self.h:
            self.lib.ba_stream_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_cameras(self, intr4, T_cj12):
        intr4 = np.ascontiguousarray(intr4, np.float64).reshape(-1, 4)
        T = np.ascontiguousarray(T_cj12, np.float64).reshape(-1, 12)
        check(self.lib.ba_stream_set_cameras(self.h, intr4.shape[0], _dp(intr4), _dp(T)),
              "ba_stream_set_cameras")

    def set_poses(self, T_jw12, fixed):
        T = np.ascontiguousarray(T_jw12, np.float64).reshape(-1, 12)
        f = np.ascontiguousarray(fixed, np.uint8)
        self.n_pose = T.shape[0]
        check(self.lib.ba_stream_set_poses(self.h, T.shape[0], _dp(T), _up(f)),
              "ba_stream_set_poses")

    def set_points(self, X3, fixed):
        X = np.ascontiguousarray(X3, np.float64).reshape(-1, 3)
        f = np.ascontiguousarray(fixed, np.uint8)
        self.n_pt = X.shape[0]
        check(self.lib.ba_stream_set_points(self.h, X.shape[0], _dp(X), _up(f)),
              "ba_stream_set_points")

    def set_observations(self, cam, pose, pt, uv):
        cam = np.ascontiguousarray(cam, np.int32)
        pose = np.ascontiguousarray(pose, np.int32)
        pt = np.ascontiguousarray(pt, np.int32)
        uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 2)
        check(self.lib.ba_stream_set_observations(self.h, cam.shape[0], _ip(cam), _ip(pose),
                                                  _ip(pt), _dp(uv)),
              "ba_stream_set_observations")

    def finalize(self):
        check(self.lib.ba_stream_finalize(self.h), "ba_stream_finalize")

    def solve(self, opt, cap=None):
        cap = cap or max(1, opt.max_num_iterations)
        rows = (BaIterInfo * cap)()
        n = C.c_int(0)
        conv = C.c_int(0)
        check(self.lib.ba_stream_solve(self.h, C.byref(opt), rows, cap, C.byref(n),
                                       C.byref(conv)), "ba_stream_solve")
        return [rows[i] for i in range(min(n.value, cap))], bool(conv.value)

    def lm_begin(self, opt):
        check(self.lib.ba_stream_lm_begin(self.h, C.byref(opt)), "ba_stream_lm_begin")

    def lm_iterate(self, n):
        check(self.lib.ba_stream_lm_iterate(self.h, n), "ba_stream_lm_iterate")

    def lm_sync(self, cap=0):
        rows = (BaIterInfo * max(cap, 1))()
        n = C.c_int(0)
        conv = C.c_int(0)
        rc = check(self.lib.ba_stream_lm_sync(self.h, rows, cap, C.byref(n), C.byref(conv)),
                   "ba_stream_lm_sync")
        return ([rows[i] for i in range(min(n.value, cap))], n.value, bool(conv.value), bool(rc))

    def get_poses(self):
        out = np.zeros((self.n_pose, 12))
        check(self.lib.ba_stream_get_poses(self.h, _dp(out)), "ba_stream_get_poses")
        return out

    def get_points(self):
        out = np.zeros((self.n_pt, 3))
        check(self.lib.ba_stream_get_points(self.h, _dp(out)), "ba_stream_get_points")
        return out

    def info(self):
        v = (C.c_int64 * 6)()
        check(self.lib.ba_stream_info(self.h, v), "ba_stream_info")
        return dict(arena_bytes=v[0], largest_chunk_bytes=v[1], all_chunks_bytes=v[2],
                    bytes_h2d=v[3], bytes_d2h=v[4], n_chunks=v[5])


class FullBundleAdjustmentSolver:
    """Mirror of reference core/full_bundle_adjustment_solver.h:127-146.

    Poses are 4x4 numpy arrays (world->camera-body pose, as in
    test/test_ba.cpp:162-167), points 3-vectors.  As in the reference the
    solver identifies them by OBJECT IDENTITY (`id(obj)` plays the role of
    the pointer key) and writes the result back INTO the same arrays at the
    end of Solve.  AddPoseArray / AddPointArray / AddObservations are bulk
    forms (integer handles) for the multi-million-observation configs.
    """

    def __init__(self, device=0, verbose=False):
        self.device = device
        self.verbose = verbose
        self.Reset()
        if self.verbose:
            print("SparseBundleAdjustmentSolver() - initialize.")

    # reference :44-70
    def Reset(self):
        self.camera_id_to_camera_map_ = {}
        self.scaler_ = SCALER
        self.inverse_scaler_ = INVERSE_SCALER
        self._pose_objs = []      # (array, row or None)
        self._pose_key = {}
        self._pose_T_jw = []      # list of (k,12) chunks
        self._pose_fixed = set()
        self._pt_objs = []
        self._pt_key = {}
        self._pt_X = []
        self._pt_fixed = set()
        self._obs_cam, self._obs_pose, self._obs_pt, self._obs_uv = \
            [], [], [], []
        self.num_total_poses_ = 0
        self.num_total_points_ = 0
        self.num_fixed_poses_ = 0
        self.num_fixed_points_ = 0
        self.num_total_observations_ = 0
        self.num_optimization_poses_ = 0
        self.num_optimization_points_ = 0
        self.is_parameter_finalized_ = False
        self._problem = None
        self._shard = (0, 1)
        self._allreduce = None
        self._stream = None

    # ---- registration ----
    def AddCamera(self, camera_index, camera):          # reference :72-85
        if camera_index in self.camera_id_to_camera_map_:
            return  # unordered_map::insert keeps the first
        c = Camera(camera.fx * SCALER, camera.fy * SCALER, camera.cx * SCALER,
                   camera.cy * SCALER, camera.pose_this_to_cam0)
        c.pose_this_to_cam0[:3, 3] *= SCALER
        self.camera_id_to_camera_map_[camera_index] = c
        if self.verbose:
            print("New camera is added.\n  fx: %g, fy: %g, cx: %g, cy: %g" %
                  (c.fx, c.fy, c.cx, c.cy))

    def _finalized_warning(self):
        sys.stderr.write(_yellow("Cannot enroll parameter. "
                                 "(is_parameter_finalized_ == true)") + "\n")

    def AddPose(self, original_pose):                   # reference :87-101
        if self.is_parameter_finalized_:
            self._finalized_warning()
            return None
        key = id(original_pose)
        if key in self._pose_key:
            return self._pose_key[key]
        T_jw = rigid_inverse(original_pose)
        T_jw[:3, 3] *= SCALER
        h = self.num_total_poses_
        self._pose_key[key] = h
        self._pose_objs.append((original_pose, None))
        self._pose_T_jw.append(_T44_to_12(T_jw))
        self.num_total_poses_ += 1
        return h

    def AddPoseArray(self, poses):
        """Bulk AddPose of an (n,4,4) array; returns integer handles."""
        if self.is_parameter_finalized_:
            self._finalized_warning()
            return None
        poses = np.asarray(poses)
        assert poses.ndim == 3 and poses.shape[1:] == (4, 4) and \
            poses.dtype == np.float64
        T_jw = rigid_inverse(poses)
        T_jw[:, :3, 3] *= SCALER
        base = self.num_total_poses_
        n = poses.shape[0]
        self._pose_objs.extend((poses, r) for r in range(n))
        self._pose_T_jw.append(_T44_to_12(T_jw))
        self.num_total_poses_ += n
        return np.arange(base, base + n, dtype=np.int32)

    def AddPoint(self, original_point):                 # reference :103-117
        if self.is_parameter_finalized_:
            self._finalized_warning()
            return None
        key = id(original_point)
        if key in self._pt_key:
            return self._pt_key[key]
        h = self.num_total_points_
        self._pt_key[key] = h
        self._pt_objs.append((original_point, h, 0))
        self._pt_X.append(np.asarray(original_point, np.float64)
                          .reshape(1, 3) * SCALER)
        self.num_total_points_ += 1
        return h

    def AddPointArray(self, points):
        if self.is_parameter_finalized_:
            self._finalized_warning()
            return None
        points = np.asarray(points)
        assert points.ndim == 2 and points.shape[1] == 3 and \
            points.dtype == np.float64
        base = self.num_total_points_
        n = points.shape[0]
        self._pt_objs.append((points, base, n))
        self._pt_X.append(points * SCALER)
        self.num_total_points_ += n
        return np.arange(base, base + n, dtype=np.int32)

    def _pose_handle(self, pose):
        if isinstance(pose, (int, np.integer)):
            return int(pose) if 0 <= pose < self.num_total_poses_ else None
        return self._pose_key.get(id(pose))

    def _point_handle(self, point):
        if isinstance(point, (int, np.integer)):
            return int(point) if 0 <= point < self.num_total_points_ else None
        return self._pt_key.get(id(point))

    def MakePoseFixed(self, original_pose):             # reference :119-134
        if self.is_parameter_finalized_:
            self._finalized_warning()
            return
        if original_pose is None:
            sys.stderr.write("Empty pointer is conveyed. Skip this one.\n")
            return
        h = self._pose_handle(original_pose)
        if h is None:
            raise RuntimeError("There is no pointer in the BA pose pool.")
        self._pose_fixed.add(h)
        self.num_fixed_poses_ += 1  # counts duplicates too, like the reference

    def MakePointFixed(self, original_point):           # reference :136-153
        if self.is_parameter_finalized_:
            self._finalized_warning()
            return
        if original_point is None:
            sys.stderr.write("Empty pointer is conveyed. Skip this one.\n")
            return
        h = self._point_handle(original_point)
        if h is None:
            raise RuntimeError("There is no pointer in the BA point pool.")
        self._pt_fixed.add(h)
        self.num_fixed_points_ += 1

    def AddObservation(self, camera_index, related_pose, related_point,
                       pixel):                           # reference :155-180
        if camera_index not in self.camera_id_to_camera_map_:
            sys.stderr.write("\033[0;31mInvalid camera index.\n\033[0m")
            return
        hp = self._pose_handle(related_pose)
        if hp is None:
            sys.stderr.write("\033[0;31mNonexisting pose.\n\033[0m")
            return
        hq = self._point_handle(related_point)
        if hq is None:
            sys.stderr.write("\033[0;31mNonexisting point.\n\033[0m")
            return
        self._obs_cam.append(np.array([camera_index], np.int64))
        self._obs_pose.append(np.array([hp], np.int32))
        self._obs_pt.append(np.array([hq], np.int32))
        self._obs_uv.append(np.asarray(pixel, np.float64).reshape(1, 2)
                            * SCALER)
        self.num_total_observations_ += 1

    def AddObservations(self, camera_index, pose_handles, point_handles,
                        pixels):
        """Bulk AddObservation; invalid entries are dropped with a warning,
        as the reference drops them one by one."""
        cam = np.broadcast_to(np.asarray(camera_index, np.int64),
                              np.shape(pose_handles)).copy()
        hp = np.asarray(pose_handles, np.int64)
        hq = np.asarray(point_handles, np.int64)
        uv = np.asarray(pixels, np.float64).reshape(-1, 2)
        known = np.array(sorted(self.camera_id_to_camera_map_), np.int64)
        ok = np.isin(cam, known) & (hp >= 0) & (hp < self.num_total_poses_) \
            & (hq >= 0) & (hq < self.num_total_points_)
        if not ok.all():
            sys.stderr.write("\033[0;31m%d invalid observations dropped.\n"
                             "\033[0m" % int((~ok).sum()))
            cam, hp, hq, uv = cam[ok], hp[ok], hq[ok], uv[ok]
        self._obs_cam.append(cam)
        self._obs_pose.append(hp.astype(np.int32))
        self._obs_pt.append(hq.astype(np.int32))
        self._obs_uv.append(uv * SCALER)
        self.num_total_observations_ += cam.shape[0]

    def ReloadParameterValues(self):
        """(new) Read the CURRENT values of every registered pose / point object again
        and hand them to the finalized problem (ba_update_values): a SLAM back end that
        re-optimises the same graph with new values pays no second FinalizeParameters
        (index assignment, plan, uploads).  The reference keeps its own copies from
        AddPose / AddPoint across Solve calls (:44-70, :87-117) and offers no such call;
        without it this facade, like the reference, continues from its internal state."""
        if not self.is_parameter_finalized_:
            return   # nothing planned yet: FinalizeParameters reads the stored copies
        T = np.zeros((self.num_total_poses_, 12))
        for h, (obj, row) in enumerate(self._pose_objs):
            Tw = np.asarray(obj if row is None else obj[row], np.float64)
            T_jw = rigid_inverse(Tw)
            T_jw[:3, 3] *= SCALER
            T[h] = _T44_to_12(T_jw[None])[0]
        X = np.zeros((self.num_total_points_, 3))
        for obj, base, cnt in self._pt_objs:
            if cnt == 0:
                X[base] = np.asarray(obj, np.float64).reshape(3) * SCALER
            else:
                X[base:base + cnt] = np.asarray(obj, np.float64) * SCALER
        self._problem.update_values(T, X)

    # ---- multi-GPU plumbing (new) ----
    def SetShard(self, rank, world, allreduce=None, stream=None):
        """allreduce: a callable hook(which, dev_ptr, n_doubles, stream) -> int, or
        an exchange object with .attach(problem) (sharding.TorchExchange /
        sharding.RcclExchange built with problem=None)."""
        self._shard = (rank, world)
        self._allreduce = allreduce
        self._stream = stream

    # ---- finalize / solve ----
    def _host_arrays(self):
        """The registered problem as C-ABI level arrays (host only, no GPU)."""
        if not self.camera_id_to_camera_map_ or not self.num_total_poses_ \
                or not self.num_total_points_:
            raise RuntimeError("cameras, poses and points must be added "
                               "before FinalizeParameters")
        cam_ids = sorted(self.camera_id_to_camera_map_)
        cam_pos = {cid: k for k, cid in enumerate(cam_ids)}
        intr = np.array([[c.fx, c.fy, c.cx, c.cy] for c in
                         (self.camera_id_to_camera_map_[i] for i in cam_ids)])
        camT = _T44_to_12(np.stack([self.camera_id_to_camera_map_[i]
                                    .pose_this_to_cam0 for i in cam_ids]))
        T_jw = np.concatenate(self._pose_T_jw, axis=0)
        X = np.concatenate(self._pt_X, axis=0)
        pf = np.zeros(self.num_total_poses_, np.uint8)
        pf[list(self._pose_fixed)] = 1
        qf = np.zeros(self.num_total_points_, np.uint8)
        qf[list(self._pt_fixed)] = 1
        if self._obs_cam:
            cam_raw = np.concatenate(self._obs_cam)
            lut = np.full(int(max(cam_ids)) + 1, -1, np.int64)
            for cid, k in cam_pos.items():
                lut[cid] = k
            ocam = lut[cam_raw].astype(np.int32)
            opose = np.concatenate(self._obs_pose)
            opt = np.concatenate(self._obs_pt)
            ouv = np.concatenate(self._obs_uv, axis=0)
        else:
            ocam = opose = opt = np.zeros(0, np.int32)
            ouv = np.zeros((0, 2))
        return intr, camT, T_jw, X, pf, qf, ocam, opose, opt, ouv

    def FinalizeParameters(self):   # reference :182-206 (private there)
        if self.is_parameter_finalized_:
            return
        intr, camT, T_jw, X, pf, qf, ocam, opose, opt, ouv = self._host_arrays()
        p = BaProblem(self.device)
        p.set_cameras(intr, camT)
        p.set_poses(T_jw, pf)
        p.set_points(X, qf)
        p.set_observations(ocam, opose, opt, ouv)
        if self._shard[1] > 1:
            p.set_shard(*self._shard)
        if self._stream is not None:
            p.set_stream(self._stream)
        p.finalize()
        if self._allreduce is not None:
            if hasattr(self._allreduce, "attach"):   # an exchange object (sharding.py)
                self._allreduce.attach(p)
            else:                                    # hook(which, ptr, n, stream) -> int
                p.set_allreduce(self._allreduce)
        self._problem = p
        self._connectivity_input = (pf, qf, opose, opt)
        self.num_optimization_poses_ = int((pf == 0).sum())
        self.num_optimization_points_ = int((qf == 0).sum())
        self.is_parameter_finalized_ = True

    def CheckPoseAndPointConnectivity(self):   # reference :310-341
        """stderr warnings for an optimisable pose that observes fewer than 5
        distinct points and for an optimisable point seen from fewer than 2
        distinct poses (fixed ones count, reference :684-693).  Indices are the
        optimisation indices (here: input order of the non-fixed entries; the
        reference's are hash-order)."""
        if self.is_parameter_finalized_:
            pf, qf, opose, opt = self._connectivity_input
        else:
            _, _, _, _, pf, qf, _, opose, opt, _ = self._host_arrays()
        n_pt = max(1, self.num_total_points_)
        pairs = np.unique(opose.astype(np.int64) * n_pt + opt.astype(np.int64))
        pts_of_pose = np.bincount(pairs // n_pt, minlength=self.num_total_poses_)
        poses_of_pt = np.bincount(pairs % n_pt, minlength=self.num_total_points_)
        j_opt = np.cumsum(pf == 0) - 1
        i_opt = np.cumsum(qf == 0) - 1
        for h in np.nonzero((pf == 0) & (pts_of_pose < 5))[0]:
            sys.stderr.write(_yellow(
                "%d-th pose: It might diverge because some frames have "
                "insufficient related points." % j_opt[h]) + "\n")
        for h in np.nonzero((qf == 0) & (poses_of_pt < 2))[0]:
            sys.stderr.write(_yellow(
                "%d-th point: It might diverge because some points have "
                "insufficient related poses." % i_opt[h]) + "\n")

    def GetSolverStatistics(self):   # reference :208-239 (returns "", Q8)
        print("| Bundle Adjustment Statistics:")
        print("| # cameras in rigid body system: %d" %
              len(self.camera_id_to_camera_map_))
        print("|             # of total poses: %d" % self.num_total_poses_)
        print("|               - # fix  poses: %d" % self.num_fixed_poses_)
        print("|               - # opt. poses: %d" %
              self.num_optimization_poses_)
        print("|            # of total points: %d" % self.num_total_points_)
        print("|              - # fix  points: %d" % self.num_fixed_points_)
        print("|              - # opt. points: %d" %
              self.num_optimization_points_)
        print("|            # of observations: %d" %
              self.num_total_observations_)
        print("|                Jacobian size: %d rows x %d cols" %
              (6 * self.num_total_observations_,
               3 * self.num_optimization_points_ +
               6 * self.num_optimization_poses_))
        print("|                Residual size: %d rows\n" %
              (2 * self.num_total_observations_))
        return ""

    def _use_gauss_newton(self, options):
        """FullBundleAdjustmentSolver::Solve ignores options.solver_type and is
        always Levenberg-Marquardt."""
        return False

    def Solve(self, options, summary=None):             # reference :630-1044
        t0 = time.perf_counter()
        if summary is not None:
            summary.max_iteration_ = options.iteration_handle.max_num_iterations
            summary.threshold_cost_change_ = \
                options.convergence_handle.threshold_cost_change
            summary.threshold_step_size_ = \
                options.convergence_handle.threshold_step_size
            summary.convergence_status_ = True
        self.FinalizeParameters()
        if self.verbose:
            self.GetSolverStatistics()
        self.CheckPoseAndPointConnectivity()            # reference :703
        p = self._problem
        c_opt = options.to_c()
        c_opt.gauss_newton = 1 if self._use_gauss_newton(options) else 0
        rows, converged = p.solve(c_opt)
        # write back through the user's objects (reference :1011-1022)
        T_jw = p.get_poses()
        T44 = _T12_to_44(T_jw)
        T44[:, :3, 3] *= INVERSE_SCALER
        T_wj = rigid_inverse(T44)
        for h, (obj, row) in enumerate(self._pose_objs):
            if h in self._pose_fixed:
                continue
            if row is None:
                obj[...] = T_wj[h]
            else:
                obj[row] = T_wj[h]
        if self._shard[1] > 1 and self._allreduce is not None:
            # every rank writes back EVERY point (reference :1018-1022): one final
            # sum-all-reduce of the owned rows
            p.gather_points()
        X, owned = p.get_points()
        Xu = X * INVERSE_SCALER
        opt_mask = np.ones(self.num_total_points_, bool)
        opt_mask[list(self._pt_fixed)] = False
        opt_mask &= owned
        for obj, base, cnt in self._pt_objs:
            if cnt == 0:      # single AddPoint object
                if opt_mask[base]:
                    obj[...] = Xu[base].reshape(np.shape(obj))
            else:             # AddPointArray block
                m = opt_mask[base:base + cnt]
                obj[m] = Xu[base:base + cnt][m]
        if summary is not None:
            for r in rows:
                info = OptimizationInfo()
                info.cost = r.cost
                info.cost_change = r.cost_change
                info.average_reprojection_error = r.average_reprojection_error
                info.abs_gradient = r.abs_gradient
                info.abs_step = r.abs_step
                info.damping_term = r.damping_term
                info.iter_time = r.iter_time_ms
                info.iteration_status = IterationStatus(r.iteration_status)
                info.rho, info.model_change, info.trial_cost = \
                    r.rho, r.model_change, r.trial_cost
                summary.optimization_info_list_.append(info)
            summary.convergence_status_ = converged
            summary.total_time_in_millisecond_ = \
                (time.perf_counter() - t0) * 1e3
        return True   # the reference always returns true (:1043)


class PoseOnlyBundleAdjustmentSolver:
    """Mirror of reference core/pose_only_bundle_adjustment_solver.h:25-67
    (monocular and stereo 6-DoF entry points)."""

    def __init__(self, device=0):
        self._p = BaProblem(device)
        self.debug_poses_ = []

    def GetDebugPoses(self):
        return self.debug_poses_

    def Solve_Monocular_6Dof(self, reference_position_list, matched_pixel_list,
                             fx, fy, cx, cy, reference_to_current_pose,
                             mask_inlier, options, summary=None):
        """reference core/pose_only_bundle_adjustment_solver.cpp:8-170.
        `reference_to_current_pose` is a 4x4 float array updated in place;
        `mask_inlier` a list/array resized to n (True) and updated in place.
        """
        t0 = time.perf_counter()
        X = np.asarray(reference_position_list, np.float32).reshape(-1, 3)
        uv = np.asarray(matched_pixel_list, np.float32).reshape(-1, 2)
        if X.shape[0] != uv.shape[0]:
            raise RuntimeError(
                "In PoseOnlyBundleAdjustmentSolver::"
                "SolveMonocularPoseOnlyBundleAdjustment6Dof(), "
                "world_position_list.size() != current_pixel_list.size()")
        n = X.shape[0]
        if summary is not None:
            summary.max_iteration_ = options.iteration_handle.max_num_iterations
            summary.threshold_cost_change_ = \
                options.convergence_handle.threshold_cost_change
            summary.threshold_step_size_ = \
                options.convergence_handle.threshold_step_size
            summary.convergence_status_ = True
        m = np.ones(n, np.uint8)
        k = min(len(mask_inlier), n)
        m[:k] = np.asarray(mask_inlier[:k], np.uint8)
        T12 = _T44_to_12(np.asarray(reference_to_current_pose,
                                    np.float64)).astype(np.float32)
        res = self._p.pose_only_mono6(X, uv, fx, fy, cx, cy, T12, m,
                                      options.to_c(), want_debug=True)
        self.debug_poses_ = [_T12_to_44(d)[0] for d in res["debug"]]
        if isinstance(mask_inlier, list):
            mask_inlier[:] = [bool(v) for v in res["mask"]]
        else:
            mask_inlier[...] = res["mask"]
        if res["success"]:
            reference_to_current_pose[...] = _T12_to_44(res["T12"])[0]
        if summary is not None:
            for cost, dchg, step in res["rows"]:
                info = OptimizationInfo()
                info.cost = cost
                info.cost_change = abs(dchg)
                info.average_reprojection_error = cost
                info.abs_step = step
                info.abs_gradient = 0
                info.damping_term = -1
                info.iter_time = 0.0
                info.iteration_status = IterationStatus.UPDATE
                summary.optimization_info_list_.append(info)
            summary.convergence_status_ = res["converged"]
            summary.total_time_in_millisecond_ = \
                (time.perf_counter() - t0) * 1e3
        return res["success"]

    def Solve_Stereo_6Dof(self, reference_position_list, matched_left_pixel_list,
                          matched_right_pixel_list, fx_left, fy_left, cx_left,
                          cy_left, fx_right, fy_right, cx_right, cy_right,
                          left_to_right_pose, reference_to_current_left_pose,
                          mask_inlier_left, mask_inlier_right, options,
                          summary=None):
        """reference core/pose_only_bundle_adjustment_solver.cpp:172-399.
        Poses are 4x4 arrays (the left pose is updated in place), the masks
        lists/arrays resized to n (True) and updated in place; a right pixel
        with a negative coordinate means "not matched in the right image"."""
        t0 = time.perf_counter()
        X = np.asarray(reference_position_list, np.float32).reshape(-1, 3)
        ul = np.asarray(matched_left_pixel_list, np.float32).reshape(-1, 2)
        ur = np.asarray(matched_right_pixel_list, np.float32).reshape(-1, 2)
        n = X.shape[0]
        if summary is not None:
            summary.max_iteration_ = options.iteration_handle.max_num_iterations
            summary.threshold_cost_change_ = \
                options.convergence_handle.threshold_cost_change
            summary.threshold_step_size_ = \
                options.convergence_handle.threshold_step_size
            summary.convergence_status_ = True

        def fit(mask):
            m = np.ones(n, np.uint8)
            k = min(len(mask), n)
            m[:k] = np.asarray(mask[:k], np.uint8)
            return m

        to12 = lambda T: _T44_to_12(np.asarray(T, np.float64)).astype(np.float32)
        res = self._p.pose_only_stereo6(
            X, ul, ur, [fx_left, fy_left, cx_left, cy_left],
            [fx_right, fy_right, cx_right, cy_right], to12(left_to_right_pose),
            to12(reference_to_current_left_pose), fit(mask_inlier_left),
            fit(mask_inlier_right), options.to_c(), want_debug=True)
        self.debug_poses_ = [_T12_to_44(d)[0] for d in res["debug"]]
        for mask, key in ((mask_inlier_left, "mask_l"), (mask_inlier_right, "mask_r")):
            if isinstance(mask, list):
                mask[:] = [bool(v) for v in res[key]]
            else:
                mask[...] = res[key]
        if res["success"]:
            reference_to_current_left_pose[...] = _T12_to_44(res["T12"])[0]
        if summary is not None:
            for cost, dchg, step in res["rows"]:
                info = OptimizationInfo()
                info.cost = cost
                info.cost_change = abs(dchg)
                info.average_reprojection_error = cost
                info.abs_step = step
                info.abs_gradient = 0
                info.damping_term = -1
                info.iter_time = 0.0
                info.iteration_status = IterationStatus.UPDATE
                summary.optimization_info_list_.append(info)
            summary.convergence_status_ = res["converged"]
            summary.total_time_in_millisecond_ = \
                (time.perf_counter() - t0) * 1e3
        return res["success"]


class FullBundleAdjustmentSolverRefactor(FullBundleAdjustmentSolver):
    """Mirror of reference core/full_bundle_adjustment_solver_refactor.h:
    117-136: the same device path behind the refactored names, plus the
    solver_type switch of its Solve (reference ..._refactor.cpp:944-982):
    LEVENBERG_MARQUARDT, or GAUSS_NEWTON = every step accepted with lambda fixed
    at initial_lambda (the default of Options)."""

    def RegisterCamera(self, camera_id, camera):
        return self.AddCamera(camera_id, camera)

    def RegisterWorldToBodyPose(self, original_pose):
        return self.AddPose(original_pose)

    def RegisterWorldPoint(self, original_point):
        return self.AddPoint(original_point)

    def _use_gauss_newton(self, options):
        if options.solver_type == SolverType.LEVENBERG_MARQUARDT:
            return False
        if options.solver_type == SolverType.GAUSS_NEWTON:
            return True
        raise RuntimeError("FullBundleAdjustmentSolverRefactor::Solve: "
                           "solver_type must be GAUSS_NEWTON or "
                           "LEVENBERG_MARQUARDT")

    def SolveByGradientDescent(self, options, summary=None):
        raise NotImplementedError(
            "SolveByGradientDescent (reference ..._refactor.cpp:1073-1370) is "
            "not on the MI355X hot path")
