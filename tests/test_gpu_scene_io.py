"""A problem read from a BAL text file, solved by the HIP
path through the C ABI and through the reference-style facade, against the CPU
oracle.  The fixture has 12 BAL cameras = 12 solver cameras (more than the 8
that fit the LDS camera table) with radial distortion removed at load time."""
import os

import numpy as np
import pytest

from bundle_adjustment_solver_amd import (Camera, FullBundleAdjustmentSolver, Options,
                                          Summary, scene_io, scenes)
from bundle_adjustment_solver_amd._lib import make_options
from bundle_adjustment_solver_amd.solver import BaProblem
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu

BAL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bal_small.txt")


def relerr(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_bal_problem_matches_oracle(built):
    ld = scene_io.load_bal(BAL, n_fixed_poses=2)
    pr = scenes.scaled_problem(ld)
    g = BaProblem(0)
    g.set_cameras(pr["cam_intr"], pr["cam_T"])
    g.set_poses(pr["pose_T"], pr["pose_fixed"])
    g.set_points(pr["pt_X"], pr["pt_fixed"])
    g.set_observations(pr["obs_cam"], pr["obs_pose"], pr["obs_pt"], pr["obs_uv"])
    g.finalize()
    o = O.Oracle(pr)
    assert relerr(g.stage_cost(), o.cost()) < 1e-12
    kw = dict(max_iter=20, thr_step=1e-8, thr_cost=1e-8)
    rows, conv = g.solve(make_options(**kw))
    orows, oconv = o.solve(O.make_options(**kw))
    assert len(rows) == len(orows) and conv == oconv
    for k, (a, b) in enumerate(zip(rows, orows)):
        assert a.iteration_status == b.iteration_status, k
        assert relerr(a.cost, b.cost) < 1e-7, k
    assert relerr(g.get_poses(), o.get_poses()) < 1e-6      # fp64, tolerance 1e-4 in north_star
    assert relerr(g.get_points()[0], o.get_points()) < 1e-6
    assert rows[-1].cost < 0.05 * rows[0].cost


def test_bal_problem_through_the_facade(built, tmp_path):
    """load_bal -> AddCamera/AddPose/AddPoint/AddObservation -> Solve -> the
    optimised scene written back with save_bal has smaller residuals."""
    ld = scene_io.load_bal(BAL, n_fixed_poses=2)
    s = FullBundleAdjustmentSolver()
    n = ld["intr"].shape[0]
    for c in range(n):
        s.AddCamera(c, Camera(*ld["intr"][c].astype(np.float32),
                              pose_this_to_cam0=ld["T_cj"][c].astype(np.float32)))
    poses = [ld["T_wc_init"][k].astype(np.float32) for k in range(n)]
    pts = [ld["X_init"][k].astype(np.float32) for k in range(ld["X_init"].shape[0])]
    for T in poses:
        s.AddPose(T)
    for X in pts:
        s.AddPoint(X)
    for k in range(2):
        s.MakePoseFixed(poses[k])
    for c, j, i, uv in zip(ld["obs_cam"], ld["obs_pose"], ld["obs_pt"], ld["obs_uv"]):
        s.AddObservation(int(c), poses[j], pts[i], uv.astype(np.float32))
    opt = Options()
    opt.iteration_handle.max_num_iterations = 20
    summ = Summary()
    assert s.Solve(opt, summ)
    out = dict(ld)
    out["T_wc_opt"] = np.stack(poses).astype(np.float64)
    out["X_opt"] = np.stack(pts).astype(np.float64)
    r0 = np.abs(scene_io.reprojection_residuals(ld)).sum()
    r1 = np.abs(scene_io.reprojection_residuals(out, "T_wc_opt", "X_opt")).sum()
    assert r1 < 0.05 * r0
    path = str(tmp_path / "optimised.txt")
    scene_io.save_bal(path, out, poses="T_wc_opt", points="X_opt")
    back = scene_io.load_bal(path)
    r2 = np.abs(scene_io.reprojection_residuals(back)).sum()
    assert abs(r2 - r1) < 1e-3 * r1 + 1e-6
