"""Developer probe: reduced system with NO fixed pose and lambda at its floor
(gauge freedom -> S numerically singular): HIP Cholesky (dropped pivots
counted) vs the oracle's pivoted LDLT."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bundle_adjustment_solver_amd import scenes
from bundle_adjustment_solver_amd.solver import BaProblem
from oracle import oracle_py as O

for stereo in (True, False):
    sc = scenes.synthetic_ba_scene(20, 600, 5 if stereo else 10, stereo, seed=31, n_fixed=0,
                                   pose_noise=0.02, point_noise=0.05)
    pr = scenes.scaled_problem(sc)
    for lam in (1e-10, 1e-6, 1e-3):
        g = BaProblem(0)
        g.set_cameras(pr["cam_intr"], pr["cam_T"]); g.set_poses(pr["pose_T"], pr["pose_fixed"])
        g.set_points(pr["pt_X"], pr["pt_fixed"])
        g.set_observations(pr["obs_cam"], pr["obs_pose"], pr["obs_pt"], pr["obs_uv"]); g.finalize()
        o = O.Oracle(pr)
        o.linearize(1.0); o.damp_invert(lam); o.schur(); o.solve_reduced(); o.backsub()
        g.stage_linearize(lam, 1.0); g.stage_schur(); g.get_dropped_pivots(reset=True)
        g.stage_solve_reduced(); g.stage_backsub_update()
        S, rhs = o.get_S()
        x, y = g.get_xy(); ox, oy = o.get_xy()
        ev = np.linalg.eigvalsh(S)
        rg = np.abs(S @ x.reshape(-1) - rhs).max() / np.abs(rhs).max()
        ro = np.abs(S @ ox.reshape(-1) - rhs).max() / np.abs(rhs).max()
        o.backup(); o.update()
        print("stereo", stereo, "lam %g" % lam, "eig min/max %.3e %.3e" % (ev[0], ev[-1]),
              "dropped", g.get_dropped_pivots(), "res gpu %.2e oracle %.2e" % (rg, ro),
              "|x| gpu %.3e oracle %.3e" % (np.abs(x).max(), np.abs(ox).max()),
              "trial cost gpu %.6e oracle %.6e" % (g.stage_scalars()[0], o.cost()))
