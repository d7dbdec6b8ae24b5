"""Developer probe: the run-away regime on the 16-pose tail of C4 (tests/test_gpu_parity.py::
test_thresholds_off_runaway_regime_matches_oracle): GPU and oracle rows side by side around
the first status mismatch.   python tools/runaway_sub_probe.py [iters]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_solver_amd import scenes
from bundle_adjustment_solver_amd._lib import make_options
from bundle_adjustment_solver_amd.solver import BaProblem
from oracle import oracle_py as O

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 320
sc = scenes.pose_window_subscene(scenes.config_scene("C4"), 984, 1000)
pr = scenes.scaled_problem(sc)
p = BaProblem(0)
p.set_cameras(pr["cam_intr"], pr["cam_T"]); p.set_poses(pr["pose_T"], pr["pose_fixed"])
p.set_points(pr["pt_X"], pr["pt_fixed"])
p.set_observations(pr["obs_cam"], pr["obs_pose"], pr["obs_pt"], pr["obs_uv"])
p.finalize()
kw = dict(max_iter=iters, thr_step=-1.0, thr_cost=-1.0)
rows, _ = p.solve(make_options(**kw))
o = O.Oracle(pr)
orows, _ = o.solve(O.make_options(**kw))
mis = next((k for k, (a, b) in enumerate(zip(rows, orows)) if a.iteration_status != b.iteration_status), None)
print("first status mismatch:", mis)
lo, hi = (max(0, mis - 8), min(iters, mis + 8)) if mis is not None else (iters - 10, iters)
for k in range(lo, hi):
    a, b = rows[k], orows[k]
    print("%3d gpu st %d lam %.3g trial %.12g step %.4g rho %.4g model %.4g | cpu st %d lam %.3g trial %.12g step %.4g rho %.4g model %.4g"
          % (k, a.iteration_status, a.damping_term, a.trial_cost, a.abs_step, a.rho, a.model_change,
             b.iteration_status, b.damping_term, b.trial_cost, b.abs_step, b.rho, b.model_change))
X, oX = p.get_points()[0], o.get_points()
n, on = np.linalg.norm(X, axis=1), np.linalg.norm(oX, axis=1)
i = int(np.nanargmax(on))
print("farthest oracle landmark", i, oX[i], "gpu", X[i], " non-finite gpu points", int((~np.isfinite(X).all(1)).sum()))
if "--stages" in sys.argv:
    # where does the first non-finite value appear?  (state = accepted point after `iters`)
    lam = rows[-1].damping_term
    p.stage_linearize(lam, 1.0)
    A, a = p.get_A(); Cc, b = p.get_C(); Ci, Cib = p.get_Cinv(); pi, pj, W = p.get_pairs()
    for name, v in (("A", A), ("a", a), ("C", Cc), ("b", b), ("Cinv", Ci), ("Cinv b", Cib), ("W", W)):
        print(name, "finite:", bool(np.isfinite(v).all()), "max |.| %.3g" % np.nanmax(np.abs(v)), "min nonzero |.| %.3g" % np.min(np.abs(v[v != 0])) if (v != 0).any() else "")
    p.stage_schur(); S, rhs = p.get_S()
    print("S finite:", bool(np.isfinite(S).all()), "rhs finite:", bool(np.isfinite(rhs).all()))
    p.stage_solve_reduced(); p.stage_backsub_update(); x, y = p.get_xy()
    print("x finite:", bool(np.isfinite(x).all()), "y finite:", bool(np.isfinite(y).all()), "max |y| %.3g" % np.nanmax(np.abs(y)))
    bad = np.nonzero(~np.isfinite(y).all(1))[0]
    print("non-finite y rows:", bad[:10])
    print("scalars (trial cost, model, pose step, point step):", p.stage_scalars())
    o.linearize(1.0); o.damp_invert(lam); o.schur(); o.solve_reduced(); o.backsub()
    ox, oy = o.get_xy()
    print("oracle max |y| %.3g" % np.abs(oy).max(), " oracle C of the far landmark", o.get_C()[0][i].ravel(), "gpu", Cc[i].ravel())
    print("oracle Cinv", o.get_Cinv()[0][i].ravel(), "gpu", Ci[i].ravel())
    print("oracle y", oy[i], "gpu", y[i])
