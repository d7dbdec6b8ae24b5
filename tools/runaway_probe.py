"""Developer probe of the thresholds-off regime (bench.py's): run `iters` LM iterations
of a config on the GPU (and, with --oracle, on the CPU oracle in fast-solve mode) and
print status / lambda / cost / abs_step every `every` iterations and around the first
non-finite trial cost.   python tools/runaway_probe.py C4 1.0 260 [--oracle]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_solver_amd import scenes
from bundle_adjustment_solver_amd._lib import make_options
from bundle_adjustment_solver_amd.solver import BaProblem

cfg, scale, iters = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
pr = scenes.scaled_problem(scenes.config_scene(cfg, scale))
p = BaProblem(0)
p.set_cameras(pr["cam_intr"], pr["cam_T"]); p.set_poses(pr["pose_T"], pr["pose_fixed"])
p.set_points(pr["pt_X"], pr["pt_fixed"])
p.set_observations(pr["obs_cam"], pr["obs_pose"], pr["obs_pt"], pr["obs_uv"])
p.finalize()
rows, _ = p.solve(make_options(max_iter=iters, thr_step=-1.0, thr_cost=-1.0))
orows = None
if "--oracle" in sys.argv:
    from oracle import oracle_py as O
    o = O.Oracle(pr); o.set_fast_solve(True)
    orows, _ = o.solve(O.make_options(max_iter=iters, thr_step=-1.0, thr_cost=-1.0))
first = next((k for k, r in enumerate(rows) if not np.isfinite(r.trial_cost)), None)
print("first non-finite trial cost on the GPU:", first, " dropped pivots", p.get_dropped_pivots())
show = set(range(0, len(rows), max(1, iters // 40)))
if first is not None:
    show |= set(range(max(0, first - 4), min(len(rows), first + 5)))
for k in sorted(show):
    r = rows[k]
    line = "%4d gpu st %d lam %.4g cost %.10g trial %.10g step %.4g rho %.3g" % (
        k, r.iteration_status, r.damping_term, r.cost, r.trial_cost, r.abs_step, r.rho)
    if orows:
        q = orows[k]
        line += " | cpu st %d lam %.4g cost %.10g trial %.10g step %.4g" % (
            q.iteration_status, q.damping_term, q.cost, q.trial_cost, q.abs_step)
    print(line)
X = p.get_points()[0]
print("max |X| on the GPU: %.4g (scaled units)" % np.abs(X[np.isfinite(X).all(1)]).max(),
      " non-finite points:", int((~np.isfinite(X).all(1)).sum()))
