"""Developer aid: where ba_finalize spends its time (BA_PLAN_TIMES laps) at a bench
configuration.  python tools/finalize_probe.py [C4 | C4@12 (scale)] [passes ...]
(BA_PLAN_THREADS is read once per process: set it in the environment.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BA_PLAN_TIMES"] = "1"
from bundle_adjustment_solver_amd import scenes  # noqa: E402
from bundle_adjustment_solver_amd.solver import BaProblem  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"   # "C4" or "C4@12" (scale)
threads = sys.argv[2:] or ["16"]
scale = float(cfg.split("@")[1]) if "@" in cfg else 1.0
pr = scenes.scaled_problem(scenes.config_scene(cfg.split("@")[0], scale))
for rep, nt in enumerate(["16"] + threads):  # the first pass warms the allocator / page cache
    os.environ["BA_PLAN_THREADS"] = nt
    p = BaProblem(0)
    p.set_cameras(pr["cam_intr"], pr["cam_T"])
    p.set_poses(pr["pose_T"], pr["pose_fixed"])
    p.set_points(pr["pt_X"], pr["pt_fixed"])
    t0 = time.time()
    p.set_observations(pr["obs_cam"], pr["obs_pose"], pr["obs_pt"], pr["obs_uv"])
    t1 = time.time()
    p.finalize()
    t2 = time.time()
    print("== %s threads %s%s: set_observations %.3f s, finalize %.3f s" %
          (cfg, nt, " (warm-up pass)" if rep == 0 else "", t1 - t0, t2 - t1), flush=True)
    del p
