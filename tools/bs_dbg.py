"""Developer tool: phase time stamps of k_backsub_update (workgroup 2000).
Needs libba_hip.so built with -DBA_BS_DBG."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_solver_amd import scenes, _lib
from bundle_adjustment_solver_amd.solver import BaProblem
sc = scenes.config_scene("C4")
pr = scenes.scaled_problem(sc)
p = BaProblem(0)
p.set_cameras(pr["cam_intr"], pr["cam_T"]); p.set_poses(pr["pose_T"], pr["pose_fixed"])
p.set_points(pr["pt_X"], pr["pt_fixed"])
p.set_observations(pr["obs_cam"], pr["obs_pose"], pr["obs_pt"], pr["obs_uv"])
p.finalize()
for _ in range(3):
    p.stage_linearize(100.0, 1.0); p.stage_schur(); p.stage_solve_reduced(); p.stage_backsub_update()
lib = _lib.load()
out = (ctypes.c_longlong * 96)()
lib.ba_debug_read_bs.argtypes = [ctypes.c_void_p]
print("rc", lib.ba_debug_read_bs(out))
t = np.array(out[:]); n = int((t != 0).sum()); d = np.diff(t[:n])
names = ["recs+issue"] + ["own issue", "xgather+lds_wr", "prefetch next", "bar1", "u-phase", "bar2", "own sum", "tail", "block_sums"] * 4
print("total", t[n - 1] - t[0])
for k in range(n - 1):
    print("  %-16s %6d" % (names[k] if k < len(names) else "?", d[k]))
