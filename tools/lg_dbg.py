"""Developer tool: phase time stamps of k_lin_grp (workgroup 300, wave 0).
Needs a library built with -DBA_LG_DBG (make dbg DBGFLAGS=-DBA_LG_DBG; BA_HIP_LIB=<path>)."""
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
    p.stage_linearize(100.0, 1.0)
lib = _lib.load()
out = (ctypes.c_longlong * 64)()
lib.ba_debug_read_lg.argtypes = [ctypes.c_void_p]
print("rc", lib.ba_debug_read_lg(out))
t = np.array(out[:]); n = int((t != 0).sum()); d = np.diff(t[:n])
print("stamps", n, "total", t[n - 1] - t[0])
print("setup (entry -> first step)", d[0])
body = d[1:n - 2]
for k in range(0, len(body) - 2, 3):
    print("  step %2d: compute %6d   sums + C/b store %6d   W copy-out %6d" % (k // 3, body[k], body[k + 1], body[k + 2]))
print("tail:", d[n - 2:])
