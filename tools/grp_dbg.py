"""Developer probe: per-phase time stamps of k_schur_grp (library built with -DBA_GRP_DBG)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_solver_amd import scenes, _lib
from bundle_adjustment_solver_amd.solver import BaProblem
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
sc = scenes.config_scene(cfg)
pr = scenes.scaled_problem(sc)
p = BaProblem(0)
p.set_cameras(pr["cam_intr"], pr["cam_T"]); p.set_poses(pr["pose_T"], pr["pose_fixed"])
p.set_points(pr["pt_X"], pr["pt_fixed"])
p.set_observations(pr["obs_cam"], pr["obs_pose"], pr["obs_pt"], pr["obs_uv"])
p.finalize()
for _ in range(3):
    p.stage_linearize(100.0, 1.0)
    p.stage_schur()
lib = _lib.load()
out = (ctypes.c_longlong * 256)()
lib.ba_debug_read_grp.argtypes = [ctypes.c_void_p]
print("rc", lib.ba_debug_read_grp(out))
t = np.array(out[:])
n = int((t != 0).sum())
print("stamps", n, "total cycles", t[n - 1] - t[0])
d = np.diff(t[:n])
if cfg == "W20":  # k_schur_grp_wide: start, [staged, barrier, mfma, barrier] per stage, scatter begin / end
    print("  prologue %d" % d[0])
    st = d[1:-2]
    for c in range(len(st) // 6):
        print("  stage %2d: " % c + "  ".join("%s %6d" % (nm, st[c * 6 + k]) for k, nm in enumerate(["stage", "prefetch", "lds drain", "barrier", "mfma issue", "barrier"])))
    print("  to scatter %d  scatter %d" % (d[-2], d[-1]))
    sys.exit(0)
names = ["Vphase", "prefetch", "mfma", "next"]
for c in range((n - 1) // 4):
    print("  chunk %2d: " % c + "  ".join("%s %6d" % (names[k], d[c * 4 + k]) for k in range(4)))
