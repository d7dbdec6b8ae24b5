"""Developer probe: streamed vs resident first trial costs for several chunk counts."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_solver_amd import scenes
from bundle_adjustment_solver_amd._lib import make_options
from bundle_adjustment_solver_amd.solver import BaProblem, BaStream
sc = scenes.synthetic_ba_scene(30, 2000, 5, True, seed=23, pixel_sigma=0.3)
pr = scenes.scaled_problem(sc)
def load(p):
    p.set_cameras(pr["cam_intr"], pr["cam_T"]); p.set_poses(pr["pose_T"], pr["pose_fixed"])
    p.set_points(pr["pt_X"], pr["pt_fixed"])
    p.set_observations(pr["obs_cam"], pr["obs_pose"], pr["obs_pt"], pr["obs_uv"]); p.finalize(); return p
opt = make_options(max_iter=4, thr_step=0, thr_cost=0)
frows, _ = load(BaProblem(0)).solve(opt)
print("resident", ["%.9g" % r.trial_cost for r in frows], ["%.9g" % r.cost for r in frows])
for K in [int(a) for a in sys.argv[1:]] or [2, 3, 4, 5]:
    st = load(BaStream(0, K, 64 << 20))
    rows, _ = st.solve(opt)
    print("K", K, ["%.9g" % r.trial_cost for r in rows], ["%.9g" % r.cost for r in rows], st.info())
    # sharded resident handles, summed on the host with the stage API
    sh = []
    for k in range(K):
        p = BaProblem(0); p.set_cameras(pr["cam_intr"], pr["cam_T"]); p.set_poses(pr["pose_T"], pr["pose_fixed"])
        p.set_points(pr["pt_X"], pr["pt_fixed"]); p.set_observations(pr["obs_cam"], pr["obs_pose"], pr["obs_pt"], pr["obs_uv"])
        p.set_shard(k, K); p.finalize(); sh.append(p)
    print("   sum of the shards' stage costs %.9g  (resident %.9g)" % (sum(p.stage_cost() for p in sh), load(BaProblem(0)).stage_cost()))
