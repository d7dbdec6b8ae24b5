"""Developer tool: device memory before/after create/finalize/solve/destroy cycles."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bundle_adjustment_solver_amd import scenes
from bundle_adjustment_solver_amd._lib import make_options
from bundle_adjustment_solver_amd.solver import BaProblem
from oracle import oracle_py as O

def free_bytes():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0]

sc = scenes.synthetic_ba_scene(30, 4000, 5, True, seed=3)
pr = scenes.scaled_problem(sc)
opt = make_options(max_iter=4, thr_step=0, thr_cost=0)

def make(finalize=True):
    p = BaProblem(0)
    if finalize:
        p.set_cameras(pr["cam_intr"], pr["cam_T"]); p.set_poses(pr["pose_T"], pr["pose_fixed"])
        p.set_points(pr["pt_X"], pr["pt_fixed"])
        p.set_observations(pr["obs_cam"], pr["obs_pose"], pr["obs_pt"], pr["obs_uv"])
        p.finalize()
    return p

def po():
    p = make(False)
    p.pose_only_mono6(np.random.rand(100, 3).astype(np.float32) + [0, 0, 2],
                      np.random.rand(100, 2).astype(np.float32) * 100, 300, 300, 320, 240,
                      np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], np.float32),
                      np.ones(100, np.uint8), opt)
    p.close()

def both():
    p = make()
    p.solve(opt)
    p.pose_only_mono6(np.random.rand(100, 3).astype(np.float32) + [0, 0, 2],
                      np.random.rand(100, 2).astype(np.float32) * 100, 300, 300, 320, 240,
                      np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], np.float32),
                      np.ones(100, np.uint8), opt)
    p.close()

for name, fn in [("solve+pose-only", both), ("create/destroy", lambda: make(False).close()), ("pose-only", po),
                 ("finalize", lambda: make().close()),
                 ("finalize+solve", lambda: (lambda p: (p.solve(opt), p.close()))(make()))]:
    fn()
    b = free_bytes()
    for _ in range(20):
        fn()
    a = free_bytes()
    for _ in range(20):
        fn()
    a2 = free_bytes()
    print("%-16s retained %.2f MB after 20 cycles, %.2f MB after 40" % (name, (b - a) / 2**20, (b - a2) / 2**20))
