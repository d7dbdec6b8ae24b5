"""Latency of the pose-only monocular 6-DoF path (config C5): one
ba_pose_only_mono6 call = H2D copy + one persistent GN kernel + D2H, vs the
single-threaded CPU oracle on the same inputs."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_solver_amd import scenes  # noqa: E402
from bundle_adjustment_solver_amd._lib import make_options  # noqa: E402
from bundle_adjustment_solver_amd.solver import BaProblem  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

g = BaProblem(0)
for n in (10_000, 300_000):
    sc = scenes.pose_only_scene(n, seed=2024)
    T12 = np.concatenate([sc["T_init"][:3, :3].reshape(9), sc["T_init"][:3, 3]])
    opt = make_options(max_iter=100, thr_step=1e-6, thr_cost=1e-6, huber=1.0,
                         outlier=2.5)
    oopt = O.make_options(max_iter=100, thr_step=1e-6, thr_cost=1e-6,
                          huber=1.0, outlier=2.5)
    ts = []
    for r in range(7):
        t = time.perf_counter()
        res = g.pose_only_mono6(sc["X"], sc["uv"], sc["fx"], sc["fy"], sc["cx"],
                                sc["cy"], T12, np.ones(n, np.uint8), opt)
        ts.append(time.perf_counter() - t)
    t = time.perf_counter()
    ref = O.pose_only_mono6(sc["X"], sc["uv"], sc["fx"], sc["fy"], sc["cx"],
                            sc["cy"], sc["T_init"], np.ones(n, np.uint8), oopt)
    tc = time.perf_counter() - t
    print("n=%d  iters gpu/cpu %d/%d  gpu call median %.3f ms (min %.3f)  "
          "cpu oracle %.3f ms  max|dT| %.2e" %
          (n, res["n_iter"], ref["n_iter"], np.median(ts) * 1e3,
           min(ts) * 1e3, tc * 1e3, np.abs(res["T12"] - ref["T12"]).max()))

# stereo 6-DoF (Solve_Stereo_6Dof path), 20 % of the points without a right match
for n in (10_000, 300_000):
    sc = scenes.pose_only_stereo_scene(n, seed=2025, right_missing_frac=0.2)
    intr = [sc["fx"], sc["fy"], sc["cx"], sc["cy"]]
    to12 = lambda T: np.concatenate([T[:3, :3].reshape(9), T[:3, 3]])
    opt = make_options(max_iter=100, thr_step=1e-6, thr_cost=1e-6, huber=1.0,
                         outlier=2.5)
    oopt = O.make_options(max_iter=100, thr_step=1e-6, thr_cost=1e-6,
                          huber=1.0, outlier=2.5)
    ones = np.ones(n, np.uint8)
    ts = []
    for r in range(7):
        t = time.perf_counter()
        res = g.pose_only_stereo6(sc["X"], sc["uv"], sc["uv_right"], intr, intr,
                                  to12(sc["T_lr"]), to12(sc["T_init"]), ones, ones, opt)
        ts.append(time.perf_counter() - t)
    t = time.perf_counter()
    ref = O.pose_only_stereo6(sc["X"], sc["uv"], sc["uv_right"], intr, intr,
                              sc["T_lr"], sc["T_init"], ones, ones, oopt)
    tc = time.perf_counter() - t
    print("stereo n=%d  iters gpu/cpu %d/%d  gpu call median %.3f ms (min %.3f)  "
          "cpu oracle %.3f ms  max|dT| %.2e" %
          (n, res["n_iter"], ref["n_iter"], np.median(ts) * 1e3,
           min(ts) * 1e3, tc * 1e3, np.abs(res["T12"] - ref["T12"]).max()))
