"""Micro-benchmark of the dense fp64-MFMA Cholesky solve alone
(ba_dense_spd_solve): n x n random SPD system, device time from hipEvents."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_solver_amd.solver import BaProblem  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=5970)
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
rng = np.random.default_rng(0)
n = args.n
t = time.time()
Q = rng.standard_normal((n, 256))
A = Q @ Q.T + n * np.eye(n) + np.diag(rng.uniform(0, 1, n))
b = rng.standard_normal(n)
g = BaProblem(0)
ms = []
for r in range(args.reps):
    x, m = g.dense_spd_solve(A, b)
    ms.append(m)
res = np.abs(A @ x - b).max() / np.abs(b).max()
fl = n ** 3 / 3.0 + 4.0 * n * n
best = min(ms)
print("n=%d  ms=%s  best %.3f ms  %.2f TFLOP/s  residual %.2e  (setup %.1fs)" %
      (n, ["%.3f" % v for v in ms], best, fl / best / 1e9, res, time.time() - t))
