"""CPU tests pinning the ORACLE (oracle/ba_oracle.cpp).

The reference ships no golden vectors and cannot be built here (SURVEY.md
§8c) -> PARITY UNPINNED by the reference.  What pins the oracle instead:
  * its analytic Jacobian blocks against central finite differences of an
    independent numpy projection (reference formulas :743-831);
  * its Eigen-style pivoted LDLT against numpy (incl. the pseudo-inverse rule
    for zero pivots, reference :854 / SURVEY Q6);
  * the documented quirks Q1 (B_ji overwrite), Q2 (previous_cost advances on
    SKIPPED), Q6;
  * convergence to ground truth on noise-free scenes;
  * the committed golden fixtures (tests/golden/, self-generated regression
    vectors — see tests/golden/make_golden.py).
"""
import json
import os

import numpy as np
import pytest

from bundle_adjustment_solver_amd import scenes
from oracle import oracle_py as O

HERE = os.path.dirname(os.path.abspath(__file__))


def so3_exp(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th**2 * K @ K


def se3_exp(xi):
    v, w = xi[:3], xi[3:]
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = so3_exp(w)
    if th < 1e-12:
        V = np.eye(3) + 0.5 * K
    else:
        V = (np.eye(3) + (1 - np.cos(th)) / th**2 * K +
             (th - np.sin(th)) / th**3 * K @ K)
    return R, V @ v


def residual(pr, k, T12=None, X=None):
    cam = pr["obs_cam"][k]
    T = pr["pose_T"][pr["obs_pose"][k]] if T12 is None else T12
    Xi = pr["pt_X"][pr["obs_pt"][k]] if X is None else X
    R, t = T[:9].reshape(3, 3), T[9:]
    Rc, tc = pr["cam_T"][cam][:9].reshape(3, 3), pr["cam_T"][cam][9:]
    fx, fy, cx, cy = pr["cam_intr"][cam]
    Xc = Rc @ (R @ Xi + t) + tc
    return np.array([fx * Xc[0] / Xc[2] + cx, fy * Xc[1] / Xc[2] + cy]) \
        - pr["obs_uv"][k]


def numeric_jacobians(pr, k, h=1e-6):
    T = pr["pose_T"][pr["obs_pose"][k]]
    X = pr["pt_X"][pr["obs_pt"][k]]
    Jp = np.zeros((2, 6))
    for a in range(6):
        cols = []
        for s in (+1, -1):
            xi = np.zeros(6)
            xi[a] = s * h
            dR, dt = se3_exp(xi)
            R, t = T[:9].reshape(3, 3), T[9:]
            T2 = np.concatenate([(dR @ R).reshape(9), dR @ t + dt])
            cols.append(residual(pr, k, T12=T2))
        Jp[:, a] = (cols[0] - cols[1]) / (2 * h)
    Jx = np.zeros((2, 3))
    for a in range(3):
        cols = []
        for s in (+1, -1):
            X2 = X.copy()
            X2[a] += s * h
            cols.append(residual(pr, k, X=X2))
        Jx[:, a] = (cols[0] - cols[1]) / (2 * h)
    return Jp, Jx


@pytest.fixture(scope="module")
def tiny(built):
    sc = scenes.synthetic_ba_scene(8, 30, 5, True, seed=21, n_fixed=2)
    return scenes.scaled_problem(sc)


def test_jacobians_vs_finite_differences(tiny):
    """A_j, a_j, C_i, b_i assembled from numeric Jacobians == oracle blocks."""
    pr = tiny
    o = O.Oracle(pr)
    huber = 1e9           # weight 1 everywhere
    o.linearize(huber)
    o.damp_invert(0.0)    # lambda = 0 -> plain mirror
    A, a = o.get_A()
    Cm, b = o.get_C()
    jopt = np.cumsum(pr["pose_fixed"] == 0) - 1
    iopt = np.cumsum(pr["pt_fixed"] == 0) - 1
    An, an = np.zeros_like(A), np.zeros_like(a)
    Cn, bn = np.zeros_like(Cm), np.zeros_like(b)
    for k in range(pr["obs_cam"].shape[0]):
        r = residual(pr, k)
        Jp, Jx = numeric_jacobians(pr, k)
        p, q = pr["obs_pose"][k], pr["obs_pt"][k]
        if not pr["pose_fixed"][p]:
            An[jopt[p]] += Jp.T @ Jp
            an[jopt[p]] -= Jp.T @ r
        if not pr["pt_fixed"][q]:
            Cn[iopt[q]] += Jx.T @ Jx
            bn[iopt[q]] -= Jx.T @ r
    assert np.abs(A - An).max() / np.abs(An).max() < 1e-6
    assert np.abs(a - an).max() / np.abs(an).max() < 1e-6
    assert np.abs(Cm - Cn).max() / np.abs(Cn).max() < 1e-6
    assert np.abs(b - bn).max() / np.abs(bn).max() < 1e-6


def test_cross_block_is_last_writer(tiny):
    """Q1: W_ji equals w Q^T R of the LAST inserted observation of the pair
    (right camera here), not the sum over both cameras."""
    pr = tiny
    o = O.Oracle(pr)
    o.linearize(1e9)
    pi, pj, W = o.get_pairs()
    jopt = np.cumsum(pr["pose_fixed"] == 0) - 1
    iopt = np.cumsum(pr["pt_fixed"] == 0) - 1
    last = {}
    for k in range(pr["obs_cam"].shape[0]):
        p, q = pr["obs_pose"][k], pr["obs_pt"][k]
        if pr["pose_fixed"][p] or pr["pt_fixed"][q]:
            continue
        last[(iopt[q], jopt[p])] = k
    assert len(last) == len(pi)
    for n in range(0, len(pi), 7):
        k = last[(pi[n], pj[n])]
        assert pr["obs_cam"][k] == 1          # right camera inserted last
        Jp, Jx = numeric_jacobians(pr, k)
        assert np.abs(W[n] - Jp.T @ Jx).max() / np.abs(W[n]).max() < 1e-5


def test_ldlt_matches_numpy_and_pseudo_inverse():
    rng = np.random.default_rng(1)
    for n in (1, 3, 6, 17, 60):
        Q = rng.standard_normal((n, n))
        A = Q @ Q.T + 0.1 * np.eye(n)
        B = rng.standard_normal((n, 2))
        X = O.ldlt_solve(A, B)
        assert np.abs(A @ X - B).max() < 1e-9 * max(1, np.abs(B).max())
    # symmetric indefinite is fine for LDLT as well
    A = np.array([[2.0, 1, 0], [1, -3, 0.5], [0, 0.5, 1]])
    Bv = np.array([1.0, 2, 3])
    assert np.allclose(O.ldlt_solve(A, Bv)[:, 0], np.linalg.solve(A, Bv))
    # zero matrix -> Eigen's LDLT solve returns 0 (D pseudo-inverted)
    assert (O.ldlt_solve(np.zeros((3, 3)), np.ones(3)) == 0).all()
    # decoupled zero row/col -> that component is 0, the rest is solved
    A = np.diag([4.0, 0.0, 2.0])
    x = O.ldlt_solve(A, np.array([8.0, 5.0, 6.0]))[:, 0]
    assert np.allclose(x, [2.0, 0.0, 3.0])


def test_unobserved_landmark_gives_zero_inverse(built):
    sc = scenes.synthetic_ba_scene(8, 20, 5, False, seed=4, n_fixed=2)
    keep = sc["obs_pt"] != 3
    for k in ("obs_cam", "obs_pose", "obs_pt", "obs_uv"):
        sc[k] = sc[k][keep]
    o = O.Oracle(scenes.scaled_problem(sc))
    o.linearize(1.0)
    o.damp_invert(10.0)
    o.schur(); o.solve_reduced(); o.backsub()
    Ci, cb = o.get_Cinv()
    assert (Ci[3] == 0).all() and (cb[3] == 0).all()
    x, y = o.get_xy()
    assert np.isfinite(x).all() and (y[3] == 0).all()


def test_lm_control_quirks(built):
    """Q2: previous_cost advances even on a rejected step, so the row after
    a SKIPPED iteration reports cost_change against the rejected trial cost;
    SKIPPED rows carry the overwritten fields of reference :995-1000."""
    sc = scenes.synthetic_ba_scene(10, 80, 5, True, seed=2)
    pr = scenes.scaled_problem(sc)
    o = O.Oracle(pr)
    # tiny lambda at the start makes early steps overshoot -> rejections
    rows, conv = o.solve(O.make_options(max_iter=30, thr_step=0, thr_cost=0,
                                        lambda0=1e-9, inc=3.0, dec=0.33))
    n_obs = pr["obs_cam"].shape[0]
    prev = None
    saw_skip = False
    for r in rows:
        if prev is not None:
            expect = abs(r.trial_cost - prev.trial_cost)
            if r.iteration_status != 2:
                assert abs(r.cost_change - expect) <= 1e-12 * max(1, expect)
        if r.iteration_status == 2:
            saw_skip = True
            assert r.cost_change == 0
            assert abs(r.average_reprojection_error -
                       np.sqrt(r.cost / n_obs)) < 1e-12
            assert r.rho <= 0.25 or np.isnan(r.rho)
        prev = r
    # lambda never leaves [1e-10, 100]
    assert all(1e-10 <= r.damping_term <= 100.0 for r in rows)
    assert isinstance(saw_skip, bool)


def test_oracle_converges_to_ground_truth(built):
    sc = scenes.synthetic_ba_scene(20, 800, 5, True, seed=3)
    o = O.Oracle(scenes.scaled_problem(sc))
    rows, _ = o.solve(O.make_options(max_iter=60, thr_step=1e-9,
                                     thr_cost=1e-9))
    err = np.linalg.norm(o.get_points() / 0.01 - sc["X_true"], axis=1)
    err0 = np.linalg.norm(sc["X_init"] - sc["X_true"], axis=1)
    assert rows[-1].cost < 1e-2 * rows[0].cost
    assert np.median(err) < 0.1 * np.median(err0)


def test_stage_functions_compose_to_solve(tiny):
    """Driving the stage entry points by hand reproduces ba_oracle_solve."""
    pr = tiny
    o1, o2 = O.Oracle(pr), O.Oracle(pr)
    opt = O.make_options(max_iter=6, thr_step=0, thr_cost=0)
    rows, _ = o1.solve(opt)
    lam, prev = 100.0, o2.cost()
    dec, inc = float(np.float32(0.33)), float(np.float32(3.0))
    for r in rows:
        o2.linearize(1.0); o2.damp_invert(lam); o2.schur()
        o2.solve_reduced(); o2.backsub(); o2.backup(); o2.update()
        cur, model = o2.cost(), o2.model_change()
        rho = (cur - prev) * 100.0 / model
        if not rho > 0.25:
            o2.revert()
        if rho > 0.5:
            lam = max(1e-10, lam * dec)
        elif rho <= 0.25:
            lam = min(100.0, lam * inc)
        assert abs(cur - r.trial_cost) <= 1e-12 * abs(cur)
        assert abs(lam - r.damping_term) <= 1e-15 * lam
        prev = cur
    assert np.array_equal(o1.get_points(), o2.get_points())


def test_golden_fixture(built):
    """Committed regression vectors (self-generated, see make_golden.py)."""
    with open(os.path.join(HERE, "golden", "ba_golden_small.json")) as f:
        gold = json.load(f)
    sc = scenes.synthetic_ba_scene(**gold["scene_args"])
    pr = scenes.scaled_problem(sc)
    assert pr["obs_cam"].shape[0] == gold["n_obs"]
    o = O.Oracle(pr)
    assert abs(o.cost() - gold["cost0"]) <= 1e-9 * gold["cost0"]
    rows, conv = o.solve(O.make_options(**gold["options"]))
    assert len(rows) == len(gold["rows"])
    for r, g in zip(rows, gold["rows"]):
        assert r.iteration_status == g["status"]
        assert abs(r.trial_cost - g["trial_cost"]) <= 1e-8 * abs(g["trial_cost"])
        assert abs(r.damping_term - g["lambda"]) <= 1e-12 * g["lambda"]
    P = o.get_poses()
    X = o.get_points()
    assert np.abs(P - np.array(gold["final_poses"])).max() < 1e-8
    assert np.abs(X[:len(gold["final_points_head"])] -
                  np.array(gold["final_points_head"])).max() < 1e-8


def test_pose_only_oracle_recovers_pose(built):
    sc = scenes.pose_only_scene(2000, seed=8)
    opt = O.make_options(max_iter=100, thr_step=1e-6, thr_cost=1e-6,
                         huber=1.0, outlier=2.5)
    res = O.pose_only_mono6(sc["X"], sc["uv"], sc["fx"], sc["fy"], sc["cx"],
                            sc["cy"], sc["T_init"], np.ones(2000, np.uint8),
                            opt)
    assert res["success"] and res["converged"]
    T = res["T12"]
    assert np.abs(T[:9].reshape(3, 3) - sc["T_true"][:3, :3]).max() < 1e-3
    assert np.abs(T[9:] - sc["T_true"][:3, 3]).max() < 1e-3
    # Q9: no Summary row on the converging iteration
    assert len(res["rows"]) == res["n_iter"] - 1


def test_stereo_pose_only_oracle_recovers_pose_and_has_mono_limit():
    """reference core/pose_only_bundle_adjustment_solver.cpp:172-399 restated:
    converges to the true pose on noise-free data; with no right match at all it
    is the monocular solver (the right camera contributes nothing, :298)."""
    from bundle_adjustment_solver_amd import scenes
    sc = scenes.pose_only_stereo_scene(3000, seed=5, right_missing_frac=0.3)
    intr = [sc["fx"], sc["fy"], sc["cx"], sc["cy"]]
    opt = O.make_options(max_iter=100, thr_step=1e-6, thr_cost=1e-6, huber=1.0,
                         outlier=2.5)
    ones = np.ones(3000, np.uint8)
    r = O.pose_only_stereo6(sc["X"], sc["uv"], sc["uv_right"], intr, intr,
                            sc["T_lr"], sc["T_init"], ones, ones, opt)
    Tt = np.concatenate([sc["T_true"][:3, :3].reshape(9), sc["T_true"][:3, 3]])
    assert r["success"] and r["converged"] and np.abs(r["T12"] - Tt).max() < 1e-5
    assert r["mask_r"][sc["right_missing"]].all()
    sc2 = scenes.pose_only_stereo_scene(3000, seed=5, right_missing_frac=1.0)
    r2 = O.pose_only_stereo6(sc2["X"], sc2["uv"], sc2["uv_right"], intr, intr,
                             sc2["T_lr"], sc2["T_init"], ones, ones, opt)
    m = O.pose_only_mono6(sc2["X"], sc2["uv"], sc2["fx"], sc2["fy"], sc2["cx"],
                          sc2["cy"], sc2["T_init"], ones, opt)
    assert np.abs(r2["T12"] - m["T12"]).max() < 1e-6 and r2["n_iter"] == m["n_iter"]


def test_stereo_pose_only_golden_vector():
    """Committed regression vector of the stereo pose-only oracle."""
    import json
    import os
    from bundle_adjustment_solver_amd import scenes
    with open(os.path.join(os.path.dirname(__file__), "golden",
                           "pose_only_stereo_golden.json")) as f:
        gold = json.load(f)
    sc = scenes.pose_only_stereo_scene(gold["n"], seed=gold["seed"],
                                       right_missing_frac=gold["right_missing_frac"])
    intr = [sc["fx"], sc["fy"], sc["cx"], sc["cy"]]
    ones = np.ones(gold["n"], np.uint8)
    opt = O.make_options(max_iter=100, thr_step=1e-6, thr_cost=1e-6, huber=1.0,
                         outlier=2.5)
    r = O.pose_only_stereo6(sc["X"], sc["uv"], sc["uv_right"], intr, intr,
                            sc["T_lr"], sc["T_init"], ones, ones, opt)
    assert r["n_iter"] == gold["n_iter"] and r["converged"] == gold["converged"]
    assert np.abs(r["T12"] - np.array(gold["T12"])).max() < 1e-6
    assert int(r["mask_l"].sum()) == gold["n_inlier_left"]
    assert int(r["mask_r"].sum()) == gold["n_inlier_right"]


# --------------------------------------------------------------------------
# fast-solve mode of the oracle (envelope LDL^T without pivoting): the test
# shortcut that lets the BASELINE-size trajectories be followed for many
# iterations.  It must agree with the restated Eigen pivoted LDLT.
# --------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["C1", "C2"])
def test_fast_solve_matches_the_pivoted_ldlt(name, built):
    """Same S, rhs -> x of the envelope LDL^T equals x of the reference-style
    pivoted LDLT to <= 1e-9 (max-norm relative) on C1 (dense S) and on the full
    C2 (banded S), at a large and at a small damping."""
    pr = scenes.scaled_problem(scenes.config_scene(name))
    o = O.Oracle(pr)
    for lam in (100.0, 1e-3):
        o.linearize(1.0)
        o.damp_invert(lam)
        o.schur()
        o.set_fast_solve(False)
        o.solve_reduced()
        x_ref = o.get_xy()[0].copy()
        o.set_fast_solve(True)
        o.solve_reduced()
        x_fast = o.get_xy()[0].copy()
        assert np.abs(x_ref).max() > 0
        assert np.abs(x_fast - x_ref).max() <= 1e-9 * np.abs(x_ref).max(), (name, lam)
        S, rhs = o.get_S()
        res = np.abs(S @ x_fast.reshape(-1) - rhs).max() / np.abs(rhs).max()
        assert res < 1e-9


def test_fast_solve_trajectory_equals_the_faithful_one(built):
    """Whole LM loop on C1 and on a mono window scene: identical status and
    lambda sequences, costs to 1e-9, final parameters to 1e-8."""
    for sc in (scenes.test_ba_scene(),
               scenes.synthetic_ba_scene(30, 1500, 10, False, seed=41)):
        pr = scenes.scaled_problem(sc)
        outs = []
        for fast in (False, True):
            o = O.Oracle(pr)
            o.set_fast_solve(fast)
            rows, conv = o.solve(O.make_options(max_iter=30, thr_step=1e-7,
                                                thr_cost=1e-7))
            outs.append((rows, conv, o.get_poses(), o.get_points()))
        (ra, ca, Pa, Xa), (rb, cb, Pb, Xb) = outs
        assert len(ra) == len(rb) and ca == cb
        for a, b in zip(ra, rb):
            assert a.iteration_status == b.iteration_status
            assert a.damping_term == b.damping_term
            # relative to the trial cost, floored at the roundoff level of the
            # starting cost (noise-free scenes converge to ~1e-9 of it)
            assert abs(a.trial_cost - b.trial_cost) <= 1e-9 * abs(a.trial_cost) \
                + 1e-12 * ra[0].cost
        assert np.abs(Pa - Pb).max() <= 1e-8 * np.abs(Pa).max()
        assert np.abs(Xa - Xb).max() <= 1e-8 * np.abs(Xa).max()


def test_fast_solve_zero_pivot_gives_zero_step(built):
    """A pose without observations has a zero row / column in S: both solvers
    return x_j = 0 for it (pseudo-inverse rule)."""
    sc = scenes.synthetic_ba_scene(12, 60, 5, False, seed=5)
    keep = sc["obs_pose"] != 11
    for k in ("obs_cam", "obs_pose", "obs_pt", "obs_uv"):
        sc[k] = sc[k][keep]
    pr = scenes.scaled_problem(sc)
    o = O.Oracle(pr)
    o.linearize(1.0); o.damp_invert(2.0); o.schur()
    xs = []
    for fast in (False, True):
        o.set_fast_solve(fast)
        o.solve_reduced()
        xs.append(o.get_xy()[0].copy())
    assert (xs[0][-1] == 0).all() and (xs[1][-1] == 0).all()
    assert np.abs(xs[0] - xs[1]).max() <= 1e-9 * np.abs(xs[0]).max()


def test_nan_cost_takes_the_skipped_branch_with_lambda_unchanged():
    """Reference :930-953 with rho = NaN: `rho > 0.25` is false (revert, SKIPPED),
    `rho > 0.5` and `rho <= 0.25` are both false (lambda unchanged); :1005 advances
    previous_cost to the NaN trial cost.  (The regime bench.py's thresholds-off loop
    ends in once a run-away landmark overflows; GPU twin:
    tests/test_gpu_parity.py::test_nan_cost_branch_of_the_control_step_matches_oracle.)"""
    sc = scenes.synthetic_ba_scene(12, 300, 5, True, seed=7)
    pr = scenes.scaled_problem(sc)
    pr["obs_uv"] = pr["obs_uv"].copy()
    pr["obs_uv"][1234, 0] = np.nan
    o = O.Oracle(pr)
    rows, conv = o.solve(O.make_options(max_iter=5, thr_step=-1.0, thr_cost=-1.0))
    assert len(rows) == 5 and not conv
    for r in rows:
        assert r.iteration_status == 2 and r.damping_term == 100.0
        assert np.isnan(r.trial_cost) and np.isnan(r.rho)


@pytest.mark.parametrize("stereo", [False, True])
def test_oracle_minimiser_matches_an_independent_least_squares_solver(stereo, built):
    """INDEPENDENT (non-restatement) pin of the whole bundle-adjustment path of the oracle:
    with the robust weight off its fixed point is the minimiser of the sum of squared
    reprojection errors; scipy.optimize.least_squares (trust-region reflective, finite-
    difference Jacobian of the plain numpy projection below, its own pose chart) started from
    the same values ends in the same minimum — squared-error sums to 1e-6 relative, poses and
    points to 1e-4 of the scene scale (the LM loop with thresholds off cycles accept / reject
    at its noise floor).  tests/test_gpu_parity.py holds the same check for the HIP path."""
    from scipy.optimize import least_squares
    sc = scenes.synthetic_ba_scene(8, 70, 5, stereo, seed=91, pixel_sigma=0.5, pose_noise=0.01, point_noise=0.03)
    pr = scenes.scaled_problem(sc)
    intr, camT = pr["cam_intr"], pr["cam_T"]
    T0, X0 = pr["pose_T"].copy(), pr["pt_X"].copy()
    free_p = np.flatnonzero(pr["pose_fixed"] == 0)
    free_x = np.flatnonzero(pr["pt_fixed"] == 0)
    oc, op, ox, uv = pr["obs_cam"], pr["obs_pose"], pr["obs_pt"], pr["obs_uv"]

    def rodrigues(w):
        th = np.linalg.norm(w)
        K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        if th < 1e-12:
            return np.eye(3) + K
        return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)

    def unpack(z):
        T, X = T0.copy(), X0.copy()
        for a, j in enumerate(free_p):
            w, v = z[6 * a:6 * a + 3], z[6 * a + 3:6 * a + 6]
            T[j, :9] = (rodrigues(w) @ T0[j, :9].reshape(3, 3)).reshape(9)
            T[j, 9:] = T0[j, 9:] + v
        X[free_x] = X0[free_x] + z[6 * len(free_p):].reshape(-1, 3)
        return T, X

    def res(T, X):
        Xj = np.einsum("kab,kb->ka", T[op, :9].reshape(-1, 3, 3), X[ox]) + T[op, 9:]
        Xc = np.einsum("kab,kb->ka", camT[oc, :9].reshape(-1, 3, 3), Xj) + camT[oc, 9:]
        u = intr[oc, 0] * Xc[:, 0] / Xc[:, 2] + intr[oc, 2]
        v = intr[oc, 1] * Xc[:, 1] / Xc[:, 2] + intr[oc, 3]
        return np.concatenate([u - uv[:, 0], v - uv[:, 1]])

    z0 = np.zeros(6 * len(free_p) + 3 * len(free_x))
    ls = least_squares(lambda z: res(*unpack(z)), z0, method="trf", jac="3-point",
                       xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=400)
    T_ls, X_ls = unpack(ls.x)
    o = O.Oracle(pr)
    o.solve(O.make_options(max_iter=80, thr_step=0, thr_cost=0, huber=1e9))
    To, Xo = o.get_poses(), o.get_points()
    sse = lambda T, X: float((res(T, X) ** 2).sum())
    assert sse(T_ls, X_ls) < 0.5 * sse(T0, X0)
    assert abs(sse(To, Xo) - sse(T_ls, X_ls)) <= 1e-6 * sse(T_ls, X_ls)
    scale = np.abs(X0).max()
    assert np.abs(To - T_ls).max() <= 1e-4 * scale and np.abs(Xo - X_ls).max() <= 1e-4 * scale
