"""Host-side logic that needs no GPU: scene generators, the Python mirror of
the reference facade (registration, error conventions), Options / Summary."""
import numpy as np
import pytest

from bundle_adjustment_solver_amd import scenes
from bundle_adjustment_solver_amd.solver import (Camera,
                                                 FullBundleAdjustmentSolver,
                                                 IterationStatus,
                                                 OptimizationInfo, Options,
                                                 Summary, rigid_inverse)


def test_c1_scene_counts_match_reference_scene():
    """the test_ba.cpp scene has 60 poses, 660 landmarks,
    34 019 observations (17 082 left + 16 937 right)."""
    sc = scenes.test_ba_scene()
    assert sc["T_wc_true"].shape[0] == 60 and sc["X_true"].shape[0] == 660
    assert sc["obs_cam"].shape[0] == 34019
    assert int((sc["obs_cam"] == 0).sum()) == 17082
    assert int((sc["obs_cam"] == 1).sum()) == 16937
    pairs = set(zip(sc["obs_pose"].tolist(), sc["obs_pt"].tolist()))
    assert len(pairs) == 17127
    assert len({p for p in pairs if p[0] >= 5}) == 16557
    seen = np.unique(sc["obs_pt"])
    assert 660 - seen.size == 130          # never-observed landmarks
    # insertion order: pose-major, left then right (test_ba.cpp:254-273)
    key = sc["obs_pose"].astype(np.int64) * 2 + sc["obs_cam"]
    assert (np.diff(key) >= 0).all()


def test_synthetic_scene_exact_sizes_and_visibility():
    sc = scenes.synthetic_ba_scene(30, 500, 5, True, seed=1)
    assert sc["obs_cam"].shape[0] == 500 * 5 * 2
    assert (sc["obs_uv"] > 0).all()
    assert (sc["obs_uv"][:, 0] < 640).all() and (sc["obs_uv"][:, 1] < 480).all()
    cnt = np.bincount(sc["obs_pt"], minlength=500)
    assert (cnt == 10).all()
    sc2 = scenes.synthetic_ba_scene(30, 500, 5, True, seed=1)
    assert np.array_equal(sc["obs_uv"], sc2["obs_uv"])     # seeded
    # reprojection of the TRUE geometry is exact (pixel noise 0)
    pr = scenes.scaled_problem(dict(sc, T_wc_init=sc["T_wc_true"],
                                    X_init=sc["X_true"]))
    k = 777
    T = pr["pose_T"][pr["obs_pose"][k]]
    Xc = T[:9].reshape(3, 3) @ pr["pt_X"][pr["obs_pt"][k]] + T[9:]
    cT = pr["cam_T"][pr["obs_cam"][k]]
    Xc = cT[:9].reshape(3, 3) @ Xc + cT[9:]
    fx, fy, cx, cy = pr["cam_intr"][pr["obs_cam"][k]]
    uv = np.array([fx * Xc[0] / Xc[2] + cx, fy * Xc[1] / Xc[2] + cy])
    assert np.abs(uv - pr["obs_uv"][k]).max() < 1e-9


def test_config_sizes():
    for name, (n_pose, n_pt, window, stereo, _) in scenes.CONFIGS.items():
        n_obs = n_pt * window * (2 if stereo else 1)
        assert n_obs == {"C2": 500_000, "C3": 2_000_000, "C4": 5_000_000}[name]


def test_options_defaults_are_the_reference_floats():
    o = Options().to_c()
    assert o.max_num_iterations == 50
    assert o.initial_lambda == 100.0
    assert o.decrease_ratio_lambda == np.float32(0.33)   # Q4: float fields
    assert float(o.decrease_ratio_lambda) != 0.33
    assert o.increase_ratio_lambda == 3.0
    assert o.threshold_huber_loss == 1.0
    assert o.threshold_outlier_rejection == 2.0
    assert o.threshold_step_size == np.float32(1e-5)


def test_summary_brief_report_format():
    s = Summary()
    for k in range(3):
        info = OptimizationInfo()
        info.cost, info.cost_change = 10.0 / (k + 1), 1.0
        info.average_reprojection_error = 0.1
        info.abs_step, info.abs_gradient, info.damping_term = 1e-3, 0, 33.0
        info.iter_time = 1.5
        info.iteration_status = [IterationStatus.UPDATE_TRUST_MORE,
                                 IterationStatus.SKIPPED,
                                 IterationStatus.UPDATE][k]
        s.optimization_info_list_.append(info)
    s.max_iteration_ = 3
    s.convergence_status_ = False
    s.total_time_in_millisecond_ = 12.0
    rep = s.BriefReport()
    assert rep.startswith("itr ")
    assert "Analytic Solver Report:" in rep and "Iterations      : 3" in rep
    assert "NO_CONVERGENCE" in rep and "MAX ITERATION is reached" in rep
    assert " SKIP " in rep
    assert abs(s.GetTotalTimeInSecond() - 0.012) < 1e-12


def test_rigid_inverse_is_isometry_inverse():
    sc = scenes.test_ba_scene()
    T = sc["T_wc_true"][:7]
    Ti = rigid_inverse(T)
    I = np.einsum("nij,njk->nik", T, Ti)
    assert np.abs(I - np.eye(4)).max() < 1e-12


def test_facade_registration_and_error_conventions(capsys):
    """reference :87-180: duplicates ignored, nullptr fix -> stderr + skip,
    unknown pointer fix -> runtime_error, invalid observation -> dropped."""
    s = FullBundleAdjustmentSolver()
    cam = Camera(525.0, 525.0, 320.0, 240.0)
    s.AddCamera(0, cam)
    s.AddCamera(0, Camera(1, 1, 1, 1))          # duplicate index: first kept
    assert s.camera_id_to_camera_map_[0].fx == pytest.approx(5.25)  # x0.01
    poses = [np.eye(4) for _ in range(3)]
    pts = [np.array([1.0, 2.0, 3.0]), np.array([0.0, 0.0, 5.0])]
    for p in poses:
        s.AddPose(p)
    s.AddPose(poses[0])                          # same object: no new pose
    for x in pts:
        s.AddPoint(x)
    assert s.num_total_poses_ == 3 and s.num_total_points_ == 2
    s.MakePoseFixed(poses[0])
    s.MakePointFixed(None)                       # test_ba.cpp:252
    assert "Empty pointer" in capsys.readouterr().err
    with pytest.raises(RuntimeError):
        s.MakePoseFixed(np.eye(4))               # unknown "pointer"
    with pytest.raises(RuntimeError):
        s.MakePointFixed(np.zeros(3))
    s.AddObservation(0, poses[1], pts[0], np.array([100.0, 50.0]))
    s.AddObservation(7, poses[1], pts[0], np.array([1.0, 1.0]))   # bad cam
    s.AddObservation(0, np.eye(4), pts[0], np.array([1.0, 1.0]))  # bad pose
    s.AddObservation(0, poses[1], np.zeros(3), np.array([1.0, 1.0]))
    err = capsys.readouterr().err
    assert "Invalid camera index" in err and "Nonexisting pose" in err \
        and "Nonexisting point" in err
    assert s.num_total_observations_ == 1
    assert s._obs_uv[0][0, 0] == pytest.approx(1.0)               # x0.01
    # T_jw = pose^-1 with translation x0.01 (reference :96-97)
    Tw = np.eye(4)
    Tw[:3, 3] = [1.0, 2.0, 3.0]
    s2 = FullBundleAdjustmentSolver()
    s2.AddPose(Tw)
    assert np.allclose(s2._pose_T_jw[0][0, 9:], [-0.01, -0.02, -0.03])


def test_facade_bulk_api_bookkeeping():
    s = FullBundleAdjustmentSolver()
    s.AddCamera(0, Camera(525.0, 525.0, 320.0, 240.0))
    P = np.tile(np.eye(4), (4, 1, 1))
    X = np.random.default_rng(0).uniform(1, 2, (10, 3))
    hp = s.AddPoseArray(P)
    hx = s.AddPointArray(X)
    assert hp.tolist() == [0, 1, 2, 3] and hx.tolist() == list(range(10))
    s.AddObservations(0, np.array([0, 1, 9]), np.array([0, 20, 1]),
                      np.zeros((3, 2)))
    assert s.num_total_observations_ == 1        # two invalid rows dropped


def test_documents_cite_existing_profile_files():
    """DESIGN.md / README.md / bench.py name files under profiles/: every one of
    them must be committed (the per-round sets replace each other)."""
    import glob
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cited = set()
    for name in ("DESIGN.md", "README.md", "bench.py", "profiles/README.md"):
        text = open(os.path.join(root, name)).read()
        cited |= set(re.findall(r"profiles/(r0[0-9]_[A-Za-z0-9_{},.]+?\.(?:json|csv|txt))", text))
    assert cited
    for c in sorted(cited):
        m = re.search(r"\{([^}]*)\}", c)
        names = [c[:m.start()] + v + c[m.end():] for v in m.group(1).split(",")] if m else [c]
        for n in names:
            assert glob.glob(os.path.join(root, "profiles", n)), "missing profiles/" + n


def test_connectivity_warnings(capsys):
    """CheckPoseAndPointConnectivity (reference :310-341): an optimisable pose
    with fewer than 5 distinct related points and an optimisable point with fewer
    than 2 distinct related poses are reported on stderr; fixed entries are not,
    and fixed partners count as related (reference :684-693)."""
    s = FullBundleAdjustmentSolver()
    s.AddCamera(0, Camera(525.0, 525.0, 320.0, 240.0))
    s.AddCamera(1, Camera(525.0, 525.0, 320.0, 240.0))
    hp = s.AddPoseArray(np.tile(np.eye(4), (4, 1, 1)))
    hx = s.AddPointArray(np.random.default_rng(0).uniform(1, 2, (8, 3)))
    s.MakePoseFixed(int(hp[0]))
    s.MakePointFixed(int(hx[7]))
    px = np.zeros((1, 2))
    # poses 0 (fixed), 1, 2 see points 0..5; pose 3 sees points 0..3 only (4 < 5,
    # twice each through the two cameras: still 4 distinct points)
    for j in (0, 1, 2):
        for i in range(6):
            s.AddObservations(0, [hp[j]], [hx[i]], px)
    for i in range(4):
        s.AddObservations(0, [hp[3]], [hx[i]], px)
        s.AddObservations(1, [hp[3]], [hx[i]], px)
    # point 6: seen by pose 1 only (through both cameras: one related pose);
    # point 7: never seen but FIXED -> no warning
    s.AddObservations(0, [hp[1]], [hx[6]], px)
    s.AddObservations(1, [hp[1]], [hx[6]], px)
    capsys.readouterr()
    s.CheckPoseAndPointConnectivity()
    err = capsys.readouterr().err
    # optimisation indices: input order of the non-fixed entries
    assert "2-th pose: It might diverge because some frames have insufficient related points." in err
    assert err.count("-th pose:") == 1
    assert "6-th point: It might diverge because some points have insufficient related poses." in err
    assert err.count("-th point:") == 1
