"""BAL scene I/O: host-only checks.  The reference has no
on-disk format and no fixture for it; the committed tests/golden/bal_small.txt
is self-generated (tests/golden/make_golden.py bal) — parity unpinned."""
import os

import numpy as np
import pytest

from bundle_adjustment_solver_amd import scene_io, scenes
from oracle import oracle_py as O

HERE = os.path.dirname(os.path.abspath(__file__))
BAL = os.path.join(HERE, "golden", "bal_small.txt")


def test_rodrigues_round_trip():
    rng = np.random.default_rng(0)
    r = rng.normal(size=(200, 3))
    r[0] = 0.0
    r[1] = [1e-10, 0, 0]
    r[2] = np.array([0.6, -0.8, 0.0]) * (np.pi - 1e-9)   # almost a half turn
    R = scene_io.rodrigues_to_matrix(r)
    assert np.abs(R @ np.transpose(R, (0, 2, 1)) - np.eye(3)).max() < 1e-12
    assert np.abs(np.linalg.det(R) - 1.0).max() < 1e-12
    R2 = scene_io.rodrigues_to_matrix(scene_io.matrix_to_rodrigues(R))
    assert np.abs(R - R2).max() < 1e-7


@pytest.mark.parametrize("stereo", [False, True])
def test_save_load_keeps_every_residual(tmp_path, stereo):
    sc = scenes.synthetic_ba_scene(10, 60, 4, stereo, seed=5, n_fixed=2,
                                   pixel_sigma=0.5)
    sc["intr"] = sc["intr"].astype(np.float64).copy()
    sc["intr"][:, 1] *= 1.07       # fx != fy: save_bal must fold it into v
    path = str(tmp_path / "scene.txt")
    cam_of_obs, pairs = scene_io.save_bal(path, sc)
    assert pairs.shape[0] == (2 if stereo else 1) * 10
    ld = scene_io.load_bal(path, n_fixed_poses=3)
    assert ld["obs_uv"].shape == sc["obs_uv"].shape
    assert (ld["obs_pose"] == cam_of_obs).all() and (ld["obs_cam"] == cam_of_obs).all()
    assert ld["pose_fixed"].sum() == 3 and not ld["pt_fixed"].any()
    r0 = scene_io.reprojection_residuals(sc)
    r1 = scene_io.reprojection_residuals(ld)
    scale = sc["intr"][sc["obs_cam"], 0] / sc["intr"][sc["obs_cam"], 1]
    assert np.abs(r1[:, 0] - r0[:, 0]).max() < 1e-8
    assert np.abs(r1[:, 1] - r0[:, 1] * scale).max() < 1e-8


def test_fixture_loads_and_undistorts():
    raw = scene_io.parse_bal(BAL)
    assert raw["cameras"].shape == (12, 9) and raw["points"].shape == (90, 3)
    assert raw["xy"].shape == (450, 2)
    assert np.abs(raw["cameras"][:, 7]).max() > 0          # the file is distorted
    with pytest.raises(ValueError, match="distortion"):
        scene_io.load_bal(BAL, undistort=False)
    ld = scene_io.load_bal(BAL, n_fixed_poses=2)
    # forward model of the file on the undistorted measurements
    f, k1, k2 = (ld["bal"][k][ld["obs_cam"]] for k in ("f", "k1", "k2"))
    p = np.stack([ld["obs_uv"][:, 0], -ld["obs_uv"][:, 1]], axis=1) / f[:, None]
    r2 = (p * p).sum(axis=1)
    back = p * (f * (1 + k1 * r2 + k2 * r2 * r2))[:, None]
    assert np.abs(back - raw["xy"]).max() < 1e-8
    # the scene was written from its INITIAL estimate: residuals are the
    # perturbation, a few pixels to a few hundred, never NaN
    r = scene_io.reprojection_residuals(ld)
    assert np.isfinite(r).all() and 0.1 < np.abs(r).max() < 2000.0


def test_oracle_solves_the_fixture():
    ld = scene_io.load_bal(BAL, n_fixed_poses=2)
    pr = scenes.scaled_problem(ld)
    o = O.Oracle(pr)
    c0 = o.cost()
    rows, _ = o.solve(O.make_options(max_iter=30, thr_step=1e-9, thr_cost=1e-9))
    assert rows[-1].cost < 0.05 * c0


@pytest.mark.parametrize("text,msg", [
    ("", "header"),
    ("1 1 1\n0 0 1.0 2.0\n", "expected"),
    ("1 1 1\n0 3 1.0 2.0\n" + "0\n" * 12, "out of range"),
    ("1 1 1\n2 0 1.0 2.0\n" + "0\n" * 12, "out of range"),
])
def test_malformed_files_are_refused(tmp_path, text, msg):
    path = tmp_path / "bad.txt"
    path.write_text(text)
    with pytest.raises(ValueError, match=msg):
        scene_io.parse_bal(str(path))


def test_empty_problem(tmp_path):
    path = tmp_path / "empty.txt"
    path.write_text("0 0 0\n")
    ld = scene_io.load_bal(str(path))
    assert ld["obs_uv"].shape == (0, 2) and ld["X_init"].shape == (0, 3)
