"""Scene I/O in the "Bundle Adjustment in the Large" (BAL) text format
(SURVEY.md §8f N4).

The reference has no on-disk format (its scenes are generated in
test/test_ba.cpp); BAL is the de-facto exchange format of public BA problems:

    <num_cameras> <num_points> <num_observations>
    <camera_index> <point_index> <x> <y>          (one line per observation)
    <9 values per camera, one per line>           Rodrigues r(3), t(3), f, k1, k2
    <3 values per point, one per line>

with the camera model  P = R X + t,  p = -P / P.z,  pixel = f r(p) p,
r(p) = 1 + k1 |p|^2 + k2 |p|^4   (camera looks down -z, no principal point).

The hot path has the reference's camera model: pinhole (fx fy cx cy), +z
forward, fixed intrinsics, no distortion.  Mapping, exact for fixed
intrinsics:

  * one solver camera AND one pose per BAL camera: pose T_jw = (D R, D t) with
    D = diag(1,-1,-1) (turns the -z camera into a +z camera), intrinsics
    (f, f, 0, 0), identity extrinsics, pixel (u, v) = (x, -y);
  * radial distortion is removed from the MEASUREMENTS once (Newton on
    rho (1 + k1 rho^2 + k2 rho^4) = rho_d): with intrinsics held fixed, as this
    solver does, that is the same least-squares problem up to the weighting of
    the residual by r(p).

Only numpy; nothing here touches the GPU or the oracle.
"""
import numpy as np


def rodrigues_to_matrix(r):
    """(n,3) angle-axis vectors -> (n,3,3) rotation matrices."""
    r = np.asarray(r, np.float64).reshape(-1, 3)
    th = np.linalg.norm(r, axis=1)
    K = np.zeros((r.shape[0], 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -r[:, 2], r[:, 1]
    K[:, 1, 0], K[:, 1, 2] = r[:, 2], -r[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -r[:, 1], r[:, 0]
    small = th < 1e-8
    ths = np.where(small, 1.0, th)
    a = np.where(small, 1.0 - th * th / 6.0, np.sin(ths) / ths)
    b = np.where(small, 0.5 - th * th / 24.0, (1.0 - np.cos(ths)) / (ths * ths))
    eye = np.broadcast_to(np.eye(3), K.shape)
    return eye + a[:, None, None] * K + b[:, None, None] * (K @ K)


def matrix_to_rodrigues(R):
    """(n,3,3) rotation matrices -> (n,3) angle-axis vectors."""
    R = np.asarray(R, np.float64).reshape(-1, 3, 3)
    out = np.zeros((R.shape[0], 3))
    for k, M in enumerate(R):
        c = np.clip((np.trace(M) - 1.0) * 0.5, -1.0, 1.0)
        th = np.arccos(c)
        w = np.array([M[2, 1] - M[1, 2], M[0, 2] - M[2, 0], M[1, 0] - M[0, 1]])
        if th < 1e-8:
            out[k] = 0.5 * w
        elif np.pi - th < 1e-6:  # near pi: axis from the symmetric part
            A = (M + np.eye(3)) * 0.5
            ax = np.sqrt(np.maximum(np.diag(A), 0.0))
            i = int(np.argmax(ax))
            ax = A[i] / ax[i]
            if np.dot(ax, w) < 0:
                ax = -ax
            out[k] = th * ax / np.linalg.norm(ax)
        else:
            out[k] = th * w / (2.0 * np.sin(th))
    return out


def _undistort(xy, f, k1, k2):
    """Measurements f r(p) p -> f p (per observation arrays)."""
    pd = xy / f[:, None]
    rd = np.linalg.norm(pd, axis=1)
    rho = rd.copy()
    for _ in range(20):  # Newton on g(rho) = rho (1 + k1 rho^2 + k2 rho^4) - rd
        r2 = rho * rho
        g = rho * (1.0 + k1 * r2 + k2 * r2 * r2) - rd
        dg = 1.0 + 3.0 * k1 * r2 + 5.0 * k2 * r2 * r2
        rho = rho - g / np.where(np.abs(dg) < 1e-12, 1.0, dg)
    scale = np.where(rd > 0, rho / np.where(rd > 0, rd, 1.0), 1.0)
    return xy * scale[:, None]


_D = np.diag([1.0, -1.0, -1.0])


def parse_bal(path):
    """Raw content of a BAL file: dict(cam_index, pt_index, xy, cameras(n,9),
    points(n,3)).  Raises ValueError on a malformed file."""
    with open(path, "r") as fh:
        tok = fh.read().split()
    if len(tok) < 3:
        raise ValueError("BAL: missing header")
    n_cam, n_pt, n_obs = int(tok[0]), int(tok[1]), int(tok[2])
    need = 3 + 4 * n_obs + 9 * n_cam + 3 * n_pt
    if n_cam < 0 or n_pt < 0 or n_obs < 0 or len(tok) != need:
        raise ValueError("BAL: expected %d values for %d cameras / %d points / "
                         "%d observations, found %d" % (need, n_cam, n_pt, n_obs, len(tok)))
    o = np.array(tok[3:3 + 4 * n_obs], dtype=np.float64).reshape(n_obs, 4)
    ci, pi = o[:, 0].astype(np.int64), o[:, 1].astype(np.int64)
    if n_obs and (ci.min() < 0 or ci.max() >= n_cam or pi.min() < 0 or pi.max() >= n_pt):
        raise ValueError("BAL: observation refers to a camera / point out of range")
    p = 3 + 4 * n_obs
    cams = np.array(tok[p:p + 9 * n_cam], dtype=np.float64).reshape(n_cam, 9)
    pts = np.array(tok[p + 9 * n_cam:], dtype=np.float64).reshape(n_pt, 3)
    return dict(cam_index=ci.astype(np.int32), pt_index=pi.astype(np.int32),
                xy=np.ascontiguousarray(o[:, 2:4]), cameras=cams, points=pts)


def load_bal(path, undistort=True, n_fixed_poses=0):
    """BAL file -> scene dict in the layout of scenes.py (feed it to
    scenes.scaled_problem or to the Add* facade).  `n_fixed_poses` leading
    cameras are held fixed (BAL itself fixes no gauge).  With undistort=False a
    file with non-zero k1 / k2 is refused."""
    raw = parse_bal(path)
    cams = raw["cameras"]
    n_cam = cams.shape[0]
    f, k1, k2 = cams[:, 6], cams[:, 7], cams[:, 8]
    xy = raw["xy"]
    ci = raw["cam_index"]
    if np.any(k1 != 0) or np.any(k2 != 0):
        if not undistort:
            raise ValueError("BAL: radial distortion present and undistort=False "
                             "(the solver's camera model is a pinhole)")
        xy = _undistort(xy, f[ci], k1[ci], k2[ci])
    R = _D @ rodrigues_to_matrix(cams[:, 0:3])
    t = cams[:, 3:6] @ _D.T
    T_jw = np.tile(np.eye(4), (n_cam, 1, 1))
    T_jw[:, :3, :3] = R
    T_jw[:, :3, 3] = t
    T_wc = np.tile(np.eye(4), (n_cam, 1, 1))  # pose = camera-to-world, as in scenes.py
    T_wc[:, :3, :3] = np.transpose(R, (0, 2, 1))
    T_wc[:, :3, 3] = -np.einsum("nij,nj->ni", T_wc[:, :3, :3], t)
    intr = np.stack([f, f, np.zeros(n_cam), np.zeros(n_cam)], axis=1)
    return dict(
        intr=intr, T_cj=np.tile(np.eye(4), (n_cam, 1, 1)),
        T_wc_init=T_wc, X_init=raw["points"].copy(),
        pose_fixed=np.arange(n_cam) < n_fixed_poses,
        pt_fixed=np.zeros(raw["points"].shape[0], bool),
        obs_cam=ci.copy(), obs_pose=ci.copy(), obs_pt=raw["pt_index"].copy(),
        obs_uv=np.stack([xy[:, 0], -xy[:, 1]], axis=1),
        bal=dict(f=f.copy(), k1=k1.copy(), k2=k2.copy()))


def save_bal(path, scene, poses="T_wc_init", points="X_init"):
    """Scene dict (scenes.py layout) -> BAL file.  Every (camera, pose) pair that
    occurs in the observations becomes one BAL camera with
    T = T_cj[camera] * pose^-1; pixels are shifted by (cx, cy) and v is scaled
    by fx / fy so that one focal length describes the camera (exact).
    Returns (bal_camera_of_observation, pairs) with pairs[k] = (camera, pose)."""
    intr = np.asarray(scene["intr"], np.float64)
    T_cj = np.asarray(scene["T_cj"], np.float64)
    T_wc = np.asarray(scene[poses], np.float64)
    X = np.asarray(scene[points], np.float64)
    oc, op = np.asarray(scene["obs_cam"]), np.asarray(scene["obs_pose"])
    key = op.astype(np.int64) * intr.shape[0] + oc
    uniq, inv = np.unique(key, return_inverse=True)
    pairs = np.stack([uniq % intr.shape[0], uniq // intr.shape[0]], axis=1)
    cam9 = np.zeros((uniq.size, 9))
    for k, (c, j) in enumerate(pairs):
        Rw, tw = T_wc[j, :3, :3], T_wc[j, :3, 3]
        T_jw = np.eye(4)
        T_jw[:3, :3] = Rw.T
        T_jw[:3, 3] = -Rw.T @ tw
        T = T_cj[c] @ T_jw
        cam9[k, 0:3] = matrix_to_rodrigues((_D @ T[:3, :3])[None])[0]
        cam9[k, 3:6] = _D @ T[:3, 3]
        cam9[k, 6] = intr[c, 0]
    uv = np.asarray(scene["obs_uv"], np.float64)
    x = uv[:, 0] - intr[oc, 2]
    y = -(uv[:, 1] - intr[oc, 3]) * (intr[oc, 0] / intr[oc, 1])
    with open(path, "w") as fh:
        fh.write("%d %d %d\n" % (uniq.size, X.shape[0], uv.shape[0]))
        for k in range(uv.shape[0]):
            fh.write("%d %d %.17g %.17g\n" % (inv[k], scene["obs_pt"][k], x[k], y[k]))
        for row in cam9:
            for v in row:
                fh.write("%.17g\n" % v)
        for row in X:
            for v in row:
                fh.write("%.17g\n" % v)
    return inv.astype(np.int32), pairs.astype(np.int32)


def reprojection_residuals(scene, poses="T_wc_init", points="X_init"):
    """(n_obs, 2) pixel residuals of a scene dict (numpy, fp64; for checks)."""
    intr = np.asarray(scene["intr"], np.float64)
    T_cj = np.asarray(scene["T_cj"], np.float64)
    T_wc = np.asarray(scene[poses], np.float64)
    X = np.asarray(scene[points], np.float64)[scene["obs_pt"]]
    j, c = scene["obs_pose"], scene["obs_cam"]
    Rw, tw = T_wc[j, :3, :3], T_wc[j, :3, 3]
    Xb = np.einsum("nji,nj->ni", Rw, X - tw)  # R^T (X - t)
    Xc = np.einsum("nij,nj->ni", T_cj[c, :3, :3], Xb) + T_cj[c, :3, 3]
    u = intr[c, 0] * Xc[:, 0] / Xc[:, 2] + intr[c, 2]
    v = intr[c, 1] * Xc[:, 1] / Xc[:, 2] + intr[c, 3]
    return np.stack([u, v], axis=1) - np.asarray(scene["obs_uv"], np.float64)
