"""Seeded synthetic scenes for the BA hot path.

* test_ba_scene        — the stereo wall scene of reference test/test_ba.cpp:
                         53-232 (60 poses, 660 landmarks, 34 019 observations),
                         seeded instead of std::random_device.
* synthetic_ba_scene   — configs C2..C4: the same camera rig and trajectory
                         law extended to N poses; every landmark is seen by a
                         window of consecutive poses.
* pose_only_scene      — config C5: reference
                         test/test_compare_ceres_vs_native.cpp:20-95.
All quantities are in USER units (metres, pixels, world->camera 4x4 poses);
the solver facade applies the reference's 0.01 scaling.
"""
import numpy as np

SEED_BASE = 20240600
FX = FY = 525.0
CX, CY = 320.0, 240.0
WIDTH, HEIGHT = 640, 480
BASELINE = 0.12


def _rot(axis, ang):
    c, s = np.cos(ang), np.sin(ang)
    if axis == "x":
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], float)
    if axis == "y":
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], float)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], float)


def stereo_cameras(stereo=True):
    """reference test/test_ba.cpp:79-98: (intr [n,4], T_cj [n,4,4])."""
    intr = [[FX, FY, CX, CY]]
    T = [np.eye(4)]
    if stereo:
        intr.append([FX, FY, CX, CY])
        Tr = np.eye(4)
        Tr[0, 3] = -BASELINE  # inverse of left->right translate(+0.12, 0, 0)
        T.append(Tr)
    return np.array(intr), np.stack(T)


def trajectory(n_pose):
    """World->camera poses T_wc, reference test/test_ba.cpp:132-171."""
    base_to_cam = np.eye(4)
    base_to_cam[:3, :3] = _rot("y", np.pi / 2) @ _rot("z", -np.pi / 2)
    x_step = float(np.float32(0.005))
    y_step = float(np.float32(0.2))
    yaw_step = float(np.float32(0.005))
    w2b = np.eye(4)
    w2b[:3, :3] = _rot("z", -0.1)
    w2b[:3, 3] = [-4.0, -2.5, 0.0]
    Rz = _rot("z", yaw_step)
    out = np.zeros((n_pose, 4, 4))
    for k in range(n_pose):
        w2b[:3, :3] = w2b[:3, :3] @ Rz
        w2b[0, 3] += x_step
        w2b[1, 3] += y_step
        out[k] = w2b @ base_to_cam
    return out


def _inv(T):
    R = T[..., :3, :3]
    t = T[..., :3, 3]
    out = np.zeros_like(T)
    Rt = np.swapaxes(R, -1, -2)
    out[..., :3, :3] = Rt
    out[..., :3, 3] = -np.einsum("...ij,...j->...i", Rt, t)
    out[..., 3, 3] = 1.0
    return out


def _wall_points():
    """reference test/test_ba.cpp:53-77 (float loop counters)."""
    pts = []
    z = np.float32(1.7)
    while z <= np.float32(5.7):
        y = np.float32(0.0)
        while y <= np.float32(26.0):
            pts.append([8.5, float(y), float(z)])
            y = np.float32(y + np.float32(0.4))
        z = np.float32(z + np.float32(0.4))
    return np.array(pts, float)


def test_ba_scene(seed=SEED_BASE + 1, pixel_sigma=0.0):
    """Scene of reference test/test_ba.cpp (config C1)."""
    rng = np.random.default_rng(seed)
    n_pose, n_fixed = 60, 5
    intr, T_cj = stereo_cameras(True)
    T_wc_true = trajectory(n_pose)
    X_true = _wall_points()
    n_pt = X_true.shape[0]
    T_wc_init = T_wc_true.copy()
    T_wc_init[n_fixed:, :3, 3] += rng.uniform(-0.1, 0.1, (n_pose - n_fixed, 3)) \
        .astype(np.float32)
    X_init = X_true + rng.uniform(-0.5, 0.5, (n_pt, 3)).astype(np.float32)
    T_cw = _inv(T_wc_true)
    cams, poses, pts, pix = [], [], [], []
    for j in range(n_pose):
        local = X_true @ T_cw[j, :3, :3].T + T_cw[j, :3, 3]
        for c in range(2):
            Xc = local @ T_cj[c, :3, :3].T + T_cj[c, :3, 3]
            inv_z = (1.0 / Xc[:, 2]).astype(np.float32).astype(np.float64)
            u = intr[c, 0] * Xc[:, 0] * inv_z + intr[c, 2]
            v = intr[c, 1] * Xc[:, 1] * inv_z + intr[c, 3]
            if pixel_sigma > 0:
                u = u + rng.normal(0, pixel_sigma, n_pt)
                v = v + rng.normal(0, pixel_sigma, n_pt)
            seen = (u < WIDTH) & (u > 0) & (v < HEIGHT) & (v > 0)
            idx = np.nonzero(seen)[0]
            cams.append(np.full(idx.size, c, np.int32))
            poses.append(np.full(idx.size, j, np.int32))
            pts.append(idx.astype(np.int32))
            pix.append(np.stack([u[idx], v[idx]], axis=1))
    return dict(
        intr=intr, T_cj=T_cj, T_wc_true=T_wc_true, T_wc_init=T_wc_init,
        X_true=X_true, X_init=X_init,
        pose_fixed=np.arange(n_pose) < n_fixed,
        pt_fixed=np.zeros(n_pt, bool),
        obs_cam=np.concatenate(cams), obs_pose=np.concatenate(poses),
        obs_pt=np.concatenate(pts), obs_uv=np.concatenate(pix, axis=0))


def synthetic_ba_scene(n_pose, n_pt, window, stereo, seed, n_fixed=5,
                       pixel_sigma=0.0, pose_noise=0.1, point_noise=0.5, dropout=0.0):
    """Configs C2..C4: every landmark is seen by `window`
    consecutive poses in every camera -> n_obs = n_pt * window * n_cam.
    dropout > 0 (config C4R): every observation is dropped independently with that
    probability (occlusion / track loss), a landmark keeps at least two — the
    observation patterns then differ from landmark to landmark."""
    rng = np.random.default_rng(seed)
    intr, T_cj = stereo_cameras(stereo)
    n_cam = intr.shape[0]
    T_wc_true = trajectory(n_pose)
    T_cw = _inv(T_wc_true)
    T_cam_w = np.einsum("cij,pjk->pcik", T_cj, T_cw)   # [pose, cam] world->cam
    half = window // 2
    X_true = np.zeros((n_pt, 3))
    first = np.zeros(n_pt, np.int64)
    uv_all = np.zeros((n_pt, window, n_cam, 2))
    done = 0
    while done < n_pt:
        m = int((n_pt - done) * 1.3) + 64
        center = rng.integers(half, n_pose - (window - half) + 1, m)
        depth = rng.uniform(4.0, 12.0, m)
        pu = rng.uniform(0, WIDTH, m)
        pv = rng.uniform(0, HEIGHT, m)
        Xc = np.stack([(pu - CX) / FX * depth, (pv - CY) / FY * depth, depth], 1)
        Tw = T_wc_true[center]
        Xw = np.einsum("mij,mj->mi", Tw[:, :3, :3], Xc) + Tw[:, :3, 3]
        f0 = center - half
        ok = np.ones(m, bool)
        uvw = np.zeros((m, window, n_cam, 2))
        for w in range(window):
            T = T_cam_w[f0 + w]                      # [m, cam, 4, 4]
            Xl = np.einsum("mcij,mj->mci", T[:, :, :3, :3], Xw) + T[:, :, :3, 3]
            z = Xl[..., 2]
            u = intr[None, :, 0] * Xl[..., 0] / z + intr[None, :, 2]
            v = intr[None, :, 1] * Xl[..., 1] / z + intr[None, :, 3]
            good = (z > 0) & (u > 0) & (u < WIDTH) & (v > 0) & (v < HEIGHT)
            ok &= good.all(axis=1)
            uvw[:, w, :, 0] = u
            uvw[:, w, :, 1] = v
        idx = np.nonzero(ok)[0][:n_pt - done]
        k = idx.size
        X_true[done:done + k] = Xw[idx]
        first[done:done + k] = f0[idx]
        uv_all[done:done + k] = uvw[idx]
        done += k
    # observation list, insertion order pose-major, left then right, landmark
    lm = np.repeat(np.arange(n_pt), window * n_cam)
    wi = np.tile(np.repeat(np.arange(window), n_cam), n_pt)
    ci = np.tile(np.arange(n_cam), n_pt * window)
    pose = first[lm] + wi
    uv = uv_all.reshape(-1, 2)
    if pixel_sigma > 0:
        uv = uv + rng.normal(0, pixel_sigma, uv.shape)
    if dropout > 0:
        keep = rng.uniform(size=lm.size) >= dropout
        # (a landmark keeps its first two observations whatever the draw)
        first_two = np.tile(np.arange(window * n_cam) < 2, n_pt)
        keep |= first_two
        lm, ci, pose, uv = lm[keep], ci[keep], pose[keep], uv[keep]
    order = np.lexsort((lm, ci, pose))
    T_wc_init = T_wc_true.copy()
    T_wc_init[n_fixed:, :3, 3] += rng.uniform(-pose_noise, pose_noise,
                                              (n_pose - n_fixed, 3))
    X_init = X_true + rng.uniform(-point_noise, point_noise, (n_pt, 3))
    return dict(
        intr=intr, T_cj=T_cj, T_wc_true=T_wc_true, T_wc_init=T_wc_init,
        X_true=X_true, X_init=X_init,
        pose_fixed=np.arange(n_pose) < n_fixed,
        pt_fixed=np.zeros(n_pt, bool),
        obs_cam=ci[order].astype(np.int32),
        obs_pose=pose[order].astype(np.int32),
        obs_pt=lm[order].astype(np.int32), obs_uv=uv[order])


# name -> (n_pose, n_pt, window, stereo, seed)   (BASELINE.json configs)
CONFIGS = {
    "C2": (200, 50_000, 10, False, SEED_BASE + 2),
    "C3": (500, 200_000, 5, True, SEED_BASE + 3),
    "C4": (1000, 500_000, 5, True, SEED_BASE + 4),
}


def dense_covisibility_scene(n_pose, n_pt, views, seed, n_fixed=5, pose_noise=0.02,
                            point_noise=0.1, pixel_sigma=0.0):
    """Off-the-happy-path scene: the rig circles a cloud of points and every
    landmark is seen from `views` RANDOM poses (loop closures everywhere), so
    that with enough landmarks every pair of poses shares some: the reduced
    camera system has no zero block and the structure-aware solve degenerates
    to the plain dense sweep.  One camera."""
    rng = np.random.default_rng(seed)
    intr = np.array([[FX, FY, CX, CY]])
    T_cj = np.eye(4)[None]
    ang = 2 * np.pi * np.arange(n_pose) / n_pose
    T_wc_true = np.tile(np.eye(4), (n_pose, 1, 1))
    radius = 12.0
    for k in range(n_pose):
        # camera on a circle of radius 12 m, looking at the centre
        c = np.array([radius * np.cos(ang[k]), radius * np.sin(ang[k]), 0.3 * np.sin(3 * ang[k])])
        z = -c / np.linalg.norm(c)
        x = np.cross([0.0, 0.0, 1.0], z)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        T_wc_true[k, :3, :3] = np.stack([x, y, z], axis=1)
        T_wc_true[k, :3, 3] = c
    X_true = rng.uniform(-2.0, 2.0, (n_pt, 3))
    T_cw = _inv(T_wc_true)
    pose = np.stack([rng.choice(n_pose, views, replace=False) for _ in range(n_pt)])
    pose.sort(axis=1)
    lm = np.repeat(np.arange(n_pt), views)
    pj = pose.reshape(-1)
    Xl = np.einsum("nij,nj->ni", T_cw[pj, :3, :3], X_true[lm]) + T_cw[pj, :3, 3]
    uv = np.stack([FX * Xl[:, 0] / Xl[:, 2] + CX, FY * Xl[:, 1] / Xl[:, 2] + CY], 1)
    assert (Xl[:, 2] > 0).all()
    if pixel_sigma > 0:
        uv = uv + rng.normal(0, pixel_sigma, uv.shape)
    order = np.lexsort((lm, pj))
    T_wc_init = T_wc_true.copy()
    T_wc_init[n_fixed:, :3, 3] += rng.uniform(-pose_noise, pose_noise, (n_pose - n_fixed, 3))
    return dict(
        intr=intr, T_cj=T_cj, T_wc_true=T_wc_true, T_wc_init=T_wc_init,
        X_true=X_true, X_init=X_true + rng.uniform(-point_noise, point_noise, (n_pt, 3)),
        pose_fixed=np.arange(n_pose) < n_fixed, pt_fixed=np.zeros(n_pt, bool),
        obs_cam=np.zeros(lm.size, np.int32), obs_pose=pj[order].astype(np.int32),
        obs_pt=lm[order].astype(np.int32), obs_uv=uv[order])


# configurations OFF the headline's happy path (bench.py --config ...)
OFFPATH = {
    # C4 with per-observation dropout (p = 0.15): real visibility — occlusion, image
    # borders, track loss — breaks the exact repetition of observation patterns that the
    # covisibility groups of the headline scene rely on
    "C4R": ("window_dropout", (1000, 500_000, 5, True, SEED_BASE + 4, 0.15)),
    # mono, windows of 20 poses: a landmark's 210 block pairs exceed the 128 register
    # slots of a super-run and its pose set the 10 poses of a covisibility group ->
    # everything goes through k_schur_partial's global triple list
    "W20": ("window", (500, 100_000, 20, False, SEED_BASE + 7)),
    # 1000 poses, every landmark seen from 8 random poses: S is fully dense
    "DENSE1K": ("dense", (1000, 60_000, 8, SEED_BASE + 8)),
}


def config_scene(name, scale=1.0, pixel_sigma=0.0, landmark_factor=1):
    """Scene of a BASELINE.json config; scale<1 shrinks poses and landmarks
    proportionally (parity-test sizes); pixel_sigma is the measurement noise in
    pixels (0, and a second run at 0.5); landmark_factor multiplies
    landmarks and observations at FIXED poses (bench.py --weak: the per-GPU share
    of the sharded work stays that of the one-GPU problem)."""
    if name == "C1":
        if landmark_factor != 1:
            raise ValueError("C1 is the fixed scene of test_ba.cpp: no landmark_factor")
        return test_ba_scene(pixel_sigma=pixel_sigma)
    if name in OFFPATH:
        kind, par = OFFPATH[name]
        if kind == "dense":
            n_pose, n_pt, views, seed = par
            return dense_covisibility_scene(max(20, int(round(n_pose * scale))),
                                            max(200, int(round(n_pt * scale))) * landmark_factor,
                                            views, seed, pixel_sigma=pixel_sigma)
        if kind == "window_dropout":
            n_pose, n_pt, window, stereo, seed, drop = par
            return synthetic_ba_scene(max(window + 6, int(round(n_pose * scale))),
                                      max(16, int(round(n_pt * scale))) * landmark_factor, window,
                                      stereo, seed, pixel_sigma=pixel_sigma, dropout=drop)
        n_pose, n_pt, window, stereo, seed = par
        return synthetic_ba_scene(max(window + 6, int(round(n_pose * scale))),
                                  max(16, int(round(n_pt * scale))) * landmark_factor, window,
                                  stereo, seed, pixel_sigma=pixel_sigma)
    n_pose, n_pt, window, stereo, seed = CONFIGS[name]
    n_pose = max(window + 6, int(round(n_pose * scale)))
    n_pt = max(16, int(round(n_pt * scale))) * landmark_factor
    return synthetic_ba_scene(n_pose, n_pt, window, stereo, seed,
                              pixel_sigma=pixel_sigma)


def pose_window_subscene(scene, p0, p1, n_fixed=5):
    """The part of a window scene that lives on poses [p0, p1): those poses (the
    first n_fixed of them fixed at their true values) and every landmark all of
    whose observations fall inside; indices re-based.  Used to cut small test
    problems out of a BASELINE config that keep a specific landmark's geometry."""
    n_pt = scene["X_true"].shape[0]
    first = np.full(n_pt, np.iinfo(np.int64).max)
    last = np.full(n_pt, -1)
    np.minimum.at(first, scene["obs_pt"], scene["obs_pose"])
    np.maximum.at(last, scene["obs_pt"], scene["obs_pose"])
    keep_pt = (first >= p0) & (last < p1)
    new_index = np.cumsum(keep_pt) - 1
    keep_obs = keep_pt[scene["obs_pt"]]
    out = dict(
        intr=scene["intr"], T_cj=scene["T_cj"],
        T_wc_true=scene["T_wc_true"][p0:p1].copy(),
        T_wc_init=scene["T_wc_init"][p0:p1].copy(),
        X_true=scene["X_true"][keep_pt], X_init=scene["X_init"][keep_pt],
        pose_fixed=np.arange(p1 - p0) < n_fixed,
        pt_fixed=np.zeros(int(keep_pt.sum()), bool),
        obs_cam=scene["obs_cam"][keep_obs],
        obs_pose=(scene["obs_pose"][keep_obs] - p0).astype(np.int32),
        obs_pt=new_index[scene["obs_pt"][keep_obs]].astype(np.int32),
        obs_uv=scene["obs_uv"][keep_obs])
    out["T_wc_init"][:n_fixed] = out["T_wc_true"][:n_fixed]
    return out


def scaled_problem(scene):
    """Apply the facade's host preprocessing (reference
    core/full_bundle_adjustment_solver.cpp:72-117,155-180): 0.01 scaling,
    T_jw = pose^-1; returns the C-ABI level arrays."""
    s = 0.01
    intr = scene["intr"] * s
    T_cj = scene["T_cj"].copy()
    T_cj[:, :3, 3] *= s
    T_jw = _inv(scene["T_wc_init"])
    T_jw[:, :3, 3] *= s
    to12 = lambda T: np.concatenate(
        [T[:, :3, :3].reshape(-1, 9), T[:, :3, 3]], axis=1)
    return dict(
        cam_intr=np.ascontiguousarray(intr), cam_T=to12(T_cj),
        pose_T=to12(T_jw), pose_fixed=scene["pose_fixed"].astype(np.uint8),
        pt_X=np.ascontiguousarray(scene["X_init"] * s),
        pt_fixed=scene["pt_fixed"].astype(np.uint8),
        obs_cam=scene["obs_cam"].astype(np.int32),
        obs_pose=scene["obs_pose"].astype(np.int32),
        obs_pt=scene["obs_pt"].astype(np.int32),
        obs_uv=np.ascontiguousarray(scene["obs_uv"] * s))


def pose_only_scene(n=10_000, seed=SEED_BASE + 5, pixel_sigma=0.0):
    """Config C5, reference test/test_compare_ceres_vs_native.cpp:20-95."""
    rng = np.random.default_rng(seed)
    fx = fy = 338.0
    cx, cy = 320.0, 240.0
    T_true = np.eye(4, dtype=np.float32)
    T_true[:3, :3] = _rot("y", -0.5).astype(np.float32)
    T_true[:3, 3] = [0.2, 0.3, -1.9]
    X = np.stack([rng.uniform(-1.7, 1.7, n), rng.uniform(-1.3, 1.3, n),
                  rng.uniform(0, 5.0, n) + 1.2], axis=1).astype(np.float32)
    Ti = _inv(T_true.astype(np.float64)).astype(np.float32)
    L = X @ Ti[:3, :3].T + Ti[:3, 3]
    inv_z = (np.float32(1.0) / L[:, 2]).astype(np.float32)
    uv = np.stack([fx * L[:, 0] * inv_z + cx, fy * L[:, 1] * inv_z + cy],
                  axis=1).astype(np.float32)
    if pixel_sigma > 0:
        uv = (uv + rng.normal(0, pixel_sigma, uv.shape)).astype(np.float32)
    return dict(X=X, uv=uv, fx=fx, fy=fy, cx=cx, cy=cy, T_true=T_true,
                T_init=np.eye(4, dtype=np.float32))


def hover_scene(n_pose, n_pt, n_cam, seed, n_fixed=2, visible_frac=1.0,
                pose_noise=0.02, point_noise=0.1):
    """Test scene with arbitrary camera count and dense visibility: the rig
    hovers around the origin looking down +z at a cloud of points, every camera
    of every pose sees every point (or a random `visible_frac` of them).  Used
    for the structure edge cases the window scenes cannot produce: landmarks
    seen by more than 128 poses, rigs with more than 8 cameras, ragged
    landmarks."""
    rng = np.random.default_rng(seed)
    intr = np.tile(np.array([[FX, FY, CX, CY]]), (n_cam, 1))
    T_cj = np.tile(np.eye(4), (n_cam, 1, 1))
    T_cj[:, 0, 3] = -0.05 * np.arange(n_cam)           # cameras side by side
    T_wc_true = np.tile(np.eye(4), (n_pose, 1, 1))
    for k in range(n_pose):
        ang = rng.uniform(-0.03, 0.03, 3)
        T_wc_true[k, :3, :3] = _rot("x", ang[0]) @ _rot("y", ang[1]) @ _rot("z", ang[2])
        T_wc_true[k, :3, 3] = rng.uniform(-0.3, 0.3, 3)
    X_true = np.stack([rng.uniform(-1.5, 1.5, n_pt), rng.uniform(-1.0, 1.0, n_pt),
                       rng.uniform(6.0, 10.0, n_pt)], 1)
    T_cw = _inv(T_wc_true)
    oc, op, oq, uv = [], [], [], []
    for j in range(n_pose):
        for c in range(n_cam):
            T = T_cj[c] @ T_cw[j]
            Xl = X_true @ T[:3, :3].T + T[:3, 3]
            u = intr[c, 0] * Xl[:, 0] / Xl[:, 2] + intr[c, 2]
            v = intr[c, 1] * Xl[:, 1] / Xl[:, 2] + intr[c, 3]
            sel = np.nonzero(rng.uniform(size=n_pt) < visible_frac)[0]
            oc.append(np.full(sel.size, c)); op.append(np.full(sel.size, j))
            oq.append(sel); uv.append(np.stack([u[sel], v[sel]], 1))
    T_wc_init = T_wc_true.copy()
    T_wc_init[n_fixed:, :3, 3] += rng.uniform(-pose_noise, pose_noise,
                                              (n_pose - n_fixed, 3))
    return dict(
        intr=intr, T_cj=T_cj, T_wc_true=T_wc_true, T_wc_init=T_wc_init,
        X_true=X_true, X_init=X_true + rng.uniform(-point_noise, point_noise, (n_pt, 3)),
        pose_fixed=np.arange(n_pose) < n_fixed, pt_fixed=np.zeros(n_pt, bool),
        obs_cam=np.concatenate(oc).astype(np.int32),
        obs_pose=np.concatenate(op).astype(np.int32),
        obs_pt=np.concatenate(oq).astype(np.int32), obs_uv=np.concatenate(uv))


def pose_only_stereo_scene(n=10_000, seed=SEED_BASE + 6, pixel_sigma=0.0,
                           right_missing_frac=0.2):
    """Stereo variant of the pose-only scene (reference
    core/pose_only_bundle_adjustment_solver.cpp:172-399 conventions): the right
    camera sees X_r = left_to_right^-1 * X_l with left_to_right = translate(+0.12,
    0, 0) (reference test/test_ba.cpp:88-97); a fraction of the points has no
    right match, marked by a negative right pixel (:298)."""
    sc = pose_only_scene(n, seed=seed, pixel_sigma=pixel_sigma)
    rng = np.random.default_rng(seed + 1000)
    T_lr = np.eye(4, dtype=np.float32)
    T_lr[0, 3] = BASELINE
    Ti = _inv(sc["T_true"].astype(np.float64)).astype(np.float32)
    L = sc["X"] @ Ti[:3, :3].T + Ti[:3, 3]
    Tr = _inv(T_lr.astype(np.float64)).astype(np.float32)
    Lr = L @ Tr[:3, :3].T + Tr[:3, 3]
    inv_z = (np.float32(1.0) / Lr[:, 2]).astype(np.float32)
    uvr = np.stack([sc["fx"] * Lr[:, 0] * inv_z + sc["cx"],
                    sc["fy"] * Lr[:, 1] * inv_z + sc["cy"]], axis=1).astype(np.float32)
    if pixel_sigma > 0:
        uvr = (uvr + rng.normal(0, pixel_sigma, uvr.shape)).astype(np.float32)
    miss = rng.uniform(size=n) < right_missing_frac
    uvr[miss] = -1.0
    sc.update(uv_right=uvr, T_lr=T_lr, right_missing=miss)
    return sc
