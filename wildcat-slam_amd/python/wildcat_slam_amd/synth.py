"""Deterministic synthetic inputs for the hot path: generator G1 "room raycast" and
G2 "patch lattice", plus window builders that turn them into the (surfels, poses, IMU states, sample times)
the matcher and the solver consume.  Pure numpy; no reference code involved.
"""
import numpy as np

from . import records as R

SEED = 0x57494C44
VS = float(np.float32(0.8))  # root voxel size as the reference sees it: (double)0.8f
T0 = 1.6e9  # epoch-sized stamps, as ROS delivers them


# --------------------------------------------------------------------------------------------------------------------
# small SO(3) helpers (numpy, vectorised) — used only to synthesise trajectories
def so3_exp_mat(w):
    w = np.asarray(w, np.float64)
    th = np.linalg.norm(w, axis=-1, keepdims=True)
    th = np.maximum(th, 1e-300)
    a = w / th
    K = np.zeros(w.shape[:-1] + (3, 3))
    K[..., 0, 1], K[..., 0, 2] = -a[..., 2], a[..., 1]
    K[..., 1, 0], K[..., 1, 2] = a[..., 2], -a[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -a[..., 1], a[..., 0]
    s, c = np.sin(th)[..., None], np.cos(th)[..., None]
    return np.eye(3) + s * K + (1 - c) * (K @ K)


def mat_to_quat(Rm):
    """rotation matrices (..., 3, 3) -> quaternions (..., 4) as (w, x, y, z), w >= 0."""
    Rm = np.asarray(Rm)
    q = np.empty(Rm.shape[:-2] + (4,))
    tr = Rm[..., 0, 0] + Rm[..., 1, 1] + Rm[..., 2, 2]
    # robust branchless-ish: compute all four candidates and pick the largest
    cand = np.stack(
        [
            1 + tr,
            1 + Rm[..., 0, 0] - Rm[..., 1, 1] - Rm[..., 2, 2],
            1 - Rm[..., 0, 0] + Rm[..., 1, 1] - Rm[..., 2, 2],
            1 - Rm[..., 0, 0] - Rm[..., 1, 1] + Rm[..., 2, 2],
        ],
        -1,
    )
    k = np.argmax(cand, -1)
    m = np.sqrt(np.take_along_axis(cand, k[..., None], -1)[..., 0]) * 2
    w0 = np.stack([m / 4, (Rm[..., 2, 1] - Rm[..., 1, 2]) / m, (Rm[..., 0, 2] - Rm[..., 2, 0]) / m, (Rm[..., 1, 0] - Rm[..., 0, 1]) / m], -1)
    w1 = np.stack([(Rm[..., 2, 1] - Rm[..., 1, 2]) / m, m / 4, (Rm[..., 0, 1] + Rm[..., 1, 0]) / m, (Rm[..., 0, 2] + Rm[..., 2, 0]) / m], -1)
    w2 = np.stack([(Rm[..., 0, 2] - Rm[..., 2, 0]) / m, (Rm[..., 0, 1] + Rm[..., 1, 0]) / m, m / 4, (Rm[..., 1, 2] + Rm[..., 2, 1]) / m], -1)
    w3 = np.stack([(Rm[..., 1, 0] - Rm[..., 0, 1]) / m, (Rm[..., 0, 2] + Rm[..., 2, 0]) / m, (Rm[..., 1, 2] + Rm[..., 2, 1]) / m, m / 4], -1)
    allc = np.stack([w0, w1, w2, w3], -2)
    q = np.take_along_axis(allc, k[..., None, None], -2)[..., 0, :]
    q = q * np.where(q[..., :1] < 0, -1.0, 1.0)
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


# --------------------------------------------------------------------------------------------------------------------
# trajectory used by G1 and by the window builders
def traj_pos(t):
    t = np.asarray(t, np.float64)
    return np.stack([0.5 * t, 0.3 * np.sin(0.4 * t), 0.1 * np.sin(0.7 * t)], -1)


def traj_rotvec(t):
    t = np.asarray(t, np.float64)
    return np.stack([0.05 * np.sin(0.5 * t), 0.04 * np.sin(0.3 * t), 0.2 * t], -1)


def traj_rot(t):
    return so3_exp_mat(traj_rotvec(t))


def make_points(xyz, t):
    pts = np.zeros(len(t), R.POINT)
    xyz = np.asarray(xyz, np.float32)
    pts["x"], pts["y"], pts["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    pts["time"] = t
    pts["ring"] = np.arange(len(t)) % 32
    return pts


def concat_points(*parts):
    """np.concatenate drops the explicit 48-byte layout of POINT; this keeps it."""
    out = np.zeros(sum(len(p) for p in parts), R.POINT)
    o = 0
    for p in parts:
        out[o:o + len(p)] = p
        o += len(p)
    return out


# --------------------------------------------------------------------------------------------------------------------
def g2_lattice(n_roots, m=32, seed=SEED, span=60, patches_per_root=8, t_start=T0, duration=0.5, noise=0.005, sample_seed=None,
               pose_fn=None):
    """G2 "patch lattice": `n_roots` distinct 0.8 m root voxels inside a cube of `span` voxels per side centred on
    the origin; each root carries `patches_per_root` planar patches (one per 0.4 m child octant, m points each).
    Returns (points[POINT], info) with info = dict(centres, normals, root_keys)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    total = span**3
    assert n_roots <= total
    flat = rng.choice(total, size=n_roots, replace=False)
    kz, ky, kx = flat % span, (flat // span) % span, flat // (span * span)
    keys = np.stack([kx, ky, kz], -1).astype(np.int64) - span // 2
    root_c = (keys + 0.5) * VS
    octs = np.array([[(o >> 2) & 1, (o >> 1) & 1, o & 1] for o in range(8)], np.float64) * 2 - 1
    if patches_per_root < 8:
        octs = octs[:patches_per_root]
    P = len(octs)
    centres = (root_c[:, None, :] + 0.2 * octs[None, :, :]).reshape(-1, 3)  # (n_roots*P, 3)
    npatch = len(centres)
    nrm = rng.normal(size=(npatch, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    # orthonormal basis of each patch plane
    helper = np.where(np.abs(nrm[:, :1]) < 0.9, np.array([[1.0, 0, 0]]), np.array([[0, 1.0, 0]]))
    u = np.cross(nrm, helper)
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    v = np.cross(nrm, u)
    if sample_seed is not None:  # same world patches, fresh samples on them (a re-observation by another sweep)
        rng = np.random.Generator(np.random.PCG64(sample_seed))
    rad = 0.15 * np.sqrt(rng.random((npatch, m)))
    ang = 2 * np.pi * rng.random((npatch, m))
    h = noise * rng.normal(size=(npatch, m))
    xyz = centres[:, None, :] + (rad * np.cos(ang))[..., None] * u[:, None, :] + (rad * np.sin(ang))[..., None] * v[:, None, :] + h[..., None] * nrm[:, None, :]
    n = npatch * m
    t = t_start + duration * (np.arange(n, dtype=np.float64) / n)
    if pose_fn is not None:  # express every patch through the (erroneous) pose estimate of its time: x -> A x + b, per patch
        A, b = pose_fn(t.reshape(npatch, m)[:, 0])
        xyz = np.einsum("pij,pmj->pmi", A, xyz) + b[:, None, :]
    xyz = xyz.reshape(-1, 3)
    return make_points(xyz, t), dict(centres=centres, normals=nrm, root_keys=keys, patches_per_root=P, m=m)


def g1_room(n_points, seed=SEED, t_start=T0, duration=0.5, noise=0.01, beams=32, spin_hz=10.0):
    """G1 "room raycast": 32-beam spinning lidar inside a 40 x 30 x 10 m box room with 4 interior walls, sensor on
    the trajectory above; returns world-frame points (already 'undistorted' with the true pose), time-ordered."""
    rng = np.random.Generator(np.random.PCG64(seed))
    i = np.arange(n_points)
    t_rel = duration * i / n_points
    beam = i % beams
    elev = np.deg2rad(-16 + 32.0 * beam / (beams - 1))
    az = 2 * np.pi * spin_hz * t_rel + 0.3 * beam
    d_body = np.stack([np.cos(elev) * np.cos(az), np.cos(elev) * np.sin(az), np.sin(elev)], -1)
    Rw = traj_rot(t_rel)
    o = traj_pos(t_rel) + np.array([0, 0, 1.5])
    d = np.einsum("nij,nj->ni", Rw, d_body)
    # axis-aligned planes: (axis, coordinate, lo, hi bounds in the two other axes)
    planes = [
        (0, -20.0, None), (0, 20.0, None), (1, -15.0, None), (1, 15.0, None), (2, 0.0, None), (2, 10.0, None),
        (0, -8.0, ((-15, 2), (0, 10))), (0, 9.0, ((-3, 15), (0, 10))), (1, -6.0, ((-20, -2), (0, 10))), (1, 5.0, ((3, 20), (0, 10))),
    ]
    best = np.full(n_points, np.inf)
    for axis, coord, bounds in planes:
        with np.errstate(divide="ignore", invalid="ignore"):
            s = (coord - o[:, axis]) / d[:, axis]
        with np.errstate(invalid="ignore"):
            hit = o + s[:, None] * d
        ok = (s > 0.3) & np.isfinite(s)
        if bounds is not None:
            others = [a for a in range(3) if a != axis]
            for a, (lo, hi) in zip(others, bounds):
                ok &= (hit[:, a] >= lo) & (hit[:, a] <= hi)
        best = np.where(ok & (s < best), s, best)
    rng_noise = noise * rng.normal(size=n_points)
    good = np.isfinite(best) & (best < 120.0)
    xyz = o + (best + rng_noise)[:, None] * d
    t = t_start + t_rel
    return make_points(xyz[good], t[good])


def g2_scan_sequence(n_scans, n_roots, m=32, seed=SEED, scan_dur=0.5, t_start=T0, sample_dt=0.08, pose_err=(0.01, 2e-4)):
    """`n_scans` sweeps of n_roots * 8 * m points each that re-observe the SAME world patches (fresh samples per sweep), as a
    sensor on the analytic trajectory would see them through a pose ESTIMATE that is off by a smooth error (the drift the
    window solve removes): a point of sweep k is T_est(t) T_true(t)^-1 x_world.  Returns dict(scans=[POINT arrays],
    imu, grav, sample_times): the input of a full odometry step (extraction -> poses -> matching -> solve) at point level."""
    rng = np.random.Generator(np.random.PCG64(seed + 977))
    ep, er = pose_err
    ph = rng.random(6) * 2 * np.pi

    def perturb(tr):
        dp = ep * np.stack([np.sin(0.9 * tr + ph[0]), np.sin(1.3 * tr + ph[1]), np.sin(0.7 * tr + ph[2])], -1)
        dr = er * np.stack([np.sin(1.1 * tr + ph[3]), np.sin(0.8 * tr + ph[4]), np.sin(1.7 * tr + ph[5])], -1)
        return dp, dr

    def pose_fn(t):
        tr = t - T0
        pos_t, R_t = traj_pos(tr), traj_rot(tr)
        dp, dr = perturb(tr)
        R_e = so3_exp_mat(dr) @ R_t
        A = np.einsum("pij,pkj->pik", R_e, R_t)  # R_e R_t^T
        return A, (pos_t + dp) - np.einsum("pij,pj->pi", A, pos_t)

    dur = n_scans * scan_dur
    ns = int(np.floor(dur / sample_dt + 1e-9)) + 2
    sample_times = t_start + sample_dt * np.arange(ns)
    lo = sample_times[0] + 1e-3
    scans = []
    for k in range(n_scans):
        a = max(t_start + k * scan_dur, lo)
        b = min(t_start + (k + 1) * scan_dur, sample_times[-1] - 1e-3)
        pts, _ = g2_lattice(n_roots, m=m, seed=seed, t_start=a, duration=b - a, sample_seed=seed + 1000 + k, pose_fn=pose_fn)
        scans.append(pts)
    imu, grav = imu_states(t_start, sample_times[-1] + 0.01, t_origin=T0, perturb=perturb)
    return dict(scans=scans, imu=imu, grav=grav, sample_times=sample_times)


# --------------------------------------------------------------------------------------------------------------------
def imu_states(t_start, t_end, rate=200.0, t_origin=T0, perturb=None, seed=SEED + 1):
    """IMU states (pose + measurements) on the analytic trajectory at `rate` Hz covering [t_start, t_end].
    `perturb(t) -> (dpos (n,3), drotvec (n,3))` adds a smooth pose error (the thing the solve removes)."""
    n = int(np.floor((t_end - t_start) * rate + 1e-9)) + 1
    t = t_start + np.arange(n) / rate
    tr = t - t_origin
    imu = np.zeros(n, R.IMU_STATE)
    imu["t"] = t
    pos, Rm = traj_pos(tr), traj_rot(tr)
    if perturb is not None:
        dp, dr = perturb(tr)
        pos = pos + dp
        Rm = so3_exp_mat(dr) @ Rm
    imu["pos"] = pos
    imu["quat"] = mat_to_quat(Rm)
    # measurements by central differences of the (perturbed) discrete poses, so that the IMU factors are consistent
    h = 1.0 / rate
    acc_w = np.zeros((n, 3))
    acc_w[1:-1] = (pos[2:] + pos[:-2] - 2 * pos[1:-1]) / (h * h)
    acc_w[0], acc_w[-1] = acc_w[1], acc_w[-2]
    grav = np.array([0, 0, -9.81])
    imu["acc"] = np.einsum("nji,nj->ni", Rm, acc_w - grav)
    dR = np.einsum("nji,njk->nik", Rm[:-1], Rm[1:])
    ang = np.arccos(np.clip((np.trace(dR, axis1=1, axis2=2) - 1) / 2, -1, 1))
    ax = np.stack([dR[:, 2, 1] - dR[:, 1, 2], dR[:, 0, 2] - dR[:, 2, 0], dR[:, 1, 0] - dR[:, 0, 1]], -1)
    with np.errstate(invalid="ignore", divide="ignore"):
        w = np.where(ang[:, None] > 1e-12, ax / (2 * np.sin(ang))[:, None] * ang[:, None], ax / 2) / h
    gyr = np.zeros((n, 3))
    gyr[:-1] = w
    gyr[-1] = w[-1]
    imu["gyr"] = gyr
    return imu, grav


def surfel_window(n_scans, patches_per_scan, seed=SEED, sample_dt=0.08, scan_dur=0.5, t_start=T0, cube=30.0,
                  pose_err=(0.01, 2e-4), fixed_patches=0):
    """Surfel-level window (skips extraction): `n_scans` sweeps each re-observing the same `patches_per_scan` world
    patches.  Returns dict(surf, pose, in_body, imu, sample_times, grav, fix_surf, fix_pose, true_pairs).
    Surfels are already in the body frame with the (perturbed) pose of their timestamp attached — the state the
    reference is in after UpdateSurfelPoses (lidar_odometry.cc:527)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    P = patches_per_scan
    c_w = (rng.random((P, 3)) - 0.5) * cube
    n_w = rng.normal(size=(P, 3))
    n_w /= np.linalg.norm(n_w, axis=1, keepdims=True)
    n_w *= np.where(np.einsum("ij,ij->i", n_w, c_w) < 0, -1.0, 1.0)[:, None]
    dur = n_scans * scan_dur
    ns = int(np.floor(dur / sample_dt + 1e-9)) + 2
    sample_times = t_start + sample_dt * np.arange(ns)
    t_end = sample_times[-1] + 0.01

    ep, er = pose_err
    ph = rng.random(6) * 2 * np.pi

    def perturb(tr):
        dp = ep * np.stack([np.sin(0.9 * tr + ph[0]), np.sin(1.3 * tr + ph[1]), np.sin(0.7 * tr + ph[2])], -1)
        dr = er * np.stack([np.sin(1.1 * tr + ph[3]), np.sin(0.8 * tr + ph[4]), np.sin(1.7 * tr + ph[5])], -1)
        return dp, dr

    imu, grav = imu_states(t_start, t_end, perturb=perturb)

    def build(times, pidx, rng):
        tr = times - t_start + (t_start - T0)
        pos_t, R_t = traj_pos(tr), traj_rot(tr)
        dp, dr = perturb(tr)
        pos_e, R_e = pos_t + dp, so3_exp_mat(dr) @ R_t
        n = len(times)
        cw = c_w[pidx] + 0.002 * rng.normal(size=(n, 3))
        nw = n_w[pidx] + 0.002 * rng.normal(size=(n, 3))
        nw /= np.linalg.norm(nw, axis=1, keepdims=True)
        surf = np.zeros(n, R.SURFEL)
        surf["t"] = times
        surf["center"] = np.einsum("nji,nj->ni", R_t, cw - pos_t)  # true body-frame observation
        nb = np.einsum("nji,nj->ni", R_t, nw)
        surf["normal"] = nb
        # disc covariance: r^2/4 in-plane, sigma^2 along the normal
        covb = (0.15**2 / 4) * (np.eye(3)[None] - nb[:, :, None] * nb[:, None, :]) + (0.005**2) * nb[:, :, None] * nb[:, None, :]
        surf["cov"] = covb.reshape(n, 9)
        surf["resolution"] = 0.4
        surf["sigma"] = 0.005
        pose = np.zeros(n, R.POSE)
        pose["pos"] = pos_e
        pose["quat"] = mat_to_quat(R_e)
        return surf, pose

    times, pidx = [], []
    lo = sample_times[0] + 1e-4
    for k in range(n_scans):
        a = max(t_start + k * scan_dur, lo)
        b = min(t_start + (k + 1) * scan_dur, sample_times[-1] - 1e-4)
        tk = np.sort(a + (b - a) * rng.random(P))
        perm = rng.permutation(P)
        times.append(tk)
        pidx.append(perm)
    times, pidx = np.concatenate(times), np.concatenate(pidx)
    surf, pose = build(times, pidx, rng)
    out = dict(surf=surf, pose=pose, in_body=np.ones(len(surf), np.uint8), imu=imu, sample_times=sample_times, grav=grav,
               patch_index=pidx, patch_centres=c_w, patch_normals=n_w)
    if fixed_patches:
        # fixed-window surfels: older than the sliding window, exact world pose (identity body pose)
        fidx = rng.choice(P, size=fixed_patches, replace=fixed_patches > P)
        ft = t_start - 5.0 + 4.0 * np.sort(rng.random(fixed_patches))
        fs = np.zeros(fixed_patches, R.SURFEL)
        fs["t"] = ft
        fs["center"] = c_w[fidx] + 0.002 * rng.normal(size=(fixed_patches, 3))
        fn = n_w[fidx]
        fs["normal"] = fn
        covb = (0.15**2 / 4) * (np.eye(3)[None] - fn[:, :, None] * fn[:, None, :]) + (0.005**2) * fn[:, :, None] * fn[:, None, :]
        fs["cov"] = covb.reshape(-1, 9)
        fs["resolution"], fs["sigma"] = 0.4, 0.005
        fp = np.zeros(fixed_patches, R.POSE)
        fp["quat"][:, 0] = 1.0
        out.update(fix_surf=fs, fix_pose=fp, fix_patch_index=fidx)
    return out


# --------------------------------------------------------------------------------------------------------------------
# raw sensor stream for the host facade (LidarOdometry): lidar messages in the LIDAR frame + 200 Hz IMU messages
EXT_R = np.array([[-5.32125e-08, -1, 0], [-1, -5.32125e-08, 0], [0, 0, -1]])  # lio_config.h:23-28 (lidar -> imu)
EXT_T = np.array([-0.001, -0.00855, 0.055])


def _room_hits(o, d, z_floor=-1.5):
    planes = [
        (0, -20.0, None), (0, 20.0, None), (1, -15.0, None), (1, 15.0, None), (2, z_floor, None), (2, z_floor + 10.0, None),
        (0, -8.0, ((-15, 2), (z_floor, z_floor + 10))), (0, 9.0, ((-3, 15), (z_floor, z_floor + 10))),
        (1, -6.0, ((-20, -2), (z_floor, z_floor + 10))), (1, 5.0, ((3, 20), (z_floor, z_floor + 10))),
    ]
    best = np.full(len(o), np.inf)
    for axis, coord, bounds in planes:
        with np.errstate(divide="ignore", invalid="ignore"):
            s = (coord - o[:, axis]) / d[:, axis]
            hit = o + s[:, None] * d
        ok = (s > 0.3) & np.isfinite(s)
        if bounds is not None:
            others = [a for a in range(3) if a != axis]
            for a, (lo, hi) in zip(others, bounds):
                ok &= (hit[:, a] >= lo) & (hit[:, a] <= hi)
        best = np.where(ok & (s < best), s, best)
    return best


def raw_stream(duration, pts_per_s=600_000, seed=SEED, t_start=T0, msg_dt=0.1, gyro_bias=(0.0, 0.0, 0.0), range_noise=0.01,
               imu_rate=200.0, beams=32, spin_hz=10.0, imu_clock_skew=1e-6):
    """-> (list of lidar messages [POINT arrays, lidar frame], imu dict(t, acc, gyr), truth callable t -> (pos, R))"""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = int(duration * pts_per_s)
    tr = duration * np.arange(n) / n
    beam = np.arange(n) % beams
    elev = np.deg2rad(-16 + 32.0 * beam / (beams - 1))
    az = 2 * np.pi * spin_hz * tr + 0.3 * beam
    d_imu = np.stack([np.cos(elev) * np.cos(az), np.cos(elev) * np.sin(az), np.sin(elev)], -1)
    Rw, o = traj_rot(tr), traj_pos(tr)
    d_w = np.einsum("nij,nj->ni", Rw, d_imu)
    rng_true = _room_hits(o, d_w)
    good = np.isfinite(rng_true) & (rng_true < 100.0)
    p_imu = d_imu * (rng_true + range_noise * rng.normal(size=n))[:, None]
    p_lidar = (p_imu - EXT_T) @ EXT_R  # R^T (p - t)
    pts = make_points(p_lidar[good], t_start + tr[good])
    msgs = []
    tt = pts["time"] - t_start
    k = 0
    while k * msg_dt < duration:
        sel = (tt >= k * msg_dt) & (tt < (k + 1) * msg_dt)
        msgs.append(pts[sel].copy())
        k += 1
    # IMU at imu_rate from the analytic trajectory (central differences)
    # the IMU clock is offset by half a period: a lidar point stamped exactly on an IMU sample trips the reference's own
    # CHECK(idx >= 1) in UndistortSweep (lidar_odometry.cc:150)
    # ... and a tiny clock skew keeps IMU stamps from landing exactly on the sample-state grid (first_imu + k * 0.08), where
    # the reference's CHECK_EQ(corrected_last_idx, size - 2) (lidar_odometry.cc:210) would abort
    ti = (np.arange(int(duration * imu_rate) + 3) / imu_rate) * (1 + imu_clock_skew) - 0.5 / imu_rate + 1.37e-4
    h = 1e-4
    acc_w = (traj_pos(ti + h) + traj_pos(ti - h) - 2 * traj_pos(ti)) / (h * h)
    Rm = traj_rot(ti)
    acc = np.einsum("nji,nj->ni", Rm, acc_w + np.array([0, 0, 9.81]))
    dR = np.einsum("nji,njk->nik", traj_rot(ti - h), traj_rot(ti + h))
    gyr = np.stack([dR[:, 2, 1] - dR[:, 1, 2], dR[:, 0, 2] - dR[:, 2, 0], dR[:, 1, 0] - dR[:, 0, 1]], -1) / (4 * h)
    imu = dict(t=t_start + ti, acc=acc, gyr=gyr + np.asarray(gyro_bias))

    def truth(t):
        return traj_pos(t - t_start), traj_rot(t - t_start)

    return msgs, imu, truth
