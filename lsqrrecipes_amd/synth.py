"""Seeded synthetic workloads for the five BASELINE.json configs.

The distributions follow the reference's own generators:
  plane   examples/planeEstimation.cxx:152-200
  sphere  examples/sphereEstimation.cxx:136-184
  dense   testing/DenseLinearEquationSystemParametersEstimatorTest.cxx:98-119,
          examples/linearEquationSystemSolver.cxx:44-53,79-81
  US      testing/SinglePointTargetUSCalibrationParametersEstimatorTest.cxx:556-666
Generator: numpy Philox, fixed seeds (data only -- nothing here is on the hot path).
"""
import numpy as np

SEEDS = {"plane": 0x5EED0001, "sphere": 0x5EED0002, "dense": 0x5EED0003, "us": 0x5EED0004,
         "line": 0x5EED0005}


def _rng(seed):
    return np.random.Generator(np.random.Philox(seed))


def plane(n, outlier_frac, seed=SEEDS["plane"], dim=3, sigma=0.4, box=1000.0, outlier_dist=20.0):
    """-> (points (n,dim) float64 shuffled, true params [n,a], is_inlier mask)."""
    g = _rng(seed)
    normal = g.uniform(0.0, 1.0, dim)
    normal /= np.linalg.norm(normal)
    a = g.uniform(-box, box, dim)
    n_out = int(round(n * outlier_frac))
    n_in = n - n_out
    rp = g.uniform(-box, box, (n_in, dim))
    noise = g.normal(0.0, sigma, (n_in, dim))
    tmp = rp - a
    inl = a + noise + (tmp - (tmp @ normal)[:, None] * normal)
    out = np.empty((0, dim))
    while out.shape[0] < n_out:
        cand = g.uniform(-box, box, (max(n_out, 16), dim))
        keep = np.abs((cand - a) @ normal) >= outlier_dist
        out = np.vstack([out, cand[keep]])
    out = out[:n_out]
    pts = np.vstack([inl, out])
    lab = np.concatenate([np.ones(n_in, bool), np.zeros(n_out, bool)])
    perm = g.permutation(n)
    return np.ascontiguousarray(pts[perm]), np.concatenate([normal, a]), lab[perm]


def line(n, outlier_frac, seed=SEEDS["line"], dim=3, sigma=0.4, box=1000.0, outlier_dist=20.0):
    g = _rng(seed)
    d = g.uniform(0.0, 1.0, dim)
    d /= np.linalg.norm(d)
    a = g.uniform(-box, box, dim)
    n_out = int(round(n * outlier_frac))
    n_in = n - n_out
    t = g.uniform(-box, box, n_in)
    inl = a + t[:, None] * d + g.normal(0.0, sigma, (n_in, dim))
    out = np.empty((0, dim))
    while out.shape[0] < n_out:
        cand = g.uniform(-box, box, (max(n_out, 16), dim))
        v = cand - a
        perp = v - (v @ d)[:, None] * d
        out = np.vstack([out, cand[np.linalg.norm(perp, axis=1) >= outlier_dist]])
    out = out[:n_out]
    pts = np.vstack([inl, out])
    lab = np.concatenate([np.ones(n_in, bool), np.zeros(n_out, bool)])
    perm = g.permutation(n)
    return np.ascontiguousarray(pts[perm]), np.concatenate([d, a]), lab[perm]


def sphere(n, outlier_frac, seed=SEEDS["sphere"], dim=3, sigma=0.4, box=1000.0,
           outlier_dist=20.0):
    g = _rng(seed)
    c = g.uniform(-box, box, dim)
    r = g.uniform(0.0, box)
    n_out = int(round(n * outlier_frac))
    n_in = n - n_out
    u = g.uniform(-1.0, 1.0, (n_in, dim))
    u /= np.linalg.norm(u, axis=1)[:, None]
    inl = c + r * u + g.normal(0.0, sigma, (n_in, dim))
    out = np.empty((0, dim))
    while out.shape[0] < n_out:
        cand = g.uniform(-box, box, (max(n_out, 16), dim))
        keep = np.abs(np.linalg.norm(cand - c, axis=1) - r) >= outlier_dist
        out = np.vstack([out, cand[keep]])
    out = out[:n_out]
    pts = np.vstack([inl, out])
    lab = np.concatenate([np.ones(n_in, bool), np.zeros(n_out, bool)])
    perm = g.permutation(n)
    return np.ascontiguousarray(pts[perm]), np.concatenate([c, [r]]), lab[perm]


def dense(m, ncols, outlier_frac=0.05, seed=SEEDS["dense"], noise=0.05, outlier_scale=20.0):
    """-> (augmented rows (m, ncols+1), true x, is_inlier)."""
    g = _rng(seed)
    A = g.uniform(-1.0, 1.0, (m, ncols))
    x = g.uniform(-1.0, 1.0, ncols)
    b = (A @ x) * (1.0 + g.uniform(-noise, noise, m))
    n_out = int(round(m * outlier_frac))
    lab = np.ones(m, bool)
    if n_out:
        idx = g.choice(m, n_out, replace=False)
        b[idx] *= outlier_scale
        lab[idx] = False
    return np.ascontiguousarray(np.hstack([A, b[:, None]])), x, lab


def euler_zyx(wz, wy, wx):
    """R = Rz*Ry*Rx (common/Frame.cxx:87-113)."""
    cz, sz, cy, sy, cx, sx = np.cos(wz), np.sin(wz), np.cos(wy), np.sin(wy), np.cos(wx), np.sin(wx)
    return np.array([[cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx],
                     [sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx],
                     [-sy, cy * sx, cy * cx]])


def us_single(m, outlier_frac=0.0, seed=SEEDS["us"], pixel_sigma=1.0):
    """Cross-wire phantom frames -> (records (m,15) float64, true 11-vector, is_inlier).
    Record = Frame{rotation[3][3], translation[3], int outputFormat(+pad)} + Point2D (120 B)."""
    g = _rng(seed)
    mx, my = 0.143, 0.139
    w3 = g.uniform(0.0, np.pi, 3)
    t3 = g.uniform(-100.0, 100.0, 3)
    t1 = g.uniform(-100.0, 100.0, 3)
    R3 = euler_zyx(*w3)
    rec = np.zeros((m, 15))
    n_out = int(round(m * outlier_frac))
    lab = np.ones(m, bool)
    uv = np.stack([g.uniform(0.0, 640.0, m), g.uniform(0.0, 480.0, m)], axis=1)
    w2 = g.uniform(0.0, np.pi, (m, 3))
    for i in range(m):
        R2 = euler_zyx(*w2[i])
        q3 = R3 @ np.array([mx * uv[i, 0], my * uv[i, 1], 0.0]) + t3
        t2 = t1 - R2 @ q3
        rec[i, 0:9] = R2.reshape(9)
        rec[i, 9:12] = t2
    rec[:, 13:15] = uv + g.normal(0.0, pixel_sigma, (m, 2))
    if n_out:
        idx = g.choice(m, n_out, replace=False)
        rec[idx, 9:12] = g.uniform(-100.0, 100.0, (n_out, 3))
        lab[idx] = False
    truth = np.concatenate([t1, t3, w3, [mx, my]])
    return np.ascontiguousarray(rec), truth, lab


def us_single_fast(m, outlier_frac=0.0, seed=SEEDS["us"], pixel_sigma=1.0):
    """Vectorised us_single for m ~ 1e6 (same distributions, different stream order)."""
    g = _rng(seed)
    mx, my = 0.143, 0.139
    w3 = g.uniform(0.0, np.pi, 3)
    t3 = g.uniform(-100.0, 100.0, 3)
    t1 = g.uniform(-100.0, 100.0, 3)
    R3 = euler_zyx(*w3)
    uv = np.stack([g.uniform(0.0, 640.0, m), g.uniform(0.0, 480.0, m)], axis=1)
    w2 = g.uniform(0.0, np.pi, (m, 3))
    cz, sz = np.cos(w2[:, 0]), np.sin(w2[:, 0])
    cy, sy = np.cos(w2[:, 1]), np.sin(w2[:, 1])
    cx, sx = np.cos(w2[:, 2]), np.sin(w2[:, 2])
    R2 = np.empty((m, 3, 3))
    R2[:, 0, 0] = cz * cy; R2[:, 0, 1] = cz * sy * sx - sz * cx; R2[:, 0, 2] = cz * sy * cx + sz * sx
    R2[:, 1, 0] = sz * cy; R2[:, 1, 1] = sz * sy * sx + cz * cx; R2[:, 1, 2] = sz * sy * cx - cz * sx
    R2[:, 2, 0] = -sy; R2[:, 2, 1] = cy * sx; R2[:, 2, 2] = cy * cx
    q3 = (np.stack([mx * uv[:, 0], my * uv[:, 1], np.zeros(m)], axis=1) @ R3.T) + t3
    t2 = t1 - np.einsum("mij,mj->mi", R2, q3)
    rec = np.zeros((m, 15))
    rec[:, 0:9] = R2.reshape(m, 9)
    rec[:, 9:12] = t2
    rec[:, 13:15] = uv + g.normal(0.0, pixel_sigma, (m, 2))
    lab = np.ones(m, bool)
    n_out = int(round(m * outlier_frac))
    if n_out:
        idx = g.choice(m, n_out, replace=False)
        rec[idx, 9:12] = g.uniform(-100.0, 100.0, (n_out, 3))
        lab[idx] = False
    return np.ascontiguousarray(rec), np.concatenate([t1, t3, w3, [mx, my]]), lab


def us_pointer(m, outlier_frac=0.0, seed=SEEDS["us"] + 1, pixel_sigma=1.0):
    """Calibrated-pointer frames -> (records (m,18), true 8-vector, is_inlier)."""
    g = _rng(seed)
    mx, my = 0.143, 0.139
    w3 = g.uniform(0.0, np.pi, 3)
    t3 = g.uniform(-100.0, 100.0, 3)
    R3 = euler_zyx(*w3)
    rec = np.zeros((m, 18))
    uv = np.stack([g.uniform(0.0, 640.0, m), g.uniform(0.0, 480.0, m)], axis=1)
    w2 = g.uniform(0.0, np.pi, (m, 3))
    t2 = g.uniform(-100.0, 100.0, (m, 3))
    for i in range(m):
        R2 = euler_zyx(*w2[i])
        q3 = R3 @ np.array([mx * uv[i, 0], my * uv[i, 1], 0.0]) + t3
        rec[i, 0:9] = R2.reshape(9)
        rec[i, 9:12] = t2[i]
        rec[i, 15:18] = R2 @ q3 + t2[i]
    rec[:, 13:15] = uv + g.normal(0.0, pixel_sigma, (m, 2))
    lab = np.ones(m, bool)
    n_out = int(round(m * outlier_frac))
    if n_out:
        idx = g.choice(m, n_out, replace=False)
        rec[idx, 15:18] = g.uniform(-100.0, 100.0, (n_out, 3))
        lab[idx] = False
    return np.ascontiguousarray(rec), np.concatenate([t3, w3, [mx, my]]), lab


def quat_to_matrix(q):
    """[s,qx,qy,qz] -> 3x3 rotation (common/Frame.cxx:750-771)."""
    s, qx, qy, qz = q
    return np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - s * qz), 2 * (qx * qz + s * qy)],
                     [2 * (qx * qy + s * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - s * qx)],
                     [2 * (qx * qz - s * qy), 2 * (qy * qz + s * qx), 1 - 2 * (qx * qx + qy * qy)]])


def absolute_orientation(n, outlier_frac, seed=0x5EED0006, bounds=100.0, max_t=1000.0, sigma=0.5,
                         outlier_shift=5.0):
    """Paired points second = T first + noise (testing/AbsoluteOrientationParametersEstimatorTest.cxx:
    33-79, examples/AbsoluteOrientation.cxx:36-90) -> (records (n,6), [s,qx,qy,qz,t], is_inlier)."""
    g = _rng(seed)
    qx = g.uniform(0.0, 1.0)
    qy = g.uniform(0.0, np.sqrt(1.0 - qx * qx))
    qz = g.uniform(0.0, np.sqrt(1.0 - qx * qx - qy * qy))
    q = np.array([np.sqrt(1.0 - qx * qx - qy * qy - qz * qz), qx, qy, qz])
    t = g.uniform(-max_t, max_t, 3)
    R = quat_to_matrix(q)
    first = g.uniform(-bounds, bounds, (n, 3))
    second = first @ R.T + t + g.normal(0.0, sigma, (n, 3)) if sigma > 0 else first @ R.T + t
    n_out = int(round(n * outlier_frac))
    lab = np.ones(n, bool)
    if n_out:
        idx = g.permutation(n)[:n_out]
        second[idx] += g.uniform(-1, 1, (n_out, 3)) * 10 * outlier_shift + outlier_shift
        lab[idx] = False
    return np.ascontiguousarray(np.hstack([first, second])), np.concatenate([q, t]), lab


def frames_from_pose_rows(rows):
    """rows 'x y z qx qy qz qs' (examples/readme.txt:30-34) -> Frame records (n,13): rotation 9,
    translation 3, one unused slot (int outputFormat + padding)."""
    rows = np.asarray(rows, dtype=np.float64).reshape(-1, 7)
    out = np.zeros((len(rows), 13))
    for i, (x, y, z, qx, qy, qz, qs) in enumerate(rows):
        out[i, :9] = quat_to_matrix([qs, qx, qy, qz]).ravel()
        out[i, 9:12] = (x, y, z)
    return out


def pivot(n, outlier_frac, seed=0x5EED0007, sigma=0.15):
    """Synthetic pivoting poses: R_i tip + t_i = pivot (+ noise), outliers with displaced t."""
    g = _rng(seed)
    tip = np.array([-17.0, 1.0, -157.0])
    piv = np.array([147.0, -63.0, -1042.0])
    out = np.zeros((n, 13))
    lab = np.ones(n, bool)
    n_out = int(round(n * outlier_frac))
    lab[g.permutation(n)[:n_out]] = False
    for i in range(n):
        q = g.normal(size=4)
        q /= np.linalg.norm(q)
        R = quat_to_matrix(q)
        t = piv - R @ tip + g.normal(0.0, sigma, 3)
        if not lab[i]:
            t += g.uniform(-40, 40, 3)
        out[i, :9] = R.ravel()
        out[i, 9:12] = t
    return out, np.concatenate([tip, piv]), lab


def rays(n, outlier_frac, seed=0x5EED0008, max_range=1000.0, sigma=0.3):
    """Rays p + t n that (approximately) meet in one point (testing/RayIntersectionParametersTest.cxx:
    27-56) -> (records (n,6) = [p, n], point (3), is_inlier)."""
    g = _rng(seed)
    target = g.uniform(-max_range, max_range, 3)
    p = g.uniform(-max_range, max_range, (n, 3))
    lab = np.ones(n, bool)
    n_out = int(round(n * outlier_frac))
    lab[g.permutation(n)[:n_out]] = False
    aim = np.where(lab[:, None], target + g.normal(0.0, sigma, (n, 3)) if sigma > 0 else target,
                   g.uniform(-max_range, max_range, (n, 3)))
    d = aim - p
    d /= np.linalg.norm(d, axis=1)[:, None]
    return np.ascontiguousarray(np.hstack([p, d])), target, lab


def _zyx(az, ay, ax):
    cx, cy, cz, sx, sy, sz = np.cos(ax), np.cos(ay), np.cos(az), np.sin(ax), np.sin(ay), np.sin(az)
    return np.array([[cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx],
                     [sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx],
                     [-sy, cy * sx, cy * cx]])


def phantom_params(omega1_yx, t1_z, t3, omega3_zyx, m_x, m_y):
    """the 41-vector of PlanePhantomUSCalibrationParametersEstimator from its 11 minimal entries
    (PlanePhantomUSCalibrationParametersEstimator.cxx:383-452)"""
    R1 = np.array([-np.sin(omega1_yx[0]), np.cos(omega1_yx[0]) * np.sin(omega1_yx[1]),
                   np.cos(omega1_yx[0]) * np.cos(omega1_yx[1])])
    R3 = _zyx(*omega3_zyx)
    p = [omega1_yx[0], omega1_yx[1], t1_z, *t3, *omega3_zyx, m_x, m_y]
    for a in range(3):
        p += [m_x * R3[j, 0] * R1[a] for j in range(3)]
    for a in range(3):
        p += [m_y * R3[j, 1] * R1[a] for j in range(3)]
    for a in range(3):
        p += [t3[j] * R1[a] for j in range(3)]
    p += list(R1)
    return np.array(p)


def plane_phantom(n, outlier_frac, seed=0x5EED0009, pixel_sigma=1.0):
    """Plane-phantom US calibration frames (testing/PlanePhantomUSCalibrationParametersEstimatorTest.cxx:
    382-548): image point q, pose T2 such that T1 T2 T3 q lies on the phantom plane z = 0.
    -> (records (n,15) as the US single layout, true 41-vector, is_inlier)."""
    g = _rng(seed)
    m_x, m_y = 0.143, 0.139
    o3 = g.uniform(0.0, np.pi, 3)          # x, y, z
    t3 = g.uniform(-100, 100, 3)
    t1 = g.uniform(-100, 100, 3)
    o1 = g.uniform(0.0, np.pi, 3)          # x, y, z
    R3 = _zyx(o3[2], o3[1], o3[0])
    T3 = np.column_stack([m_x * R3[:, 0], m_y * R3[:, 1], R3[:, 2], t3])
    R1 = _zyx(o1[2], o1[1], o1[0])
    truth = phantom_params([o1[1], o1[0]], t1[2], t3, [o3[2], o3[1], o3[0]], m_x, m_y)
    rec = np.zeros((n, 15))
    lab = np.ones(n, bool)
    n_out = int(round(n * outlier_frac))
    lab[g.permutation(n)[:n_out]] = False
    for i in range(n):
        q = np.array([g.uniform(0, 640), g.uniform(0, 480)])
        pi = np.array([g.uniform(-100, 100), g.uniform(-100, 100), 0.0])
        if not lab[i]:
            pi[2] = g.uniform(20.0, 100.0) * (1 if g.random() < 0.5 else -1)   # off the plane
        o2 = g.uniform(0.0, np.pi, 3)
        R2 = _zyx(o2[2], o2[1], o2[0])
        qt = T3 @ np.array([q[0], q[1], 0.0, 1.0])
        pt = R1.T @ (pi - t1)               # T1^-1 p
        t2 = pt - R2 @ qt
        rec[i, 0:9] = R2.ravel()
        rec[i, 9:12] = t2
        rec[i, 13:15] = q + (g.normal(0.0, pixel_sigma, 2) if pixel_sigma > 0 else 0.0)
    return rec, truth, lab


def plane_phantom_fast(n, outlier_frac, seed=0x5EED0009, pixel_sigma=1.0):
    """Vectorised plane_phantom for n ~ 1e6 (same distributions, different stream order)."""
    g = _rng(seed)
    m_x, m_y = 0.143, 0.139
    o3 = g.uniform(0.0, np.pi, 3)
    t3 = g.uniform(-100, 100, 3)
    t1 = g.uniform(-100, 100, 3)
    o1 = g.uniform(0.0, np.pi, 3)
    R3, R1 = _zyx(o3[2], o3[1], o3[0]), _zyx(o1[2], o1[1], o1[0])
    truth = phantom_params([o1[1], o1[0]], t1[2], t3, [o3[2], o3[1], o3[0]], m_x, m_y)
    uv = np.stack([g.uniform(0.0, 640.0, n), g.uniform(0.0, 480.0, n)], axis=1)
    on_plane = np.stack([g.uniform(-100, 100, n), g.uniform(-100, 100, n), np.zeros(n)], axis=1)
    lab = np.ones(n, bool)
    n_out = int(round(n * outlier_frac))
    if n_out:
        idx = g.choice(n, n_out, replace=False)
        lab[idx] = False
        on_plane[idx, 2] = g.uniform(20.0, 100.0, n_out) * np.where(g.random(n_out) < 0.5, 1.0, -1.0)
    w2 = g.uniform(0.0, np.pi, (n, 3))
    cz, sz = np.cos(w2[:, 0]), np.sin(w2[:, 0])
    cy, sy = np.cos(w2[:, 1]), np.sin(w2[:, 1])
    cx, sx = np.cos(w2[:, 2]), np.sin(w2[:, 2])
    R2 = np.empty((n, 3, 3))
    R2[:, 0, 0] = cz * cy; R2[:, 0, 1] = cz * sy * sx - sz * cx; R2[:, 0, 2] = cz * sy * cx + sz * sx
    R2[:, 1, 0] = sz * cy; R2[:, 1, 1] = sz * sy * sx + cz * cx; R2[:, 1, 2] = sz * sy * cx - cz * sx
    R2[:, 2, 0] = -sy; R2[:, 2, 1] = cy * sx; R2[:, 2, 2] = cy * cx
    q3 = (np.stack([m_x * uv[:, 0], m_y * uv[:, 1], np.zeros(n)], axis=1) @ R3.T) + t3
    in_tracker = (on_plane - t1) @ R1              # rows: R1^T (p - t1)
    rec = np.zeros((n, 15))
    rec[:, 0:9] = R2.reshape(n, 9)
    rec[:, 9:12] = in_tracker - np.einsum("mij,mj->mi", R2, q3)
    rec[:, 13:15] = uv + (g.normal(0.0, pixel_sigma, (n, 2)) if pixel_sigma > 0 else 0.0)
    return np.ascontiguousarray(rec), truth, lab


def phantom_check(est, truth, trans_eps=3.0, ang_eps=0.08726646259971647884618453842445, scale_eps=1.0):
    """The acceptance test of testing/PlanePhantomUSCalibrationParametersEstimatorTest.cxx:277-379:
    only T3 is checked (t3 within 3 mm, one of the two Euler solutions within 5 degrees, scales
    within 1.0); the rotation is rebuilt from the derived products est[11..13], est[20..22], est[38]."""
    est = np.asarray(est, dtype=np.float64)
    if est.size == 0:
        return False
    r1 = est[11:14] / (est[9] * est[38])
    r2 = est[20:23] / (est[10] * est[38])
    R = np.column_stack([r1, r2, np.cross(r1, r2)])
    small, half_pi = 0.008726535498373935, np.pi / 2
    h = np.hypot(R[0, 0], R[1, 0])
    y1, y2 = np.arctan2(-R[2, 0], h), np.arctan2(-R[2, 0], -h)
    if abs(y1 - half_pi) > small and abs(y1 + half_pi) > small:
        c1, c2 = np.cos(y1), np.cos(y2)
        z1, x1 = np.arctan2(R[1, 0] / c1, R[0, 0] / c1), np.arctan2(R[2, 1] / c1, R[2, 2] / c1)
        z2, x2 = np.arctan2(R[1, 0] / c2, R[0, 0] / c2), np.arctan2(R[2, 1] / c2, R[2, 2] / c2)
    else:
        z1 = z2 = 0.0
        x1 = x2 = np.arctan2(R[0, 1], R[1, 1])
    t = truth[6:9]
    # modulo 2 pi (the reference compares raw differences, which fails spuriously at the +-pi seam)
    ad = lambda a: np.abs(np.remainder(a - t + np.pi, 2 * np.pi) - np.pi)
    ang = np.all(ad(np.array([z1, y1, x1])) < ang_eps) or np.all(ad(np.array([z2, y2, x2])) < ang_eps)
    return bool(np.all(np.abs(est[3:6] - truth[3:6]) < trans_eps) and ang
                and abs(est[9] - truth[9]) < scale_eps and abs(est[10] - truth[10]) < scale_eps)
