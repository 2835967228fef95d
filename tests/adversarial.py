"""Adversarial VALUES for the drop-in calls: non-finite / huge / denormal / negative coordinates,
degenerate point sets, degenerate images.  One table, used twice: by the CPU suite on the kernel emulator
(tests/test_kernel_emulation.py -- it converts float -> int like gfx950) and by the -m gpu suite through the C ABI
(tests/test_gpu_round6.py).  Both hold the product's kernels to the oracle, whose conversions are x86's.  Test data only."""
import numpy as np

NAN, INF = np.nan, np.inf

# calcOpticalFlowPyrLK start points (feature.cpp:136): NaN, +-inf, beyond int32, just inside int32, denormal, negative zero,
# the edge of the +-winSize admissibility window, and one ordinary point at the end (it must survive)
LK_POINTS = np.array([[NAN, 40], [40, NAN], [NAN, NAN], [INF, 30], [-INF, 30], [30, INF], [1e30, 20], [-1e30, 20], [20, 1e30],
                      [3e9, 20], [-3e9, 20], [2147483520.0, 20], [1e-42, 1e-42], [-0.0, 5], [-1e-42, 50], [-10.999, 50],
                      [-11.0, 50], [-11.001, 60], [60.5, -11.5], [120, 64]], np.float32)
LK_N_HOPELESS = 12   # the first 12 fail every hop in the reference (status 0 four times)


def degenerate_images(h, w, seed=1):
    """name -> 8-bit image: constant, saturated, 1-pixel checkerboard, 1-pixel stripes (aperture), binary noise, uniform noise"""
    rng = np.random.default_rng(seed)
    yy, xx = np.indices((h, w))
    return {
        "zeros": np.zeros((h, w), np.uint8),
        "gray": np.full((h, w), 128, np.uint8),
        "white": np.full((h, w), 255, np.uint8),
        "checker1": (((yy + xx) & 1) * 255).astype(np.uint8),
        "stripes1": ((xx & 1) * 255).astype(np.uint8) + np.zeros((h, w), np.uint8),
        "binary": np.where(rng.random((h, w)) < 0.5, 0, 255).astype(np.uint8),
        "noise": rng.integers(0, 256, (h, w), dtype=np.uint8),
        "step": np.where(xx < w // 2, 0, 255).astype(np.uint8),
    }


# triangulatePoints + convertPointsFromHomogeneous (main.cpp:169-171): (left, right) pixel pairs --
# ordinary, ZERO disparity (w = 0 -> scale 1), NEGATIVE disparity (negative depth), NaN / inf / huge in either image,
# the origin, denormals, the principal point in both, a vertical mismatch
TRI_LEFT = np.array([[100, 50], [100, 50], [100, 50], [NAN, 50], [100, NAN], [INF, 50], [1e30, 50], [0, 0], [100, 50],
                     [-100, -50], [1e-40, 1e-40], [607.1928, 185.2157], [100, 50], [100, 50], [-INF, -INF], [3e38, 3e38]], np.float32)
TRI_RIGHT = np.array([[90, 50], [100, 50], [110, 50], [90, 50], [90, 50], [90, 50], [90, 50], [0, 0], [NAN, NAN],
                      [-110, -50], [0, 0], [607.1928, 185.2157], [90, -INF], [90, 70], [INF, INF], [-3e38, 3e38]], np.float32)

# carried feature set of vo_detect_bucket (points + ages): where the reference would index its bucket vector out of range
# (undefined behaviour there) the feature is ignored -- vo_hip.h states it
BUCKET_POINTS = np.array([[NAN, 10], [10, NAN], [INF, 10], [-INF, 5], [1e30, 1], [-1e30, 1], [-5, -5], [-0.5, 3], [159.9, 95.9],
                          [160, 96], [200, 50], [50, 200], [3e9, 3e9], [80, 40], [81, 40], [1e6, 0], [0, 1e6], [1199999, 3]],
                         np.float32)
BUCKET_AGES = np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, -5, 2147483647, -2147483648, 1, 1, 1, 3, 3], np.int32)


def pnp_cases(orc, planted_problem, K):
    """name -> (xyz, uv, iterations): solvePnPRansac inputs (visualOdometry.cpp:176-178)"""
    cases = {}
    X, uv, r, t, _ = planted_problem(orc, 60, 0.2, 0.1, 4)
    a = X.copy()
    a[3] = NAN
    a[17, 2] = INF
    a[30] = [1e30, 0, 1]
    a[41, 2] = -5            # negative depth
    a[50] = 0                # the camera centre
    cases["xyz NaN / inf / 1e30 / negative depth"] = (a, uv, 60)
    b = uv.copy()
    b[5] = NAN
    b[9] = INF
    b[12] = -1e30
    cases["uv NaN / inf / huge"] = (X, b, 60)
    Xd, uvd = X.copy(), uv.copy()
    Xd[10:30], uvd[10:30] = Xd[10], uvd[10]
    cases["20 duplicates"] = (Xd, uvd, 60)
    s = np.linspace(0, 1, 40)[:, None]
    Xc = (np.array([[-5, -1, 8]]) + s * np.array([[10, 2, 30]])).astype(np.float32)
    uvc = orc.project_points(Xc, r, t, K).astype(np.float32)
    cases["collinear"] = (Xc, uvc, 40)
    Xp = X.copy()
    Xp[:, 2] = 20.0
    cases["coplanar"] = (Xp, orc.project_points(Xp, r, t, K).astype(np.float32), 40)
    for n in (4, 5, 6):
        Xs, uvs, _, _, _ = planted_problem(orc, n, 0, 0, 40 + n)
        cases["n=%d" % n] = (Xs, uvs, 40)
        X2, uv2 = Xs.copy(), uvs.copy()
        X2[1], uv2[1] = X2[0], uv2[0]
        cases["n=%d duplicate" % n] = (X2, uv2, 40)
        X3 = Xs.copy()
        X3[2] = NAN
        cases["n=%d NaN" % n] = (X3, uvs, 40)
        cases["n=%d collinear" % n] = (Xc[:n * 3:3], uvc[:n * 3:3], 40)
    cases["all the same point"] = (np.tile(X[:1], (20, 1)), np.tile(uv[:1], (20, 1)), 40)
    cases["all zero"] = (np.zeros((20, 3), np.float32), np.zeros((20, 2), np.float32), 40)
    return cases


def essential_cases(em_scene):
    """name -> (pts0, pts1): findEssentialMat + recoverPose inputs (visualOdometry.cpp:152-153)"""
    cases = {}
    p1, p2, R, t, F, PP = em_scene(3, 60, 0.2)
    a = p1.copy()
    a[4] = NAN
    a[7, 0] = INF
    a[9] = 1e30
    cases["NaN / inf / huge"] = (a, p2)
    a, b = p1.copy(), p2.copy()
    a[10:40], b[10:40] = a[10], b[10]
    cases["30 duplicates"] = (a, b)
    s = np.linspace(0, 1, 30)[:, None]
    a = (np.array([[100, 100]]) + s * np.array([[600, 150]])).astype(np.float32)
    cases["collinear"] = (a, a + np.float32([3, 1]))
    for n in (5, 6):
        q1, q2, *_ = em_scene(20 + n, n, 0.0)
        cases["n=%d" % n] = (q1, q2)
        c, d = q1.copy(), q2.copy()
        c[1], d[1] = c[0], d[0]
        cases["n=%d duplicate" % n] = (c, d)
        c = q1.copy()
        c[2] = NAN
        cases["n=%d NaN" % n] = (c, q2)
    cases["identical sets"] = (p1, p1.copy())
    cases["all the same point"] = (np.tile(p1[:1], (20, 1)), np.tile(p2[:1], (20, 1)))
    cases["zeros"] = (np.zeros((20, 2), np.float32), np.zeros((20, 2), np.float32))
    return cases, F, PP


def same(a, b, tol=0.0):
    """a == b within tol where both are finite; NaN compares as NaN, infinities by value"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.shape != b.shape or not np.array_equal(np.isnan(a), np.isnan(b)):
        return False
    m = ~np.isnan(a)
    inf = np.isinf(a[m]) | np.isinf(b[m])
    if not np.array_equal(a[m][inf], b[m][inf]):
        return False
    d = np.abs(a[m][~inf] - b[m][~inf])
    return d.size == 0 or d.max() <= tol


def pose_jacobian_singular_values(X, rvec, tvec, K):
    """singular values of the 2m x 6 Jacobian of the pinhole projection of X [m, 3] with respect to (rvec, tvec), by central
    differences in float64: how well the points determine the pose AT that pose (a flat valley of the reprojection error --
    sigma_max / sigma_min in the thousands -- is where a Levenberg-Marquardt run stops wherever its last step happened to end)"""
    X = np.asarray(X, np.float64)
    K = np.asarray(K, np.float64)

    def proj(p):
        th = np.linalg.norm(p[:3])
        k = p[:3] / th if th > 0 else np.zeros(3)
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        Y = X @ R.T + p[3:]
        return np.stack([K[0, 0] * Y[:, 0] / Y[:, 2] + K[0, 2], K[1, 1] * Y[:, 1] / Y[:, 2] + K[1, 2]], 1).reshape(-1)
    p = np.concatenate([np.asarray(rvec, np.float64), np.asarray(tvec, np.float64)])
    J = np.zeros((2 * len(X), 6))
    for k in range(6):
        d = np.zeros(6)
        d[k] = 1e-6
        J[:, k] = (proj(p + d) - proj(p - d)) / 2e-6
    return np.linalg.svd(J, compute_uv=False), J


def bucket_grid_ok(w, h, bucket_size, fpb):
    """the documented limits of the device bucketing (include/vo_hip.h, vo_detect_params)"""
    if bucket_size < 1 or fpb < 1 or fpb > 8:
        return False
    n = (h // bucket_size + 1) * (w // bucket_size + 1)
    return n <= 1024 or (n <= 4096 and n * fpb <= 8192)
