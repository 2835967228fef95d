"""Oracle pinning, part 2: SVD / triangulation / Rodrigues / projection / EPnP / solvePnPRansac
against numpy / scipy and planted ground truth."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

K_KITTI = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1]], np.float32)


@pytest.mark.parametrize("shape", [(3, 3), (4, 4), (6, 4), (6, 5), (6, 6), (12, 12)])
def test_svd_matches_numpy(orc, shape):
    rng = np.random.default_rng(shape[0] * 13 + shape[1])
    A = rng.normal(size=shape)
    w, u, vt = orc.svd(A)
    assert np.allclose(w, np.linalg.svd(A, compute_uv=False), rtol=1e-12, atol=1e-13)
    assert np.all(np.diff(w) <= 0)
    assert np.allclose((u * w) @ vt, A, atol=1e-12)
    assert np.allclose(u.T @ u, np.eye(shape[1]), atol=1e-12)
    assert np.allclose(vt @ vt.T, np.eye(shape[1]), atol=1e-12)


def test_svd_rank_deficient_null_space(orc):
    """EPnP's 12x12 M^T M for 5 points has rank <= 10: the trailing left vectors must span its null space"""
    rng = np.random.default_rng(3)
    M = rng.normal(size=(10, 12))
    B = M.T @ M
    w, u, vt = orc.svd(B)
    assert w[10] < 1e-12 * w[0] and w[11] < 1e-12 * w[0]
    assert np.abs(B @ u[:, 10:]).max() < 1e-10 * w[0]


def test_triangulate_matches_numpy_dlt(orc, kitti_world):
    P_l, P_r = kitti_world.proj_matrices()
    rng = np.random.default_rng(0)
    n = 300
    pl = rng.uniform([0, 0], [1241, 376], (n, 2)).astype(np.float32)
    pr = pl.copy()
    pr[:, 0] -= rng.uniform(4, 90, n).astype(np.float32)
    pr[:, 1] += rng.normal(0, 0.3, n).astype(np.float32)
    got = orc.triangulate(P_l, P_r, pl, pr)
    ref = np.zeros((n, 3))
    for i in range(n):
        A = np.stack([pl[i, 0] * P_l[2] - P_l[0], pl[i, 1] * P_l[2] - P_l[1],
                      pr[i, 0] * P_r[2] - P_r[0], pr[i, 1] * P_r[2] - P_r[1]]).astype(np.float64)
        X = np.linalg.svd(A)[2][3]
        ref[i] = X[:3] / X[3]
    assert np.allclose(got, ref, rtol=2e-5, atol=1e-5)
    # main.cpp:73-74 geometry: close to disparity depth Z = -bf / (xl - xr)
    z = 386.1448 / (pl[:, 0] - pr[:, 0])
    assert np.median(np.abs(got[:, 2] - z) / z) < 1e-5


def test_triangulate_project_roundtrip(orc, kitti_world):
    P_l, P_r = kitti_world.proj_matrices()
    rng = np.random.default_rng(1)
    X = rng.uniform([-10, -2, 4], [10, 2, 60], (200, 3))
    Xh = np.c_[X, np.ones(200)]
    pl = (P_l.astype(np.float64) @ Xh.T).T
    pr = (P_r.astype(np.float64) @ Xh.T).T
    pl, pr = pl[:, :2] / pl[:, 2:], pr[:, :2] / pr[:, 2:]
    got = orc.triangulate(P_l, P_r, pl.astype(np.float32), pr.astype(np.float32))
    # f32 pixel rounding (~3e-5 px) over disparities of 6..97 px
    assert np.max(np.abs(got - X) / np.abs(X).max(1, keepdims=True)) < 2e-4


def test_convert_from_homogeneous_zero_w(orc):
    import ctypes as C
    p = np.array([[2, 4, 6, 2], [1, 2, 3, 0]], np.float32)
    out = np.zeros((2, 3), np.float32)
    orc.lib().orc_convert_points_from_homogeneous(p.ctypes.data_as(C.c_void_p), 2, out.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out, [[1, 2, 3], [1, 2, 3]])  # w == 0 -> scale 1


def test_rodrigues_matches_scipy_and_roundtrip(orc):
    rng = np.random.default_rng(2)
    for _ in range(50):
        r = rng.normal(0, 0.8, 3)
        R = orc.rodrigues(r)
        assert np.allclose(R, Rotation.from_rotvec(r).as_matrix(), atol=1e-14)
        assert np.allclose(orc.rodrigues(R), r, atol=1e-12)
    assert np.array_equal(orc.rodrigues(np.zeros(3)), np.eye(3))
    r = np.array([np.pi, 0, 0]) * (1 - 1e-9)  # near-pi branch (s < 1e-5)
    assert np.allclose(np.abs(orc.rodrigues(orc.rodrigues(r))), np.abs(r), atol=1e-6)


def test_rodrigues_jacobian_finite_differences(orc):
    r = np.array([0.11, -0.23, 0.31])
    R, J = orc.rodrigues_jac(r)
    for i in range(3):
        d = np.zeros(3)
        d[i] = 1e-6
        num = (orc.rodrigues(r + d) - orc.rodrigues(r - d)).reshape(9) / 2e-6
        assert np.allclose(J[i], num, atol=1e-8)


def test_project_points_matches_numpy(orc):
    rng = np.random.default_rng(4)
    X = rng.uniform([-8, -2, 4], [8, 2, 40], (100, 3)).astype(np.float32)
    r, t = rng.normal(0, 0.05, 3), rng.normal(0, 0.5, 3)
    got = orc.project_points(X, r, t, K_KITTI)
    Xc = X.astype(np.float64) @ Rotation.from_rotvec(r).as_matrix().T + t
    uv = Xc[:, :2] / Xc[:, 2:] * [K_KITTI[0, 0], K_KITTI[1, 1]] + [K_KITTI[0, 2], K_KITTI[1, 2]]
    assert np.allclose(got, uv, rtol=0, atol=2e-4)


def test_epnp_exact_on_noise_free_points(orc):
    rng = np.random.default_rng(5)
    for n in (5, 6, 12):
        X = rng.uniform([-8, -2, 4], [8, 2, 40], (n, 3)).astype(np.float32)
        r, t = rng.normal(0, 0.05, 3), rng.normal(0, 0.5, 3)
        uv = orc.project_points(X, r, t, K_KITTI)
        R, tt = orc.epnp(X, uv, K_KITTI)
        assert np.allclose(orc.rodrigues(R), r, atol=2e-4) and np.allclose(tt, t, atol=5e-3), n


def test_cv_rng_subsets_are_valid_and_deterministic(orc):
    a = orc.ransac_subsets(1200, 500)
    b = orc.ransac_subsets(1200, 500)
    assert np.array_equal(a, b)
    assert a.min() >= 0 and a.max() < 1200
    assert all(len(set(row)) == 5 for row in a)
    # first draw of cv::RNG(0xffffffffffffffff): state' = 0xffffffff*4164903690 + 0xffffffff
    s = (0xFFFFFFFF * 4164903690 + 0xFFFFFFFF) & 0xFFFFFFFFFFFFFFFF
    assert a[0, 0] == (s & 0xFFFFFFFF) % 1200
    small = orc.ransac_subsets(6, 200)  # heavy duplicate rejection
    assert all(len(set(row)) == 5 for row in small)


def planted_problem(orc, n, outlier_frac, noise, seed):
    rng = np.random.default_rng(seed)
    X = rng.uniform([-10, -2, 4], [10, 2, 50], (n, 3)).astype(np.float32)
    r = np.array([0.002, -0.03, 0.001]) + rng.normal(0, 0.002, 3)
    t = np.array([0.02, -0.01, -0.9]) + rng.normal(0, 0.02, 3)
    uv = orc.project_points(X, r, t, K_KITTI).astype(np.float64)
    uv += rng.normal(0, noise, uv.shape)
    out = rng.random(n) < outlier_frac
    uv[out] += rng.uniform(-40, 40, (out.sum(), 2))
    return X, uv.astype(np.float32), r, t, out


def test_solve_pnp_ransac_noise_free(orc):
    X, uv, r, t, _ = planted_problem(orc, 400, 0.0, 0.0, 6)
    rc, rv, tv, inl, dbg = orc.solve_pnp_ransac(X, uv, K_KITTI)
    assert rc == 1 and len(inl) == 400
    assert np.allclose(rv, r, atol=1e-6) and np.allclose(tv, t, atol=1e-5)  # f32 pixel rounding only
    assert dbg[0] <= 5  # adaptive stop: all inliers -> log(1-p)/log(1-1) -> 0 iterations left


def test_solve_pnp_ransac_with_outliers(orc):
    X, uv, r, t, out = planted_problem(orc, 800, 0.35, 0.1, 7)
    rc, rv, tv, inl, dbg = orc.solve_pnp_ransac(X, uv, K_KITTI)
    assert rc == 1
    assert np.allclose(rv, r, atol=2e-4) and np.allclose(tv, t, atol=5e-3)
    assert not out[inl].any() or out[inl].mean() < 0.01
    assert 5 < dbg[0] <= 500 and 0 <= dbg[1] < dbg[0]


def test_solve_pnp_ransac_small_n(orc):
    X, uv, r, t, _ = planted_problem(orc, 5, 0.0, 0.0, 8)
    rc, rv, tv, inl, _ = orc.solve_pnp_ransac(X, uv, K_KITTI)
    assert rc == 1 and np.array_equal(inl, np.arange(5))
    assert np.allclose(rv, r, atol=1e-3) and np.allclose(tv, t, atol=2e-2)
    rc, *_ = orc.solve_pnp_ransac(X[:3], uv[:3], K_KITTI)
    assert rc < 0  # CV_Assert(npoints >= 4)


def test_solve_pnp_ransac_no_model(orc):
    rng = np.random.default_rng(9)
    X = rng.uniform([-10, -2, 4], [10, 2, 50], (60, 3)).astype(np.float32)
    uv = rng.uniform([0, 0], [1241, 376], (60, 2)).astype(np.float32)
    rc, rv, tv, inl, dbg = orc.solve_pnp_ransac(X, uv, K_KITTI)
    assert rc == 0 and len(inl) == 0 and dbg[0] == 500


def test_pnp_refinement_is_the_least_squares_minimum_scipy_finds(orc):
    """independent check of the final solvePnP(ITERATIVE) refinement (CvLevMarq restated in oracle/orc_pnp.c, the chain the
    device's select_refine_kernel mirrors): on the RANSAC inliers, scipy's own trust-region least-squares solver -- numpy
    projection, scipy rotation vectors, numerical Jacobian: no code shared with the oracle -- started from the oracle's
    pose must not move it, and started from the planted pose must arrive at it"""
    from scipy.optimize import least_squares
    fx, fy, cx, cy = K_KITTI[0, 0], K_KITTI[1, 1], K_KITTI[0, 2], K_KITTI[1, 2]
    # (moderate outlier shares only: the refinement starts from the LAST hypothesis RANSAC evaluated -- rvec / tvec are shared
    # buffers in solvePnPRansac -- and with half of the points wrong that start can be too far off for 20 iterations)
    for n, frac, noise, seed in ((600, 0.3, 0.15, 21), (300, 0.0, 0.3, 22), (1500, 0.2, 0.1, 23), (60, 0.1, 0.05, 24)):
        X, uv, r, t, _ = planted_problem(orc, n, frac, noise, seed)
        rc, rv, tv, inl, _ = orc.solve_pnp_ransac(X, uv, K_KITTI)
        assert rc == 1 and len(inl) >= 0.4 * n
        Xi, ui = X[inl].astype(np.float64), uv[inl].astype(np.float64)

        def residual(p):
            Xc = Xi @ Rotation.from_rotvec(p[:3]).as_matrix().T + p[3:]
            return np.concatenate([Xc[:, 0] / Xc[:, 2] * fx + cx - ui[:, 0], Xc[:, 1] / Xc[:, 2] * fy + cy - ui[:, 1]])
        mine = np.concatenate([rv, tv])
        for start in (mine, np.concatenate([r, t])):
            sol = least_squares(residual, start, method="trf", xtol=1e-15, ftol=1e-15, gtol=1e-15, x_scale=[1e-3] * 3 + [1e-2] * 3)
            # CvLevMarq stops at 20 iterations or a relative step of FLT_EPSILON: agreement far below the 1e-6 rad / 1e-5 m
            # the GPU-vs-oracle pose tolerance works with
            assert np.allclose(sol.x[:3], rv, rtol=0, atol=2e-7) and np.allclose(sol.x[3:], tv, rtol=0, atol=5e-6), (n, sol.x - mine)
            assert np.sum(sol.fun ** 2) <= np.sum(residual(mine) ** 2) * (1 + 1e-9)
