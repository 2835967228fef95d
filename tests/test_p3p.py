"""solvePnPRansac with EXACTLY four correspondences (reference src/visualOdometry.cpp:176: OpenCV then switches to its
P3P kernel and, model_points being npoints, returns solvePnP(SOLVEPNP_P3P) directly) -- the checker's restatement
(oracle/orc_p3p.c) pinned by what does not depend on it: numpy's polynomial roots, planted poses, the geometric
constraints every candidate must satisfy; and the device-side code (csrc/vo_p3p.h compiled for the host) against it."""
import ctypes as C

import numpy as np
import pytest

K4 = np.array([718.856, 718.856, 607.1928, 185.2157])
KM = np.array([[K4[0], 0, K4[2]], [0, K4[1], K4[3]], [0, 0, 1]], np.float32)


def rot(rv):
    th = np.linalg.norm(rv)
    if th < 1e-12:
        return np.eye(3)
    k = rv / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def planted(rng, noise=0.0):
    """4 scene points in front of a KITTI-like camera, a small frame-to-frame motion, their projections (f64)"""
    X = np.stack([rng.uniform(-8, 8, 4), rng.uniform(-2, 2, 4), rng.uniform(5, 40, 4)], 1)
    rv = rng.normal(0, 0.02, 3)
    t = np.array([rng.normal(0, 0.05), rng.normal(0, 0.03), -rng.uniform(0.5, 1.2)])
    Xc = X @ rot(rv).T + t
    uv = np.stack([K4[0] * Xc[:, 0] / Xc[:, 2] + K4[2], K4[1] * Xc[:, 1] / Xc[:, 2] + K4[3]], 1)
    return X, uv + rng.normal(0, noise, uv.shape) if noise else uv, rv, t


def test_quartic_roots_equal_numpy(orc):
    rng = np.random.default_rng(1)
    n_real = 0
    for _ in range(400):
        r = rng.uniform(-3, 3, 4)
        if rng.random() < 0.5:  # two complex roots
            p = np.poly1d([1, -2 * r[0], r[0] ** 2 + abs(r[1]) + 0.1]) * np.poly1d(np.poly(r[2:]))
            want = np.sort(r[2:])
        else:
            p = np.poly1d(np.poly(r))
            want = np.sort(r)
        c = p.coeffs * rng.uniform(0.5, 2.0)
        got = np.sort(orc.solve_deg4(*c))
        assert len(got) == len(want)
        assert np.allclose(got, want, atol=2e-6), (got, want)
        n_real += len(got)
    assert n_real > 1000
    # degree drops: a = 0 -> cubic, a = b = 0 -> quadratic
    assert np.allclose(np.sort(orc.solve_deg4(0, 1, -6, 11, -6)), [1, 2, 3], atol=1e-9)
    assert np.allclose(np.sort(orc.solve_deg4(0, 0, 1, -3, 2)), [1, 2], atol=1e-12)


def test_p3p_recovers_planted_poses_and_every_candidate_fits_its_three_points(orc):
    rng = np.random.default_rng(7)
    best_err, fit = [], []
    for _ in range(300):
        X, uv, rv, t = planted(rng)
        R, ts = orc.p3p_solve(K4, uv, X)
        assert 1 <= len(R) <= 4
        for Rk, tk in zip(R, ts):
            assert abs(np.linalg.det(Rk) - 1) < 1e-9 and np.abs(Rk @ Rk.T - np.eye(3)).max() < 1e-9
            Xc = X[:3] @ Rk.T + tk
            assert (Xc[:, 2] > 0).all()
            p = np.stack([K4[0] * Xc[:, 0] / Xc[:, 2] + K4[2], K4[1] * Xc[:, 1] / Xc[:, 2] + K4[3]], 1)
            fit.append(np.abs(p - uv[:3]).max())             # each candidate reprojects ITS three points
        # sorted by the fourth point's error: the first is the planted pose
        best_err.append(max(np.abs(R[0] - rot(rv)).max(), np.abs(ts[0] - t).max()))
        e4 = []
        for Rk, tk in zip(R, ts):
            q = X[3] @ Rk.T + tk
            e4.append(((q[0] / q[2] - (uv[3, 0] - K4[2]) / K4[0]) ** 2 + (q[1] / q[2] - (uv[3, 1] - K4[3]) / K4[1]) ** 2))
        assert all(e4[i] <= e4[i + 1] for i in range(len(e4) - 1))
    # Gao's method through a closed-form quartic is exact in the bulk and loses digits near double roots / small b0 (a
    # known property of this solver, which OpenCV shares: the same formulas with numpy.roots instead of the Ferrari
    # solver satisfy the distance constraints to 1e-12 in the median and still show a tail) -- so the bars are quantiles
    assert np.median(fit) < 1e-8 and np.percentile(fit, 90) < 1e-4, (np.median(fit), np.percentile(fit, 90))
    assert np.median(best_err) < 1e-9 and np.percentile(best_err, 90) < 1e-5, (np.median(best_err), np.max(best_err))


def test_solve_pnp_ransac_with_four_points_is_p3p(orc):
    rng = np.random.default_rng(11)
    acc = []
    for _ in range(100):
        X, uv, rv, t = planted(rng)
        n, rvs, tvs = orc.solve_p3p(X, uv, KM)
        assert n >= 1
        rc, r_out, t_out, inl, dbg = orc.solve_pnp_ransac(X.astype(np.float32), uv.astype(np.float32), KM,
                                                         tvec=[9, 9, 9])
        assert rc == 1 and list(inl) == [0, 1, 2, 3]
        assert np.array_equal(r_out, rvs[0]) and np.array_equal(t_out, tvs[0])   # first of the sorted solutions, no refinement
        # f32 storage of the points and the f32 normalised coordinates of undistortPoints bound the accuracy
        acc.append(max(np.abs(r_out - rv).max(), np.abs(t_out - t).max()))
        # sorted by the summed squared pixel error over all four points
        errs = []
        for r_k, t_k in zip(rvs, tvs):
            Xc = X.astype(np.float32).astype(np.float64) @ rot(r_k).T + t_k
            p = np.stack([K4[0] * Xc[:, 0] / Xc[:, 2] + K4[2], K4[1] * Xc[:, 1] / Xc[:, 2] + K4[3]], 1)
            errs.append(((p - uv.astype(np.float32).astype(np.float64)) ** 2).sum())
        assert all(errs[i] <= errs[i + 1] * (1 + 1e-9) for i in range(len(errs) - 1))
    assert np.median(acc) < 1e-4, np.median(acc)
    # no solution (the three rays cannot span the triangle): solvePnP returns false, rvec / tvec untouched, no inliers
    X = np.array([[0, 0, 10], [1, 0, 10], [0, 1, 10], [1, 1, 10]], np.float32)
    uv = np.array([[600, 180], [600.001, 180], [600, 180.001], [600.001, 180.001]], np.float32)
    rc, r_out, t_out, inl, dbg = orc.solve_pnp_ransac(X, uv, KM, rvec=[0.1, 0.2, 0.3], tvec=[1, 2, 3])
    if rc == 0:
        assert np.array_equal(r_out, [0.1, 0.2, 0.3]) and np.array_equal(t_out, [1, 2, 3]) and len(inl) == 0
    # fewer than four points: CV_Assert(npoints >= 4)
    assert orc.solve_pnp_ransac(X[:3], uv[:3], KM)[0] == -1


def test_device_p3p_source_on_the_host_equals_the_checker(orc, host_check):
    """csrc/vo_p3p.h (what p3p_frame runs on the device) compiled by g++ against oracle/orc_p3p.c on planted quadruples: same number of
    solutions, same first solution.  The header's cubic uses vo_math.h's cbrt / acos / cos (the same bits on the device), the
    oracle glibc's pow / acos / cos like OpenCV: both within one ulp, so ~95 % of the quadruples agree to the bit and the
    rest to what the quartic's closed form makes of one ulp -- WORST case <= 1e-6 like every other pose path (observed
    5e-9 over 3000 quadruples)"""
    rng = np.random.default_rng(23)
    host_check.hc_p3p4.restype = C.c_int
    exact = total = 0
    worst = 0.0
    for _ in range(1500):
        X, uv, rv, t = planted(rng, noise=0.3 if rng.random() < 0.5 else 0.0)
        Xf, uvf = np.ascontiguousarray(X, np.float32), np.ascontiguousarray(uv, np.float32)
        n, rvs, tvs = orc.solve_p3p(Xf, uvf, KM)
        r_d, t_d = np.full(3, 7.0), np.full(3, 7.0)
        nd = host_check.hc_p3p4(Xf.ctypes.data_as(C.c_void_p), uvf.ctypes.data_as(C.c_void_p),
                                KM.ctypes.data_as(C.c_void_p), r_d.ctypes.data_as(C.c_void_p), t_d.ctypes.data_as(C.c_void_p))
        assert nd == n
        if n == 0:
            assert np.array_equal(r_d, np.full(3, 7.0)) and np.array_equal(t_d, np.full(3, 7.0))  # untouched
            continue
        total += 1
        exact += np.array_equal(r_d, rvs[0]) and np.array_equal(t_d, tvs[0])
        worst = max(worst, np.abs(r_d - rvs[0]).max(), np.abs(t_d - tvs[0]).max())
    print("vo_p3p.h on the host vs the checker: %d of %d bit-identical, worst %.3g" % (exact, total, worst))
    assert total > 1400 and exact >= 0.9 * total and worst <= 1e-6
    x = np.zeros(4)
    host_check.hc_p3p_deg4.argtypes = [C.c_double] * 5 + [C.c_void_p]
    assert host_check.hc_p3p_deg4(1.0, -10.0, 35.0, -50.0, 24.0, x.ctypes.data_as(C.c_void_p)) == 4
    assert np.allclose(np.sort(x), [1, 2, 3, 4], atol=1e-9)
