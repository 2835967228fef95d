"""f4 (`mono_rotation`, reference visualOdometry.cpp:146-157): the oracle's findEssentialMat / recoverPose
restatement pinned by analytic ground truth, and the device-side header (vo_fivept.h, compiled for the host)
compared with it bit for bit."""
import ctypes as C

import numpy as np
import pytest


def _rot(rv):
    rv = np.asarray(rv, np.float64)
    th = np.linalg.norm(rv)
    k = rv / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def _scene(seed, n=300, rv=(0.01, -0.03, 0.005), t=(0.05, -0.02, 0.9)):
    rng = np.random.default_rng(seed)
    R = _rot(rv)
    t = np.asarray(t, np.float64)
    t = t / np.linalg.norm(t)
    X = np.c_[rng.uniform(-8, 8, n), rng.uniform(-2, 2, n), rng.uniform(5, 40, n)]
    x1 = X[:, :2] / X[:, 2:]
    X2 = X @ R.T + t
    x2 = X2[:, :2] / X2[:, 2:]
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    E = tx @ R
    return R, t, x1, x2, E / np.linalg.norm(E), rng


F, PP = 718.856, (607.1928, 185.2157)


def vp(a):
    return a.ctypes.data_as(C.c_void_p)


def test_five_point_solutions_satisfy_the_constraints_and_contain_the_truth(orc):
    for seed in range(8):
        R, t, x1, x2, Et, rng = _scene(seed, rv=rng_rv(seed), t=rng_t(seed))
        Es = orc.five_point(x1[:5], x2[:5])
        assert 1 <= len(Es) <= 10
        best = 1e9
        for E in Es:
            assert abs(np.linalg.norm(E) - 1) < 1e-12
            for i in range(5):
                assert abs(np.r_[x2[i], 1] @ E @ np.r_[x1[i], 1]) < 1e-12
            assert abs(np.linalg.det(E)) < 1e-9
            assert np.abs(2 * E @ E.T @ E - np.trace(E @ E.T) * E).max() < 1e-8
            best = min(best, np.abs(E - Et).max(), np.abs(E + Et).max())
        assert best < 1e-8, (seed, best)


def rng_rv(seed):
    return np.random.default_rng(100 + seed).normal(0, 0.03, 3)


def rng_t(seed):
    return np.r_[np.random.default_rng(200 + seed).normal(0, 0.2, 2), 1.0]


def test_find_essential_mat_and_recover_pose_recover_the_planted_motion(orc):
    for seed in range(4):
        R, t, x1, x2, Et, rng = _scene(seed, n=400, rv=rng_rv(seed), t=rng_t(seed))
        p1 = (x1 * F + PP).astype(np.float32)
        p2 = (x2 * F + PP).astype(np.float32)
        p2[:60] += rng.uniform(-40, 40, (60, 2)).astype(np.float32)  # gross outliers
        ok, E, mask, dbg = orc.find_essential_mat(p1, p2, F, PP)
        assert ok == 1
        assert mask[60:].sum() >= 330 and mask[:60].sum() <= 6
        assert min(np.abs(E - Et).max(), np.abs(E + Et).max()) < 1e-3  # minimal (5-point) model of f32-quantised pixels
        assert dbg[0] < 1000  # adaptive stop
        good, Rr, tr, m2 = orc.recover_pose(E, p1, p2, F, PP, mask)
        assert np.abs(Rr - R).max() < 1e-3
        assert np.abs(tr - t).max() < 5e-2
        assert good == int((m2 != 0).sum()) and good >= 0.95 * mask.sum()
        assert set(np.unique(m2)) <= {0, 1}  # bitwise_and of the 0/1 RANSAC mask with 0/255 compare results


def test_decompose_gives_two_rotations_and_unit_translation(orc):
    R, t, x1, x2, Et, _ = _scene(3)
    R1, R2, tt = orc.decompose_essential_mat(Et)
    for Q in (R1, R2):
        assert np.abs(Q @ Q.T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(Q) - 1) < 1e-12
    assert min(np.abs(R1 - R).max(), np.abs(R2 - R).max()) < 1e-9
    assert min(np.abs(tt - t).max(), np.abs(tt + t).max()) < 1e-9


def test_device_five_point_is_bit_identical_to_the_oracle(orc, host_check):
    n_checked = 0
    for seed in range(40):
        R, t, x1, x2, Et, rng = _scene(seed, n=12, rv=rng_rv(seed), t=rng_t(seed))
        idx = rng.permutation(12)[:5]
        q1 = np.ascontiguousarray(x1[idx] + rng.normal(0, 1e-3, (5, 2)))
        q2 = np.ascontiguousarray(x2[idx] + rng.normal(0, 1e-3, (5, 2)))
        ref = orc.five_point(q1, q2)
        Es = np.zeros((10, 9))
        k = host_check.hc_five_point(vp(q1), vp(q2), vp(Es))
        assert k == len(ref)
        assert np.array_equal(Es[:k].reshape(k, 3, 3), ref)
        n_checked += k
    assert n_checked > 80


def test_device_sampson_decompose_cheirality_match_the_oracle(orc, host_check):
    from oracle import oracle as O
    host_check.hc_sampson.restype = C.c_float
    O.lib().orc_sampson_error.restype = C.c_float
    R, t, x1, x2, Et, rng = _scene(5, n=200)
    for i in range(200):
        x4 = np.r_[x1[i], x2[i] + rng.normal(0, 1e-3, 2)]
        a = host_check.hc_sampson(vp(Et), vp(x4))
        b = O.lib().orc_sampson_error(vp(Et), C.c_double(x4[0]), C.c_double(x4[1]), C.c_double(x4[2]),
                                      C.c_double(x4[3]))
        assert a == b
    R1, R2, tt = orc.decompose_essential_mat(Et)
    r1, r2, t2 = np.zeros((3, 3)), np.zeros((3, 3)), np.zeros(3)
    host_check.hc_decompose(vp(Et), vp(r1), vp(r2), vp(t2))
    assert np.array_equal(r1, R1) and np.array_equal(r2, R2) and np.array_equal(t2, tt)
    # cheirality: the true configuration passes, the mirrored translation fails
    P_true = np.ascontiguousarray(np.c_[R, t])
    P_flip = np.ascontiguousarray(np.c_[R, -t])
    ok_true = sum(host_check.hc_cheirality(vp(P_true), vp(np.r_[x1[i], x2[i]]), C.c_double(50.0)) for i in range(200))
    ok_flip = sum(host_check.hc_cheirality(vp(P_flip), vp(np.r_[x1[i], x2[i]]), C.c_double(50.0)) for i in range(200))
    assert ok_true == 200 and ok_flip == 0
