"""The VO_HD headers the kernels are built from (vo_linalg.h / vo_epnp.h / vo_tri.h), compiled by
g++ (tests/host_check) and compared with the oracle: same operation order + no FMA contraction =>
bit-identical on the CPU.  This is a unit test of device code, not a product path."""
import ctypes as C

import numpy as np

from conftest import vp

K = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1]], np.float32)


def test_epnp5_bit_identical(orc, host_check):
    rng = np.random.default_rng(0)
    for _ in range(300):
        xyz = rng.uniform([-8, -2, 4], [8, 2, 40], (5, 3)).astype(np.float32)
        rv, tv = rng.normal(0, 0.02, 3), rng.normal(0, 0.5, 3)
        uv = (orc.project_points(xyz, rv, tv, K) + rng.normal(0, 0.4, (5, 2))).astype(np.float32)
        R, t = orc.epnp(xyz, uv, K)
        r0 = orc.rodrigues(R)
        r1, t1 = np.zeros(3), np.zeros(3)
        host_check.hc_epnp5(vp(xyz), vp(uv), vp(K), vp(r1), vp(t1))
        assert np.array_equal(r0, r1) and np.array_equal(t, t1)


def test_triangulate_bit_identical(orc, host_check, kitti_world):
    P_l, P_r = kitti_world.proj_matrices()
    rng = np.random.default_rng(1)
    n = 2000
    pl = rng.uniform([0, 0], [1241, 376], (n, 2)).astype(np.float32)
    pr = pl.copy()
    pr[:, 0] -= rng.uniform(0.5, 120, n).astype(np.float32)
    pr[:, 1] += rng.normal(0, 0.3, n).astype(np.float32)
    a = orc.triangulate(P_l, P_r, pl, pr)
    b = np.zeros((n, 3), np.float32)
    host_check.hc_triangulate(vp(P_l), vp(P_r), vp(pl), vp(pr), n, vp(b))
    assert np.array_equal(a, b)


def test_rodrigues_bit_identical(orc, host_check):
    rng = np.random.default_rng(2)
    for _ in range(100):
        r = rng.normal(0, 0.5, 3)
        R0, J0 = orc.rodrigues_jac(r)
        R1, J1 = np.zeros((3, 3)), np.zeros((3, 9))
        host_check.hc_rodrigues_v2m(vp(r), vp(R1), vp(J1))
        assert np.array_equal(R0, R1) and np.array_equal(J0, J1)
        back = np.zeros(3)
        host_check.hc_rodrigues_m2v(vp(R1), vp(back))
        assert np.array_equal(back, orc.rodrigues(R0))
