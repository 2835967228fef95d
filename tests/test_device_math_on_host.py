"""The VO_HD headers the kernels are built from (vo_linalg.h / vo_epnp.h / vo_tri.h), compiled by
g++ (tests/host_check) and compared with the oracle: same operation order + no FMA contraction =>
bit-identical on the CPU -- except where the CPU path calls libm: since round 4 the headers take sin / cos / acos (Rodrigues)
from csrc/vo_math.h (IEEE operations only, so that gfx950 and this host build compute the SAME bits; the GPU suite holds the
device to this build bit for bit), while the oracle calls glibc's like OpenCV does: both are within one ulp of the exact
value, so those outputs agree to a few ulp instead of to the bit.  This is a unit test of device code, not a product path."""
import ctypes as C

import numpy as np

from conftest import vp

K = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1]], np.float32)


def test_epnp5_matches_oracle(orc, host_check):
    rng = np.random.default_rng(0)
    for _ in range(300):
        xyz = rng.uniform([-8, -2, 4], [8, 2, 40], (5, 3)).astype(np.float32)
        rv, tv = rng.normal(0, 0.02, 3), rng.normal(0, 0.5, 3)
        uv = (orc.project_points(xyz, rv, tv, K) + rng.normal(0, 0.4, (5, 2))).astype(np.float32)
        R, t = orc.epnp(xyz, uv, K)
        r0 = orc.rodrigues(R)
        r1, t1 = np.zeros(3), np.zeros(3)
        host_check.hc_epnp5(vp(xyz), vp(uv), vp(K), vp(r1), vp(t1))
        assert np.array_equal(t, t1)                   # no libm on the way to t
        assert np.abs(r0 - r1).max() <= 4e-16          # rvec = Rodrigues(R): one acos (vo_math.h vs glibc, a few ulp of <= pi)


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


def test_rodrigues_matches_oracle_to_a_few_ulp(orc, host_check):
    rng = np.random.default_rng(2)
    for _ in range(100):
        r = rng.normal(0, 0.5, 3)
        R0, J0 = orc.rodrigues_jac(r)
        R1, J1 = np.zeros((3, 3)), np.zeros((3, 9))
        host_check.hc_rodrigues_v2m(vp(r), vp(R1), vp(J1))
        # sin / cos / acos: vo_math.h here, glibc in the oracle -- a few ulp of values <= 1 (R), <= ~2 (J), <= pi (rvec)
        assert np.abs(R0 - R1).max() <= 4e-16 and np.abs(J0 - J1).max() <= 2e-15
        back = np.zeros(3)
        host_check.hc_rodrigues_m2v(vp(R1), vp(back))
        assert np.abs(back - orc.rodrigues(R0)).max() <= 2e-15
        assert np.abs(back - r).max() <= 1e-14         # and the round trip returns the vector


# ---------------------------------------------------------------------------------------------
# vo_lkmath.h: the packed v_perm / v_dot2 pixel arithmetic of the LK kernel against the plain
# DESCALE formulas of OpenCV's LKTrackerInvoker (lkpyramid.cpp), incl. the extreme operands
def _weights(rng, n):
    a, b = rng.random(n, dtype=np.float32), rng.random(n, dtype=np.float32)
    a[:4], b[:4] = [0, 0, 1 - 2**-20, 0.5], [0, 1 - 2**-20, 0, 0.5]
    one = np.float32(1)
    s = np.float32(1 << 14)
    w00 = np.rint((one - a) * (one - b) * s).astype(np.int32)
    w01 = np.rint(a * (one - b) * s).astype(np.int32)
    w10 = np.rint((one - a) * b * s).astype(np.int32)
    w11 = (1 << 14) - w00 - w01 - w10
    return np.ascontiguousarray(np.stack([w00, w01, w10, w11], 1).astype(np.int32))


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def test_lk_bilinear_u8_exact(host_check):
    rng = np.random.default_rng(5)
    n = 20000
    top = rng.integers(0, 256, (n, 8), dtype=np.uint8)
    bot = rng.integers(0, 256, (n, 8), dtype=np.uint8)
    top[:8], bot[:8] = 255, 255
    top[8:16], bot[8:16] = 0, 255
    w = _weights(rng, n)
    w[4:8] = [[16384, 0, 0, 0], [0, 16384, 0, 0], [0, 0, 16384, 0], [0, 0, 0, 16384]]
    # three roundings can add up to 2^14 + 1, leaving iw11 = -1 (seen on the MI355X at a = b ~ 0.006)
    w[8:12] = [[16385, 0, 0, -1], [16189, 98, 98, -1], [1, 16383, 1, -1], [8192, 8192, 1, -1]]
    assert (w.sum(1) == 16384).all()
    got = np.zeros((n, 7), np.int16)
    host_check.hc_bilinear7_u8(vp(top), vp(bot), vp(w), n, vp(got))
    t, b = top.astype(np.int64), bot.astype(np.int64)
    ref = _descale(t[:, :7] * w[:, [0]] + t[:, 1:] * w[:, [1]] + b[:, :7] * w[:, [2]] + b[:, 1:] * w[:, [3]], 9)
    assert np.array_equal(got, ref)
    assert ref.max() == 8160 and ref[8:12].min() >= 0


def test_lk_bilinear_deriv_exact(host_check):
    rng = np.random.default_rng(6)
    n = 20000
    # true Scharr samples are in [-4080, 4080]; stored pre-multiplied by 4
    dx = rng.integers(-4080, 4081, (2, n, 8)).astype(np.int64)
    dy = rng.integers(-4080, 4081, (2, n, 8)).astype(np.int64)
    dx[:, :4], dy[:, :4] = 4080, -4080
    dx[:, 4:8], dy[:, 4:8] = -4080, 4080
    packed = (((dx * 4) & 0xffff) | (((dy * 4) & 0xffff) << 16)).astype(np.uint32)
    w = _weights(rng, n)
    ix, iy = np.zeros((n, 7), np.int16), np.zeros((n, 7), np.int16)
    host_check.hc_bilinear7_deriv(vp(np.ascontiguousarray(packed[0])), vp(np.ascontiguousarray(packed[1])), vp(w), n,
                                  vp(ix), vp(iy))
    for got, d in ((ix, dx), (iy, dy)):
        ref = _descale(d[0][:, :7] * w[:, [0]] + d[0][:, 1:] * w[:, [1]] + d[1][:, :7] * w[:, [2]] +
                       d[1][:, 1:] * w[:, [3]], 14)
        assert np.array_equal(got, ref)


def test_lk_diff_dot_exact(host_check):
    rng = np.random.default_rng(7)
    n = 5000
    val = rng.integers(0, 8161, (n, 7)).astype(np.int16)
    I = rng.integers(0, 8161, (n, 7)).astype(np.int16)
    ix = rng.integers(-4080, 4081, (n, 7)).astype(np.int16)
    val[0], I[0], ix[0] = 8160, 0, 4080       # largest per-lane partial: 7 * 8160 * 4080 < 2^28
    val[1], I[1], ix[1] = 0, 8160, 4080
    b1 = np.zeros(n, np.int32)
    host_check.hc_diff_dot(vp(val), vp(I), vp(ix), n, vp(b1))
    ref = ((val.astype(np.int64) - I) * ix).sum(1)
    assert np.array_equal(b1, ref) and abs(ref).max() < 2**28


def test_scharr_packed_matches_oracle(orc, host_check):
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, (9, 11), dtype=np.uint8)
    img[4, 5], img[3:6, 4], img[3:6, 6] = 0, 255, 0
    d = orc.scharr(img).astype(np.int64)
    host_check.hc_scharr4.restype = C.c_uint32
    for y in range(1, 8):
        for x in range(1, 10):
            p = img[y - 1:y + 2, x - 1:x + 2].astype(np.int32)
            p8 = np.ascontiguousarray([p[0, 0], p[0, 1], p[0, 2], p[1, 0], p[1, 2], p[2, 0], p[2, 1], p[2, 2]], np.int32)
            v = host_check.hc_scharr4(vp(p8))
            gx, gy = np.array([v & 0xffff, v >> 16], np.uint16).view(np.int16)
            assert gx == 4 * d[y, x, 0] and gy == 4 * d[y, x, 1]


def test_min_eig_pretest_never_contradicts_the_exact_expression():
    """lk.hip skips the correctly rounded sqrt + divide of OpenCV's minEig when a cheap bound shows the feature
    cannot be rejected; the bound (same f32 operations as the kernel) must imply the exact test's outcome,
    in particular for structure tensors right at the threshold"""
    rng = np.random.default_rng(3)
    f = np.float32
    thr = f(1e-3)
    n = 4_000_000
    # eigenvalues: lam_min around 882 * thr (log-uniform over 4 decades), lam_max above it, random orientation
    lam_min = (f(882.0) * thr * f(10.0) ** rng.uniform(-2, 2, n)).astype(f)
    lam_max = (lam_min * f(10.0) ** rng.uniform(0, 6, n)).astype(f)
    th = rng.uniform(0, np.pi, n)
    c, s = np.cos(th), np.sin(th)
    A11 = (lam_max * c * c + lam_min * s * s).astype(f) * f(0.5)   # (A11 + A22 - sqrt(..)) = 2 * lam_min * 0.5
    A22 = (lam_max * s * s + lam_min * c * c).astype(f) * f(0.5)
    A12 = ((lam_max - lam_min) * c * s).astype(f) * f(0.5)
    t = A22 + A11
    d = A11 - A22
    s2 = d * d + f(4.0) * A12 * A12
    min_eig_n = thr * (f(1.001) * f(882.0))
    u = t - (min_eig_n + t * f(3.814697265625e-6))
    fast_ok = (u > 0) & (s2 < u * u * f(0.9999))
    exact = (t - np.sqrt(s2)) / f(882.0)          # numpy f32 sqrt / divide are correctly rounded
    rejected = exact < thr
    assert not np.any(fast_ok & rejected)
    assert fast_ok.mean() > 0.3 and rejected.mean() > 0.2   # both sides of the threshold are exercised
    near = np.abs(exact / thr - 1) < 5e-4
    assert near.sum() > 200 and not np.any(fast_ok & near)  # the band around the threshold always takes the exact path
