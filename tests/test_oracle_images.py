"""Oracle pinning, part 1: pyramid / Scharr / LK against independent numpy re-derivations and
analytic ground truth (the reference ships no golden vectors -- PARITY UNPINNED vs real OpenCV)."""
import numpy as np
import pytest


def np_pyr_down(img):
    """independent restatement: REFLECT_101 pad, separable [1 4 6 4 1], (v + 128) >> 8, even taps"""
    a = np.pad(img.astype(np.int64), 2, mode="reflect")
    k = np.array([1, 4, 6, 4, 1], np.int64)
    h, w = img.shape
    hor = sum(k[i] * a[:, i:i + w] for i in range(5))          # (h+4, w)
    ver = sum(k[i] * hor[i:i + h, :] for i in range(5))         # (h, w)
    return ((ver[::2, ::2] + 128) >> 8).astype(np.uint8)


def np_scharr(img):
    a = np.pad(img.astype(np.int64), 1, mode="reflect")
    h, w = img.shape
    t0 = (a[0:h, :] + a[2:h + 2, :]) * 3 + a[1:h + 1, :] * 10
    t1 = a[2:h + 2, :] - a[0:h, :]
    ix = t0[:, 2:] - t0[:, :-2]
    iy = (t1[:, 2:] + t1[:, :-2]) * 3 + t1[:, 1:-1] * 10
    return np.stack([ix, iy], -1).astype(np.int16)


@pytest.mark.parametrize("shape", [(376, 1241), (47, 156), (33, 32), (5, 7), (2, 2), (188, 621)])
def test_pyr_down_matches_numpy(orc, shape):
    rng = np.random.default_rng(sum(shape))
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    got = orc.pyr_down(img)
    assert got.shape == ((shape[0] + 1) // 2, (shape[1] + 1) // 2)
    assert np.array_equal(got, np_pyr_down(img))


def test_pyr_down_constant_and_levels(orc):
    img = np.full((376, 1241), 173, np.uint8)
    pyr = orc.build_pyramid(img, 3)
    assert [p.shape for p in pyr] == [(376, 1241), (188, 621), (94, 311), (47, 156)]
    assert all((p == 173).all() for p in pyr)


@pytest.mark.parametrize("shape", [(94, 311), (21, 40), (3, 3), (2, 5)])
def test_scharr_matches_numpy(orc, shape):
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    assert np.array_equal(orc.scharr(img), np_scharr(img))


def smooth_image(w, h, dx=0.0, dy=0.0, seed=0):
    """analytic band-limited pattern sampled at (x - dx, y - dy): exact sub-pixel translation"""
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    xs = xs - dx
    ys = ys - dy
    img = np.zeros((h, w))
    for _ in range(24):
        fx, fy = rng.uniform(-0.35, 0.35, 2)
        img += rng.uniform(0.3, 1.0) * np.cos(fx * xs + fy * ys + rng.uniform(0, 6.28))
    img = 128 + 100 * img / np.abs(img).max()
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("flow", [(0.0, 0.0), (3.25, -1.5), (-7.4, 5.1), (12.3, 0.7)])
def test_lk_recovers_known_translation(orc, flow):
    w, h = 320, 240
    I = smooth_image(w, h)
    J = smooth_image(w, h, flow[0], flow[1])
    rng = np.random.default_rng(5)
    pts = np.stack([rng.uniform(50, w - 50, 60), rng.uniform(50, h - 50, 60)], 1).astype(np.float32)
    out, st, err = orc.calc_optical_flow_pyr_lk(I, J, pts)
    assert st.all()
    d = out - pts
    # LK on 8-bit quantised data with the 0.01 px stop criterion: sub-0.25 px worst case
    e = np.abs(d - np.array(flow, np.float32))
    assert e.max() < 0.25 and np.median(e) < 0.05


def test_lk_status_rules(orc):
    w, h = 320, 240
    I = smooth_image(w, h)
    J = smooth_image(w, h, 1.0, 0.5)
    flat = np.full((h, w), 90, np.uint8)
    pts = np.array([[160, 120], [-40, 50], [500, 100], [100, -35], [100, 400], [5.5, 4.5], [318, 238]], np.float32)
    out, st, _ = orc.calc_optical_flow_pyr_lk(I, J, pts)
    assert st[0] == 1
    assert not st[1:5].any()          # prev point outside the +-winSize admissibility window at level 0
    assert st[5] == 1 and st[6] == 1  # border points are tracked through the REFLECT_101 border
    out2, st2, _ = orc.calc_optical_flow_pyr_lk(flat, flat, pts[:1])
    assert st2[0] == 0                # min-eigenvalue test (textureless window)
    assert np.array_equal(out2[0], pts[0])  # the propagated guess is returned unchanged


def test_lk_float_accumulators_stay_close(orc):
    """x86 OpenCV accumulates in f32; the determinism recipe uses exact integers"""
    w, h = 320, 240
    I = smooth_image(w, h, seed=3)
    J = smooth_image(w, h, 2.6, -3.3, seed=3)
    rng = np.random.default_rng(1)
    pts = np.stack([rng.uniform(40, w - 40, 200), rng.uniform(40, h - 40, 200)], 1).astype(np.float32)
    a, sa, _ = orc.calc_optical_flow_pyr_lk(I, J, pts, accum_mode=0)
    b, sb, _ = orc.calc_optical_flow_pyr_lk(I, J, pts, accum_mode=1)
    assert np.array_equal(sa, sb)
    assert np.abs(a - b).max() < 2e-3


def test_lk_empty_and_threads(orc):
    I = smooth_image(200, 100)
    out, st, _ = orc.calc_optical_flow_pyr_lk(I, I, np.zeros((0, 2), np.float32))
    assert out.shape == (0, 2) and st.shape == (0,)
    pts = np.array([[50, 50], [100.5, 40.25], [150, 60]], np.float32)
    a, _, _ = orc.calc_optical_flow_pyr_lk(I, smooth_image(200, 100, 1.5, 0), pts, nthreads=1)
    b, _, _ = orc.calc_optical_flow_pyr_lk(I, smooth_image(200, 100, 1.5, 0), pts, nthreads=4)
    assert np.array_equal(a, b)
