"""The real kernel sources (pyramid.hip, lk.hip) executed on the CPU through the coroutine SIMT
emulator of tests/host_check (hip_emu.h / kernel_emu.cpp) and compared with the oracle: bordered
pyramid layout, Scharr images, lane mapping, packed pixel arithmetic, DPP reductions, search-tile
handling -- everything of the LK data path except the silicon.  The -m gpu tests repeat the same
comparisons on the MI355X through the C ABI; this file is what lets a kernel change be checked
before any GPU time is spent.  Unit test of device code, not a product path."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import BUILD_DIR, ROOT, SAN_FLAGS, vp


@pytest.fixture(scope="module")
def kemu():
    src_dir = os.path.join(ROOT, "tests", "host_check")
    out_dir = BUILD_DIR
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libkernel_emu.so")
    csrc = os.path.join(ROOT, "visual_odom_amd", "csrc")
    deps = [os.path.join(src_dir, f) for f in ("kernel_emu.cpp", "hip_emu.h")]
    deps += [os.path.join(csrc, f) for f in ("lk.hip", "pyramid.hip", "fast.hip", "vo_dev.h", "vo_kernels.h", "vo_lkmath.h", "vo_svd_wide.h",
                                               "vo_linalg.h", "vo_epnp.h", "pnp.hip", "vo_p3p.h", "vo_math.h",
                                               "vo_seqtail.h", "vo_integrate.h", "post.hip", "vo_tri.h",
                                               "essential.hip", "vo_fivept.h", "seq.hip")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-Wno-unknown-pragmas", "-Wno-attributes"] + SAN_FLAGS + ["-o", so,
                               os.path.join(src_dir, "kernel_emu.cpp")])
    lib = C.CDLL(so)
    lib.ke_run.restype = C.c_int
    return lib


def ke_run(lib, imgs, pts=None, max_level=3, want_level=-1, max_count=30, eps=0.01, min_eig=1e-3, full_chain=1):
    imgs = np.ascontiguousarray(np.stack(imgs), np.uint8)
    n_img, h, w = imgs.shape
    pts = np.zeros((0, 2), np.float32) if pts is None else np.ascontiguousarray(pts, np.float32)
    n = len(pts)
    lvl = np.zeros((h, w), np.uint8)
    der = np.zeros((h, w), np.uint32)
    lw, lh = C.c_int(0), C.c_int(0)
    trk = np.zeros((4, max(n, 1), 2), np.float32)
    st = np.zeros((4, max(n, 1)), np.uint8)
    levels = lib.ke_run(vp(imgs), n_img, w, h, max_level, want_level, vp(lvl), vp(der), C.byref(lw), C.byref(lh),
                        vp(pts), n, max_count, C.c_double(eps), C.c_float(min_eig), full_chain, vp(trk), vp(st))
    lw, lh = lw.value, lh.value
    return dict(levels=levels, lvl=lvl.ravel()[:lw * lh].reshape(lh, lw), der=der.ravel()[:lw * lh].reshape(lh, lw),
                trk=trk[:, :n], status=st[:, :n])


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(params=[2, 3, 0, 1], ids=["fused-pass", "fused-pass-dispatch-order", "column-walk", "lds-tile"])
def pyr_kernel(kemu, request):
    """the product's pyramid chain (round 4: the fused passes, pyr_pass_kernel level by level, in XCD-aware and in dispatch
    order of the workgroups) and the three-kernel chain it replaced with either of its pyr_down kernels"""
    kemu.ke_set_pyr_lds(request.param)
    yield request.param
    kemu.ke_set_pyr_lds(2)


@pytest.mark.parametrize("shape", [(100, 200), (97, 131), (45, 53), (150, 70)])
def test_emulated_pyramid_and_scharr_match_oracle(kemu, orc, shape, pyr_kernel):
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    ref = orc.build_pyramid(img, 3)
    expect = 1
    hh, ww = shape
    while expect < 4 and (ww + 1) // 2 > 21 and (hh + 1) // 2 > 21:
        ww, hh, expect = (ww + 1) // 2, (hh + 1) // 2, expect + 1
    for l in range(expect):
        r = ke_run(kemu, [img], want_level=l)
        assert r["levels"] == expect
        assert np.array_equal(r["lvl"], ref[l]), l
        d = orc.scharr(ref[l]).astype(np.int64) * 4
        packed = ((d[..., 0] & 0xffff) | ((d[..., 1] & 0xffff) << 16)).astype(np.uint32)
        assert np.array_equal(r["der"], packed), l


@pytest.mark.parametrize("shape", [(100, 203), (61, 96), (48, 45)])
def test_emulated_borders(kemu, orc, shape, pyr_kernel):
    """every pixel of the bordered allocation that a kernel may read: REFLECT_101 of the level around it (VO_BY rows above /
    below, VO_BX columns left, at least VO_BY right -- the fill runs to the end of the row), derivatives zero outside the
    image (calcSharrDeriv's constant border of the derivative image, which LK reads for off-image windows)"""
    BX, BY = 32, 24
    rng = np.random.default_rng(shape[1])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    ref = orc.build_pyramid(img, 3)
    h, w = shape
    for l in range(len(ref)):
        cap = 4 << 20
        pix, der = np.zeros(cap, np.uint8), np.zeros(cap, np.uint32)
        lw, lh, ls = C.c_int(0), C.c_int(0), C.c_int(0)
        levels = kemu.ke_bordered_level(vp(img), w, h, 3, l, vp(pix), vp(der), cap, C.byref(lw), C.byref(lh), C.byref(ls))
        if levels <= l:
            break
        lw, lh, ls = lw.value, lh.value, ls.value
        assert (lh, lw) == ref[l].shape and ls % 16 == 0 and ls - BX - lw >= BY
        pix = pix[:ls * (lh + 2 * BY)].reshape(lh + 2 * BY, ls)
        der = der[:ls * (lh + 2 * BY)].reshape(lh + 2 * BY, ls)
        right = ls - BX - lw

        def refl(n, lo, hi):   # cv::borderInterpolate(BORDER_REFLECT_101), repeated for levels narrower than the border
            idx = np.arange(lo, hi)
            if n == 1:
                return np.zeros_like(idx)
            period = 2 * (n - 1)
            idx = np.mod(idx, period)
            return np.where(idx >= n, period - idx, idx)
        expect = ref[l][refl(lh, -BY, lh + BY)][:, refl(lw, -BX, lw + right)]
        assert np.array_equal(pix, expect), l
        inside = np.zeros_like(der, bool)
        inside[BY:BY + lh, BX:BX + lw] = True
        assert not der[~inside].any(), l
        d = orc.scharr(ref[l]).astype(np.int64) * 4
        packed = ((d[..., 0] & 0xffff) | ((d[..., 1] & 0xffff) << 16)).astype(np.uint32)
        assert np.array_equal(der[BY:BY + lh, BX:BX + lw], packed), l


def _oracle_hops(orc, L0, R0, L1, R1, pts, **kw):
    p1, s1, _ = orc.calc_optical_flow_pyr_lk(L0, R0, pts, **kw)
    p2, s2, _ = orc.calc_optical_flow_pyr_lk(R0, R1, p1, **kw)
    p3, s3, _ = orc.calc_optical_flow_pyr_lk(R1, L1, p2, **kw)
    p4, s4, _ = orc.calc_optical_flow_pyr_lk(L1, L0, p3, **kw)
    return np.stack([p1, p2, p3, p4]), np.stack([s1, s2, s3, s4])


@pytest.fixture(params=[0, 1, 2, 3], ids=["one-feature-per-wave", "two-features-per-wave", "hop-0-then-hops-1-3", "four-launches"])
def lk_variant(request, kemu):
    """the LK emulation tests run on lk_circular_kernel, on lk_circular_pair_kernel and on lk_hops_kernel (the synchronous
    calls' split chain: hop 0, then hops 1 .. 3 from what the first launch left in the track / status arrays, which start as
    garbage; and one launch per hop)"""
    kemu.ke_set_lk_pair(request.param)
    yield request.param
    kemu.ke_set_lk_pair(0)


def test_emulated_lk_kernel_bit_exact(kemu, orc, small_seq, lk_variant):
    s = small_seq
    imgs = [s["L"][0], s["R"][0], s["L"][1], s["R"][1]]
    border = np.array([[0, 0], [479, 159], [2.5, 80.25], [476.2, 10.7], [240, 1.1], [250.4, 158.9],
                       [-5, 50], [100, -3], [520, 100], [12.5, 12.5], [-25, 80], [240, 185]], np.float32)
    pts = np.vstack([s["pts"][0][::6], border]).astype(np.float32)
    r = ke_run(kemu, imgs, pts)
    ref, st = _oracle_hops(orc, *imgs, pts)
    assert np.array_equal(r["status"], st)
    assert np.array_equal(bits(r["trk"]), bits(ref))
    assert st[:, :len(pts) - len(border)].mean() > 0.5
    # default mode: a feature retires at the first hop the circular filter rejects; everything up to and
    # including that hop is unchanged, later hops read status 0, so the filter's survivors are the same
    e = ke_run(kemu, imgs, pts, full_chain=0)
    neg = np.concatenate([(pts < 0).any(1)[None], (ref[:3] < 0).any(2)])           # pt0 .. pt3 negative
    bad = (st == 0) | np.vstack([neg[0] | neg[1], neg[2], neg[3], np.zeros(len(pts), bool)])
    alive = np.vstack([np.ones(len(pts), bool), ~np.logical_or.accumulate(bad, 0)[:3]])  # hop k evaluated?
    assert np.array_equal(e["status"][alive], st[alive]) and not e["status"][~alive].any()
    assert np.array_equal(bits(e["trk"][alive]), bits(ref[alive]))
    keep_full = (st != 0).all(0) & ~neg.any(0)
    keep_early = (e["status"] != 0).all(0) & ~np.concatenate([(pts < 0).any(1)[None], (e["trk"][:3] < 0).any(2)]).any(0)
    assert np.array_equal(keep_full, keep_early) and 0 < keep_full.sum() < len(pts)


def test_emulated_lk_large_motion_and_params(kemu, orc, lk_variant):
    """big flow (search tile re-fetched mid-iteration), fractional / out-of-image start points,
    non-reference parameters"""
    from test_oracle_images import smooth_image
    w, h = 256, 128
    imgs = [smooth_image(w, h, seed=9), smooth_image(w, h, 13.7, -9.2, seed=9), smooth_image(w, h, 20.1, 4.4, seed=9),
            smooth_image(w, h, -6.3, 11.8, seed=9)]
    rng = np.random.default_rng(3)
    pts = np.stack([rng.uniform(-10, w + 10, 40), rng.uniform(-10, h + 10, 40)], 1).astype(np.float32)
    r = ke_run(kemu, imgs, pts)
    ref, st = _oracle_hops(orc, *imgs, pts)
    assert np.array_equal(r["status"], st) and np.array_equal(bits(r["trk"]), bits(ref))
    r = ke_run(kemu, imgs, pts[:16], max_level=2, max_count=7, eps=0.03, min_eig=0.01)
    ref, st = _oracle_hops(orc, *imgs, pts[:16], max_level=2, max_count=7, eps=0.03, min_eig=0.01)
    assert np.array_equal(r["status"], st) and np.array_equal(bits(r["trk"]), bits(ref))


def test_emulated_lk_negative_bilinear_weight(kemu, orc, lk_variant):
    """regression found on the MI355X: the three rounded bilinear weights can add up to 2^14 + 1, which
    makes iw11 = -1; feature 133 of the large-motion GPU test walks through such an iteration"""
    from test_oracle_images import smooth_image
    w, h = 512, 256
    imgs = [smooth_image(w, h, seed=9), smooth_image(w, h, 13.7, -9.2, seed=9), smooth_image(w, h, 20.1, 4.4, seed=9),
            smooth_image(w, h, -6.3, 11.8, seed=9)]
    rng = np.random.default_rng(3)
    pts = np.stack([rng.uniform(-10, w + 10, 700), rng.uniform(-10, h + 10, 700)], 1).astype(np.float32)[[133, 7, 400]]
    r = ke_run(kemu, imgs, pts)
    ref, st = _oracle_hops(orc, *imgs, pts)
    assert np.array_equal(r["status"], st) and np.array_equal(bits(r["trk"]), bits(ref))


def test_lk_nonfinite_points(kemu, orc, lk_variant):
    """NaN / infinite / beyond-int32 / denormal / negative start points.  The emulator converts float -> int
    like gfx950 (vo_f2i: NaN -> 0, saturating), the oracle like x86 (INT_MIN): status and positions must agree all the same
    (a NaN position compares as NaN)."""
    from test_oracle_images import smooth_image
    w, h = 256, 128
    imgs = [smooth_image(w, h, seed=9), smooth_image(w, h, 3.7, -2.2, seed=9), smooth_image(w, h, 5.1, 1.4, seed=9),
            smooth_image(w, h, -2.3, 2.8, seed=9)]
    from adversarial import LK_POINTS as pts, LK_N_HOPELESS
    for full_chain in (1, 0):
        r = ke_run(kemu, imgs, pts, full_chain=full_chain)
        ref, st = _oracle_hops(orc, *imgs, pts)
        if full_chain:
            assert np.array_equal(r["status"], st)
            assert np.array_equal(r["trk"], ref, equal_nan=True)
            assert np.array_equal(np.isnan(r["trk"]), np.isnan(ref))
            assert not st[:, :LK_N_HOPELESS].any() and st[:, -1].all()   # every non-finite / huge point fails every hop
        else:
            assert np.array_equal(r["status"][0], st[0])
            assert np.array_equal(r["trk"][0], ref[0], equal_nan=True)
        keep = (r["status"] != 0).all(0) & ~(pts < 0).any(1) & ~(r["trk"][:3] < 0).any(2).any(0)
        keep_ref = (st != 0).all(0) & ~(pts < 0).any(1) & ~(ref[:3] < 0).any(2).any(0)
        assert np.array_equal(keep, keep_ref) and keep[-1]


def ke_detect(lib, img, tracked=None, ages=None, threshold=20, nonmax=1, detect=1, bucket_size=0, fpb=1, cap=40000):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    tracked = np.zeros((0, 2), np.float32) if tracked is None else np.ascontiguousarray(tracked, np.float32).reshape(-1, 2)
    ages = np.zeros(len(tracked), np.int32) if ages is None else np.ascontiguousarray(ages, np.int32)
    out_p, out_a = np.zeros((cap, 2), np.float32), np.zeros(cap, np.int32)
    n = lib.ke_detect(vp(img), w, h, threshold, nonmax, detect, vp(tracked), len(tracked), vp(ages), len(ages), bucket_size,
                      fpb, vp(out_p), vp(out_a), cap)
    assert 0 <= n <= cap
    return out_p[:n].copy(), out_a[:n].copy()


@pytest.fixture(params=[0, 1, 2], ids=["tile-64x16", "tile-64x32", "tile-128x32"])
def fast_variant(request, kemu):
    """the FAST emulation tests run on all three tile forms of the kernel -- and, with the middle one, on the 1024-thread form
    of bucket_kernel (launches of <= 4 frames; 256 threads otherwise)"""
    kemu.ke_set_fast_big(request.param)
    kemu.ke_set_bucket_threads(1024 if request.param == 1 else 256)
    yield request.param
    kemu.ke_set_fast_big(0)
    kemu.ke_set_bucket_threads(256)


def test_emulated_fast_matches_oracle(kemu, orc, small_seq, fast_variant):
    img = small_seq["L"][0]
    for thr, nonmax in ((20, 1), (35, 0), (0, 1)):
        got, _ = ke_detect(kemu, img, threshold=thr, nonmax=nonmax)
        ref = orc.fast_detect(img, thr, bool(nonmax))
        assert len(ref) > 50 and np.array_equal(got, ref), (thr, nonmax)
    rng = np.random.default_rng(4)
    noise = rng.integers(0, 256, (40, 57), dtype=np.uint8)
    got, _ = ke_detect(kemu, noise, threshold=20)
    assert np.array_equal(got, orc.fast_detect(noise, 20, True))
    flat = np.full((32, 32), 77, np.uint8)
    assert len(ke_detect(kemu, flat)[0]) == 0


@pytest.mark.parametrize("shape", [(33, 70), (40, 200), (34, 1241), (32, 4096), (45, 2000)])
def test_emulated_fast_row_packing(kemu, orc, shape):
    """fast_nms_write_kernel packs 64 / segs image rows into a wavefront (segs = 64-pixel segments per row) and ranks a
    lane's corners within the lanes of its row: widths with 2, 4, 20 (three rows, four spare lanes), 32 and 64 (one row)
    segments, heights that leave the last wavefront's rows partly outside the image; corner list in cv::FAST's row-major
    order, bit for bit"""
    h, w = shape
    rng = np.random.default_rng(h * w)
    base = rng.integers(0, 256, (h // 3 + 2, w // 3 + 2)).astype(np.float32)
    img = np.kron(base, np.ones((3, 3), np.float32))[:h, :w].astype(np.uint8)
    got, _ = ke_detect(kemu, img, threshold=20, nonmax=1, cap=65536)
    ref = orc.fast_detect(img, 20, True)
    assert len(ref) > 20 and np.array_equal(got, ref)


@pytest.mark.parametrize("h,w", [(32, 32), (33, 47), (150, 70), (64, 257), (121, 1023), (480, 642)])
def test_emulated_small_and_odd_shapes(kemu, orc, h, w, fast_variant):
    """the shapes of tests/test_gpu_round4.py::test_fused_pyramid_pass_on_small_and_odd_shapes on the emulator, whose image
    table gives every level its own exactly-sized block (one-level pyramids, widths of every residue mod 4, levels narrower
    than a wavefront's span): pyramid levels, the four LK hops next to all four image edges and the FAST corner list (every
    tile form: the tiles of the last tile column hang over the row end) bit-exact against the oracle -- and, in the
    sanitizer tier, without one access outside a level's allocation"""
    rng = np.random.default_rng(h * 1000 + w)
    base = rng.integers(0, 256, (h // 4 + 2, w // 4 + 2)).astype(np.float32)
    img = np.kron(base, np.ones((4, 4), np.float32))[:h, :w]
    img = np.clip(img + rng.normal(0, 6, (h, w)), 0, 255).astype(np.uint8)
    got, _ = ke_detect(kemu, img, threshold=20, nonmax=1, cap=65536)
    assert np.array_equal(got, orc.fast_detect(img, 20, True))
    if fast_variant:
        return                                           # (pyramids and LK do not depend on the FAST tile form)
    shifted = np.roll(img, (1, -2), (0, 1))
    ref = orc.build_pyramid(img, 3)
    n_levels, cw, ch = 1, w, h
    while n_levels < 4 and (cw + 1) // 2 > 21 and (ch + 1) // 2 > 21:
        cw, ch, n_levels = (cw + 1) // 2, (ch + 1) // 2, n_levels + 1
    xs = rng.uniform(0, w - 1, 24).astype(np.float32)
    ys = rng.uniform(0, h - 1, 24).astype(np.float32)
    xs[:8] = [0, 0.4, w - 1, w - 1.3, 2.5, w / 2, w / 2, 7.25]
    ys[:8] = [0, h - 1, 0.6, h - 1, h / 2, 0, h - 1, 3.75]
    pts = np.stack([xs, ys], 1)
    imgs = [img, shifted, shifted, img]
    for l in range(n_levels):
        r = ke_run(kemu, imgs, pts if l == 0 else None, want_level=l)
        assert r["levels"] == n_levels and np.array_equal(r["lvl"], ref[l]), l
        if l == 0:
            want, st = _oracle_hops(orc, *imgs, pts, max_level=n_levels - 1)
            assert np.array_equal(r["status"], st) and np.array_equal(bits(r["trk"]), bits(want))


def test_emulated_bucketing_matches_oracle(kemu, orc, small_seq, fast_variant):
    """bucket.cpp / feature.cpp:206-253 quirks: aliased last column, duplicate emission, slot-0 overwrite,
    age >= 10 dropped, ages longer than points (stale ages inherited by new corners)"""
    img = small_seq["L"][0]
    h, w = img.shape
    rng = np.random.default_rng(5)
    tracked = np.stack([rng.uniform(0, w - 1, 300), rng.uniform(0, h - 1, 300)], 1).astype(np.float32)
    tracked[:4] = [[w - 1, 3.5], [w - 0.5, h - 1], [0, 0], [w - 2.25, 50]]      # last-column aliases
    ages = rng.integers(0, 13, 330).astype(np.int32)                             # 30 more ages than points
    fast = orc.fast_detect(img, 20, True)
    for bucket, fpb in ((h // 10, 1), (h // 10, 6), (23, 3)):
        comb_p = np.vstack([tracked, fast])
        comb_a = np.concatenate([ages, np.zeros(len(tracked) + len(fast) - len(ages), np.int32)])
        ref_p, ref_a = orc.bucketing_features(h, w, comb_p, comb_a, bucket, fpb)
        got_p, got_a = ke_detect(kemu, img, tracked, ages, bucket_size=bucket, fpb=fpb)
        assert len(ref_p) > 30 and np.array_equal(got_p, ref_p) and np.array_equal(got_a, ref_a), (bucket, fpb)
    # no re-detection: only the carried set is bucketed
    ref_p, ref_a = orc.bucketing_features(h, w, tracked, ages[:len(tracked)], h // 10, 2)
    got_p, got_a = ke_detect(kemu, img, tracked, ages[:len(tracked)], detect=0, bucket_size=h // 10, fpb=2)
    assert np.array_equal(got_p, ref_p) and np.array_equal(got_a, ref_a)


def test_exact_wave_sums_at_the_extremes(kemu):
    """wave_sum2 / wave_sum3 (v_permlane32/16_swap + DPP trees, hi/lo split) must return (float)(int64 sum) for any
    per-lane partials below the documented bound 2^28 (8-lane sums must fit int32), including totals far beyond int32"""
    rng = np.random.default_rng(11)
    B = (1 << 28) - 1
    cases = [np.full(64, B), np.full(64, -B), np.where(np.arange(64) % 2 == 0, B, -B), np.zeros(64, np.int64)]
    cases += [rng.integers(-B, B + 1, 64) for _ in range(40)]
    cases += [rng.integers(-3, 4, 64) * (B // 3) + rng.integers(-1, 2, 64) for _ in range(10)]  # rounding ties nearby
    for k in range(0, len(cases) - 2):
        a, b, c = (np.ascontiguousarray(cases[k + j], np.int32) for j in range(3))
        out = np.zeros(5, np.float32)
        kemu.ke_wave_sums(vp(a), vp(b), vp(c), vp(out))
        ref = [np.float32(int(x.astype(np.int64).sum())) for x in (a, b, a, b, c)]
        assert [bits(o) for o in out] == [bits(r) for r in ref], k


def test_fast_compass_pair_equals_the_scalar_pretest(kemu):
    """the tile kernel tests two positions per packed 16-bit instruction (fast_compass_pair: saturating subtractions, minima);
    the exhaustive ring test below is about the scalar fast_compass_candidate -- the two agree for every centre, threshold and
    compass pixels on both sides of centre +- threshold and at the ends of the byte range (~7 M combinations)"""
    kemu.ke_fast_compass_pair_check.restype = C.c_longlong
    assert kemu.ke_fast_compass_pair_check() == 0


def test_fast_compass_pretest_is_necessary_exhaustive(kemu):
    """the pre-test fast_tile_kernel compacts on (two neighbouring compass pixels both brighter or both darker) must never
    reject a TYPE_9_16 corner: checked on all 3^16 = 43 046 721 brighter / darker / similar ring patterns with the real
    device functions; the counts pin the corner test itself (rings with >= 9 contiguous equal non-similar states)"""
    nc, nk = C.c_longlong(0), C.c_longlong(0)
    kemu.ke_fast_compass_exhaustive.restype = C.c_longlong
    bad = kemu.ke_fast_compass_exhaustive(C.byref(nc), C.byref(nk))
    assert bad == 0
    # independent count of the corner rings: cyclic ternary strings of length 16 with a run of >= 9 equal non-zero symbols
    def corner_rings():
        total = 0
        for sym in (1, 2):
            for run in range(9, 17):
                if run == 16:
                    total += 1
                else:
                    # a maximal run of exactly `run` symbols starting at one of 16 rotations, bounded by a different symbol
                    # on both sides (the same cell when run == 15), the remaining cells free; runs >= 9 cannot occur twice
                    free = 16 - run - 2
                    total += 16 * (2 * 3 ** 0 if run == 15 else 2 * 2 * 3 ** free)
        return total
    assert nc.value == corner_rings()
    assert nk.value > nc.value


@pytest.mark.parametrize("pattern", ["white", "black", "columns", "rows", "checker"])
def test_emulated_pyramid_extremes(kemu, orc, pattern, pyr_kernel):
    """the packed 16-bit arithmetic of pyr_down_kernel (6 q2 + 4 (q1 + q3) + q0 + q4 + 128 <= 65408) and scharr_kernel
    (|4 Ix|, |4 Iy| <= 16320) at the ends of their ranges"""
    h, w = 90, 150
    yy, xx = np.mgrid[0:h, 0:w]
    img = {"white": np.full((h, w), 255), "black": np.zeros((h, w)), "columns": (xx % 2) * 255, "rows": (yy % 2) * 255,
           "checker": ((xx + yy) % 2) * 255}[pattern].astype(np.uint8)
    ref = orc.build_pyramid(img, 3)
    levels = ke_run(kemu, [img], want_level=0)["levels"]   # the plan stops before a level of 21 pixels or fewer
    assert levels == 3
    for l in range(levels):
        r = ke_run(kemu, [img], want_level=l)
        assert np.array_equal(r["lvl"], ref[l]), l
        d = orc.scharr(ref[l]).astype(np.int64) * 4
        packed = ((d[..., 0] & 0xffff) | ((d[..., 1] & 0xffff) << 16)).astype(np.uint32)
        assert np.array_equal(r["der"], packed), l
    # a vertical / horizontal step edge of full contrast: the largest derivative the Scharr image can hold
    step = np.zeros((h, w), np.uint8)
    step[:, w // 2:] = 255
    r = ke_run(kemu, [step], want_level=0)
    ix = (r["der"] & 0xffff).astype(np.uint16).view(np.int16)
    assert ix.max() == 16320


def test_row_cooperative_svd12_is_bit_identical_to_the_one_lane_routine(kemu):
    """vo_svd_wide.h (EPnP's 12 x 12 Jacobi sweeps over the 16 lanes of a DPP row, every sum in the serial order through
    row broadcasts) + jacobi12_finish against jacobi_svd<12, 12, false>, the routine the one-hypothesis-per-lane kernel runs:
    the sorted, normalised rows must agree BIT FOR BIT -- on M^T M of EPnP-shaped 10 x 12 matrices (rank 10: two singular values
    at rounding level), on full-rank and on exactly singular matrices (zero rows: the pseudo-random fill of lapack.cpp)."""
    rng = np.random.default_rng(7)
    mats = []
    for _ in range(14):  # M^T M with M's sparsity: rows (a fu, 0, a (uc - u)) / (0, a fv, a (vc - v)) per control point
        M = np.zeros((10, 12))
        al = rng.normal(0.25, 0.6, (5, 4))
        uv = rng.uniform(0, 1241, (5, 2)) * [1, 0.3]
        for p in range(5):
            for q in range(4):
                M[2 * p, 3 * q] = al[p, q] * 718.856
                M[2 * p, 3 * q + 2] = al[p, q] * (607.19 - uv[p, 0])
                M[2 * p + 1, 3 * q + 1] = al[p, q] * 718.856
                M[2 * p + 1, 3 * q + 2] = al[p, q] * (185.2 - uv[p, 1])
        mats.append(M.T @ M)
    for _ in range(4):
        A = rng.normal(size=(12, 12))
        mats.append(A @ A.T)
    Z = rng.normal(size=(12, 12))
    Z[3] = 0
    Z[:, 3] = 0
    Z[7] = 0
    Z[:, 7] = 0
    mats += [(Z + Z.T) / 2 + 12 * np.diag((np.arange(12) % 4 != 3).astype(float)), np.zeros((12, 12)), np.eye(12)]
    mats = np.ascontiguousarray(np.array(mats), np.float64)
    n = len(mats)
    wide, serial = np.zeros_like(mats), np.zeros_like(mats)
    dp = C.POINTER(C.c_double)
    kemu.ke_svd12_wide(mats.ctypes.data_as(dp), n, wide.ctypes.data_as(dp), serial.ctypes.data_as(dp))
    assert np.isfinite(serial).all()
    assert np.array_equal(wide.view(np.uint64), serial.view(np.uint64))
    # and it is an SVD: rows orthonormal, the first EPnP matrix's last two rows span its null space
    U = serial[0]
    assert np.allclose(U @ U.T, np.eye(12), atol=1e-12)
    assert np.abs(mats[0] @ U[10:].T).max() < 1e-6 * np.abs(mats[0]).max()


def test_wavefront_6x6_solve_is_bit_identical_to_solve_svd(kemu):
    """the Levenberg-Marquardt step of the pose refinement: (J^T J + lambda diag) x = J^T e through the SVD, its Jacobi
    sweeps run by a wavefront (15 pairs in 9 steps, V accumulated) -- against solve_svd<6, 6> bit for bit, on normal
    matrices of projection Jacobians, on an ill-conditioned and on a singular system"""
    rng = np.random.default_rng(11)
    As, bs = [], []
    for _ in range(40):
        J = rng.normal(size=(rng.integers(8, 400), 6)) * [300, 300, 300, 40, 40, 8]
        A = J.T @ J
        A[np.diag_indices(6)] *= 1 + 10.0 ** rng.integers(-8, 2)
        As.append(A)
        bs.append(J.T @ rng.normal(size=len(J)))
    H = np.vander(np.linspace(1, 2, 6), 6)
    As += [H.T @ H, np.diag([4.0, 3.0, 0.0, 2.0, 0.0, 1.0]), np.zeros((6, 6))]
    bs += [np.ones(6), np.arange(6.0), np.ones(6)]
    A = np.ascontiguousarray(np.array(As), np.float64)
    b = np.ascontiguousarray(np.array(bs), np.float64)
    xw, xs = np.zeros_like(b), np.zeros_like(b)
    dp = C.POINTER(C.c_double)
    kemu.ke_solve6_wave(A.ctypes.data_as(dp), b.ctypes.data_as(dp), len(A), xw.ctypes.data_as(dp), xs.ctypes.data_as(dp))
    assert np.isfinite(xs).all()
    assert np.array_equal(xw.view(np.uint64), xs.view(np.uint64))
    assert np.allclose(A[0] @ xs[0], b[0], rtol=1e-8)


def test_four_kernel_epnp_composition_is_bit_identical_to_the_one_piece_solver(kemu, orc):
    """pnp.hip's small-launch EPnP on the CPU: set-up, the 12 x 12 SVD by an (emulated) 128-thread workgroup, each of the
    three approximations from the state the workspace holds, selection -- rvec and tvec BIT FOR BIT those of epnp5_solve (the
    one-kernel form, itself bit-identical to the oracle's EPnP: test_device_math_on_host.py), on noisy, exact, far and
    near-planar 5-point sets"""
    rng = np.random.default_rng(21)
    K = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1]], np.float32)
    X, U = [], []
    for case in range(28):
        lo, hi = ([-8, -2, 4], [8, 2, 40]) if case % 4 else ([-30, -6, 60], [30, 6, 200])
        xyz = rng.uniform(lo, hi, (5, 3)).astype(np.float32)
        if case % 7 == 3:
            xyz[:, 2] = xyz[0, 2] + rng.normal(0, 0.01, 5).astype(np.float32)  # almost fronto-parallel plane
        rv, tv = rng.normal(0, 0.02, 3), rng.normal(0, 0.5, 3)
        uv = orc.project_points(xyz, rv, tv, K) + (rng.normal(0, 0.4, (5, 2)) if case % 3 else 0)
        X.append(xyz)
        U.append(uv.astype(np.float32))
    X = np.ascontiguousarray(np.array(X, np.float32))
    U = np.ascontiguousarray(np.array(U, np.float32))
    split, mono = np.zeros((len(X), 6)), np.zeros((len(X), 6))
    dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
    kemu.ke_epnp_split(X.ctypes.data_as(fp), U.ctypes.data_as(fp), K.ctypes.data_as(fp), len(X), split.ctypes.data_as(dp),
                       mono.ctypes.data_as(dp))
    assert np.isfinite(mono).all()
    assert np.array_equal(split.view(np.uint64), mono.view(np.uint64))


def ke_pnp(lib, X, uv, K, iters=500, reproj=0.5, confidence=float(np.float32(0.999)), split=0, first_chunk=128):
    X = np.ascontiguousarray(X, np.float32).reshape(-1, 3)
    uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
    K = np.ascontiguousarray(K, np.float32).reshape(3, 3)
    n = len(X)
    rv, tv = np.zeros(3), np.zeros(3)
    inl = np.zeros(max(n, 1), np.int32)
    ninl = C.c_int(0)
    dbg = (C.c_int * 4)()
    lib.ke_pnp_ransac.restype = C.c_int
    rc = lib.ke_pnp_ransac(X.ctypes.data_as(C.POINTER(C.c_float)), uv.ctypes.data_as(C.POINTER(C.c_float)), n,
                           K.ctypes.data_as(C.POINTER(C.c_float)), iters, C.c_float(reproj), C.c_double(confidence), split,
                           first_chunk, rv.ctypes.data_as(C.POINTER(C.c_double)), tv.ctypes.data_as(C.POINTER(C.c_double)),
                           inl.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(ninl), dbg)
    return rc, rv, tv, inl[:ninl.value].copy(), list(dbg)


@pytest.mark.parametrize("n,outliers,noise,seed,first_chunk", [(300, 0.3, 0.15, 2, 128), (60, 0.5, 0.2, 3, 128),
                                                             (300, 0.3, 0.15, 2, 64), (150, 0.45, 0.2, 6, 64),
                                                             (6, 0.0, 0.05, 5, 128), (5, 0.0, 0.0, 8, 128)])
def test_emulated_pose_chain_matches_oracle(kemu, orc, n, outliers, noise, seed, first_chunk):
    """pnp.hip kernel by kernel on the CPU emulator (raw RNG table, wavefront subsets, one-kernel EPnP, votes, control-flow
    replay over two chunks, refinement with the wavefront 6 x 6 solve and the fixed-order workgroup sums) against the
    oracle's solvePnPRansac: same status, inlier set, iterations / winner / best count, Levenberg-Marquardt iterations;
    pose <= 1e-6 (the refinement's sums are formed in another order than the serial code's).  first_chunk = 64 is what
    launches of 128 frames and more use; (60, 0.5) and (150, 0.45) need the second chunk."""
    from test_oracle_geom import planted_problem, K_KITTI
    X, uv, r, t, _ = planted_problem(orc, n, outliers, noise, seed)
    rc, rv, tv, inl, dbg = orc.solve_pnp_ransac(X, uv, K_KITTI)
    grc, grv, gtv, ginl, gdbg = ke_pnp(kemu, X, uv, K_KITTI, first_chunk=first_chunk)
    assert grc == rc and np.array_equal(ginl, inl)
    if n > 5:  # (exactly five points: solvePnPRansac runs no RANSAC loop, the oracle's counters stay 0)
        assert tuple(gdbg[:4]) == tuple(int(x) for x in dbg[:4])
    assert np.abs(grv - rv).max() <= 1e-6 and np.abs(gtv - tv).max() <= 1e-6


def test_emulated_pose_chain_four_kernel_epnp_and_edge_cases(kemu, orc):
    """the small-launch form (epnp_prepare / svd12_wave / epnp_approx / epnp_select kernels with their workspace) through
    the same chain: results identical to the one-kernel form's, hence to the oracle's; no consensus -> status 0 with the
    last hypothesis; exactly four points -> P3P"""
    from test_oracle_geom import planted_problem, K_KITTI
    X, uv, r, t, _ = planted_problem(orc, 80, 0.35, 0.15, 12)
    rc, rv, tv, inl, dbg = orc.solve_pnp_ransac(X, uv, K_KITTI, iterations=40)  # (40 hypotheses: the emulated SVD is slow)
    a = ke_pnp(kemu, X, uv, K_KITTI, iters=40, split=1)
    b = ke_pnp(kemu, X, uv, K_KITTI, iters=40, split=0)
    s = ke_pnp(kemu, X, uv, K_KITTI, iters=40, split=2)  # the slim form (round 4): matrices in the global workspace
    assert a[0] == b[0] == s[0] == rc and np.array_equal(a[3], inl) and np.array_equal(b[3], inl) and np.array_equal(s[3], inl)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[4] == b[4]   # the forms: bit for bit
    assert np.array_equal(s[1], b[1]) and np.array_equal(s[2], b[2]) and s[4] == b[4]
    assert np.abs(a[1] - rv).max() <= 1e-6 and np.abs(a[2] - tv).max() <= 1e-6
    # a frame that needs hypotheses behind the first chunk: small launches run everything behind it in ONE launch
    # (ransac_rest_kernel: subsets, EPnP, votes by every workgroup, the control-flow replay by the one that arrives last) --
    # bit for bit what the four kernels per chunk give, and the oracle's control flow
    X2, uv2, _, _, _ = planted_problem(orc, 60, 0.5, 0.2, 3)
    rc, rv, tv, inl, dbg = orc.solve_pnp_ransac(X2, uv2, K_KITTI, iterations=200)
    a = ke_pnp(kemu, X2, uv2, K_KITTI, iters=200, split=1, first_chunk=8)
    b = ke_pnp(kemu, X2, uv2, K_KITTI, iters=200, split=0, first_chunk=8)
    assert dbg[0] > 8, "this problem must reach the second chunk"
    assert a[0] == b[0] == rc and np.array_equal(a[3], inl) and np.array_equal(b[3], inl)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[4] == b[4]
    assert tuple(a[4][:4]) == tuple(int(x) for x in dbg[:4])
    assert np.abs(a[1] - rv).max() <= 1e-6 and np.abs(a[2] - tv).max() <= 1e-6
    rng = np.random.default_rng(9)
    Xr = rng.uniform([-10, -2, 4], [10, 2, 50], (60, 3)).astype(np.float32)
    uvr = rng.uniform([0, 0], [1241, 376], (60, 2)).astype(np.float32)
    rc, rv, tv, inl, dbg = orc.solve_pnp_ransac(Xr, uvr, K_KITTI)
    grc, grv, gtv, ginl, gdbg = ke_pnp(kemu, Xr, uvr, K_KITTI)
    assert rc == 0 and grc == 0 and len(ginl) == 0
    assert np.abs(grv - rv).max() <= 1e-6 and np.abs(gtv - tv).max() <= 1e-6
    X4, uv4, _, _, _ = planted_problem(orc, 4, 0.0, 0.0, 31)
    rc, rv, tv, inl, dbg = orc.solve_pnp_ransac(X4, uv4, K_KITTI)
    grc, grv, gtv, ginl, gdbg = ke_pnp(kemu, X4, uv4, K_KITTI)
    assert grc == rc and np.array_equal(ginl, inl)
    if rc == 1:
        assert np.abs(grv - rv).max() <= 1e-9 and np.abs(gtv - tv).max() <= 1e-9


def ke_post(lib, pts, trk, status, P_l, P_r, threshold=0):
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    n = len(pts)
    trk = np.ascontiguousarray(trk, np.float32).reshape(4, n, 2)
    status = np.ascontiguousarray(status, np.uint8).reshape(4, n)
    P_l, P_r = np.ascontiguousarray(P_l, np.float32), np.ascontiguousarray(P_r, np.float32)
    outA, outB = np.zeros((5, max(n, 1), 2), np.float32), np.zeros((4, max(n, 1), 2), np.float32)
    idxA, idxB = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
    nA, nB = C.c_int(0), C.c_int(0)
    xyz = np.zeros((max(n, 1), 3), np.float32)
    lib.ke_post(vp(pts), vp(trk), vp(status), n, threshold, vp(P_l), vp(P_r), vp(outA), vp(idxA), C.byref(nA), vp(outB),
                vp(idxB), C.byref(nB), vp(xyz))
    a, b = nA.value, nB.value
    return dict(A=outA[:, :a], idxA=idxA[:a], B=outB[:, :b], idxB=idxB[:b], xyz=xyz[:b])


def test_whole_hot_path_on_the_cpu_emulator(kemu, orc, small_world, small_seq):
    """circularMatching -> filters -> triangulation -> solvePnPRansac with nothing but the product's kernel sources, run on
    the CPU emulator launch by launch (pyramid.hip, lk.hip, post.hip, pnp.hip), against the oracle: survivors and tracks bit
    for bit, triangulation identical, inlier set / RANSAC control flow / Levenberg-Marquardt iterations identical, pose
    <= 1e-6.  What the GPU parity tests assert of the library, asserted of its kernel code without a GPU."""
    s = small_seq
    imgs = [s["L"][0], s["R"][0], s["L"][1], s["R"][1]]
    pts = s["pts"][0].astype(np.float32)
    r = ke_run(kemu, imgs, pts, full_chain=0)                      # the product's default: a feature retires at its first bad hop
    P_l, P_r = small_world.proj_matrices()
    post = ke_post(kemu, pts, r["trk"], r["status"], P_l, P_r)
    ref = orc.circular_matching(*imgs, pts)
    assert np.array_equal(post["idxA"], ref["keep_idx"]) and len(ref["keep_idx"]) > 100
    for row, name in enumerate(("l0", "r0", "r1", "l1", "l0_ret")):
        assert np.array_equal(bits(post["A"][row]), bits(ref[name])), name
    (l0, r0, l1, r1), valid = orc.check_valid_and_remove(ref["l0"], ref["r0"], ref["l1"], ref["r1"], ref["l0_ret"])
    assert np.array_equal(post["idxB"], ref["keep_idx"][valid])
    for row, a in enumerate((l0, r0, l1, r1)):
        assert np.array_equal(bits(post["B"][row]), bits(a)), row
    xyz = orc.triangulate(P_l, P_r, l0, r0)
    assert np.array_equal(post["xyz"], xyz)
    K = small_world.K()
    rc, rv, tv, inl, dbg = orc.solve_pnp_ransac(xyz, l1, K)
    grc, grv, gtv, ginl, gdbg = ke_pnp(kemu, post["xyz"], post["B"][2], K)
    assert grc == rc == 1 and np.array_equal(ginl, inl)
    assert tuple(gdbg[:4]) == tuple(int(x) for x in dbg[:4])
    assert np.abs(grv - rv).max() <= 1e-6 and np.abs(gtv - tv).max() <= 1e-6


def test_adversarial_values_post_and_bucketing(kemu, orc):
    """tests/adversarial.py through post.hip (filters + triangulation: zero / negative disparity, NaN, inf, huge) and through
    bucket_kernel (carried points the reference would index its bucket vector out of range with) -- against the oracle, NaN as NaN"""
    import adversarial as adv
    P_l = np.array([[718.856, 0, 607.1928, 0], [0, 718.856, 185.2157, 0], [0, 0, 1, 0]], np.float32)
    P_r = P_l.copy()
    P_r[0, 3] = -386.1448
    pl, pr = adv.TRI_LEFT, adv.TRI_RIGHT
    n = len(pl)
    trk = np.stack([pr, pr, pl, pl])
    post = ke_post(kemu, pl, trk, np.ones((4, n), np.uint8), P_l, P_r, threshold=1 << 30)
    keep = ~((pl < 0).any(1) | (pr < 0).any(1))              # deleteUnmatchFeaturesCircle's sign tests (NaN passes them)
    assert np.array_equal(post["idxB"], np.flatnonzero(keep))
    assert adv.same(post["xyz"], orc.triangulate(P_l, P_r, pl[keep], pr[keep]))
    for name, img in adv.degenerate_images(96, 160).items():
        for bs, fpb in ((9, 1), (9, 2), (37, 1)):
            gp, ga = ke_detect(kemu, img, adv.BUCKET_POINTS, adv.BUCKET_AGES, bucket_size=bs, fpb=fpb)
            corners = orc.fast_detect(img)
            allp = np.vstack([adv.BUCKET_POINTS, corners])
            alla = np.concatenate([adv.BUCKET_AGES, np.zeros(max(0, len(allp) - len(adv.BUCKET_AGES)), np.int32)])
            op, oa = orc.bucketing_features(96, 160, allp, alla, bs, fpb)
            assert np.array_equal(gp, op, equal_nan=True) and np.array_equal(ga, oa), (name, bs, fpb)


@pytest.mark.parametrize("threads", [256, 1024])
def test_bucketing_on_fine_grids(kemu, orc, threads):
    """bucket_kernel<4096> (round 6: grids of 1 025 .. 4 096 buckets, [q][n_buckets] table in the same LDS) against the oracle:
    the reference's own rule rows / 10 on a wide, low image (601 x 64: 11 x 101 buckets), a 3-pixel bucket, up to as many per
    bucket as the table holds -- and the refusal beyond the documented limits"""
    import adversarial as adv
    rng = np.random.default_rng(41)
    kemu.ke_set_bucket_threads(threads)
    try:
        cases = ((64, 601, 6, 6), (97, 131, 3, 3), (64, 601, 6, 1), (40, 640, 4, 4))
        for (h, w, bs, fpb) in cases[:2 if threads == 1024 else 4]:
            n = (h // bs + 1) * (w // bs + 1)
            assert 1024 < n <= 4096 and adv.bucket_grid_ok(w, h, bs, fpb), (h, w, bs, fpb, n)
            img = rng.integers(0, 256, (h, w), dtype=np.uint8)
            img[:, : w // 2] = (img[:, : w // 2] // 64) * 64     # (fewer corners on one half: some buckets stay under their quota)
            pts = np.stack([rng.uniform(0, w - 0.01, 300), rng.uniform(0, h - 0.01, 300)], 1).astype(np.float32)
            pts[:len(adv.BUCKET_POINTS)] = adv.BUCKET_POINTS
            ages = rng.integers(-2, 14, 320).astype(np.int32)
            gp, ga = ke_detect(kemu, img, pts, ages, bucket_size=bs, fpb=fpb, cap=1 << 16)
            corners = orc.fast_detect(img)
            allp, alla = np.vstack([pts, corners]), np.concatenate([ages, np.zeros(len(corners), np.int32)])
            op, oa = orc.bucketing_features(h, w, allp, alla, bs, fpb)
            assert len(op) > n // 4, (h, w, bs, fpb, len(op))
            assert np.array_equal(gp, op, equal_nan=True) and np.array_equal(ga, oa), (h, w, bs, fpb)
        for (h, w, bs, fpb) in ((64, 601, 6, 8), (256, 640, 3, 1), (64, 64, 1, 2), (96, 160, 9, 9)):
            assert not adv.bucket_grid_ok(w, h, bs, fpb)
            img = np.zeros((h, w), np.uint8)
            with pytest.raises(AssertionError):
                ke_detect(kemu, img, np.zeros((0, 2), np.float32), np.zeros(0, np.int32), bucket_size=bs, fpb=fpb)
    finally:
        kemu.ke_set_bucket_threads(256)


@pytest.mark.parametrize("which", ["n=4 duplicate", "n=4 NaN", "n=5 NaN", "n=5 collinear", "all zero"])   # (the whole table: -m gpu)
def test_adversarial_values_pose_chain(kemu, orc, which):
    """degenerate solvePnPRansac inputs through pnp.hip on the emulator: same status, inliers and control flow as the oracle;
    a pose that is NaN in the reference is NaN here"""
    import adversarial as adv
    from test_oracle_geom import planted_problem, K_KITTI
    X, uv, iters = adv.pnp_cases(orc, planted_problem, K_KITTI)[which]
    rc, rv, tv, inl, dbg = orc.solve_pnp_ransac(X, uv, K_KITTI, iterations=iters)
    grc, grv, gtv, ginl, gdbg = ke_pnp(kemu, X, uv, K_KITTI, iters=iters, split=0)   # (the one-lane solver: the emulated wide SVD takes minutes on NaN)
    assert grc == rc and np.array_equal(ginl, inl)
    assert adv.same(grv, rv, 1e-6) and adv.same(gtv, tv, 1e-6)
    if len(X) > 5:
        assert tuple(gdbg[:3]) == tuple(int(x) for x in dbg[:3])


def test_adversarial_values_essential_chain(kemu, orc):
    import adversarial as adv
    from test_gpu_parity import _em_scene
    cases, F, PP = adv.essential_cases(_em_scene)
    for name, (p0, p1) in cases.items():
        rc, E, Rg, tg, mask, dbg = ke_essential(kemu, p0, p1, F, PP)
        ok, Eo, mo, odbg = orc.find_essential_mat(p0, p1, F, PP)
        assert (rc == 1) == bool(ok), name
        if ok:
            go, Ro, to, m2 = orc.recover_pose(Eo, p0, p1, F, PP, mo)
            assert adv.same(E, Eo, 1e-9) and dbg[1] == go and np.array_equal(mask, m2), name
            assert adv.same(Rg, Ro, 1e-9) and adv.same(tg, to, 1e-9), name


def ke_essential(lib, p0, p1, focal, pp, prob=0.999, threshold=1.0, max_iters=1000):
    p0 = np.ascontiguousarray(p0, np.float32).reshape(-1, 2)
    p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
    n = len(p0)
    E, R, t = np.zeros((3, 3)), np.zeros((3, 3)), np.zeros(3)
    mask = np.zeros(max(n, 1), np.uint8)
    dbg = (C.c_int * 4)()
    lib.ke_essential.restype = C.c_int
    rc = lib.ke_essential(vp(p0), vp(p1), n, C.c_double(focal), C.c_double(pp[0]), C.c_double(pp[1]), C.c_double(prob),
                          C.c_double(threshold), max_iters, vp(E), vp(R), vp(t), vp(mask), dbg)
    return rc, E, R, t, mask[:n].copy(), list(dbg)


@pytest.mark.parametrize("n,outliers,seed", [(300, 0.0, 1), (400, 0.3, 2), (60, 0.5, 3), (5, 0.0, 5), (6, 0.0, 6)])
def test_emulated_essential_chain_matches_oracle(kemu, orc, n, outliers, seed):
    """essential.hip launch by launch on the CPU emulator (`mono_rotation`, SURVEY 8 row f4: five-point samples from the
    wavefront subset kernel, Sampson votes, control-flow replay over the 128-sample chunks, mask + cheirality + selection)
    against the oracle's findEssentialMat + recoverPose: E <= 1e-9, identical masks and cheirality counts, R / t <= 1e-9"""
    from test_gpu_parity import _em_scene
    p1, p2, R, t, F, PP = _em_scene(seed, n, outliers)
    rc, E, Rg, tg, mask, dbg = ke_essential(kemu, p1, p2, F, PP)
    ok, Eo, mo, odbg = orc.find_essential_mat(p1, p2, F, PP)
    assert (rc == 1) == bool(ok)
    if not ok:
        return
    assert np.abs(E - Eo).max() <= 1e-9
    go, Ro, to, m2 = orc.recover_pose(Eo, p1, p2, F, PP, mo)
    assert dbg[1] == go and np.array_equal(mask, m2)
    assert np.abs(Rg - Ro).max() <= 1e-9 and np.abs(tg - to).max() <= 1e-9
    if n >= 60:
        assert np.abs(Rg - R).max() < 5e-3  # and it is the planted rotation


def test_emulated_frame_loop_matches_the_oracle_chain(kemu, orc, small_world):
    """The reference's frame loop (main.cpp:123-224: appendNewFeatures when fewer than 2000 features are carried, bucketing,
    circularMatching, filters, the feature carry with its ages, triangulation, solvePnPRansac, Euler gates +
    integrateOdometryStereo) for one sequence, every device-side step run from the product's kernel sources on the CPU
    emulator in the order vo_seq_step launches them -- fast.hip, pyramid.hip, lk.hip, post.hip, seq.hip (carry), pnp.hip with
    the integration tail of vo_seqtail.h -- against the oracle's functions chained the same way: carried features and ages
    bit for bit after every frame, counts identical, frame_pose <= 1e-6."""
    kemu.ke_set_fast_big(0)
    n = 3
    L, R, poses, _ = small_world.render_sequence(n)
    P_l, P_r = small_world.proj_matrices()
    K = np.ascontiguousarray(small_world.K(), np.float32)
    h, w = L[0].shape
    fpb, max_steps = 1, 8
    feat, fages = np.zeros((0, 2), np.float32), np.zeros(0, np.int32)          # emulated currentVOFeatures
    pose = np.eye(4).ravel().copy()
    rows = np.zeros((max_steps, 27))
    n_rows = C.c_int(0)
    o_pts, o_ages = np.zeros((0, 2), np.float32), np.zeros(0, np.int32)        # the oracle's
    o_pose, o_t = np.eye(4), np.zeros(3)
    dp = C.POINTER(C.c_double)
    for k in range(1, n):
        imgs = [L[k - 1], R[k - 1], L[k], R[k]]
        # ---- emulated kernels
        bp, ba = ke_detect(kemu, imgs[0], feat, fages, detect=int(len(feat) < 2000), bucket_size=h // 10, fpb=fpb)
        r = ke_run(kemu, imgs, bp, full_chain=0)
        post = ke_post(kemu, bp, r["trk"], r["status"], P_l, P_r)
        cap = max(len(bp), 1)
        outB = np.zeros((4, cap, 2), np.float32)
        outB[:, :post["B"].shape[1]] = post["B"]
        idxA = np.zeros(cap, np.int32)
        idxA[:len(post["idxA"])] = post["idxA"]
        ages_b = np.zeros(cap, np.int32)
        ages_b[:len(ba)] = ba
        fcap = 4096
        feat_o, fages_o = np.zeros((fcap, 2), np.float32), np.zeros(fcap, np.int32)
        ntr, nag = C.c_int(0), C.c_int(0)
        kemu.ke_seq_carry(vp(outB), len(post["idxB"]), vp(idxA), len(post["idxA"]), vp(ages_b), len(bp), cap, fcap, vp(feat_o),
                          vp(fages_o), C.byref(ntr), C.byref(nag))
        feat, fages = feat_o[:ntr.value].copy(), fages_o[:nag.value].copy()
        info = (C.c_int * 8)()
        dbg = (C.c_int * 4)()
        xyz = np.ascontiguousarray(post["xyz"], np.float32)
        l1 = np.ascontiguousarray(post["B"][2], np.float32)
        kemu.ke_pnp_ransac_tail.restype = C.c_int
        # (the one-kernel EPnP: the four-kernel form gives the same bits -- test above -- and takes minutes to emulate)
        rc_e = kemu.ke_pnp_ransac_tail(vp(xyz), vp(l1), len(xyz), vp(K), 0, pose.ctypes.data_as(dp), rows.ctypes.data_as(dp),
                                       C.byref(n_rows), max_steps, info, dbg)
        # ---- the oracle's chain (tests/test_gpu_sequences.py holds the GPU loop to the same one)
        if len(o_pts) < 2000:
            fast = orc.fast_detect(imgs[0], 20, True)
            o_pts = np.vstack([o_pts, fast])
            o_ages = np.concatenate([o_ages, np.zeros(len(fast), np.int32)])
        obp, oba = orc.bucketing_features(h, w, o_pts, o_ages, h // 10, fpb)
        cm = orc.circular_matching(*imgs, obp, ages=oba)
        (pl0, pr0, pl1, pr1), _ = orc.check_valid_and_remove(cm["l0"], cm["r0"], cm["l1"], cm["r1"], cm["l0_ret"])
        o_pts, o_ages = pl1, cm["ages"]
        oxyz = orc.triangulate(P_l, P_r, pl0, pr0)
        rc, rv, tv, inl, odbg = orc.solve_pnp_ransac(oxyz, pl1, K, tvec=o_t)
        o_t = tv
        Rm = orc.rodrigues(rv)
        e = orc.rotation_matrix_to_euler(Rm)
        if abs(e[1]) < 0.1 and abs(e[0]) < 0.1 and abs(e[2]) < 0.1:
            o_pose, _ = orc.integrate_odometry_stereo(o_pose, Rm, tv)
        # ---- compare
        assert np.array_equal(bits(bp), bits(obp)) and np.array_equal(ba, oba), k
        assert np.array_equal(bits(feat), bits(o_pts)) and np.array_equal(fages, o_ages), k
        assert rc_e == rc and info[3] == len(inl) and info[6] == int(odbg[0]), k
        assert np.abs(pose.reshape(4, 4) - o_pose).max() <= 1e-6, k
        assert len(o_pts) > 100
    assert n_rows.value == n - 1
    assert np.abs(rows[n - 2, :12] - o_pose[:3].ravel()).max() <= 1e-6


@pytest.mark.parametrize("shape", [(1241, 376, 3), (640, 480, 3), (1920, 1080, 4), (1280, 720, 3), (4096, 3000, 4), (33, 32, 3), (45, 4000, 3),
                                   (16384, 16384, 4), (256, 60000, 2)])
def test_pyramid_pass_workgroup_decode(kemu, shape):
    """pyr_pass_kernel turns its one-dimensional workgroup id into (image, row, wavefront) with multiply-high divisions, in
    XCD-aware order (image z on XCD z % 8) or in dispatch order: ids of the largest launch the host allows
    (pass_images_per_launch: further images go to a second launch; up to 2^31 workgroups for the absurd shapes), and of small
    launches whose image count is not a multiple of 8, against plain division; the decode of a full launch is a bijection"""
    w, h, ml = shape
    for n in (0, 1, 7, 8, 9, 514):
        assert kemu.ke_pass_decode_check(w, h, ml, n) == 0, (shape, n)


@pytest.mark.parametrize("w,h,stride,n_waves", [(1241, 13, 1241, 7), (64, 9, 80, 3), (519, 5, 519, 64), (32, 4, 32, 1), (100, 6, 131, 1000)])
def test_emulated_seq_ingest_persistent_grid(kemu, w, h, stride, n_waves):
    """seq_ingest_kernel (round 6): G single-wave workgroups walk over the rows of all new images; a row's last 8 bytes come
    from an overlapping access -- every destination pixel right, nothing outside the w columns of a row touched, for widths
    that are / are not multiples of 8 and 512, padded strides, fewer and more waves than rows (and, under the sanitizer
    tier's exactly-sized buffers, no access outside a source row's image or a destination image)"""
    rng = np.random.default_rng(w + h)
    n_pairs, pitch = 3, ((w + 56 + 15) // 16) * 16
    src = [np.ascontiguousarray(rng.integers(0, 256, (h, stride), dtype=np.uint8)) for _ in range(2 * n_pairs)]
    views = [a[:, :w] for a in src]
    if stride > w:                      # the last row of a strided image ends at its w-th byte: nothing may be read behind it
        src = [np.ascontiguousarray(a.ravel()[:(h - 1) * stride + w]) for a in src]
    dst = np.full((2 * n_pairs, h, pitch), 0xAB, np.uint8)
    ptr = lambda arrs: (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    kemu.ke_seq_ingest(ptr(src[0::2]), ptr(src[1::2]), n_pairs, w, h, stride, pitch, vp(dst), n_waves)
    for i in range(n_pairs):
        assert np.array_equal(dst[2 * i, :, :w], views[2 * i]) and np.array_equal(dst[2 * i + 1, :, :w], views[2 * i + 1]), i
    assert (dst[:, :, w:] == 0xAB).all()
