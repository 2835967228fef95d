"""Parity tests proper (-m gpu): the HIP path, called through the C ABI (ctypes on libvo_hip.so),
against the CPU oracle on the same seeded inputs.

Stated bars: pyramids and LK (positions, status, survivor indices) BIT-EXACT; triangulation
<= 1e-5 relative (observed bit-exact); pose rvec <= 1e-6 rad and tvec <= 1e-6 m with identical
inlier sets and identical RANSAC control flow (observed ~1e-16)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import contextlib


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@contextlib.contextmanager
def full_chain(ctx):
    """raw per-hop tracks are only defined for every feature when all four hops run (the reference's
    four independent calcOpticalFlowPyrLK calls); the default retires a feature at its first rejected hop"""
    ctx.set_params(lk_full_chain=1)
    try:
        yield ctx
    finally:
        ctx.set_params(lk_full_chain=0)


def oracle_hops(orc, L0, R0, L1, R1, pts, **kw):
    p1, s1, _ = orc.calc_optical_flow_pyr_lk(L0, R0, pts, **kw)
    p2, s2, _ = orc.calc_optical_flow_pyr_lk(R0, R1, p1, **kw)
    p3, s3, _ = orc.calc_optical_flow_pyr_lk(R1, L1, p2, **kw)
    p4, s4, _ = orc.calc_optical_flow_pyr_lk(L1, L0, p3, **kw)
    return (p1, p2, p3, p4), np.stack([s1, s2, s3, s4])


def run_batch_single(ctx, volib, imgs, pts, P=None, stages=None):
    h, w = imgs[0].shape
    ctx.batch_configure(4, w, h, 1)
    for i, im in enumerate(imgs):
        ctx.batch_upload_image(i, im)
    ctx.batch_set_quads([[0, 1, 2, 3]])
    ctx.batch_set_points(0, pts)
    if P is not None:
        ctx.batch_set_projection(*P)
    ctx.batch_run(volib.STAGE_ALL if stages is None else stages)
    ctx.batch_sync()


# ------------------------------------------------------------------ pyramid
@pytest.mark.parametrize("shape", [(376, 1241), (1080, 1920), (160, 480), (97, 131), (64, 64)])
def test_pyramid_bit_exact(gpu_ctx, volib, orc, shape):
    rng = np.random.default_rng(shape[0])
    imgs = [rng.integers(0, 256, shape, dtype=np.uint8) for _ in range(4)]
    run_batch_single(gpu_ctx, volib, imgs, np.zeros((0, 2), np.float32), stages=volib.STAGE_PYRAMID)
    for i in (0, 3):
        ref = orc.build_pyramid(imgs[i], 3)
        lvl = 0
        while True:
            try:
                g = gpu_ctx.batch_get_pyramid_level(i, lvl)
            except volib.VoError:
                break
            assert np.array_equal(g, ref[lvl]), (shape, i, lvl)
            lvl += 1
        # buildOpticalFlowPyramid stops when the next level would be <= the 21x21 window
        expect = 1
        hh, ww = shape
        while expect < 4 and (ww + 1) // 2 > 21 and (hh + 1) // 2 > 21:
            ww, hh, expect = (ww + 1) // 2, (hh + 1) // 2, expect + 1
        assert lvl == expect


# ------------------------------------------------------------------ LK
def test_lk_bit_exact_small(gpu_ctx, volib, orc, small_seq):
    with full_chain(gpu_ctx):
        s = small_seq
        imgs = [s["L"][0], s["R"][0], s["L"][1], s["R"][1]]
        border = np.array([[0, 0], [479, 159], [2.5, 80.25], [476.2, 10.7], [240, 1.1], [250.4, 158.9],
                           [-5, 50], [100, -3], [520, 100], [12.5, 12.5], [-25, 80], [240, 185]], np.float32)
        pts = np.vstack([s["pts"][0], border]).astype(np.float32)
        run_batch_single(gpu_ctx, volib, imgs, pts, stages=volib.STAGE_PYRAMID | volib.STAGE_LK)
        g = gpu_ctx.batch_get_tracks(0, len(pts))
        (p1, p2, p3, p4), st = oracle_hops(orc, *imgs, pts)
        assert np.array_equal(g["status4"], st)
        for name, ref in (("r0", p1), ("r1", p2), ("l1", p3), ("l0_ret", p4)):
            assert np.array_equal(bits(g[name]), bits(ref)), name
        assert st[:, :len(s["pts"][0])].mean() > 0.5


def test_lk_bit_exact_kitti_2000(gpu_ctx, volib, orc, kitti_seq):
    with full_chain(gpu_ctx):
        s = kitti_seq
        imgs = [s["L"][0], s["R"][0], s["L"][1], s["R"][1]]
        pts = s["pts"]
        assert 1800 < len(pts) < 2300
        run_batch_single(gpu_ctx, volib, imgs, pts, stages=volib.STAGE_PYRAMID | volib.STAGE_LK)
        g = gpu_ctx.batch_get_tracks(0, len(pts))
        (p1, p2, p3, p4), st = oracle_hops(orc, *imgs, pts)
        assert np.array_equal(g["status4"], st)
        for name, ref in (("r0", p1), ("r1", p2), ("l1", p3), ("l0_ret", p4)):
            assert np.array_equal(bits(g[name]), bits(ref)), name


def test_lk_large_motion_tile_refetch(gpu_ctx, volib, orc):
    """big flow forces the search tile to be re-fetched mid-iteration; fractional start points"""
    with full_chain(gpu_ctx):
        from test_oracle_images import smooth_image
        w, h = 512, 256
        I = smooth_image(w, h, seed=9)
        imgs = [I, smooth_image(w, h, 13.7, -9.2, seed=9), smooth_image(w, h, 20.1, 4.4, seed=9),
                smooth_image(w, h, -6.3, 11.8, seed=9)]
        rng = np.random.default_rng(3)
        pts = np.stack([rng.uniform(-10, w + 10, 700), rng.uniform(-10, h + 10, 700)], 1).astype(np.float32)
        run_batch_single(gpu_ctx, volib, imgs, pts, stages=volib.STAGE_PYRAMID | volib.STAGE_LK)
        g = gpu_ctx.batch_get_tracks(0, len(pts))
        (p1, p2, p3, p4), st = oracle_hops(orc, *imgs, pts)
        assert np.array_equal(g["status4"], st)
        for name, ref in (("r0", p1), ("r1", p2), ("l1", p3), ("l0_ret", p4)):
            assert np.array_equal(bits(g[name]), bits(ref)), name
        assert st.all(0).sum() > 200


def test_lk_params_other_than_reference(gpu_ctx, volib, orc, small_seq):
    with full_chain(gpu_ctx):
        s = small_seq
        imgs = [s["L"][0], s["R"][0], s["L"][1], s["R"][1]]
        pts = s["pts"][0]
        gpu_ctx.set_params(lk_max_level=2, lk_max_count=7, lk_epsilon=0.03, lk_min_eig_threshold=0.01)
        try:
            run_batch_single(gpu_ctx, volib, imgs, pts, stages=volib.STAGE_PYRAMID | volib.STAGE_LK)
            g = gpu_ctx.batch_get_tracks(0, len(pts))
            (p1, p2, p3, p4), st = oracle_hops(orc, *imgs, pts, max_level=2, max_count=7, eps=0.03, min_eig=0.01)
            assert np.array_equal(g["status4"], st)
            assert np.array_equal(bits(g["l0_ret"]), bits(p4)) and np.array_equal(bits(g["r0"]), bits(p1))
        finally:
            gpu_ctx.set_params(lk_max_level=3, lk_max_count=30, lk_epsilon=0.01, lk_min_eig_threshold=0.001)


# ------------------------------------------------------------------ drop-in calls
def test_circular_match_dropin(gpu_ctx, orc, small_seq):
    s = small_seq
    args = (s["L"][0], s["R"][0], s["L"][1], s["R"][1])
    pts = np.vstack([s["pts"][0], [[-3, 20], [20, -2], [1000, 50]]]).astype(np.float32)
    ref = orc.circular_matching(*args, pts)
    got = gpu_ctx.circular_match(*args, pts)       # default: features retire at their first rejected hop
    assert got["n_out"] == ref["n_out"] > 20
    for k in ("l0", "r0", "r1", "l1", "l0_ret"):
        assert np.array_equal(bits(got[k]), bits(ref[k])), k
    assert np.array_equal(got["keep_idx"], ref["keep_idx"])
    survivors = np.zeros(len(pts), bool)
    survivors[ref["keep_idx"]] = True          # all four statuses 1 and no negative coordinate (feature.cpp:96-104)
    assert np.array_equal(got["status4"].all(0), survivors)
    with full_chain(gpu_ctx):                       # all four hops for every feature: raw statuses too
        got = gpu_ctx.circular_match(*args, pts)
        assert got["n_out"] == ref["n_out"] and np.array_equal(got["status4"], ref["status4"])
        for k in ("l0", "r0", "r1", "l1", "l0_ret"):
            assert np.array_equal(bits(got[k]), bits(ref[k])), k
    # + checkValidMatch / removeInvalidPoints (visualOdometry.cpp:119-125)
    got2 = gpu_ctx.circular_match(*args, pts, apply_consistency=True)
    (l0, r0, l1, r1), valid = orc.check_valid_and_remove(ref["l0"], ref["r0"], ref["l1"], ref["r1"], ref["l0_ret"])
    assert got2["n_out"] == len(l0)
    assert np.array_equal(got2["l0"], l0) and np.array_equal(got2["r0"], r0) and np.array_equal(got2["l1"], l1)
    assert np.array_equal(got2["l0_ret"], ref["l0_ret"][valid])
    assert np.array_equal(got2["keep_idx"], ref["keep_idx"][valid])


def test_circular_match_empty_and_single(gpu_ctx, orc, small_seq):
    s = small_seq
    args = (s["L"][0], s["R"][0], s["L"][1], s["R"][1])
    got = gpu_ctx.circular_match(*args, np.zeros((0, 2), np.float32))
    assert got["n_out"] == 0 and got["l0"].shape == (0, 2)
    one = s["pts"][0][:1]
    got = gpu_ctx.circular_match(*args, one)
    ref = orc.circular_matching(*args, one)
    assert got["n_out"] == ref["n_out"] and np.array_equal(bits(got["l0_ret"]), bits(ref["l0_ret"]))


def test_capacity_is_enforced(gpu_ctx, volib, small_seq):
    s = small_seq
    with pytest.raises(volib.VoError) as e:
        gpu_ctx.circular_match(s["L"][0], s["R"][0], s["L"][1], s["R"][1], np.zeros((9000, 2), np.float32))
    assert e.value.code == volib.VO_ERR_ARG


def test_triangulate_dropin(gpu_ctx, orc, kitti_world):
    P_l, P_r = kitti_world.proj_matrices()
    rng = np.random.default_rng(0)
    n = 4000
    pl = rng.uniform([0, 0], [1241, 376], (n, 2)).astype(np.float32)
    pr = pl.copy()
    pr[:, 0] -= rng.uniform(1.0, 120, n).astype(np.float32)
    pr[:, 1] += rng.normal(0, 0.3, n).astype(np.float32)
    pr[7] = pl[7]  # zero disparity: homogeneous w ~ 0 -> point at infinity
    got = gpu_ctx.triangulate(P_l, P_r, pl, pr)
    ref = orc.triangulate(P_l, P_r, pl, pr)
    fin = np.ones(n, bool)
    fin[7] = False
    rel = np.abs(got[fin] - ref[fin]).max(1) / np.abs(ref[fin]).max(1)
    assert rel.max() <= 1e-5                       # stated tolerance
    assert (got[fin] == ref[fin]).all(1).mean() > 0.99  # in practice bit-identical
    assert not np.isfinite(got[7]).all() or np.abs(got[7]).max() > 1e6  # degenerate point: far away on both
    assert not np.isfinite(ref[7]).all() or np.abs(ref[7]).max() > 1e6
    assert gpu_ctx.triangulate(P_l, P_r, pl[:0], pr[:0]).shape == (0, 3)


def pose_close(got_r, got_t, ref_r, ref_t):
    return np.abs(got_r - ref_r).max() <= 1e-6 and np.abs(got_t - ref_t).max() <= 1e-6


@pytest.mark.parametrize("n,outliers,noise,seed", [(400, 0.0, 0.0, 1), (1500, 0.3, 0.15, 2), (60, 0.5, 0.2, 3),
                                                   (3000, 0.1, 0.05, 4), (6, 0.0, 0.05, 5)])
def test_pnp_ransac_dropin(gpu_ctx, orc, n, outliers, noise, seed):
    from test_oracle_geom import planted_problem, K_KITTI
    X, uv, r, t, _ = planted_problem(orc, n, outliers, noise, seed)
    rc, rv, tv, inl, dbg = orc.solve_pnp_ransac(X, uv, K_KITTI)
    found, grv, gtv, gR, ginl = gpu_ctx.pnp_ransac(X, uv, K_KITTI)
    assert found == (rc == 1)
    assert np.array_equal(ginl, inl)
    assert pose_close(grv, gtv, rv, tv)
    assert np.allclose(gR, orc.rodrigues(grv), atol=1e-15)


def test_pnp_ransac_edge_cases(gpu_ctx, volib, orc):
    from test_oracle_geom import planted_problem, K_KITTI
    X, uv, r, t, _ = planted_problem(orc, 5, 0.0, 0.0, 8)
    rc, rv, tv, inl, _ = orc.solve_pnp_ransac(X, uv, K_KITTI)
    found, grv, gtv, _, ginl = gpu_ctx.pnp_ransac(X, uv, K_KITTI)  # n == 5: direct EPnP, all inliers
    assert found and np.array_equal(ginl, np.arange(5)) and pose_close(grv, gtv, rv, tv)
    with pytest.raises(volib.VoError) as e:                          # n < 5: OpenCV would CV_Assert
        gpu_ctx.pnp_ransac(X[:3], uv[:3], K_KITTI)
    assert e.value.code == volib.VO_ERR_TOO_FEW
    rng = np.random.default_rng(9)                                    # no consensus: returns "not found"
    Xr = rng.uniform([-10, -2, 4], [10, 2, 50], (60, 3)).astype(np.float32)
    uvr = rng.uniform([0, 0], [1241, 376], (60, 2)).astype(np.float32)
    rc, rv, tv, inl, _ = orc.solve_pnp_ransac(Xr, uvr, K_KITTI)
    found, grv, gtv, _, ginl = gpu_ctx.pnp_ransac(Xr, uvr, K_KITTI)
    assert rc == 0 and not found and len(ginl) == 0
    assert pose_close(grv, gtv, rv, tv)  # both hold the last evaluated hypothesis


# ------------------------------------------------------------------ fused path
def test_track_frame_full_path_kitti(gpu_ctx, orc, kitti_world, kitti_seq):
    s = kitti_seq
    P_l, P_r = kitti_world.proj_matrices()
    args = (s["L"][0], s["R"][0], s["L"][1], s["R"][1])
    got = gpu_ctx.track_frame(*args, s["pts"], P_l, P_r)
    ref = orc.circular_matching(*args, s["pts"])
    (l0, r0, l1, r1), valid = orc.check_valid_and_remove(ref["l0"], ref["r0"], ref["l1"], ref["r1"], ref["l0_ret"])
    assert np.array_equal(got["keep_idx_circ"], ref["keep_idx"])
    for name, a in (("l0", l0), ("r0", r0), ("l1", l1), ("r1", r1)):
        assert np.array_equal(bits(got[name]), bits(a)), name
    xyz = orc.triangulate(P_l, P_r, l0, r0)
    assert np.max(np.abs(got["xyz"] - xyz) / np.abs(xyz).max(1, keepdims=True)) <= 1e-5
    rc, rv, tv, inl, _ = orc.solve_pnp_ransac(xyz, l1, kitti_world.K())
    assert rc == 1 and got["rc"] == 0
    assert np.array_equal(got["inliers"], inl) and pose_close(got["rvec"], got["tvec"], rv, tv)
    # and the answer is right: close to the planted camera motion
    from visual_odom_amd import synth
    Rg, tg = synth.relative_pose(s["poses"][0], s["poses"][1])
    assert np.abs(got["tvec"] - tg).max() < 0.02 and np.abs(got["rvec"] - orc.rodrigues(Rg)).max() < 2e-3


def test_sequence_replay_matches_oracle(gpu_ctx, orc, small_world, small_seq):
    """two consecutive frames, tracked points of frame k feed frame k+1 (visualOdometry.cpp:127)"""
    s = small_seq
    P_l, P_r = small_world.proj_matrices()
    pts_g = pts_o = s["pts"][0]
    for k in range(2):
        args = (s["L"][k], s["R"][k], s["L"][k + 1], s["R"][k + 1])
        got = gpu_ctx.track_frame(*args, pts_g, P_l, P_r)
        ref = orc.circular_matching(*args, pts_o)
        (l0, r0, l1, r1), _ = orc.check_valid_and_remove(ref["l0"], ref["r0"], ref["l1"], ref["r1"], ref["l0_ret"])
        assert np.array_equal(bits(got["l1"]), bits(l1))
        xyz = orc.triangulate(P_l, P_r, l0, r0)
        rc, rv, tv, inl, _ = orc.solve_pnp_ransac(xyz, l1, small_world.K())
        assert pose_close(got["rvec"], got["tvec"], rv, tv) and np.array_equal(got["inliers"], inl)
        pts_g, pts_o = got["l1"], l1


# ------------------------------------------------------------------ batched API + size-independent properties
def test_batch_ragged_equals_single(gpu_ctx, volib, orc, small_world, small_seq):
    s = small_seq
    P_l, P_r = small_world.proj_matrices()
    h, w = s["L"][0].shape
    gpu_ctx.batch_configure(6, w, h, 4)
    for k in range(3):
        gpu_ctx.batch_upload_image(2 * k, s["L"][k])
        gpu_ctx.batch_upload_image(2 * k + 1, s["R"][k])
    gpu_ctx.batch_set_quads([[0, 1, 2, 3], [2, 3, 4, 5], [0, 1, 2, 3], [2, 3, 0, 1]])
    sets = [s["pts"][0], s["pts"][1], s["pts"][0][:7], np.zeros((0, 2), np.float32)]  # ragged, one empty
    for f, p in enumerate(sets):
        gpu_ctx.batch_set_points(f, p)
    gpu_ctx.batch_set_projection(P_l, P_r)
    gpu_ctx.batch_run(volib.STAGE_ALL)
    gpu_ctx.batch_sync()
    quads = [(0, 1), (1, 2), (0, 1)]
    for f in range(3):
        a, b = quads[f]
        args = (s["L"][a], s["R"][a], s["L"][b], s["R"][b])
        ref = orc.circular_matching(*args, sets[f])
        (l0, r0, l1, r1), _ = orc.check_valid_and_remove(ref["l0"], ref["r0"], ref["l1"], ref["r1"], ref["l0_ret"])
        got = gpu_ctx.batch_get_filtered(f)
        assert np.array_equal(got["keep_idx_circ"], ref["keep_idx"])
        assert np.array_equal(bits(got["l1"]), bits(l1)) and np.array_equal(bits(got["r1"]), bits(r1))
        pose = gpu_ctx.batch_get_pose(f)
        if len(l0) >= 4:  # 4 survivors (the 7-point set): OpenCV's P3P switch, here without a solution -> status 0
            xyz = orc.triangulate(P_l, P_r, l0, r0)
            rc, rv, tv, inl, dbg = orc.solve_pnp_ransac(xyz, l1, small_world.K())
            assert pose["status"] == rc and np.array_equal(pose["inliers"], inl)
            assert pose_close(pose["rvec"], pose["tvec"], rv, tv)
            assert (pose["niters"], pose["best_iter"], pose["max_good"]) == tuple(int(x) for x in dbg[:3])
        else:
            assert pose["status"] < 0
    assert len(gpu_ctx.batch_get_filtered(3)["l0"]) == 0 and gpu_ctx.batch_get_pose(3)["status"] < 0


def test_full_size_properties_1080p_4000(gpu_ctx, volib, orc):
    """BASELINE config 4 shape (1920x1080, 4000 points, HBM stress): the oracle is too slow for all of
    it, so (i) a 150-point subset is checked bit-exactly (features are independent, so the subset's
    result inside the full launch must equal the oracle's), (ii) determinism: two runs are
    bit-identical, (iii) permutation equivariance, (iv) circular closure of static scenes."""
    with full_chain(gpu_ctx):
        from visual_odom_amd import synth
        w, h = 1920, 1080
        world = synth.StereoWorld(seed=5, width=w, height=h, fx=1112.0, cx=959.5, cy=539.5, bf=-597.0, tex_size=1024)
        L, R, poses, _ = world.render_sequence(2)
        pts = synth.select_keypoints(L[0], bucket=108, per_bucket=60, min_dist=3)[:4000]
        assert len(pts) == 4000
        imgs = [L[0], R[0], L[1], R[1]]
        run_batch_single(gpu_ctx, volib, imgs, pts, stages=volib.STAGE_PYRAMID | volib.STAGE_LK)
        g1 = gpu_ctx.batch_get_tracks(0, len(pts))
        run_batch_single(gpu_ctx, volib, imgs, pts, stages=volib.STAGE_PYRAMID | volib.STAGE_LK)
        g2 = gpu_ctx.batch_get_tracks(0, len(pts))
        for k in ("r0", "r1", "l1", "l0_ret", "status4"):
            assert np.array_equal(g1[k], g2[k])                                  # (ii)
        sub = np.arange(0, len(pts), len(pts) // 150)
        (p1, p2, p3, p4), st = oracle_hops(orc, *imgs, pts[sub])
        assert np.array_equal(g1["status4"][:, sub], st)
        assert np.array_equal(bits(g1["l0_ret"][sub]), bits(p4)) and np.array_equal(bits(g1["r1"][sub]), bits(p2))  # (i)
        perm = np.random.default_rng(0).permutation(len(pts))
        run_batch_single(gpu_ctx, volib, imgs, pts[perm], stages=volib.STAGE_PYRAMID | volib.STAGE_LK)
        g3 = gpu_ctx.batch_get_tracks(0, len(pts))
        assert np.array_equal(bits(g3["l0_ret"]), bits(g1["l0_ret"][perm]))      # (iii)
        # (iv) static scene (same stereo pair at t0 and t1): the circle closes for nearly every tracked point
        run_batch_single(gpu_ctx, volib, [L[0], R[0], L[0], R[0]], pts, stages=volib.STAGE_PYRAMID | volib.STAGE_LK)
        g4 = gpu_ctx.batch_get_tracks(0, len(pts))
        ok = g4["status4"].all(0)
        assert ok.mean() > 0.6
        assert np.median(np.abs(g4["l0_ret"][ok] - pts[ok]).max(1)) < 0.1


# ------------------------------------------------------------------ row f1: FAST + bucketing on the device
def test_fast_detect_dropin(gpu_ctx, orc, kitti_seq, small_seq):
    for img in (kitti_seq["L"][0], small_seq["L"][1]):
        for thr, nonmax in ((20, True), (40, False)):
            ref = orc.fast_detect(img, thr, nonmax)
            got = gpu_ctx.fast_detect(img, thr, nonmax)
            assert len(ref) > 100 and np.array_equal(got, ref), (img.shape, thr, nonmax)
    rng = np.random.default_rng(2)
    noise = rng.integers(0, 256, (64, 97), dtype=np.uint8)
    assert np.array_equal(gpu_ctx.fast_detect(noise), orc.fast_detect(noise, 20, True))


def test_detect_bucket_dropin_with_quirks(gpu_ctx, orc, kitti_seq):
    """head of matchingFeatures (visualOdometry.cpp:95-108) incl. aliasing / duplicate emission /
    slot-0 overwrite / age >= 10 / ages longer than points"""
    img = kitti_seq["L"][0]
    h, w = img.shape
    rng = np.random.default_rng(6)
    tracked = np.stack([rng.uniform(0, w - 1, 500), rng.uniform(0, h - 1, 500)], 1).astype(np.float32)
    tracked[:3] = [[w - 1, 5.5], [w - 0.25, h - 1], [1240.5, 200]]
    ages = rng.integers(0, 13, 540).astype(np.int32)
    fast = orc.fast_detect(img, 20, True)
    comb_p = np.vstack([tracked, fast])
    comb_a = np.concatenate([ages, np.zeros(len(comb_p) - len(ages), np.int32)])
    for fpb in (1, 6):
        ref_p, ref_a = orc.bucketing_features(h, w, comb_p, comb_a, h // 10, fpb)
        got_p, got_a = gpu_ctx.detect_bucket(img, tracked, ages, features_per_bucket=fpb)
        assert np.array_equal(got_p, ref_p) and np.array_equal(got_a, ref_a), fpb
    # a set of >= redetect_below points is bucketed without re-detection
    ref_p, ref_a = orc.bucketing_features(h, w, tracked, ages[:500], h // 10, 1)
    got_p, got_a = gpu_ctx.detect_bucket(img, tracked, ages[:500], redetect_below=400)
    assert np.array_equal(got_p, ref_p) and np.array_equal(got_a, ref_a)
    # empty carried set (first frame of a sequence, main.cpp:110-121)
    ref_p, ref_a = orc.bucketing_features(h, w, fast, np.zeros(len(fast), np.int32), h // 10, 1)
    got_p, got_a = gpu_ctx.detect_bucket(img, np.zeros((0, 2), np.float32), np.zeros(0, np.int32))
    assert np.array_equal(got_p, ref_p) and np.array_equal(got_a, ref_a) and 250 < len(got_p) <= 374


def test_batch_detect_stage_feeds_lk(gpu_ctx, volib, orc, small_world, small_seq):
    """VO_STAGE_DETECT leaves the bucketed set as the LK input on the device: the rest of the path must
    equal the path run on the oracle's bucketed points"""
    s = small_seq
    P_l, P_r = small_world.proj_matrices()
    h, w = s["L"][0].shape
    gpu_ctx.batch_configure(6, w, h, 2)
    for k in range(3):
        gpu_ctx.batch_upload_image(2 * k, s["L"][k])
        gpu_ctx.batch_upload_image(2 * k + 1, s["R"][k])
    gpu_ctx.batch_set_quads([[0, 1, 2, 3], [2, 3, 4, 5]])
    carried = [(s["pts"][0][:40], np.arange(40, dtype=np.int32) % 12), (np.zeros((0, 2), np.float32), np.zeros(0, np.int32))]
    for f, (p, a) in enumerate(carried):
        gpu_ctx.batch_set_features(f, p, a)
    gpu_ctx.batch_set_detect_params(features_per_bucket=2)
    gpu_ctx.batch_set_projection(P_l, P_r)
    try:
        gpu_ctx.batch_run(volib.STAGE_ALL | volib.STAGE_DETECT)
        gpu_ctx.batch_sync()
        for f, (p, a) in enumerate(carried):
            L0, R0, L1, R1 = s["L"][f], s["R"][f], s["L"][f + 1], s["R"][f + 1]
            fast = orc.fast_detect(L0, 20, True)
            comb_p = np.vstack([p, fast])
            comb_a = np.concatenate([a, np.zeros(len(fast), np.int32)])
            ref_p, ref_a = orc.bucketing_features(h, w, comb_p, comb_a, h // 10, 2)
            got_p, got_a = gpu_ctx.batch_get_features(f)
            assert np.array_equal(got_p, ref_p) and np.array_equal(got_a, ref_a)
            ref = orc.circular_matching(L0, R0, L1, R1, ref_p)
            (l0, r0, l1, r1), _ = orc.check_valid_and_remove(ref["l0"], ref["r0"], ref["l1"], ref["r1"], ref["l0_ret"])
            got = gpu_ctx.batch_get_filtered(f)
            assert np.array_equal(got["keep_idx_circ"], ref["keep_idx"])
            assert np.array_equal(bits(got["l1"]), bits(l1)) and np.array_equal(bits(got["r0"]), bits(r0))
            if len(l0) >= 4:
                xyz = orc.triangulate(P_l, P_r, l0, r0)
                rc, rv, tv, inl, _ = orc.solve_pnp_ransac(xyz, l1, small_world.K())
                pose = gpu_ctx.batch_get_pose(f)
                assert pose["status"] == rc and pose_close(pose["rvec"], pose["tvec"], rv, tv)
    finally:
        gpu_ctx.batch_set_detect_params()


# ------------------------------------------------------------------ row f3: the reference's frame loop
def test_sequence_trajectory_matches_oracle_and_ground_truth(gpu_ctx, orc, small_world):
    """main.cpp:123-224 replayed through the product (StereoOdometry: detect/bucket -> track_frame ->
    integrate) and through the checker's functions chained the same way; per-frame R|t <= 1e-6, identical
    feature state, trajectory (ATE) <= 1e-6 m vs the checker and close to the planted camera path"""
    from visual_odom_amd import odometry, synth, _lib
    n = 7
    L, R, poses, _ = small_world.render_sequence(n)
    P_l, P_r = small_world.proj_matrices()
    K = small_world.K()
    h, w = L[0].shape
    vo = odometry.StereoOdometry(P_l, P_r, ctx=gpu_ctx, streaming=True, features_per_bucket=3)    # the batch-API ring
    vo_dropin = odometry.StereoOdometry(P_l, P_r, ctx=gpu_ctx, streaming=False, keep_pair=False, features_per_bucket=3)
    o_pts, o_ages = np.zeros((0, 2), np.float32), np.zeros(0, np.int32)
    o_pose, o_t = np.eye(4), np.zeros(3)
    o_traj = [o_pose[:3].copy()]
    vo.process(L[0], R[0])
    for k in range(1, n):
        rec = vo.process(L[k], R[k])
        # ---- checker chain
        l0, r0, l1, r1 = L[k - 1], R[k - 1], L[k], R[k]
        if len(o_pts) < 2000:
            fast = orc.fast_detect(l0, 20, True)
            o_pts = np.vstack([o_pts, fast])
            o_ages = np.concatenate([o_ages, np.zeros(len(fast), np.int32)])
        bp, ba = orc.bucketing_features(h, w, o_pts, o_ages, h // 10, 3)
        cm = orc.circular_matching(l0, r0, l1, r1, bp, ages=ba)
        (pl0, pr0, pl1, pr1), _ = orc.check_valid_and_remove(cm["l0"], cm["r0"], cm["l1"], cm["r1"], cm["l0_ret"])
        o_pts, o_ages = pl1, cm["ages"]
        xyz = orc.triangulate(P_l, P_r, pl0, pr0)
        rc, rv, tv, inl, _ = orc.solve_pnp_ransac(xyz, pl1, K, tvec=o_t)
        o_t = tv
        Rm = orc.rodrigues(rv)
        e = orc.rotation_matrix_to_euler(Rm)
        if abs(e[1]) < 0.1 and abs(e[0]) < 0.1 and abs(e[2]) < 0.1:
            o_pose, _ = orc.integrate_odometry_stereo(o_pose, Rm, tv)
        o_traj.append(o_pose[:3].copy())
        # ---- per-frame parity
        assert rec["n_bucketed"] == len(bp) and rec["n_tracked"] == len(pl1) and rec["n_inliers"] == len(inl)
        assert np.array_equal(bits(vo.points), bits(o_pts)) and np.array_equal(vo.ages, o_ages)
        assert np.abs(rec["rvec"] - rv).max() <= 1e-6 and np.abs(rec["tvec"] - tv).max() <= 1e-6
        assert np.abs(vo.frame_pose - o_pose).max() <= 1e-6
    assert odometry.ate_rmse(vo.trajectory, o_traj) <= 1e-6
    # the stateless drop-in calls (four uploads + four pyramids per frame) give the same trajectory as the
    # streaming ring (one upload, two pyramids per frame)
    for k in range(n):
        vo_dropin.process(L[k], R[k])
    assert np.array_equal(np.asarray(vo_dropin.trajectory), np.asarray(vo.trajectory))
    assert np.array_equal(bits(vo_dropin.points), bits(vo.points)) and np.array_equal(vo_dropin.ages, vo.ages)
    # against the planted motion: camera-to-world poses relative to the first frame
    T0inv = np.linalg.inv(poses[0])
    gt = [(T0inv @ T)[:3] for T in poses]
    assert odometry.ate_rmse(vo.trajectory, gt) < 0.05
    assert sum(r["integrated"] for r in vo.log) == n - 1


def test_cpp_frame_loop_equals_python_mirror(gpu_ctx, vo_run_binary, small_world, tmp_path):
    """examples/vo_run.cpp (C++ host, C ABI) and visual_odom_amd.odometry.StereoOdometry (ctypes) replay the
    same sequence: the KITTI-format trajectories must agree to the printed precision"""
    import subprocess
    from visual_odom_amd import odometry
    n = 6
    L, R, poses, _ = small_world.render_sequence(n)
    P_l, P_r = small_world.proj_matrices()
    d = tmp_path / "seq"
    for cam, imgs in ((0, L), (1, R)):
        (d / ("image_%d" % cam)).mkdir(parents=True)
        for k, img in enumerate(imgs):
            h, w = img.shape
            with open(d / ("image_%d" % cam) / ("%06d.pgm" % k), "wb") as f:
                f.write(b"P5\n%d %d\n255\n" % (w, h) + np.ascontiguousarray(img).tobytes())
    out = tmp_path / "poses.txt"
    fx, cx, cy, bf = P_l[0, 0], P_l[0, 2], P_l[1, 2], P_r[0, 3]
    r = subprocess.run([vo_run_binary, str(d), repr(float(fx)), repr(float(cx)), repr(float(cy)), repr(float(bf)), str(n),
                        str(out), "3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    vo = odometry.StereoOdometry(P_l, P_r, ctx=gpu_ctx, features_per_bucket=3)
    for k in range(n):
        vo.process(L[k], R[k])
    got = odometry.load_poses(str(out))
    assert got.shape == (n, 3, 4)
    assert np.abs(got - np.asarray(vo.trajectory)).max() < 1e-8
    T0inv = np.linalg.inv(poses[0])
    assert odometry.ate_rmse(got, [(T0inv @ T)[:3] for T in poses]) < 0.05


# ------------------------------------------------------------------ f4: mono_rotation (essential matrix + recoverPose)
def _em_scene(seed, n, outliers):
    from test_essential_oracle import _scene, rng_rv, rng_t, F, PP
    R, t, x1, x2, Et, rng = _scene(seed, n=n, rv=rng_rv(seed), t=rng_t(seed))
    p1 = (x1 * F + PP).astype(np.float32)
    p2 = (x2 * F + PP).astype(np.float32)
    k = int(outliers * n)
    p2[:k] += rng.uniform(-40, 40, (k, 2)).astype(np.float32)
    p2 += rng.normal(0, 0.1, p2.shape).astype(np.float32)
    return p1, p2, R, t, F, PP


@pytest.mark.gpu
@pytest.mark.parametrize("n,outliers,seed", [(300, 0.0, 1), (2000, 0.3, 2), (60, 0.5, 3), (900, 0.7, 4), (5, 0.0, 5),
                                             (6, 0.0, 6)])
def test_essential_pose_dropin(gpu_ctx, orc, n, outliers, seed):
    p1, p2, R, t, F, PP = _em_scene(seed, n, outliers)
    found, E, Rg, tg, mask, good = gpu_ctx.essential_pose(p1, p2, F, PP)
    ok, Eo, mo, dbg = orc.find_essential_mat(p1, p2, F, PP)
    assert found == bool(ok)
    if not ok:
        return
    assert np.abs(E - Eo).max() <= 1e-9
    go, Ro, to, m2 = orc.recover_pose(Eo, p1, p2, F, PP, mo)
    assert good == go and np.array_equal(mask, m2)
    assert np.abs(Rg - Ro).max() <= 1e-9 and np.abs(tg - to).max() <= 1e-9
    if n >= 60 and outliers <= 0.5:
        assert np.abs(Rg - R).max() < 5e-3  # and it is the planted rotation


@pytest.mark.gpu
def test_essential_pose_edge_cases(gpu_ctx, volib, orc):
    p1, p2, R, t, F, PP = _em_scene(9, 40, 0.0)
    with pytest.raises(volib.VoError) as e:  # findEssentialMat needs 5 points (OpenCV returns an empty E, recoverPose throws)
        gpu_ctx.essential_pose(p1[:4], p2[:4], F, PP)
    assert e.value.code == volib.VO_ERR_TOO_FEW
    # pure noise: whatever RANSAC does, the device does the same
    rng = np.random.default_rng(5)
    a = rng.uniform(0, 1000, (80, 2)).astype(np.float32)
    b = rng.uniform(0, 1000, (80, 2)).astype(np.float32)
    found, E, Rg, tg, mask, good = gpu_ctx.essential_pose(a, b, F, PP)
    ok, Eo, mo, dbg = orc.find_essential_mat(a, b, F, PP)
    assert found == bool(ok)
    if ok:
        go, Ro, to, m2 = orc.recover_pose(Eo, a, b, F, PP, mo)
        assert np.abs(E - Eo).max() <= 1e-9 and good == go and np.array_equal(mask, m2)


@pytest.mark.gpu
def test_track_frame_mono_rotation(gpu_ctx, orc, kitti_world, kitti_seq):
    """trackingFrame2Frame(..., mono_rotation = true): rotation from recoverPose, translation from PnP"""
    s = kitti_seq
    P_l, P_r = kitti_world.proj_matrices()
    args = (s["L"][0], s["R"][0], s["L"][1], s["R"][1])
    base = gpu_ctx.track_frame(*args, s["pts"], P_l, P_r)
    gpu_ctx.set_params(mono_rotation=1)
    try:
        got = gpu_ctx.track_frame(*args, s["pts"], P_l, P_r)
        em = gpu_ctx.batch_get_essential(0, len(got["l0"]))
    finally:
        gpu_ctx.set_params(mono_rotation=0)
    # the PnP side is unchanged
    assert np.array_equal(got["rvec"], base["rvec"]) and np.array_equal(got["tvec"], base["tvec"])
    focal, pp = float(P_l[0, 0]), (float(P_l[0, 2]), float(P_l[1, 2]))
    ok, Eo, mo, dbg = orc.find_essential_mat(got["l0"], got["l1"], focal, pp)
    assert ok == 1 and em["status"] == 1
    go, Ro, to, m2 = orc.recover_pose(Eo, got["l0"], got["l1"], focal, pp, mo)
    assert np.abs(em["E"] - Eo).max() <= 1e-9 and em["n_good"] == go and np.array_equal(em["mask"], m2)
    assert np.abs(got["R"] - Ro).max() <= 1e-9 and em["niters"] == int(dbg[0])
    # and recoverPose's rotation agrees with the PnP rotation of the same motion
    assert np.abs(got["R"] - base["R"]).max() < 5e-3


@pytest.mark.gpu
def test_sequence_trajectory_mono_rotation(gpu_ctx, orc, small_world):
    """the frame loop with trackingFrame2Frame(..., mono_rotation = true): per-frame rotation = recoverPose's
    (checker chain on the same tracks), translation = PnP's, and the integrated trajectory still follows
    the planted path"""
    from visual_odom_amd import odometry
    n = 5
    L, R, poses, _ = small_world.render_sequence(n)
    P_l, P_r = small_world.proj_matrices()
    K = small_world.K()
    h, w = L[0].shape
    focal, pp = float(P_l[0, 0]), (float(P_l[0, 2]), float(P_l[1, 2]))
    vo = odometry.StereoOdometry(P_l, P_r, ctx=gpu_ctx, mono_rotation=True, features_per_bucket=3)
    try:
        o_pts, o_ages = np.zeros((0, 2), np.float32), np.zeros(0, np.int32)
        o_pose, o_t = np.eye(4), np.zeros(3)
        vo.process(L[0], R[0])
        for k in range(1, n):
            rec = vo.process(L[k], R[k])
            l0, r0, l1, r1 = L[k - 1], R[k - 1], L[k], R[k]
            if len(o_pts) < 2000:
                fast = orc.fast_detect(l0, 20, True)
                o_pts = np.vstack([o_pts, fast])
                o_ages = np.concatenate([o_ages, np.zeros(len(fast), np.int32)])
            bp, ba = orc.bucketing_features(h, w, o_pts, o_ages, h // 10, 3)
            cm = orc.circular_matching(l0, r0, l1, r1, bp, ages=ba)
            (pl0, pr0, pl1, pr1), _ = orc.check_valid_and_remove(cm["l0"], cm["r0"], cm["l1"], cm["r1"], cm["l0_ret"])
            o_pts, o_ages = pl1, cm["ages"]
            ok, E, mask, _ = orc.find_essential_mat(pl0, pl1, focal, pp)
            assert ok == 1
            _, Rm, _, _ = orc.recover_pose(E, pl0, pl1, focal, pp, mask)
            xyz = orc.triangulate(P_l, P_r, pl0, pr0)
            rc, rv, tv, inl, _ = orc.solve_pnp_ransac(xyz, pl1, K, tvec=o_t)
            o_t = tv
            e = orc.rotation_matrix_to_euler(Rm)
            if abs(e[1]) < 0.1 and abs(e[0]) < 0.1 and abs(e[2]) < 0.1:
                o_pose, _ = orc.integrate_odometry_stereo(o_pose, Rm, tv)
            assert np.abs(vo.rotation - Rm).max() <= 1e-9
            assert np.abs(rec["tvec"] - tv).max() <= 1e-6 and np.abs(vo.frame_pose - o_pose).max() <= 1e-6
    finally:
        gpu_ctx.set_params(mono_rotation=0)
    T0inv = np.linalg.inv(poses[0])
    gt = [(T0inv @ T)[:3] for T in poses]
    assert odometry.ate_rmse(vo.trajectory, gt) < 0.05


@pytest.mark.gpu
def test_exact_wave_sums_on_the_gpu(tmp_path):
    """the wave reductions LK is built on (v_permlane32_swap / v_permlane16_swap + DPP, hi/lo split), on the hardware,
    with per-lane partials at the documented bound: must equal (float)(int64 sum) bit for bit"""
    import subprocess
    src = os.path.join(ROOT, "tests", "host_check", "wave_sums_gpu.hip")
    exe = str(tmp_path / "wave_sums_gpu")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-w", "-o", exe, src])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mono", [False, True])
def test_product_frame_loop_against_the_reference_sources(gpu_ctx, orc, small_world, mono):
    """the product's frame loop (StereoOdometry over libvo_hip) against the body of the reference's main() loop run
    through the reference's OWN sources (oracle/_ref: matchingFeatures, trackingFrame2Frame, rotationMatrixToEulerAngles,
    integrateOdometryStereo compiled where they lie) over the oracle's OpenCV-algorithm restatement"""
    from visual_odom_amd import odometry
    if orc.ref_lib() is None:
        pytest.skip("oracle/_ref was not shipped")
    n = 6
    L, R, poses, _ = small_world.render_sequence(n)
    P_l, P_r = small_world.proj_matrices()
    loop = orc.RefFrameLoop(P_l[0, 0], P_l[0, 2], P_l[1, 2], P_r[0, 3], mono_rotation=mono)
    vo = odometry.StereoOdometry(P_l, P_r, ctx=gpu_ctx, mono_rotation=mono)   # reference defaults: 1 feature per bucket
    try:
        loop.process(L[0], R[0])
        vo.process(L[0], R[0])
        for k in range(1, n):
            a = loop.process(L[k], R[k])
            rec = vo.process(L[k], R[k])
            assert np.array_equal(bits(vo.points), bits(loop.points)) and np.array_equal(vo.ages, loop.ages), k
            assert rec["n_tracked"] == len(a["l1"])
            assert np.abs(rec["tvec"] - a["tvec"]).max() <= 1e-6 and np.abs(vo.rotation - a["R"]).max() <= 1e-6
            assert rec["integrated"] == a["integrated"]
            assert np.abs(vo.frame_pose - loop.frame_pose).max() <= 1e-6
    finally:
        gpu_ctx.set_params(mono_rotation=0)
    assert odometry.ate_rmse(vo.trajectory, loop.trajectory) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("mono", [False, True])
@pytest.mark.parametrize("entry", ["ref_frame_step", "ref_frame_step_adapter", "ref_frame_step_adapter+keep_pair"])
def test_reference_sources_run_on_libvo_hip(orc, small_world, mono, entry):
    """THE DROP-IN: the reference's unmodified matchingFeatures() / trackingFrame2Frame() / integrateOdometryStereo()
    (compiled where they lie; visualOdometry.cpp's circularMatching call reaches the SHIPPED adapter adapters/feature_hip.cpp,
    the OpenCV entry points reach the C ABI: tests/ref_dropin) driving the MI355X, against the same reference code over
    the CPU oracle (oracle/_ref), frame after frame.  ref_frame_step_adapter: main.cpp:169-171,181 edited as INTEGRATION.md
    says -- the adapter's triangulate_hip / trackingFrame2Frame_hip instead of the OpenCV calls."""
    import ctypes
    so = os.path.join(ROOT, "tests", "_build", "libvo_ref_dropin.so")
    if orc.ref_lib() is None or not os.path.exists(so):
        pytest.skip("built only where /root/reference exists (make -C tests/ref_dropin) and shipped with the snapshot")
    hip = ctypes.CDLL(so)
    keep = entry.endswith("+keep_pair")   # the adapter's opt-in: circularMatching_hip names the kept pair instead of its t0 images
    entry = entry.split("+")[0]
    hip.adapter_kept_calls.restype = ctypes.c_long
    hip.adapter_keep_pair(1 if keep else 0)
    kept0 = hip.adapter_kept_calls()
    n = 6
    L, R, poses, _ = small_world.render_sequence(n)
    P_l, P_r = small_world.proj_matrices()
    args = (P_l[0, 0], P_l[0, 2], P_l[1, 2], P_r[0, 3])
    cpu = orc.RefFrameLoop(*args, mono_rotation=mono)
    gpu = orc.RefFrameLoop(*args, mono_rotation=mono, lib=hip, entry=entry)
    cpu.process(L[0], R[0])
    gpu.process(L[0], R[0])
    for k in range(1, n):
        a, b = cpu.process(L[k], R[k]), gpu.process(L[k], R[k])
        for name in ("l0", "r0", "l1", "r1"):
            assert np.array_equal(bits(a[name]), bits(b[name])), (k, name)
        assert np.array_equal(bits(cpu.points), bits(gpu.points)) and np.array_equal(cpu.ages, gpu.ages)
        assert np.abs(a["tvec"] - b["tvec"]).max() <= 1e-6 and np.abs(a["R"] - b["R"]).max() <= 1e-6
        assert a["integrated"] == b["integrated"] and a["integrated"]
        assert np.abs(cpu.frame_pose - gpu.frame_pose).max() <= 1e-6
    hip.adapter_keep_pair(0)
    assert hip.adapter_kept_calls() - kept0 == (n - 2 if keep else 0)   # every call but the first went without its t0 pair
    T0inv = np.linalg.inv(poses[0])
    gt = [(T0inv @ T)[:3] for T in poses]
    from visual_odom_amd import odometry
    assert odometry.ate_rmse(gpu.trajectory, gt) < 0.05
