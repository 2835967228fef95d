"""Round 6 (-m gpu), VERDICT r05 item 1: VALUE-level parity at the ABI.  Every drop-in call of include/vo_hip.h gets the
adversarial inputs of tests/adversarial.py -- non-finite / huge / denormal / negative coordinates, degenerate point sets,
degenerate images -- and is held to the oracle with the usual bars (status, survivor indices, tracks bit-exact;
triangulation <= 1e-5 relative; pose <= 1e-6 with identical inlier sets); a NaN in the reference is a NaN here.  Then a seeded
hypothesis fuzz of 300 random (w, h, stride, n, parameters) cases through vo_track_frame.
Reference: feature.cpp:96-104,136-139, visualOdometry.cpp:44-61,152-153,176-178, main.cpp:169-171."""
import os

import numpy as np
import pytest

import adversarial as adv
from test_gpu_parity import bits, full_chain, oracle_hops, run_batch_single

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

P_L = np.array([[718.856, 0, 607.1928, 0], [0, 718.856, 185.2157, 0], [0, 0, 1, 0]], np.float32)
P_R = P_L.copy()
P_R[0, 3] = -386.1448


def circle_filter(pts, trk, st):
    """deleteUnmatchFeaturesCircle (feature.cpp:96-104) as a mask: a NaN coordinate passes the sign tests"""
    neg = (pts < 0).any(1) | (trk[0] < 0).any(1) | (trk[1] < 0).any(1) | (trk[2] < 0).any(1)
    return (st != 0).all(0) & ~neg


def oracle_frame(orc, imgs, pts, P_l, P_r, K, lk=None, threshold=0, pnp=None):
    """one frame of the reference's path from the oracle's pieces, with non-default parameters"""
    trk, st = oracle_hops(orc, *imgs, pts, **(lk or {}))
    trk = np.stack(trk)
    keep = circle_filter(pts, trk, st)
    l0, r0, r1, l1, ret = pts[keep], trk[0][keep], trk[1][keep], trk[2][keep], trk[3][keep]
    (l0, r0, l1, r1), valid = orc.check_valid_and_remove(l0, r0, l1, r1, ret, threshold)
    xyz = orc.triangulate(P_l, P_r, l0, r0)
    out = dict(status4=st, trk=trk, keep_circ=np.flatnonzero(keep), keep=np.flatnonzero(keep)[valid], l0=l0, r0=r0, l1=l1,
               r1=r1, xyz=xyz)
    if len(l0) >= 4:
        out["pnp"] = orc.solve_pnp_ransac(xyz, l1, K, **(pnp or {}))
    return out


# ------------------------------------------------------------------ LK / circularMatching
def test_lk_nonfinite_and_edge_points(gpu_ctx, volib, orc):
    """NaN / inf / beyond-int32 / denormal / negative start points among ordinary ones: raw status4 and tracks of all four
    hops (lk_full_chain = 1), survivors of the default mode, and the same through vo_track_frame"""
    from test_oracle_images import smooth_image
    w, h = 256, 128
    imgs = [smooth_image(w, h, seed=9), smooth_image(w, h, 3.7, -2.2, seed=9), smooth_image(w, h, 5.1, 1.4, seed=9),
            smooth_image(w, h, -2.3, 2.8, seed=9)]
    imgs = [imgs[0], imgs[1], imgs[3], imgs[2]]   # (l0, r0, l1, r1)
    rng = np.random.default_rng(4)
    ordinary = np.stack([rng.uniform(20, w - 20, 40), rng.uniform(20, h - 20, 40)], 1).astype(np.float32)
    pts = np.vstack([adv.LK_POINTS, ordinary, adv.LK_POINTS[::-1]])
    ref_trk, ref_st = oracle_hops(orc, *imgs, pts)
    ref_trk = np.stack(ref_trk)
    keep = circle_filter(pts, ref_trk, ref_st)
    assert not ref_st[:, :adv.LK_N_HOPELESS].any() and keep[len(adv.LK_POINTS):len(adv.LK_POINTS) + 40].sum() > 30
    with full_chain(gpu_ctx):
        got = gpu_ctx.circular_match(*imgs, pts)
        assert np.array_equal(got["status4"], ref_st)
        run_batch_single(gpu_ctx, volib, imgs, pts, stages=volib.STAGE_PYRAMID | volib.STAGE_LK)
        raw = gpu_ctx.batch_get_tracks(0, len(pts))
    assert np.array_equal(raw["status4"], ref_st)
    for k, name in enumerate(("r0", "r1", "l1", "l0_ret")):
        assert np.array_equal(raw[name], ref_trk[k], equal_nan=True), name
        assert np.array_equal(np.isnan(raw[name]), np.isnan(ref_trk[k])), name
        fin = ~np.isnan(ref_trk[k])
        assert np.array_equal(bits(raw[name])[fin], bits(ref_trk[k])[fin]), name
    for mode in (1, 0):
        gpu_ctx.set_params(lk_full_chain=mode)
        got = gpu_ctx.circular_match(*imgs, pts)
        assert np.array_equal(got["keep_idx"], np.flatnonzero(keep))
        assert np.array_equal(bits(got["l0_ret"]), bits(ref_trk[3][keep])) and np.array_equal(bits(got["l1"]), bits(ref_trk[2][keep]))
        tf = gpu_ctx.track_frame(*imgs, pts, P_L, P_R)
        assert np.array_equal(tf["keep_idx_circ"], np.flatnonzero(keep))
    gpu_ctx.set_params(lk_full_chain=0)
    # nothing but hopeless points: zero survivors, the pose call reports too few points, nothing crashes
    tf = gpu_ctx.track_frame(*imgs, adv.LK_POINTS[:adv.LK_N_HOPELESS], P_L, P_R)
    assert len(tf["keep_idx_circ"]) == 0 and len(tf["l0"]) == 0


@pytest.mark.parametrize("pair", [("zeros", "zeros"), ("white", "gray"), ("checker1", "checker1"), ("stripes1", "stripes1"),
                                  ("binary", "binary"), ("noise", "noise"), ("step", "step"), ("noise", "zeros")])
def test_lk_on_degenerate_images(gpu_ctx, volib, orc, pair):
    """constant / saturated / 1-pixel checkerboard / stripes (aperture problem) / noise / a single step edge: min-eigenvalue
    rejections, singular 2 x 2 systems, 30-iteration walks -- status and tracks bit for bit"""
    h, w = 96, 160
    d = adv.degenerate_images(h, w)
    a, b = d[pair[0]], d[pair[1]]
    imgs = [a, np.roll(b, 1, 1), np.roll(a, 2, 0), b]
    rng = np.random.default_rng(7)
    pts = np.vstack([np.stack([rng.uniform(-12, w + 12, 60), rng.uniform(-12, h + 12, 60)], 1),
                     [[w / 2 - 0.5, 10], [w / 2, h / 2], [w / 2 + 0.5, h - 1]]]).astype(np.float32)
    ref_trk, ref_st = oracle_hops(orc, *imgs, pts)
    with full_chain(gpu_ctx):
        got = gpu_ctx.circular_match(*imgs, pts)
        run_batch_single(gpu_ctx, volib, imgs, pts, stages=volib.STAGE_PYRAMID | volib.STAGE_LK)
        raw = gpu_ctx.batch_get_tracks(0, len(pts))
    assert np.array_equal(got["status4"], ref_st) and np.array_equal(raw["status4"], ref_st)
    for k, name in enumerate(("r0", "r1", "l1", "l0_ret")):
        assert np.array_equal(bits(raw[name]), bits(ref_trk[k])), name
    # FAST + bucketing on the same image, with a carried set the reference would index out of range with
    corners = orc.fast_detect(a)
    assert np.array_equal(gpu_ctx.fast_detect(a), corners)
    for bs, fpb in ((9, 1), (9, 2), (37, 1)):
        allp = np.vstack([adv.BUCKET_POINTS, corners])
        alla = np.concatenate([adv.BUCKET_AGES, np.zeros(max(0, len(allp) - len(adv.BUCKET_AGES)), np.int32)])
        op, oa = orc.bucketing_features(h, w, allp, alla, bs, fpb)
        gp, ga = gpu_ctx.detect_bucket(a, adv.BUCKET_POINTS, adv.BUCKET_AGES, bucket_size=bs, features_per_bucket=fpb)
        assert np.array_equal(gp, op, equal_nan=True) and np.array_equal(ga, oa), (bs, fpb)


# ------------------------------------------------------------------ triangulation
def test_triangulate_adversarial(gpu_ctx, orc):
    """zero disparity (w = 0: convertPointsFromHomogeneous scales by 1), negative disparity (negative depth), NaN / inf / huge /
    denormal coordinates: <= 1e-5 relative where finite, NaN where the reference is NaN, infinities by value"""
    got = gpu_ctx.triangulate(P_L, P_R, adv.TRI_LEFT, adv.TRI_RIGHT)
    ref = orc.triangulate(P_L, P_R, adv.TRI_LEFT, adv.TRI_RIGHT)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    fin = np.isfinite(ref)
    assert np.array_equal(got[~fin & ~np.isnan(ref)], ref[~fin & ~np.isnan(ref)])
    assert np.all(np.abs(got[fin] - ref[fin]) <= 1e-5 * np.maximum(1.0, np.abs(ref[fin])))
    assert ref[2, 2] < 0 and np.isnan(ref[3]).all()      # negative depth stays negative; NaN in -> NaN out
    # a large ordinary set with a sprinkling of zero / negative disparities
    rng = np.random.default_rng(2)
    pl = np.stack([rng.uniform(0, 1241, 3000), rng.uniform(0, 376, 3000)], 1).astype(np.float32)
    pr = pl - np.stack([rng.uniform(1, 80, 3000), rng.normal(0, 0.3, 3000)], 1).astype(np.float32)
    pr[::50] = pl[::50]
    pr[7::50, 0] = pl[7::50, 0] + 5
    got, ref = gpu_ctx.triangulate(P_L, P_R, pl, pr), orc.triangulate(P_L, P_R, pl, pr)
    assert np.array_equal(np.isfinite(got), np.isfinite(ref))
    fin = np.isfinite(ref)
    assert np.all(np.abs(got[fin] - ref[fin]) <= 1e-5 * np.maximum(1.0, np.abs(ref[fin])))


# ------------------------------------------------------------------ solvePnPRansac
def _pnp_case_names():
    class _O:   # the table's names do not depend on the oracle: build them from a stand-in
        @staticmethod
        def project_points(X, r, t, K):
            return np.zeros((len(X), 2))
    from test_oracle_geom import K_KITTI

    def planted(orc, n, *_):
        return np.zeros((n, 3), np.float32), np.zeros((n, 2), np.float32), np.zeros(3), np.zeros(3), None
    return list(adv.pnp_cases(_O, planted, K_KITTI))


@pytest.mark.parametrize("which", _pnp_case_names())
def test_pnp_ransac_adversarial(gpu_ctx, orc, which):
    """NaN / inf / huge / negative-depth object points, NaN / inf image points, duplicates, collinear and coplanar sets,
    n = 4 / 5 / 6 with a duplicate, a NaN, collinear: same return code, same inliers, pose <= 1e-6 (NaN where the reference's
    is NaN -- solvePnPRansac leaves the LAST hypothesis in rvec / tvec, and a hypothesis on a NaN point is NaN)"""
    from test_oracle_geom import planted_problem, K_KITTI
    X, uv, iters = adv.pnp_cases(orc, planted_problem, K_KITTI)[which]
    gpu_ctx.set_params(ransac_iterations=iters)
    try:
        found, rv, tv, R, inl = gpu_ctx.pnp_ransac(X, uv, K_KITTI)
    finally:
        gpu_ctx.set_params(ransac_iterations=500)
    rc, orv, otv, oinl, dbg = orc.solve_pnp_ransac(X, uv, K_KITTI, iterations=iters)
    assert found == (rc == 1)
    assert np.array_equal(inl, oinl)
    assert adv.same(rv, orv, 1e-6) and adv.same(tv, otv, 1e-6), (rv, orv, tv, otv)
    if found and np.isfinite(orv).all():
        assert adv.same(R, orc.rodrigues(orv), 1e-6)


# ------------------------------------------------------------------ findEssentialMat + recoverPose
def test_essential_pose_adversarial(gpu_ctx, orc):
    from test_gpu_parity import _em_scene
    cases, F, PP = adv.essential_cases(_em_scene)
    for name, (p0, p1) in cases.items():
        found, E, Rg, tg, mask, good = gpu_ctx.essential_pose(p0, p1, F, PP)
        ok, Eo, mo, dbg = orc.find_essential_mat(p0, p1, F, PP)
        assert found == bool(ok), name
        if ok:
            go, Ro, to, m2 = orc.recover_pose(Eo, p0, p1, F, PP, mo)
            assert adv.same(E, Eo, 1e-9) and good == go and np.array_equal(mask, m2), name
            assert adv.same(Rg, Ro, 1e-9) and adv.same(tg, to, 1e-9), name


def test_essential_pose_fuzz(gpu_ctx, orc):
    """300 random findEssentialMat(RANSAC) + recoverPose problems (visualOdometry.cpp:152-153, mono_rotation) through
    vo_essential_pose: 5 .. 2 000 correspondences, outlier rates to 0.7, confidences and thresholds other than the reference's --
    found / mask / n_good IDENTICAL, E, R, t <= 1e-9 (the bars of test_gpu_parity.py::test_essential_pose_dropin)."""
    from hypothesis import HealthCheck, given, settings, strategies as st
    from test_gpu_parity import _em_scene
    seen = dict(cases=0, found=0)
    n_examples = int(os.environ.get("VO_FUZZ_EXAMPLES", "300"))
    explore = os.environ.get("VO_FUZZ_SEED")

    @settings(max_examples=n_examples, derandomize=explore is None, deadline=None, database=None, suppress_health_check=list(HealthCheck))
    @given(seed=st.integers(0, 2 ** 31 - 1), n=st.sampled_from([5, 6, 8, 20, 60, 300, 900, 2000]),
           outliers=st.sampled_from([0.0, 0.0, 0.2, 0.5, 0.7]), prob=st.sampled_from([0.9, 0.999, 0.999]),
           thr=st.sampled_from([0.5, 1.0, 1.0, 3.0]))
    def run(seed, n, outliers, prob, thr):
        p1, p2, R, t, F, PP = _em_scene(seed, n, outliers)
        found, E, Rg, tg, mask, good = gpu_ctx.essential_pose(p1, p2, F, PP, prob, thr)
        ok, Eo, mo, dbg = orc.find_essential_mat(p1, p2, F, PP, prob, thr)
        assert found == bool(ok)
        seen["cases"] += 1
        if not ok:
            return
        go, Ro, to, m2 = orc.recover_pose(Eo, p1, p2, F, PP, mo)
        assert good == go and np.array_equal(mask, m2)
        assert adv.same(E, Eo, 1e-9) and adv.same(Rg, Ro, 1e-9) and adv.same(tg, to, 1e-9), (np.abs(E - Eo).max(), np.abs(Rg - Ro).max())
        seen["found"] += 1

    if explore is not None:
        from hypothesis import seed as hyp_seed
        run = hyp_seed(int(explore))(run)
    run()
    print("essential fuzz:", seen)
    assert seen["cases"] >= 0.9 * n_examples and seen["found"] >= 0.8 * seen["cases"], seen


def test_triangulate_fuzz(gpu_ctx, orc):
    """300 random calls of vo_triangulate (main.cpp:169-171): rectified projection pairs of several intrinsics and baselines,
    0 .. 3 000 point pairs with positive, zero and negative disparities, sub-pixel and far outside the image -- finite
    where the oracle is finite, <= 1e-5 relative (the bar of test_gpu_parity.py), NaN / inf where it is."""
    from hypothesis import HealthCheck, given, settings, strategies as st
    seen = dict(cases=0, points=0, identical=0)

    @settings(max_examples=300, derandomize=True, deadline=None, database=None, suppress_health_check=list(HealthCheck))
    @given(seed=st.integers(0, 2 ** 31 - 1), n=st.sampled_from([0, 1, 2, 63, 64, 65, 500, 3000]), fx=st.sampled_from([300.0, 718.856, 1400.0]),
           bf=st.sampled_from([-386.1448, -160.0, -30.0, 50.0]), zero=st.sampled_from([0.0, 0.0, 0.05, 0.5]))
    def run(seed, n, fx, bf, zero):
        rng = np.random.default_rng(seed)
        P_l = np.array([[fx, 0, 607.19, 0], [0, fx, 185.2, 0], [0, 0, 1, 0]], np.float32)
        P_r = P_l.copy()
        P_r[0, 3] = bf
        pl = np.stack([rng.uniform(-50, 1300, n), rng.uniform(-50, 430, n)], 1).astype(np.float32)
        d = rng.uniform(-5, 120, n).astype(np.float32)
        d[rng.random(n) < zero] = 0.0
        pr = pl.copy()
        pr[:, 0] -= d
        pr[:, 1] += rng.normal(0, 0.3, n).astype(np.float32)
        got = gpu_ctx.triangulate(P_l, P_r, pl, pr)
        ref = orc.triangulate(P_l, P_r, pl, pr) if n else np.zeros((0, 3), np.float32)
        assert got.shape == ref.shape
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        fin = np.isfinite(ref)
        assert np.array_equal(got[~fin & ~np.isnan(ref)], ref[~fin & ~np.isnan(ref)])
        rowmax = np.maximum(1.0, np.abs(np.where(fin, ref, 0)).max(1, keepdims=True) if n else 1.0) * np.ones_like(ref)
        assert np.all(np.abs(got[fin] - ref[fin]) <= 1e-5 * rowmax[fin])
        seen["cases"] += 1
        seen["points"] += n
        seen["identical"] += int((bits(got) == bits(ref)).all(1).sum()) if n else 0

    run()
    print("triangulate fuzz:", seen)
    assert seen["cases"] >= 290 and seen["identical"] >= 0.97 * seen["points"], seen


# ------------------------------------------------------------------ integrateOdometryStereo
def test_integrate_odometry_adversarial(volib, orc):
    """the gates of main.cpp:201 / utils.cpp:80 on NaN / inf / huge motions: a NaN fails every comparison -> not integrated"""
    rng = np.random.default_rng(3)
    eye = np.eye(3)
    cases = [(eye, [0, 0, np.nan]), (eye, [np.inf, 0, 0]), (eye, [0, 0, 1e30]), (eye, [0, 0, 0.05]), (eye, [0, 0, 10.0]),
             (eye * np.nan, [0, 0, 1]), (np.full((3, 3), np.inf), [0, 0, 1]), (np.zeros((3, 3)), [0, 0, 1]),
             (orc.rodrigues([0.0999, 0, 0]), [0, 0, 1]), (orc.rodrigues([0.1001, 0, 0]), [0, 0, 1]), (-eye, [0, 0, 1])]
    pose0 = np.eye(4)
    pose0[:3, 3] = rng.normal(0, 5, 3)
    for R, t in cases:
        e_o = orc.rotation_matrix_to_euler(R)
        gate = bool(np.all(np.abs(e_o) < 0.1))
        want, ok_o = orc.integrate_odometry_stereo(pose0, R, t) if gate else (pose0, False)
        got, ok_g, e_g = volib.integrate_odometry(pose0, R, t)
        assert ok_g == ok_o and adv.same(e_g, e_o, 0) and adv.same(got, want, 1e-12), (R, t)


def test_pnp_ransac_fuzz(gpu_ctx, orc):
    """500 random solvePnPRansac problems (visualOdometry.cpp:176-178) through vo_pnp_ransac: 5 .. 3 000 points, any outlier
    rate and noise, scenes in depth, planar scenes and scenes with duplicated points, small (VO-like) and large motions,
    several intrinsics, 1 .. 1 000 iterations, reprojection thresholds 0.25 .. 8 px, confidences 0.5 .. 0.999 -- return code
    and inlier set IDENTICAL, pose <= 1e-6 (the bar of a converged pose of sane size, as in test_track_frame_fuzz)."""
    from hypothesis import HealthCheck, given, settings, strategies as st
    seen = dict(cases=0, found=0, wild=0, wild_within=0, flat=0)
    n_examples = int(os.environ.get("VO_FUZZ_EXAMPLES", "500"))
    explore = os.environ.get("VO_FUZZ_SEED")

    @settings(max_examples=n_examples, derandomize=explore is None, deadline=None, database=None, suppress_health_check=list(HealthCheck))
    @given(seed=st.integers(0, 2 ** 31 - 1), n=st.sampled_from([5, 6, 7, 9, 16, 40, 64, 65, 150, 400, 1000, 3000]),
           outliers=st.sampled_from([0.0, 0.0, 0.1, 0.3, 0.5, 0.8]), noise=st.sampled_from([0.0, 0.05, 0.2, 1.0]),
           scene=st.sampled_from(["depth", "depth", "planar", "duplicates", "near"]), motion=st.sampled_from(["vo", "vo", "large"]),
           fx=st.sampled_from([300.0, 718.856, 1000.0]), iters=st.sampled_from([1, 10, 100, 500, 500, 1000]),
           reproj=st.sampled_from([0.25, 0.5, 0.5, 2.0, 8.0]), conf=st.sampled_from([0.5, 0.99, float(np.float32(0.999))]))
    def run(seed, n, outliers, noise, scene, motion, fx, iters, reproj, conf):
        rng = np.random.default_rng(seed)
        K = np.array([[fx, 0, 607.19], [0, fx, 185.2], [0, 0, 1]], np.float64)
        X = rng.uniform([-10, -2, 4], [10, 2, 50], (n, 3))
        if scene == "planar":
            X[:, 2] = 20.0 + 0.1 * X[:, 0]
        elif scene == "near":
            X[:, 2] = rng.uniform(0.5, 3.0, n)
        elif scene == "duplicates":
            X[n // 2:] = X[:n - n // 2]
        X = X.astype(np.float32)
        if motion == "vo":
            r = np.array([0.002, -0.03, 0.001]) + rng.normal(0, 0.002, 3)
            t = np.array([0.02, -0.01, -0.9]) + rng.normal(0, 0.02, 3)
        else:
            r, t = rng.normal(0, 0.25, 3), rng.normal(0, 1.5, 3)
        uv = orc.project_points(X, r, t, K).astype(np.float64) + rng.normal(0, noise, (n, 2))
        out = rng.random(n) < outliers
        uv[out] += rng.uniform(-40, 40, (int(out.sum()), 2))
        uv = uv.astype(np.float32)
        gpu_ctx.set_params(ransac_iterations=iters, ransac_reproj_error=reproj, ransac_confidence=conf)
        found, rv, tv, R, inl = gpu_ctx.pnp_ransac(X, uv, K)
        rc, orv, otv, oinl, dbg = orc.solve_pnp_ransac(X, uv, K, iterations=iters, reproj=reproj, confidence=conf)
        assert found == (rc == 1), (found, rc)
        assert np.array_equal(inl, oinl)
        seen["cases"] += 1
        sane = np.isfinite(orv).all() and np.isfinite(otv).all() and np.abs(orv).max() <= np.pi and np.abs(otv).max() <= 1e3 and dbg[3] < 20
        if rc == 1 and not sane:   # (no bar, but a count of how many of them meet the bar of a sane pose anyway)
            seen["wild"] += 1
            fin = np.isfinite(orv).all() and np.isfinite(otv).all()
            seen["wild_within"] += bool(fin and adv.same(rv, orv, 1e-6) and adv.same(tv, otv, 1e-6 * max(1.0, float(np.abs(otv).max()))))
            return
        tol_t = 1e-6 * max(1.0, float(np.abs(otv).max())) if np.isfinite(otv).all() else 1e-6
        if rc == 1 and not (adv.same(rv, orv, 1e-6) and adv.same(tv, otv, tol_t)):
            # A flat valley: the inliers do not determine the pose to the bar (hunt seed 3: five coplanar inliers, singular
            # values of the projection's Jacobian 812 ... 0.13 -- the two LM runs end 1.3e-6 apart along the flat direction,
            # where the reprojection moves by 2e-7 px).  Accepted only when the Jacobian says so AND the two poses reproject
            # alike to 1e-5 px; counted, and bounded below.
            sv, J = adv.pose_jacobian_singular_values(X[oinl], orv, otv, K)
            dp = np.concatenate([rv - orv, tv - otv])
            assert sv[0] > 1e3 * sv[-1] and np.abs(J @ dp).max() <= 1e-5 and np.abs(dp).max() <= 1e-4, (rv, orv, tv, otv, dbg, sv)
            seen["flat"] += 1
            return
        assert adv.same(rv, orv, 1e-6) and adv.same(tv, otv, tol_t), (rv, orv, tv, otv, dbg)
        seen["found"] += rc == 1

    if explore is not None:
        from hypothesis import seed as hyp_seed
        run = hyp_seed(int(explore))(run)
    try:
        run()
    finally:
        gpu_ctx.set_params(ransac_iterations=500, ransac_reproj_error=0.5, ransac_confidence=float(np.float32(0.999)))
    print("pnp fuzz:", seen)
    assert (seen["cases"] >= n_examples * 0.9 and seen["found"] >= 0.4 * seen["cases"] and seen["wild"] <= 0.1 * seen["cases"]
            and seen["wild"] - seen["wild_within"] <= 0.01 * seen["cases"] and seen["flat"] <= 0.002 * seen["cases"] + 1), seen


# ------------------------------------------------------------------ seeded fuzz through vo_track_frame
@pytest.fixture(scope="module")
def fuzz_world():
    from visual_odom_amd import synth
    w, h = 640, 256
    world = synth.StereoWorld(seed=61, width=w, height=h, fx=360.0, cx=319.5, cy=127.5, bf=-190.0, tex_size=1024)
    L, R, poses, _ = world.render_sequence(4)
    kps = [synth.select_keypoints(L[k], bucket=16, per_bucket=3) for k in range(3)]
    return dict(world=world, L=L, R=R, kps=kps, w=w, h=h)


def test_track_frame_fuzz(gpu_ctx, orc, fuzz_world):
    """>= 300 random cases: an ROI of a rendered stereo sequence (so w, h are arbitrary and stride != width), n in 0 .. 500
    points (keypoints, random positions, a few adversarial ones), random LK / consistency / RANSAC parameters -- every output of
    vo_track_frame against the oracle's frame.  Seeded (derandomize): the same 300 cases on every run."""
    from hypothesis import HealthCheck, given, settings, strategies as st
    fw = fuzz_world
    seen = dict(cases=0, posed=0, empty=0, wild=0, wild_within=0)

    # (VO_FUZZ_EXAMPLES / VO_FUZZ_SEED: a longer or differently seeded hunt by hand; the suite runs the same 300 cases every time)
    n_examples = int(os.environ.get("VO_FUZZ_EXAMPLES", "300"))
    explore = os.environ.get("VO_FUZZ_SEED")

    @settings(max_examples=n_examples, derandomize=explore is None, deadline=None, database=None,
              suppress_health_check=list(HealthCheck))
    # (the parameter draws lean towards values that track -- a 1-iteration, level-0-only LK on a 40-pixel crop yields a
    # handful of garbage tracks whose "pose" is chaos: still checked for status / survivors / inliers, but not a pose)
    @given(seed=st.integers(0, 2 ** 31 - 1), k=st.integers(0, 2), w=st.sampled_from([40, 64, 96, 131, 200, 320, 333, 480, 601, 640]),
           h=st.sampled_from([40, 64, 97, 128, 160, 200, 256]),
           n_kp=st.integers(0, 400), n_rand=st.integers(0, 100), n_bad=st.integers(0, 6),
           max_level=st.sampled_from([0, 1, 2, 3, 3, 3, 4]), max_count=st.sampled_from([-1, 0, 1, 3, 10, 30, 30, 30, 40]),
           eps=st.sampled_from([0.0, 0.003, 0.01, 0.01, 0.05, 11.0]),
           min_eig=st.sampled_from([0.0, 1e-4, 1e-3, 1e-2]), thr=st.integers(0, 2),
           iters=st.sampled_from([1, 7, 64, 100, 500]), reproj=st.sampled_from([0.25, 0.5, 1.0, 3.0]),
           conf=st.sampled_from([0.5, 0.99, float(np.float32(0.999))]))
    def run(seed, k, w, h, n_kp, n_rand, n_bad, max_level, max_count, eps, min_eig, thr, iters, reproj, conf):
        rng = np.random.default_rng(seed)
        x0, y0 = int(rng.integers(0, fw["w"] - w + 1)), int(rng.integers(0, fw["h"] - h + 1))
        roi = (slice(y0, y0 + h), slice(x0, x0 + w))
        imgs = [fw["L"][k][roi], fw["R"][k][roi], fw["L"][k + 1][roi], fw["R"][k + 1][roi]]   # views: stride 640
        kp = fw["kps"][k] - np.float32([x0, y0])
        kp = kp[(kp[:, 0] >= 0) & (kp[:, 0] < w) & (kp[:, 1] >= 0) & (kp[:, 1] < h)]
        kp = kp[rng.permutation(len(kp))[:n_kp]]
        rnd = np.stack([rng.uniform(-15, w + 15, n_rand), rng.uniform(-15, h + 15, n_rand)], 1).astype(np.float32)
        bad = adv.LK_POINTS[rng.integers(0, len(adv.LK_POINTS), n_bad)]
        pts = np.vstack([kp, rnd, bad]).astype(np.float32)
        pts = pts[rng.permutation(len(pts))]
        P_l, P_r = fw["world"].proj_matrices()
        P_l, P_r = P_l.copy(), P_r.copy()
        P_l[0, 2] -= x0
        P_l[1, 2] -= y0
        P_r[0, 2] -= x0
        P_r[1, 2] -= y0
        P_r[0, 3] = P_r[0, 3]   # (the baseline term bf does not move with the ROI)
        K = P_l[:, :3].copy()
        gpu_ctx.set_params(lk_max_level=max_level, lk_max_count=max_count, lk_epsilon=eps, lk_min_eig_threshold=min_eig,
                           consistency_threshold=thr, ransac_iterations=iters, ransac_reproj_error=reproj, ransac_confidence=conf)
        got = gpu_ctx.track_frame(*imgs, pts, P_l, P_r)
        ref = oracle_frame(orc, imgs, pts, P_l, P_r, K, lk=dict(max_level=max_level, max_count=max_count, eps=eps, min_eig=min_eig),
                           threshold=thr, pnp=dict(iterations=iters, reproj=reproj, confidence=conf))
        assert np.array_equal(got["keep_idx_circ"], ref["keep_circ"])
        assert np.array_equal(got["keep_idx"], ref["keep"])
        for name in ("l0", "r0", "l1", "r1"):
            assert np.array_equal(bits(got[name]), bits(ref[name])), name
        fin = np.isfinite(ref["xyz"])
        assert np.array_equal(np.isfinite(got["xyz"]), fin)
        assert np.all(np.abs(got["xyz"][fin] - ref["xyz"][fin]) <= 1e-5 * np.maximum(1.0, np.abs(ref["xyz"][fin])))
        seen["cases"] += 1
        if "pnp" not in ref:
            assert got["rc"] == -4   # VO_ERR_TOO_FEW
            seen["empty"] += 1
            return
        rc, rv, tv, inl, dbg = ref["pnp"]
        assert (got["rc"] == 0) == (rc == 1)
        assert np.array_equal(got["inliers"], inl)
        # The pose bar (<= 1e-6) is a bar for a pose.  When the reference's own Levenberg-Marquardt run ends nowhere -- a handful
        # of garbage tracks (maxCount 1, level 0 only): it stops at its 20-iteration cap, or at |rvec| beyond a turn, or at a
        # translation of kilometres -- its last digits are chaos in ANY summation order (seed 296 of this fuzz: 10 LM iterations
        # in the reference's order, 18 in the kernel's, both "poses" ~1e9; a 3 x 2 000-case hunt with other seeds found two more
        # of the kind: 20 iterations without convergence at |rvec| = 6.29; a converged 5-inlier frame at |t| = 84 m whose tvec
        # differs by 7e-6 = 8e-8 of its length) and all that can be held is the class of the answer.  So: a converged run at a
        # pose of sane size is held to 1e-6 rad and 1e-6 x max(1 m, |t|); the rest is counted, and bounded below.
        sane = np.isfinite(rv).all() and np.isfinite(tv).all() and np.abs(rv).max() <= np.pi and np.abs(tv).max() <= 1e3 and dbg[3] < 20
        if rc == 1 and not sane:
            seen["wild"] += 1
            assert not (np.isfinite(got["rvec"]).all() and np.abs(got["rvec"]).max() <= 1e-3 and np.abs(got["tvec"]).max() <= 1e-3)
            fin = np.isfinite(rv).all() and np.isfinite(tv).all()   # (no bar -- but most of them meet the bar of a sane pose anyway: counted)
            seen["wild_within"] += bool(fin and adv.same(got["rvec"], rv, 1e-6) and adv.same(got["tvec"], tv, 1e-6 * max(1.0, float(np.abs(tv).max()))))
            return
        tol_t = 1e-6 * max(1.0, float(np.abs(tv).max())) if np.isfinite(tv).all() else 1e-6
        assert adv.same(got["rvec"], rv, 1e-6) and adv.same(got["tvec"], tv, tol_t), (got["rvec"], rv, got["tvec"], tv, dbg)
        seen["posed"] += rc == 1

    if explore is not None:
        from hypothesis import seed as hyp_seed
        run = hyp_seed(int(explore))(run)
    try:
        run()
    finally:
        gpu_ctx.set_params(lk_max_level=3, lk_max_count=30, lk_epsilon=0.01, lk_min_eig_threshold=1e-3, consistency_threshold=0,
                           ransac_iterations=500, ransac_reproj_error=0.5, ransac_confidence=float(np.float32(0.999)))
    print("fuzz:", seen)
    assert (seen["cases"] >= min(n_examples, 300) and seen["posed"] >= 0.2 * seen["cases"] and seen["wild"] <= 0.25 * seen["cases"]
            and seen["wild"] - seen["wild_within"] <= 0.01 * seen["cases"]), seen   # (300 cases: 53 unconverged, 52 of them within the bar anyway)


def test_detect_bucket_fuzz(gpu_ctx, volib, orc, fuzz_world):
    """1 000 random cases (seconds) through vo_fast_detect + vo_detect_bucket (head of matchingFeatures, visualOdometry.cpp:95-108): an ROI of
    a rendered frame, of noise or of a checkerboard (arbitrary w, h, stride != width), any FAST threshold with and without
    non-maximum suppression, bucket edges from 1 pixel to beyond the image, 1 .. 40 features per bucket (beyond the documented limits: VO_ERR_ARG), 0 .. 2 500 carried
    points (a few of them the adversarial ones of tests/adversarial.py) with an ages array at least as long (quirk B3),
    re-detection below 0 / 400 / 2 000 points -- corners, bucketed points and ages BIT-EXACT against the oracle's chain."""
    from hypothesis import HealthCheck, given, settings, strategies as st
    fw = fuzz_world
    rs = np.random.default_rng(5)
    noise = rs.integers(0, 256, (fw["h"], fw["w"]), dtype=np.uint8)
    yy, xx = np.mgrid[0:fw["h"], 0:fw["w"]]
    board = (((xx // 5 + yy // 5) & 1) * 200 + 20).astype(np.uint8)
    sources = [fw["L"][0], fw["R"][1], fw["L"][2], noise, board]
    seen = dict(cases=0, redetected=0, corners=0, out=0, refused=0, fine=0)
    n_examples = int(os.environ.get("VO_FUZZ_EXAMPLES", "1000"))
    explore = os.environ.get("VO_FUZZ_SEED")

    @settings(max_examples=n_examples, derandomize=explore is None, deadline=None, database=None, suppress_health_check=list(HealthCheck))
    @given(seed=st.integers(0, 2 ** 31 - 1), src=st.integers(0, 4), w=st.sampled_from([40, 64, 97, 131, 200, 320, 333, 480, 601, 640]),
           h=st.sampled_from([40, 64, 97, 128, 160, 200, 256]), thr=st.sampled_from([0, 1, 5, 20, 20, 20, 50, 100, 254, 255]),
           nonmax=st.integers(0, 1), bs=st.sampled_from([0, 0, 1, 3, 7, 16, 25, 50, 300]), fpb=st.sampled_from([1, 1, 2, 6, 40]),
           n=st.sampled_from([0, 0, 1, 17, 150, 399, 400, 1999, 2000, 2500]), extra_ages=st.sampled_from([0, 0, 3, 40]),
           n_bad=st.integers(0, 4), redetect=st.sampled_from([0, 400, 2000, 2000]))
    def run(seed, src, w, h, thr, nonmax, bs, fpb, n, extra_ages, n_bad, redetect):
        rng = np.random.default_rng(seed)
        x0, y0 = int(rng.integers(0, fw["w"] - w + 1)), int(rng.integers(0, fw["h"] - h + 1))
        img = sources[src][y0:y0 + h, x0:x0 + w]                       # a view: stride 640
        pts = np.stack([rng.uniform(0, w - 0.01, n), rng.uniform(0, h - 0.01, n)], 1).astype(np.float32)
        if n_bad and n:
            pts[rng.integers(0, n, n_bad)] = adv.BUCKET_POINTS[rng.integers(0, len(adv.BUCKET_POINTS), n_bad)]
        ages = rng.integers(-2, 14, n + extra_ages).astype(np.int32)
        cap = 1 << 18
        fast_o = orc.fast_detect(np.ascontiguousarray(img), thr, bool(nonmax), cap=cap)
        if src == 3 and thr < 5 and not nonmax and w * h > 100000:
            return                                                       # (hundreds of thousands of corners: a capacity case, test_gpu_round2)
        fast_g = gpu_ctx.fast_detect(img, thr, bool(nonmax), cap=cap)
        assert np.array_equal(bits(fast_g), bits(fast_o)), ("corners", len(fast_g), len(fast_o))
        if n + len(fast_o) + extra_ages > gpu_ctx.max_pts:
            return
        if n < redetect:
            allp, alla = np.vstack([pts, fast_o]), np.concatenate([ages, np.zeros(len(fast_o), np.int32)])   # feature.cpp:255-262
            seen["redetected"] += 1
        else:
            allp, alla = pts, ages
        edge = bs if bs else h // 10
        if not adv.bucket_grid_ok(w, h, edge, fpb):                    # the documented limits of the device bucketing (vo_hip.h)
            with pytest.raises(volib.VoError) as e:
                gpu_ctx.detect_bucket(img, pts, ages, fast_threshold=thr, fast_nonmax=nonmax, redetect_below=redetect,
                                      bucket_size=bs, features_per_bucket=fpb)
            assert e.value.code == volib.VO_ERR_ARG
            seen["refused"] += 1
            return
        want_p, want_a = orc.bucketing_features(h, w, allp, alla, edge, fpb)
        got_p, got_a = gpu_ctx.detect_bucket(img, pts, ages, fast_threshold=thr, fast_nonmax=nonmax, redetect_below=redetect,
                                             bucket_size=bs, features_per_bucket=fpb)
        assert np.array_equal(bits(got_p), bits(want_p)), ("points", len(got_p), len(want_p))
        assert np.array_equal(got_a, want_a), "ages"
        seen["cases"] += 1
        seen["fine"] += (h // edge + 1) * (w // edge + 1) > 1024
        seen["corners"] += len(fast_o)
        seen["out"] += len(want_p)

    if explore is not None:
        from hypothesis import seed as hyp_seed
        run = hyp_seed(int(explore))(run)
    run()
    print("detect fuzz:", seen)
    assert seen["cases"] >= 0.4 * n_examples and seen["redetected"] >= 0.3 * seen["cases"] and seen["refused"] > 0 and seen["fine"] > 0 and seen["out"] > 10 * seen["cases"], seen


# ------------------------------------------------------------------ the shipped adapter (adapters/feature_hip.cpp)
def test_shipped_adapter_detect_and_bucket(orc, small_seq):
    """detectAndBucket_hip (adapters/feature_hip.cpp; head of matchingFeatures, visualOdometry.cpp:95-108) against the
    reference's OWN appendNewFeatures + bucketingFeatures compiled where they lie (oracle/_ref): first frame (empty set),
    a carried set with ages longer than points (quirk B3), a set that is large enough to skip re-detection"""
    import ctypes as C
    so = os.path.join(ROOT, "tests", "_build", "libvo_ref_dropin.so")
    if orc.ref_lib() is None or not os.path.exists(so):
        pytest.skip("built only where /root/reference exists (make -C tests/ref_dropin) and shipped with the snapshot")
    hip = C.CDLL(so)
    img = np.ascontiguousarray(small_seq["L"][0])
    h, w = img.shape
    rng = np.random.default_rng(8)
    big = np.stack([rng.uniform(0, w - 1, 2100), rng.uniform(0, h - 1, 2100)], 1).astype(np.float32)
    sets = [(np.zeros((0, 2), np.float32), np.zeros(0, np.int32)),
            (big[:150], rng.integers(0, 12, 170).astype(np.int32)),
            (big, rng.integers(0, 9, 2100).astype(np.int32))]
    cap = 1 << 16
    for pts, ages in sets:
        rp, ra = (orc.ref_append_new_features(img, pts, ages) if len(pts) < 2000 else (pts, ages))   # visualOdometry.cpp:95
        rp, ra = orc.ref_bucketing_features(h, w, rp, ra, h // 10, 1)
        P, A = np.zeros((cap, 2), np.float32), np.zeros(cap, np.int32)
        P[:len(pts)], A[:len(ages)] = pts, ages
        n_p, n_a = C.c_int(len(pts)), C.c_int(len(ages))
        rc = hip.adapter_detect_bucket(img.ctypes.data_as(C.c_void_p), w, h, P.ctypes.data_as(C.c_void_p), A.ctypes.data_as(C.c_void_p),
                                       C.byref(n_p), C.byref(n_a), cap)
        assert rc == 0
        assert np.array_equal(P[:n_p.value], rp) and np.array_equal(A[:n_a.value], ra), len(pts)


# ------------------------------------------------------------------ ADVICE r05
def test_two_frame_loops_share_one_context(volib, small_world):
    """the kept pair belongs to the context.  Two StereoOdometry loops on ONE context with images of the
    same size must not read each other's t1 pair as t0: each compares vo_kept_pair_id with the id it saw after its own call
    and falls back to the four-image call -- trajectories equal those of two loops with a context each."""
    from visual_odom_amd.odometry import StereoOdometry
    P_l, P_r = small_world.proj_matrices()
    L, R, _, _ = small_world.render_sequence(6)
    L2, R2 = [np.ascontiguousarray(a[:, ::-1]) for a in R], [np.ascontiguousarray(a[:, ::-1]) for a in L]   # another "sequence", same size
    h, w = L[0].shape
    solo = []
    for seqL, seqR in ((L, R), (L2, R2)):
        vo = StereoOdometry(P_l, P_r, max_w=w, max_h=h)
        for a, b in zip(seqL, seqR):
            vo.process(a, b)
        solo.append(np.array(vo.trajectory))
        vo.close()
    ctx = volib.Context(0, w, h, 4096, 1)
    a, b = StereoOdometry(P_l, P_r, ctx=ctx), StereoOdometry(P_l, P_r, ctx=ctx)
    ids = set()
    for k in range(len(L)):
        a.process(L[k], R[k])
        ids.add(ctx.kept_pair_id())
        b.process(L2[k], R2[k])
        ids.add(ctx.kept_pair_id())
    assert np.array_equal(np.array(a.trajectory), solo[0]) and np.array_equal(np.array(b.trajectory), solo[1])
    assert len(ids - {0}) >= 2 * (len(L) - 1) and ctx.kept_pair_id() > 0   # every call left a new pair
    # a single loop on the context keeps its pair: the id it stored is still the context's
    c = StereoOdometry(P_l, P_r, ctx=ctx)
    for k in range(3):
        c.process(L[k], R[k])
        assert k == 0 or c._kept_id == ctx.kept_pair_id()
    ctx.batch_configure(4, w, h, 1)
    ctx.batch_upload_image(0, L[0])            # the batch API takes the image table: no kept pair
    assert ctx.kept_pair_id() == 0
    ctx.close()


def test_kept_pair_split_chain_edges(volib, small_world):
    """Round 6: on the kept pair the synchronous calls launch hop 0 of the LK chain (lk_hops_kernel) before the new pair has
    crossed PCIe and hops 1 .. 3 behind it (capi_run.hip, vo_ctx::defer).  The ordinary cases are
    test_gpu_round5.py::test_kept_pair_calls_equal_four_image_calls; here the edges of the new path, each against a four-image
    call on a second context, every output bit for bit: no points, too few points (VO_ERR_TOO_FEW: the pair is kept all the
    same), the full-chain mode with points that die in hop 0, non-reference LK parameters, mono_rotation, a new point load
    (the schedule probe runs inside the call: the deferred pair goes the old way), vo_circular_match, 40 calls in a row."""
    L, R, _, _ = small_world.render_sequence(6)
    P_l, P_r = small_world.proj_matrices()
    h, w = L[0].shape
    from visual_odom_amd import synth
    base = synth.select_keypoints(L[0], bucket=20, per_bucket=3).astype(np.float32)
    rng = np.random.default_rng(66)
    wild = np.vstack([base[:120], [[-3, 10], [w + 30, 5], [np.nan, 4], [5, np.inf], [1e30, 1], [0, 0], [w - 1, h - 1]],
                      rng.uniform(-15, [w + 15, h + 15], (60, 2))]).astype(np.float32)
    keys = ("l0", "r0", "l1", "r1", "xyz", "keep_idx", "keep_idx_circ", "inliers", "rvec", "tvec", "R")
    ctx = volib.Context(0, w, h, 2048, 1)
    ref = volib.Context(0, w, h, 2048, 1)

    def both(k0, k1, pts):
        """kept call on ctx (its t0 pair = pair k0, left by the previous call) against the four-image call on ref"""
        want = ref.track_frame(L[k0], R[k0], L[k1], R[k1], pts, P_l, P_r)
        got = ctx.track_frame(None, None, L[k1], R[k1], pts, P_l, P_r)
        assert got["rc"] == want["rc"], (k0, k1, got["rc"], want["rc"])
        for key in keys:
            assert np.array_equal(got[key], want[key], equal_nan=True), (k0, k1, key)
        return got

    def params(k, **kw):
        """new parameters on both contexts; vo_set_params drops the kept pair (the pyramid plan may change), so a four-image
        call leaves pair k again"""
        ctx.set_params(**kw)
        ref.set_params(**kw)
        with pytest.raises(volib.VoError) as e:
            ctx.track_frame(None, None, L[k], R[k], base, P_l, P_r)
        assert e.value.code == volib.VO_ERR_STATE
        ctx.track_frame(L[k - 1], R[k - 1], L[k], R[k], base, P_l, P_r)

    try:
        ctx.track_frame(L[0], R[0], L[1], R[1], base, P_l, P_r)              # leaves pair 1
        assert len(both(1, 2, np.zeros((0, 2), np.float32))["l0"]) == 0      # no points (a points-only pull of nothing)
        assert both(2, 3, base[:3])["rc"] == volib.VO_ERR_TOO_FEW            # ... and the pair of the failed call is kept:
        g = both(3, 4, base)
        assert g["rc"] == volib.VO_OK and len(g["inliers"]) > 20
        params(4, lk_full_chain=1)
        both(4, 5, wild)                                                     # dead in hop 0, carried through hops 1 .. 3
        params(5, lk_full_chain=0)
        both(5, 4, wild)
        params(4, lk_max_level=2, lk_max_count=7, lk_epsilon=0.03)
        both(4, 3, base)
        both(3, 2, wild)
        params(2, lk_max_level=3, lk_max_count=30, lk_epsilon=0.01, mono_rotation=1)
        both(2, 3, base)
        params(3, mono_rotation=0)
        both(3, 2, base[::7])                                                # another point load: a probe inside the call
        wantc = ref.circular_match(L[2], R[2], L[1], R[1], wild)
        gotc = ctx.circular_match(None, None, L[1], R[1], wild)
        for key in ("l0", "r0", "r1", "l1", "l0_ret", "status4", "keep_idx"):
            assert np.array_equal(gotc[key], wantc[key], equal_nan=True), key
        order = [2, 3, 4, 5, 4, 3, 2, 1, 0, 1]
        prev = 1
        for i in range(40):
            k = order[i % len(order)]
            both(prev, k, base if i % 3 else wild)
            prev = k
    finally:
        ctx.close()
        ref.close()


def test_kept_pair_call_fuzz(volib, fuzz_world):
    """Random walks of synchronous calls: one context names the kept pair whenever the call's t0 pair IS the previous call's t1
    pair (the split LK chain of round 6: hop 0 under the new pair's PCIe pull), a second context gets all four images every
    time -- every output of every call bit for bit.  Random ROIs of a rendered sequence (w, h arbitrary, stride 640; a new ROI
    or a parameter change drops the kept pair: four images on both), 0 .. 500 points incl. adversarial ones, vo_track_frame and
    vo_circular_match, detections on the kept image and on others in between, full-chain mode on and off, mono_rotation.
    VO_FUZZ_EXAMPLES / VO_FUZZ_SEED: longer, differently seeded hunts (tools/gpu_round.sh hunt)."""
    fw = fuzz_world
    n_calls = int(os.environ.get("VO_FUZZ_EXAMPLES", "300"))
    rng = np.random.default_rng(int(os.environ.get("VO_FUZZ_SEED", "20261001")))
    P_l0, P_r0 = fw["world"].proj_matrices()
    keys = ("l0", "r0", "l1", "r1", "xyz", "keep_idx", "keep_idx_circ", "inliers", "rvec", "tvec", "R")
    ctx = volib.Context(0, fw["w"], fw["h"], 1024, 1)
    ref = volib.Context(0, fw["w"], fw["h"], 1024, 1)
    seen = dict(calls=0, kept=0, circ=0, ok=0, detect=0)
    defaults = dict(lk_max_level=3, lk_max_count=30, lk_epsilon=0.01, lk_min_eig_threshold=1e-3, lk_full_chain=0, mono_rotation=0,
                    consistency_threshold=0, ransac_iterations=500)
    try:
        roi = k_prev = None           # the pair ctx holds: frame k_prev of this ROI (None: no kept pair)
        for _ in range(n_calls):
            if roi is None or rng.random() < 0.07:
                w = int(rng.choice([64, 96, 131, 200, 320, 333, 480, 601, 640]))
                h = int(rng.choice([64, 97, 128, 160, 200, 256]))
                x0, y0 = int(rng.integers(0, fw["w"] - w + 1)), int(rng.integers(0, fw["h"] - h + 1))
                roi, k_prev = (slice(y0, y0 + h), slice(x0, x0 + w)), None
                P_l, P_r = P_l0.copy(), P_r0.copy()
                P_l[0, 2] -= x0; P_l[1, 2] -= y0; P_r[0, 2] -= x0; P_r[1, 2] -= y0
            if rng.random() < 0.08:   # new parameters: vo_set_params drops the kept pair
                prm = dict(defaults, lk_max_level=int(rng.choice([1, 2, 3, 3, 4])), lk_max_count=int(rng.choice([3, 10, 30, 30])),
                           lk_full_chain=int(rng.integers(0, 2)), mono_rotation=int(rng.random() < 0.2),
                           consistency_threshold=int(rng.integers(0, 2)), ransac_iterations=int(rng.choice([7, 100, 500])))
                ctx.set_params(**prm)
                ref.set_params(**prm)
                k_prev = None
            k0 = int(rng.integers(0, 4)) if k_prev is None else k_prev
            k1 = k0 + 1 if k0 == 0 else k0 - 1 if k0 == 3 else k0 + int(rng.choice([-1, 1]))
            kept = k_prev is not None and rng.random() < 0.9
            imgs = [fw["L"][k0][roi], fw["R"][k0][roi], fw["L"][k1][roi], fw["R"][k1][roi]]
            hh, ww = imgs[0].shape
            if kept and rng.random() < 0.15:   # detection on the kept pair's left image / on another image: the pair stays
                none = (np.zeros((0, 2), np.float32), np.zeros(0, np.int32))
                a = ctx.detect_bucket(None, *none, features_per_bucket=2)
                b = ref.detect_bucket(imgs[0], *none, features_per_bucket=2)
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
                if rng.random() < 0.5:
                    assert np.array_equal(ctx.fast_detect(imgs[3]), ref.fast_detect(imgs[3]))
                seen["detect"] += 1
            kp = fw["kps"][min(k0, 2)] - np.float32([roi[1].start, roi[0].start])
            kp = kp[(kp[:, 0] >= 0) & (kp[:, 0] < ww) & (kp[:, 1] >= 0) & (kp[:, 1] < hh)]
            kp = kp[rng.permutation(len(kp))[:int(rng.integers(0, 400))]]
            n_rand = int(rng.integers(0, 100))
            rnd = np.stack([rng.uniform(-15, ww + 15, n_rand), rng.uniform(-15, hh + 15, n_rand)], 1).astype(np.float32)
            bad = adv.LK_POINTS[rng.integers(0, len(adv.LK_POINTS), int(rng.integers(0, 6)))]
            pts = np.vstack([kp, rnd, bad]).astype(np.float32)
            pts = pts[rng.permutation(len(pts))]
            t0 = (None, None) if kept else (imgs[0], imgs[1])
            if rng.random() < 0.15:
                got = ctx.circular_match(*t0, imgs[2], imgs[3], pts)
                want = ref.circular_match(*imgs, pts)
                for key in ("l0", "r0", "r1", "l1", "l0_ret", "status4", "keep_idx"):
                    assert np.array_equal(got[key], want[key], equal_nan=True), (seen, key)
                seen["circ"] += 1
            else:
                got = ctx.track_frame(*t0, imgs[2], imgs[3], pts, P_l, P_r)
                want = ref.track_frame(*imgs, pts, P_l, P_r)
                assert got["rc"] == want["rc"], (seen, got["rc"], want["rc"])
                for key in keys:
                    assert np.array_equal(got[key], want[key], equal_nan=True), (seen, key)
                seen["ok"] += got["rc"] == 0
            seen["calls"] += 1
            seen["kept"] += kept
            k_prev = k1
    finally:
        ctx.close()
        ref.close()
    print("kept-pair fuzz:", seen)
    assert seen["kept"] >= 0.6 * n_calls and seen["ok"] >= 0.3 * n_calls and seen["circ"] >= 0.05 * n_calls, seen


def test_batch_run_after_a_dropin_call_reads_the_uploaded_quads(gpu_ctx, volib, small_seq):
    """a drop-in call runs frame 0 on a constant quadruple of its own; the next batch run without a new
    vo_batch_set_quads reads what vo_batch_set_quads last uploaded -- whatever slot pair the drop-in calls used"""
    s = small_seq
    imgs = [s["L"][0], s["R"][0], s["L"][1], s["R"][1]]
    pts = s["pts"][0]
    h, w = imgs[0].shape
    gpu_ctx.batch_configure(4, w, h, 1)
    for i, im in enumerate(imgs):
        gpu_ctx.batch_upload_image(i, im)
    gpu_ctx.batch_set_quads([[0, 1, 2, 3]])
    gpu_ctx.batch_set_points(0, pts)
    gpu_ctx.batch_run(volib.STAGE_PYRAMID | volib.STAGE_LK)
    gpu_ctx.batch_sync()
    want = gpu_ctx.batch_get_tracks(0, len(pts))
    gpu_ctx.circular_match(*imgs, pts)                       # slots (0, 1) -> (2, 3): quadruple {0, 1, 2, 3}
    gpu_ctx.circular_match(None, None, imgs[0], imgs[1], pts)   # kept pair (2, 3) -> new pair in (0, 1): quadruple {2, 3, 0, 1}
    gpu_ctx.batch_configure(4, w, h, 1)                      # same shape: early return
    for i, im in enumerate(imgs):
        gpu_ctx.batch_upload_image(i, im)
    gpu_ctx.batch_set_points(0, pts)
    gpu_ctx.batch_run(volib.STAGE_PYRAMID | volib.STAGE_LK)  # no vo_batch_set_quads: {0, 1, 2, 3} as uploaded above
    gpu_ctx.batch_sync()
    got = gpu_ctx.batch_get_tracks(0, len(pts))
    for k in ("r0", "r1", "l1", "l0_ret", "status4"):
        assert np.array_equal(got[k], want[k]), k


# ------------------------------------------------------------------ eight real ranks on GPU 0
def test_one_rank_over_rccl():
    """The N > 1 code path of bench.py over the backend a real node uses: `nccl` (= RCCL) cannot put two ranks on one GPU, so
    the shared-GPU tests run over gloo -- this one forces the process group at world size 1 (VO_DIST_FORCE=1, the driver's
    launch line with --nproc-per-node 1): init_process_group("nccl") beside libvo_hip's own HIP runtime, barrier, the MAX / SUM
    all_reduce and the all_gather on DEVICE tensors, the config-5 leg and the per-rank core slices."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, VO_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("VO_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--frames", "16", "--steps", "3",
           "--warmup", "1", "--no-cpu-baseline", "--sustain", "0", "--no-replay-leg", "--validate", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    b = json.loads(lines[0])
    assert b["dist_backend"] == "nccl" and b["ranks"] == 1 and b["n_gpus"] == 1
    frames = b["value"] * b["ms_per_step"] * 1e-3 * b["steps"]
    assert abs(frames - 16 * 3) < 1e-6 * frames and b["validated_frames"] == 2
    c5 = [c for c in b["configs"] if c["name"] == "config5_one_sequence_per_gpu"]
    assert len(c5) == 1 and len(c5[0]["per_gpu_value"]) == 1 and abs(c5[0]["per_gpu_value"][0] - c5[0]["value"]) <= 1e-6 * c5[0]["value"]
    assert len(b["host_cores_per_rank"]["cpus"]) == 1


def test_eight_ranks_of_bench_share_gpu_zero():
    """the driver's N = 8 launch line with EIGHT real contexts, all on GPU 0 (VO_ALLOW_SHARED_GPU=1, gloo
    for the barrier and the reductions): the real run_batch with per-rank validation, the real config-5 leg (one sequence per
    rank through the lock-step loop), eight disjoint core slices in the printed line.  No scaling claim: one GPU."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, VO_ALLOW_SHARED_GPU="1", VO_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--frames", "32", "--steps", "3",
           "--warmup", "1", "--no-cpu-baseline", "--sustain", "0", "--no-replay-leg", "--validate", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    b = json.loads(lines[0])
    assert b["ranks"] == 8 and b["n_gpus"] == 1 and b["scaling"] == "weak"
    assert b["validated_frames"] == 2                       # the smallest of the eight ranks' counts
    frames = b["value"] * b["ms_per_step"] * 1e-3 * b["steps"]
    assert abs(frames - 8 * 32 * 3) < 1e-6 * frames
    c5 = [c for c in b["configs"] if c["name"] == "config5_one_sequence_per_gpu"]
    assert len(c5) == 1 and c5[0]["ranks"] == 8 and len(c5[0]["per_gpu_value"]) == 8 and all(v > 0 for v in c5[0]["per_gpu_value"])
    assert c5[0]["value"] <= sum(c5[0]["per_gpu_value"]) * 1.0001 and c5[0]["validated_frames"] >= 1
    hc = b["host_cores_per_rank"]
    assert len(hc["cpus"]) == 8 and all(c >= 1 for c in hc["cpus"])
    spans = sorted(zip(hc["first_cpu"], hc["last_cpu"]))
    if (os.cpu_count() or 1) >= 8:
        assert all(a[1] < b_[0] for a, b_ in zip(spans, spans[1:])), spans     # eight disjoint slices
    print("eight ranks on GPU 0: batch %.0f frames/s; config 5: %.0f aggregate, per rank %s; core slices %s"
          % (b["value"], c5[0]["value"], ["%.0f" % v for v in c5[0]["per_gpu_value"]], spans))


# ------------------------------------------------------------------ soak + determinism
def test_soak_5000_lockstep_steps_are_deterministic_and_leak_free(volib):
    """VERDICT r05 item 7.  The reference's loop runs 9 000 frames (main.cpp:123); here 5 000 lock-step steps of 8 sequences
    on a 33-pair feed, TWICE: device memory flat after warm-up, trajectories of the two runs bit-identical (the schedule
    probe of the first ~200 steps and its pipeline drains included), a run into max_steps exhaustion (VO_ERR_STATE) and
    vo_seq_reset mid-run, then a context created and destroyed 50 times without a leak on the device or the host."""
    import psutil
    import torch
    from visual_odom_amd import synth
    w, h, S, Q, STEPS, CHUNK = 640, 192, 8, 32, 5000, 1000
    world = synth.StereoWorld(seed=77, width=w, height=h, fx=370.0, cx=319.5, cy=95.5, bf=-200.0, tex_size=1024)
    lefts, rights, _, _ = world.render_sequence(Q + 1)
    P_l, P_r = world.proj_matrices()
    dev = torch.device("cuda", 0)
    src = [(torch.from_numpy(np.ascontiguousarray(lefts[k])).to(dev), torch.from_numpy(np.ascontiguousarray(rights[k])).to(dev))
           for k in range(Q + 1)]
    torch.cuda.synchronize()

    def feed(s, k):                                   # ping-pong over the 33 pairs, every sequence phase-shifted
        j = (k + 4 * s) % (2 * Q)
        return j if j <= Q else 2 * Q - j

    def run(ctx, exhaust):
        ctx.batch_set_detect_params()
        ctx.seq_configure(S, w, h, ring=3, max_steps=CHUNK + 8)
        ctx.batch_set_projection(P_l, P_r)
        tables = [ctx.seq_pair_table(range(S), [src[feed(s, k)][0].data_ptr() for s in range(S)],
                                     [src[feed(s, k)][1].data_ptr() for s in range(S)]) for k in range(2 * Q)]
        rows, free_after_warmup, k = [], None, 0
        while k < STEPS:
            for _ in range(CHUNK):
                ctx.seq_push_pairs(tables[k % (2 * Q)], w, 2)
                ctx.seq_step()
                k += 1
                if k == 300:
                    ctx.seq_sync()
                    free_after_warmup = torch.cuda.mem_get_info(0)[0]
            if exhaust and k == 2 * CHUNK:            # run into the end of the trajectory rows: the step is refused, nothing breaks
                refused = 0
                for extra in range(12):
                    ctx.seq_push_pairs(tables[(k + extra) % (2 * Q)], w, 2)
                    try:
                        ctx.seq_step()
                    except volib.VoError as e:
                        assert e.code == volib.VO_ERR_STATE
                        refused += 1
                        break
                assert refused == 1
            ctx.seq_sync()
            for s in range(S):
                r, info = ctx.seq_get_trajectory(s)
                rows.append((r[:CHUNK - 1].copy(), info[:CHUNK - 1].copy()))   # (the same rows in both runs, whatever the exhaustion run appended)
            ctx.seq_reset(-1)                         # rewind: all rows available again, empty feature sets, identity pose
        ctx.seq_sync()
        return rows, free_after_warmup, torch.cuda.mem_get_info(0)[0]

    ctx = volib.Context(0, w, h, 4096, S)
    try:
        rows_a, free_warm, free_end = run(ctx, exhaust=False)
        assert free_end >= free_warm - (1 << 20), (free_warm, free_end)        # flat: not a MiB lost over 4 700 steps
        rows_b, _, free_end_b = run(ctx, exhaust=True)
        assert free_end_b >= free_warm - (1 << 20)
    finally:
        ctx.close()
    assert len(rows_a) == len(rows_b) == S * (STEPS // CHUNK)
    n_int = 0
    for (ra, ia), (rb, ib) in zip(rows_a, rows_b):
        assert ra.shape == rb.shape and ra.shape[0] == CHUNK - 1
        assert np.array_equal(ra.view(np.uint64), rb.view(np.uint64)) and np.array_equal(ia, ib)
        n_int += int((ia[:, 5] & 2 != 0).sum())
    assert n_int > 0.5 * S * STEPS                                            # and it was odometry: most motions integrated
    # create / destroy
    proc = psutil.Process()
    imgs = [lefts[0], rights[0], lefts[1], rights[1]]
    pts = synth.select_keypoints(lefts[0], bucket=h // 10, per_bucket=3)
    first = None
    for it in range(50):
        c = volib.Context(0, w, h, 4096, S)
        got = c.track_frame(*imgs, pts, P_l, P_r)
        c.seq_configure(2, w, h, ring=2, max_steps=8)
        c.close()
        if first is None:
            first = got
        assert np.array_equal(got["l1"], first["l1"]) and np.array_equal(got["tvec"], first["tvec"])
        if it == 9:
            torch.cuda.synchronize()
            free10, rss10 = torch.cuda.mem_get_info(0)[0], proc.memory_info().rss
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(0)[0] >= free10 - (2 << 20)
    assert proc.memory_info().rss <= rss10 + (256 << 20), (rss10, proc.memory_info().rss)   # (a leaked context is ~10 MB of host memory: 40 of them would show)


# ------------------------------------------------------------------ ingest paths (round 6: persistent grid, copy-engine block)
def test_ingest_paths_agree_at_32_sequences(volib, small_world):
    """32 sequences -- the size from which a step of pageable pairs crosses the link as ONE copy-engine transfer of the staging
    half -- fed four ways: device memory, page-locked memory (192-wave persistent kernel), pageable memory through
    vo_seq_push_pairs (threaded staging + copy engine), pageable memory pushed pair by pair with a padded stride (staging
    repack).  Same pairs, so the trajectories and the carried feature sets must be identical bit for bit; and a step in which
    one sequence brings a page-locked pair falls back to the kernel path with the same result."""
    import torch
    S, n = 32, 6
    w, h = 480, 160
    L, R, _, _ = small_world.render_sequence(n + 3)
    P_l, P_r = small_world.proj_matrices()
    dev = torch.device("cuda", 0)

    def pair(s, k):                               # sequence s is the same street, three phases
        return L[k + s % 3], R[k + s % 3]
    results = {}
    for mode in ("device", "pinned", "pageable", "pageable_strided", "mixed"):
        ctx = volib.Context(0, w, h, 4096, S)
        try:
            ctx.batch_set_detect_params(features_per_bucket=2)
            ctx.seq_configure(S, w, h, ring=3, max_steps=16)
            ctx.batch_set_projection(P_l, P_r)
            keep = []
            for k in range(n):
                if mode == "device":
                    t = [(torch.from_numpy(np.ascontiguousarray(pair(s, k)[0])).to(dev), torch.from_numpy(np.ascontiguousarray(pair(s, k)[1])).to(dev)) for s in range(S)]
                    torch.cuda.synchronize()
                    tab = ctx.seq_pair_table(range(S), [a.data_ptr() for a, _ in t], [b.data_ptr() for _, b in t])
                    ctx.seq_push_pairs(tab, w, 2)
                elif mode == "pinned":
                    t = [(torch.from_numpy(np.ascontiguousarray(pair(s, k)[0])).pin_memory(), torch.from_numpy(np.ascontiguousarray(pair(s, k)[1])).pin_memory()) for s in range(S)]
                    tab = ctx.seq_pair_table(range(S), [a.data_ptr() for a, _ in t], [b.data_ptr() for _, b in t])
                    ctx.seq_push_pairs(tab, w, 1)
                elif mode == "pageable":
                    t = [(np.ascontiguousarray(pair(s, k)[0]), np.ascontiguousarray(pair(s, k)[1])) for s in range(S)]
                    tab = ctx.seq_pair_table(range(S), [a.ctypes.data for a, _ in t], [b.ctypes.data for _, b in t])
                    ctx.seq_push_pairs(tab, w, 0)
                elif mode == "pageable_strided":
                    t = []
                    for s in range(S):
                        pad = np.zeros((2, h, w + 37), np.uint8)
                        pad[0, :, :w], pad[1, :, :w] = pair(s, k)
                        t.append(pad)
                        ctx.seq_push_pair(s, pad[0, :, :w], pad[1, :, :w])
                else:                             # 31 pageable pairs + one page-locked one: not a block, the kernel path
                    t = [(np.ascontiguousarray(pair(s, k)[0]), np.ascontiguousarray(pair(s, k)[1])) for s in range(S - 1)]
                    for s in range(S - 1):
                        ctx.seq_push_pair(s, *t[s])
                    p = (torch.from_numpy(np.ascontiguousarray(pair(S - 1, k)[0])).pin_memory(), torch.from_numpy(np.ascontiguousarray(pair(S - 1, k)[1])).pin_memory())
                    ctx.seq_push_pair(S - 1, p[0].numpy(), p[1].numpy(), pinned=True)
                    t.append(p)
                keep.append(t)                    # (page-locked / device sources stay alive until their step has run)
                ctx.seq_step()
            ctx.seq_sync()
            traj = [ctx.seq_get_trajectory(s) for s in range(S)]
            state = [ctx.seq_get_state(s) for s in range(S)]
            results[mode] = (traj, state)
        finally:
            ctx.close()
    base_t, base_s = results["device"]
    assert all(len(r) == n - 1 for r, _ in base_t) and sum(int((i[:, 5] & 2 != 0).sum()) for _, i in base_t) > S * (n - 1) // 2
    for mode in ("pinned", "pageable", "pageable_strided", "mixed"):
        traj, state = results[mode]
        for s in range(S):
            assert np.array_equal(traj[s][0].view(np.uint64), base_t[s][0].view(np.uint64)) and np.array_equal(traj[s][1], base_t[s][1]), (mode, s)
            assert np.array_equal(state[s][0], base_s[s][0]) and np.array_equal(state[s][1], base_s[s][1]), (mode, s)
