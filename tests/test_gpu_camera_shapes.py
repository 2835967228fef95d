"""Parity (-m gpu) on the configurations nobody had run before round 3: the reference's OTHER shipped camera shapes
-- calibration/zed.yaml (1280 x 720) and calibration/rgbd.yaml (640 x 480, src/rgbd_standalone.cpp:184-196) -- at the
reference-default bucketing (bucket_size = rows / 10, 1 feature per bucket, visualOdometry.cpp:106-107), and a REAL
stereo photograph (Middlebury "Motorcycle", 741 x 500, read from scikit-image's sample data where the image installs it).

Every stage of the path through the C ABI against the checker (oracle/, and the reference's own sources in oracle/_ref):
FAST corners / bucketed set / circular-matching survivors / tracks BIT-EXACT, triangulation <= 1e-5 relative,
RANSAC control flow + inlier set identical, rvec / tvec <= 1e-6; the frame loop (vo_seq_*) bit-exact in feature state
after every frame against the reference's main() loop body."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import camera_shapes as cs  # noqa: E402

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _check_track_frame(ctx, orc, quad, pts, P_l, P_r, tag):
    l0, r0, l1, r1 = quad
    got = ctx.track_frame(l0, r0, l1, r1, pts, P_l, P_r)
    ref = orc.circular_matching(l0, r0, l1, r1, pts)
    assert np.array_equal(got["keep_idx_circ"], ref["keep_idx"]), (tag, "circular-matching survivors")
    (a, b, c, d), keep = orc.check_valid_and_remove(ref["l0"], ref["r0"], ref["l1"], ref["r1"], ref["l0_ret"])
    for name, arr in (("l0", a), ("r0", b), ("l1", c), ("r1", d)):
        assert np.array_equal(bits(got[name]), bits(arr)), (tag, name)
    if len(a) < 5:
        return got, ref, None
    xyz = orc.triangulate(P_l, P_r, a, b)
    den = np.abs(xyz).max(1, keepdims=True)
    assert np.max(np.abs(got["xyz"] - xyz) / den) <= 1e-5, (tag, "triangulation")
    rc, rv, tv, inl, dbg = orc.solve_pnp_ransac(xyz, c, np.ascontiguousarray(P_l[:, :3]))
    assert (got["rc"] == 0) == (rc == 1), (tag, "solvePnPRansac's return value")
    assert np.array_equal(got["inliers"], inl), (tag, "inlier set")
    assert np.abs(got["rvec"] - rv).max() <= 1e-6 and np.abs(got["tvec"] - tv).max() <= 1e-6, (tag, "pose")
    return got, ref, (rc, rv, tv, inl)


@pytest.mark.parametrize("name", ["zed", "rgbd"])
def test_stagewise_parity_at_the_shipped_calibrations(volib, orc, name):
    cal = cs.CALIBRATIONS[name]
    w, h = cal["width"], cal["height"]
    world = cs.world(name, seed=41)
    L, R, poses, _ = world.render_sequence(3)
    P_l, P_r = world.proj_matrices()
    ctx = volib.Context(0, w, h, 4096, 1)
    try:
        for k in (0, 1):
            quad = (L[k], R[k], L[k + 1], R[k + 1])
            # FAST (feature.cpp:39-47), with and without non-maximum suppression
            for nonmax in (True, False):
                g = ctx.fast_detect(L[k], 20, nonmax, cap=1 << 17)
                o = orc.fast_detect(L[k], 20, nonmax, cap=1 << 17)
                assert np.array_equal(g, o), (name, k, nonmax, len(g), len(o))
            fast = orc.fast_detect(L[k], 20, True)
            assert len(fast) > 1500
            # appendNewFeatures + bucketingFeatures at rows / 10 (visualOdometry.cpp:95-108), carried set of the
            # previous frame included (ages decide which feature a bucket keeps, bucket.cpp:26-51)
            carried = fast[::37][:60] + np.float32(0.25)
            ages_in = (np.arange(len(carried)) % 7 + 1).astype(np.int32)
            for fpb in (1, 3):
                gp, ga = ctx.detect_bucket(L[k], carried, ages_in, features_per_bucket=fpb)
                op = np.vstack([carried, fast])
                oa = np.concatenate([ages_in, np.zeros(len(fast), np.int32)])
                bp, ba = orc.bucketing_features(h, w, op, oa, h // 10, fpb)
                assert np.array_equal(bits(gp), bits(bp)) and np.array_equal(ga, ba), (name, k, fpb)
            # the reference-default set through the whole per-frame path
            bp, ba = orc.bucketing_features(h, w, fast, np.zeros(len(fast), np.int32), h // 10, 1)
            cells = (h // (h // 10) + 1) * (w // (h // 10) + 1)
            assert 100 < len(bp) <= cells
            got, ref, pnp = _check_track_frame(ctx, orc, quad, bp, P_l, P_r, (name, k, "default"))
            assert pnp is not None and pnp[0] == 1 and len(pnp[3]) > 30
            # ... and a denser set (6 per bucket) so that every LK tile / level path of the shape is visited
            bp6, _ = orc.bucketing_features(h, w, fast, np.zeros(len(fast), np.int32), h // 10, 6)
            _check_track_frame(ctx, orc, quad, bp6, P_l, P_r, (name, k, "6 per bucket"))
            # raw per-hop tracks and status of all four hops (the reference's four independent calls)
            ctx.set_params(lk_full_chain=1)
            cm = ctx.circular_match(*quad, bp6)
            ctx.set_params(lk_full_chain=0)
            pts_h = bp6
            st_ref = []
            for a_img, b_img in ((quad[0], quad[1]), (quad[1], quad[3]), (quad[3], quad[2]), (quad[2], quad[0])):
                pts_h, st, _ = orc.calc_optical_flow_pyr_lk(a_img, b_img, pts_h)
                st_ref.append(st)
            assert np.array_equal(cm["status4"], np.stack(st_ref)), (name, k, "status4")
    finally:
        ctx.close()


@pytest.mark.parametrize("name", ["zed", "rgbd"])
def test_frame_loop_at_the_shipped_calibrations_equals_the_reference_loop(volib, orc, name):
    """3 sequences x 12 frames at each shape through vo_seq_* against the reference's own main() loop body
    (oracle/_ref): feature state bit-exact after every frame, frame_pose <= 1e-6, ATE <= 1e-6 m"""
    from visual_odom_amd import odometry
    if orc.ref_lib() is None:
        pytest.skip("oracle/_ref was not shipped")
    cal = cs.CALIBRATIONS[name]
    w, h = cal["width"], cal["height"]
    S, N = 3, 12
    worlds = [cs.world(name, seed=200 + 11 * s) for s in range(S)]
    seqs = [wd.render_sequence(N) for wd in worlds]
    P_l, P_r = worlds[0].proj_matrices()
    ctx = volib.Context(0, w, h, 4096, S)
    try:
        vo = odometry.MultiSequenceOdometry(P_l, P_r, S, w, h, ctx=ctx, ring=3, max_steps=32)
        loops = [orc.RefFrameLoop(P_l[0, 0], P_l[0, 2], P_l[1, 2], P_r[0, 3]) for _ in range(S)]
        for k in range(N):
            for s in range(S):
                vo.push(s, seqs[s][0][k], seqs[s][1][k])
                loops[s].process(seqs[s][0][k], seqs[s][1][k])
            vo.step()
            for s in range(S):
                pts, ages, pose = vo.state(s)
                assert np.array_equal(bits(pts), bits(loops[s].points)), (name, k, s, "points")
                assert np.array_equal(ages, loops[s].ages), (name, k, s, "ages")
                assert np.abs(pose - loops[s].frame_pose).max() <= 1e-6, (name, k, s, "frame_pose")
        for s in range(S):
            traj = vo.trajectory(s)
            assert len(traj) == N and odometry.ate_rmse(traj, loops[s].trajectory) <= 1e-6
            log = vo.log(s)
            assert all(r["overflow"] == 0 for r in log) and sum(r["integrated"] for r in log) >= N - 2
            T0inv = np.linalg.inv(seqs[s][2][0])
            gt = [(T0inv @ T)[:3] for T in seqs[s][2]]
            assert odometry.ate_rmse(traj, gt) < 0.5  # and it is the planted motion
    finally:
        ctx.close()


def test_real_photograph_through_every_stage(volib, orc):
    """Middlebury Motorcycle pair (t1 = sub-pixel shift + 1.2 % zoom of the same pair): saturated / flat / specular /
    occluded regions -- min-eigenvalue rejections and lost tracks the procedural texture never produces"""
    quad = cs.real_quadruple()
    assert quad is not None, "scikit-image sample data (motorcycle_left/right.png) not found on this box"
    from visual_odom_amd import synth
    l0, r0, l1, r1 = quad
    h, w = l0.shape
    P_l, P_r = synth.proj_matrices(**cs.REAL_CALIB)
    small = volib.Context(0, w, h, 1024, 1)   # corner list of 23 156 entries (w h / 16) < the 45 905 corners below
    try:
        with pytest.raises(volib.VoError) as e:
            small.fast_detect(l0, 7, False, cap=1 << 17)
        assert e.value.code == volib.VO_ERR_OVERFLOW  # never a silently truncated list
    finally:
        small.close()
    ctx = volib.Context(0, w, h, 16384, 1)
    try:
        for img in (l0, r0, l1):
            for nonmax in (True, False):
                for thr in (20, 7):
                    assert np.array_equal(ctx.fast_detect(img, thr, nonmax, cap=1 << 17),
                                          orc.fast_detect(img, thr, nonmax, cap=1 << 17)), (nonmax, thr)
        fast = orc.fast_detect(l0, 20, True)
        for fpb in (1, 6):
            gp, ga = ctx.detect_bucket(l0, np.zeros((0, 2), np.float32), np.zeros(0, np.int32), features_per_bucket=fpb)
            bp, ba = orc.bucketing_features(h, w, fast, np.zeros(len(fast), np.int32), h // 10, fpb)
            assert np.array_equal(bits(gp), bits(bp)) and np.array_equal(ga, ba), fpb
            got, ref, pnp = _check_track_frame(ctx, orc, quad, bp, P_l, P_r, ("photo", fpb))
            if fpb == 6:
                assert len(ref["l0"]) < len(bp) and pnp is not None and pnp[0] == 1
        # every FAST corner of the photograph (4 308 points, many of them on edges and in flat regions): raw status of all
        # four hops and every track, bit for bit
        ctx.set_params(lk_full_chain=1)
        cm = ctx.circular_match(l0, r0, l1, r1, fast)
        ctx.set_params(lk_full_chain=0)
        pts_h, st_ref, trk_ref = fast, [], []
        for a_img, b_img in ((l0, r0), (r0, r1), (r1, l1), (l1, l0)):
            pts_h, st, _ = orc.calc_optical_flow_pyr_lk(a_img, b_img, pts_h)
            st_ref.append(st)
            trk_ref.append(pts_h)
        st_ref = np.stack(st_ref)
        assert np.array_equal(cm["status4"], st_ref)
        assert (st_ref == 0).sum() > 20  # the photograph does produce rejections
        ref = orc.circular_matching(l0, r0, l1, r1, fast)
        assert np.array_equal(cm["keep_idx"], ref["keep_idx"])
        for name in ("l0", "r0", "r1", "l1", "l0_ret"):
            assert np.array_equal(bits(cm[name]), bits(ref[name])), name
        # the device-side detect -> track batch path (VO_STAGE_DETECT | VO_STAGE_ALL) on the photograph
        ctx.batch_configure(4, w, h, 1)
        for i, im in enumerate(quad):
            ctx.batch_upload_image(i, im)
        ctx.batch_set_quads([[0, 1, 2, 3]])
        ctx.batch_set_features(0, np.zeros((0, 2), np.float32), np.zeros(0, np.int32))
        ctx.batch_set_detect_params(features_per_bucket=4)
        ctx.batch_set_projection(P_l, P_r)
        ctx.batch_run(volib.STAGE_ALL | volib.STAGE_DETECT)
        ctx.batch_sync()
        gp, ga = ctx.batch_get_features(0)
        bp, ba = orc.bucketing_features(h, w, fast, np.zeros(len(fast), np.int32), h // 10, 4)
        assert np.array_equal(bits(gp), bits(bp))
        flt = ctx.batch_get_filtered(0)
        ref = orc.circular_matching(l0, r0, l1, r1, bp)
        (a, b, c, d), _ = orc.check_valid_and_remove(ref["l0"], ref["r0"], ref["l1"], ref["r1"], ref["l0_ret"])
        assert np.array_equal(flt["keep_idx_circ"], ref["keep_idx"])
        for name, arr in (("l0", a), ("r0", b), ("l1", c), ("r1", d)):
            assert np.array_equal(bits(flt[name]), bits(arr)), name
        xyz = orc.triangulate(P_l, P_r, a, b)
        rc, rv, tv, inl, dbg = orc.solve_pnp_ransac(xyz, c, np.ascontiguousarray(P_l[:, :3]))
        pose = ctx.batch_get_pose(0)
        assert pose["status"] == rc and np.array_equal(pose["inliers"], inl)
        assert (pose["niters"], pose["best_iter"], pose["max_good"]) == tuple(int(x) for x in dbg[:3])
        assert np.abs(pose["rvec"] - rv).max() <= 1e-6 and np.abs(pose["tvec"] - tv).max() <= 1e-6
        ctx.batch_set_detect_params()
    finally:
        ctx.close()


def test_real_photograph_frame_loop_equals_the_reference_loop(volib, orc):
    """the photograph as a 6-frame "sequence" (progressive zoom + drift of the same pair) through vo_seq_* against
    the reference's own loop body: features carried from frame to frame, re-detection, ages"""
    from visual_odom_amd import odometry
    pair = cs.real_stereo_pair()
    assert pair is not None, "scikit-image sample data not found on this box"
    if orc.ref_lib() is None:
        pytest.skip("oracle/_ref was not shipped")
    from visual_odom_amd import synth
    l0, r0 = pair
    h, w = l0.shape
    c = (cs.REAL_CALIB["cx"], cs.REAL_CALIB["cy"])
    frames = [(l0, r0)] + [(cs.warp_subpixel(l0, 1.7 * k, -0.9 * k, 1.0 + 0.011 * k, c),
                            cs.warp_subpixel(r0, 1.7 * k, -0.9 * k, 1.0 + 0.011 * k, c)) for k in range(1, 6)]
    P_l, P_r = synth.proj_matrices(**cs.REAL_CALIB)
    ctx = volib.Context(0, w, h, 4096, 1)
    try:
        vo = odometry.MultiSequenceOdometry(P_l, P_r, 1, w, h, ctx=ctx, ring=3, max_steps=16)
        loop = orc.RefFrameLoop(P_l[0, 0], P_l[0, 2], P_l[1, 2], P_r[0, 3])
        for k, (L, R) in enumerate(frames):
            vo.push(0, L, R)
            loop.process(L, R)
            vo.step()
            pts, ages, pose = vo.state(0)
            assert np.array_equal(bits(pts), bits(loop.points)), (k, "points")
            assert np.array_equal(ages, loop.ages), (k, "ages")
            assert np.abs(pose - loop.frame_pose).max() <= 1e-6, (k, "frame_pose")
        assert len(vo.trajectory(0)) == len(frames)
    finally:
        ctx.close()
