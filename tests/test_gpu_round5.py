"""Round 5 on the MI355X, through the C ABI:
  * small launches run everything behind the first RANSAC chunk in ONE launch (ransac_rest_kernel: every workgroup draws the
    chunk's subsets, solves and votes its 64 hypotheses, the one that arrives last replays the control flow) and solve the
    four-point frames inside the first replay kernel: control flow, inlier sets and poses against the oracle for frames that do
    and do not reach past 128 hypotheses, alone and side by side in one launch;
  * the synchronous call's input path (images pulled out of pinned staging by a kernel each, the points riding with the last
    one, no synchronisation before the results) against the batch API's path: identical results, also with strided images,
    zero points, and right after a run that left a VO_STAGE_DETECT feature set current."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _hard_problem(orc, seed, n=90, outliers=0.62):
    from test_oracle_geom import planted_problem
    return planted_problem(orc, n, outliers, 0.2, seed)


def test_rest_of_the_solve_in_one_launch_matches_the_oracle(volib, orc):
    """a frame whose adaptive iteration count reaches past the first chunk (62 % outliers: OpenCV goes on for several hundred
    iterations) through the single-frame call (the four-kernel first chunk + ransac_rest_kernel), alternating with an easy frame
    that leaves the rest kernel at once: inliers and pose against the oracle, call after call"""
    from test_oracle_geom import planted_problem, K_KITTI
    hard = [_hard_problem(orc, s) for s in (3, 11)]
    want = [orc.solve_pnp_ransac(X, uv, K_KITTI) for X, uv, _, _, _ in hard]
    assert all(int(w[4][0]) > 128 for w in want), [int(w[4][0]) for w in want]      # they do need the second chunk
    ctx = volib.Context(0, 640, 480, 2048, 4)
    try:
        for (X, uv, _, _, _), (rc, rv, tv, inl, dbg) in zip(hard, want):
            found, grv, gtv, gR, ginl = ctx.pnp_ransac(X, uv, K_KITTI)
            assert found == (rc == 1) and np.array_equal(ginl, inl)
            assert np.abs(grv - rv).max() <= 1e-6 and np.abs(gtv - tv).max() <= 1e-6
        # the same call again and an easy frame after it: the arrival counter of the frame's state is back at zero
        Xe, uve, _, _, _ = planted_problem(orc, 400, 0.1, 0.1, 5)
        rce, rve, tve, inle, _ = orc.solve_pnp_ransac(Xe, uve, K_KITTI)
        for _ in range(2):
            found, grv, gtv, _, ginl = ctx.pnp_ransac(Xe, uve, K_KITTI)
            assert found == (rce == 1) and np.array_equal(ginl, inle) and np.abs(grv - rve).max() <= 1e-6
            found, grv, gtv, _, ginl = ctx.pnp_ransac(hard[0][0], hard[0][1], K_KITTI)
            assert np.array_equal(ginl, want[0][3]) and np.abs(gtv - want[0][2]).max() <= 1e-6
    finally:
        ctx.close()


def test_track_frame_input_path_equals_the_batch_path(volib, small_world, small_seq):
    """vo_track_frame (sync-free inputs: pull kernels, points with the last image, everything on one stream without events)
    against the same frame through vo_batch_* (hipMemcpyAsync uploads, vo_batch_set_points, the batch streams): every output
    identical -- contiguous and strided images, no points at all, and after a DETECT run whose feature set is current"""
    L, R = small_seq["L"], small_seq["R"]
    pts = small_seq["pts"][0]
    P_l, P_r = small_world.proj_matrices()
    h, w = L[0].shape
    ctx = volib.Context(0, w + 64, h + 8, 2048, 2)
    ref = volib.Context(0, w + 64, h + 8, 2048, 2)
    try:
        def batch(imgs, p):
            ref.batch_configure(4, w, h, 1)
            for i, im in enumerate(imgs):
                ref.batch_upload_image(i, im)
            ref.batch_set_quads([[0, 1, 2, 3]])
            ref.batch_set_points(0, p)
            ref.batch_set_projection(P_l, P_r)
            ref.batch_run(volib.STAGE_ALL)
            ref.batch_sync()
            return ref.batch_get_filtered(0), ref.batch_get_pose(0)

        def same(got, flt, pose):
            for k in ("l0", "r0", "l1", "r1"):
                assert np.array_equal(got[k], flt[k]), k
            assert np.array_equal(got["xyz"], flt["xyz"]) and np.array_equal(got["keep_idx"], flt["keep_idx"])
            assert np.array_equal(got["inliers"], pose["inliers"])
            assert np.array_equal(got["rvec"], pose["rvec"]) and np.array_equal(got["tvec"], pose["tvec"])

        imgs = [L[0], R[0], L[1], R[1]]
        flt, pose = batch(imgs, pts)
        assert len(pose["inliers"]) > 20
        for _ in range(3):                                    # (the first call probes the schedule, the later ones do not)
            same(ctx.track_frame(*imgs, pts, P_l, P_r), flt, pose)
        # strided views of a padded buffer: the staging repack takes the rows apart
        pad = [np.zeros((h + 3, w + 37), np.uint8) for _ in range(4)]
        for p, im in zip(pad, imgs):
            p[2:2 + h, 5:5 + w] = im
        views = [p[2:2 + h, 5:5 + w] for p in pad]
        same(ctx.track_frame(*views, pts, P_l, P_r), flt, pose)
        # a different frame right behind it (the staging slots and the pinned points are reused call after call)
        imgs2 = [L[1], R[1], L[2], R[2]]
        flt2, pose2 = batch(imgs2, small_seq["pts"][1])
        same(ctx.track_frame(*imgs2, small_seq["pts"][1], P_l, P_r), flt2, pose2)
        same(ctx.track_frame(*imgs, pts, P_l, P_r), flt, pose)
        # no points: nothing tracked, CV_Assert(npoints >= 4) of solvePnPRansac as the call's error code
        got = ctx.track_frame(*imgs, np.zeros((0, 2), np.float32), P_l, P_r)
        assert got["rc"] == volib.VO_ERR_TOO_FEW and len(got["l0"]) == 0 and len(got["inliers"]) == 0
        # a DETECT run leaves its bucketed set current (pts_sel >= 0): the call takes the general path for the points
        ctx.batch_configure(4, w, h, 1)
        for i, im in enumerate(imgs):
            ctx.batch_upload_image(i, im)
        ctx.batch_set_quads([[0, 1, 2, 3]])
        ctx.batch_set_features(0, np.zeros((0, 2), np.float32), np.zeros(0, np.int32))
        ctx.batch_set_projection(P_l, P_r)
        ctx.batch_run(volib.STAGE_PYRAMID | volib.STAGE_DETECT)
        ctx.batch_sync()
        same(ctx.track_frame(*imgs, pts, P_l, P_r), flt, pose)
        same(ctx.track_frame(*imgs, pts, P_l, P_r), flt, pose)
    finally:
        ctx.close()
        ref.close()


def test_three_frames_one_small_launch_at_a_tight_threshold(volib, orc, small_world, small_seq):
    """three frames in ONE small launch (the four-kernel first chunk, then ransac_rest_kernel over all of them) at a reprojection
    threshold that leaves few inliers: frames that go on past the first chunk next to frames that may not (a sparse 40-point
    frame among them) -- RANSAC counters, inlier sets and poses of each against the oracle at the same threshold.  (The
    four-point frames, P3P inside the first replay kernel: tests/test_gpu_round3.py.)"""
    L, R = small_seq["L"], small_seq["R"]
    P_l, P_r = small_world.proj_matrices()
    h, w = L[0].shape
    pts = small_seq["pts"][0]
    rng = np.random.default_rng(17)
    ctx = volib.Context(0, w, h, 2048, 4)
    try:
        ctx.set_params(ransac_reproj_error=0.12)               # a tight threshold: few inliers, hundreds of iterations
        ctx.batch_configure(6, w, h, 3)
        for i, im in enumerate([L[0], R[0], L[1], R[1], L[2], R[2]]):
            ctx.batch_upload_image(i, im)
        ctx.batch_set_quads([[0, 1, 2, 3], [2, 3, 4, 5], [0, 1, 2, 3]])
        ctx.batch_set_points(0, pts)
        ctx.batch_set_points(1, small_seq["pts"][1])
        ctx.batch_set_points(2, pts[rng.permutation(len(pts))[:40]])
        ctx.batch_set_projection(P_l, P_r)
        ctx.batch_run(volib.STAGE_ALL)
        ctx.batch_sync()
        K = small_world.K()
        iters = []
        for f in range(3):
            flt, pose = ctx.batch_get_filtered(f), ctx.batch_get_pose(f)
            if len(flt["l0"]) < 4:
                continue
            xyz = orc.triangulate(P_l, P_r, flt["l0"], flt["r0"])
            rc, rv, tv, inl, dbg = orc.solve_pnp_ransac(xyz, flt["l1"], K, reproj=0.12)
            assert pose["status"] == rc and np.array_equal(pose["inliers"], inl), f
            assert (pose["niters"], pose["best_iter"], pose["max_good"]) == tuple(int(x) for x in dbg[:3]), f
            if rc == 1:
                assert np.abs(pose["rvec"] - rv).max() <= 1e-6 and np.abs(pose["tvec"] - tv).max() <= 1e-6, f
            iters.append(pose["niters"])
        assert max(iters) > 128, iters                         # at least one frame went through the rest kernel's work
    finally:
        ctx.close()


def test_one_sequence_on_the_partitioned_streams_equals_two_sequences(volib, small_world):
    """vo_seq_configure(1) runs on the CU-partitioned twin of the context's streams (post-LK streams and tracking streams on
    disjoint halves of the compute units), every other configuration on the plain set: the same pairs give the same state and
    trajectory bit for bit through 1 sequence, 2 sequences and 1 sequence again on ONE context, and a synchronous call between
    the modes (back on the plain set) equals a fresh context's"""
    from visual_odom_amd import synth
    L, R, _, _ = small_world.render_sequence(7)
    pts0 = synth.select_keypoints(L[0], bucket=16, per_bucket=2)
    P_l, P_r = small_world.proj_matrices()
    h, w = L[0].shape
    ctx = volib.Context(0, w, h, 2048, 2)
    fresh = volib.Context(0, w, h, 2048, 2)
    try:
        def loop(n_seq):
            ctx.seq_configure(n_seq, w, h, 3, 64)
            ctx.batch_set_projection(P_l, P_r)
            for k in range(7):
                for s in range(n_seq):
                    ctx.seq_push_pair(s, L[k], R[k])
                ctx.seq_step()
            ctx.seq_sync()
            out = []
            for s in range(n_seq):
                p, a, pose = ctx.seq_get_state(s)
                rows, info = ctx.seq_get_trajectory(s)
                out.append((p, a, pose, rows, info))
            return out

        one = loop(1)
        imgs = [L[0], R[0], L[1], R[1]]
        want = fresh.track_frame(*imgs, pts0, P_l, P_r)
        got = ctx.track_frame(*imgs, pts0, P_l, P_r)
        for k in ("l0", "r0", "l1", "r1", "xyz", "inliers", "rvec", "tvec"):
            assert np.array_equal(got[k], want[k]), k
        two = loop(2)
        again = loop(1)
        assert len(one[0][3]) == 6          # 7 pairs: 6 motions
        for a, b in ((one[0], two[0]), (one[0], two[1]), (one[0], again[0])):
            for x, y in zip(a, b):
                assert x.shape == y.shape and x.tobytes() == y.tobytes()
    finally:
        ctx.close()
        fresh.close()


def test_kept_pair_calls_equal_four_image_calls(volib, small_world):
    """the drop-in calls without t0 images (the previous call's t1 pair is this call's t0 pair, kept on the device with its
    pyramids -- main.cpp:157-158) against the same calls with all four images: every output of vo_track_frame /
    vo_circular_match / vo_detect_bucket / vo_fast_detect identical frame after frame, strided t1 images, a detection on an
    image of its own in between, and VO_ERR_STATE / VO_ERR_ARG wherever no such pair exists"""
    from visual_odom_amd import synth
    L, R, _, _ = small_world.render_sequence(6)
    P_l, P_r = small_world.proj_matrices()
    h, w = L[0].shape
    keys = ("l0", "r0", "l1", "r1", "xyz", "keep_idx", "keep_idx_circ", "inliers", "rvec", "tvec", "R")
    ctx = volib.Context(0, w + 40, h + 8, 2048, 1)
    ref = volib.Context(0, w + 40, h + 8, 2048, 1)
    try:
        with pytest.raises(volib.VoError) as e:                 # nothing kept yet
            ctx.track_frame(None, None, L[1], R[1], np.zeros((4, 2), np.float32), P_l, P_r)
        assert e.value.code == volib.VO_ERR_STATE
        pts, ages = np.zeros((0, 2), np.float32), np.zeros(0, np.int32)
        for k in range(5):
            t0 = (L[k], R[k]) if k == 0 else (None, None)
            if k == 3:                                           # a padded buffer's views as the new pair
                pad = [np.zeros((h + 5, w + 29), np.uint8) for _ in range(2)]
                pad[0][3:3 + h, 7:7 + w], pad[1][3:3 + h, 7:7 + w] = L[k + 1], R[k + 1]
                t1 = (pad[0][3:3 + h, 7:7 + w], pad[1][3:3 + h, 7:7 + w])
            else:
                t1 = (L[k + 1], R[k + 1])
            # detection + bucketing on the kept pair's left image against the same on the image itself
            want_p, want_a = ref.detect_bucket(L[k], pts, ages, features_per_bucket=2)
            got_p, got_a = ctx.detect_bucket(None if k else L[k], pts, ages, features_per_bucket=2)
            assert np.array_equal(got_p, want_p) and np.array_equal(got_a, want_a) and len(got_p) > 50, k
            if k == 2:    # FAST alone on the kept image, and a detection on some other image: the kept pair stays
                assert np.array_equal(ctx.fast_detect(None), ref.fast_detect(L[k]))
                other_p, _ = ctx.detect_bucket(R[4], pts, ages, features_per_bucket=2)
                assert np.array_equal(other_p, ref.detect_bucket(R[4], pts, ages, features_per_bucket=2)[0])
            want = ref.track_frame(L[k], R[k], L[k + 1], R[k + 1], want_p, P_l, P_r)
            got = ctx.track_frame(*t0, *t1, got_p, P_l, P_r)
            assert got["rc"] == want["rc"] and len(got["inliers"]) > 20, k
            for key in keys:
                assert np.array_equal(got[key], want[key]), (k, key)
            pts, ages = want["l1"].copy(), (want_a + 1)[want["keep_idx_circ"]]
        # circularMatching alone: frame 4 -> 5 on the kept pair, raw status and survivors
        wantc = ref.circular_match(L[5], R[5], L[4], R[4], pts)
        gotc = ctx.circular_match(None, None, L[4], R[4], pts)
        for key in ("l0", "r0", "r1", "l1", "l0_ret", "status4", "keep_idx"):
            assert np.array_equal(gotc[key], wantc[key]), key
        # one of the two t0 images missing: an argument error; the batch API / another size / the sequence loop take the
        # image table over: a state error until a call with four images has run again
        with pytest.raises(volib.VoError) as e:
            ctx.track_frame(None, R[4], L[5], R[5], pts, P_l, P_r)
        assert e.value.code == volib.VO_ERR_ARG
        ctx.track_frame(None, None, L[5], R[5], pts, P_l, P_r)                       # (still fine after the refused call)
        ctx.batch_upload_image(1, R[0])
        for call in (lambda: ctx.track_frame(None, None, L[1], R[1], pts, P_l, P_r),
                     lambda: ctx.detect_bucket(None, pts, ages), lambda: ctx.fast_detect(None)):
            with pytest.raises(volib.VoError) as e:
                call()
            assert e.value.code == volib.VO_ERR_STATE
        ctx.track_frame(L[0], R[0], L[1], R[1], pts, P_l, P_r)
        ctx.track_frame(None, None, L[2], R[2], pts, P_l, P_r)
        with pytest.raises(volib.VoError) as e:                                     # another size
            ctx.track_frame(None, None, L[3][:h - 8], R[3][:h - 8], pts, P_l, P_r)
        assert e.value.code == volib.VO_ERR_STATE
        ctx.track_frame(None, None, L[3], R[3], pts, P_l, P_r)                       # (the refused call changed nothing)
        ctx.seq_configure(1, w, h, 2, 8)                                             # ring 2 x 1 sequence: the same 4 images
        with pytest.raises(volib.VoError) as e:
            ctx.track_frame(None, None, L[4], R[4], pts, P_l, P_r)
        assert e.value.code == volib.VO_ERR_STATE
        got = ctx.track_frame(L[4], R[4], L[5], R[5], want_p, P_l, P_r)
        for key in keys:
            assert np.array_equal(got[key], want[key]), key
    finally:
        ctx.close()
        ref.close()


def test_frame_loop_with_the_kept_pair_equals_the_other_loops(volib, small_world):
    """StereoOdometry over 7 pairs: the drop-in calls with the kept pair (two uploads per frame), the stateless calls (four)
    and the streaming ring give the same trajectory, feature sets and per-frame records"""
    from visual_odom_amd import odometry
    L, R, _, _ = small_world.render_sequence(7)
    P_l, P_r = small_world.proj_matrices()
    h, w = L[0].shape
    runs = []
    for kw in (dict(streaming=False, keep_pair=True), dict(streaming=False, keep_pair=False), dict(streaming=True)):
        vo = odometry.StereoOdometry(P_l, P_r, 0, w, h, 2048, features_per_bucket=2, **kw)
        try:
            for k in range(7):
                vo.process(L[k], R[k])
            runs.append((np.array(vo.trajectory), vo.points.copy(), vo.ages.copy(),
                         [(r["n_bucketed"], r["n_tracked"], r["n_inliers"], r["integrated"]) for r in vo.log]))
        finally:
            vo.close()
    assert len(runs[0][0]) == 7 and runs[0][3][0][2] > 20
    for other in runs[1:]:
        assert runs[0][0].tobytes() == other[0].tobytes()
        assert np.array_equal(runs[0][1], other[1]) and np.array_equal(runs[0][2], other[2]) and runs[0][3] == other[3]
