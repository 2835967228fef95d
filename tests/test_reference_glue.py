"""oracle/_ref: the reference's own src/feature.cpp, bucket.cpp, visualOdometry.cpp and utils.cpp, compiled where they
lie against a stand-in for the OpenCV declarations (oracle/ref_shim), with the OpenCV algorithms forwarding to the
oracle's restatement.  These tests pin the oracle's RESTATED glue (oracle/orc_glue.c: call order of the four LK hops,
deleteUnmatchFeaturesCircle's erase / age semantics, Bucket::add_feature, bucketingFeatures' aliased indexing and
duplicate emission, appendNewFeatures) against the real sources on identical inputs.  OpenCV's own arithmetic is
not part of what is pinned here (it is absent from the reference tree: parity of LK / FAST stays unpinned)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref(orc):
    if orc.ref_lib() is None:
        pytest.skip("oracle/_ref is only buildable where /root/reference exists and was not shipped")
    return orc


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_circular_matching_restatement_equals_the_reference_source(ref, small_seq):
    s = small_seq
    args = (s["L"][0], s["R"][0], s["L"][1], s["R"][1])
    # in-image points, points at / beyond the borders (negative results, status 0), and an ages array that is
    # LONGER than the points array (quirk B3: the tail survives untouched apart from the += 1)
    border = np.array([[0, 0], [479, 159], [2.5, 80.25], [-5, 50], [100, -3], [520, 100], [240, 185], [476.2, 10.7]], np.float32)
    pts = np.vstack([s["pts"][0][:150], border]).astype(np.float32)
    rng = np.random.default_rng(2)
    for ages in (None, rng.integers(0, 12, len(pts)).astype(np.int32), rng.integers(0, 12, len(pts) + 17).astype(np.int32)):
        a = ref.circular_matching(*args, pts, ages=ages)
        b = ref.ref_circular_matching(*args, pts, ages=ages)
        assert a["n_out"] == b["n_out"] and 0 < a["n_out"] < len(pts)
        for k in ("l0", "r0", "r1", "l1", "l0_ret"):
            assert np.array_equal(bits(a[k]), bits(b[k])), k
        assert np.array_equal(a["ages"], b["ages"])


def test_circular_matching_degenerate_inputs(ref, small_seq):
    s = small_seq
    args = (s["L"][0], s["R"][0], s["L"][1], s["R"][1])
    for pts in (np.zeros((0, 2), np.float32), np.array([[-50, -50]], np.float32), s["pts"][0][:1]):
        a = ref.circular_matching(*args, pts)
        b = ref.ref_circular_matching(*args, pts)
        assert a["n_out"] == b["n_out"]
        assert np.array_equal(bits(a["l1"]), bits(b["l1"])) and np.array_equal(a["ages"], b["ages"])


@pytest.mark.parametrize("rows,cols,bucket,fpb", [(376, 1241, 37, 1), (376, 1241, 37, 6), (160, 480, 16, 2), (97, 131, 10, 3)])
def test_bucketing_restatement_equals_the_reference_source(ref, rows, cols, bucket, fpb):
    rng = np.random.default_rng(rows + fpb)
    for n, extra in ((0, 0), (40, 0), (3000, 0), (3000, 25)):
        # points inside the image (the reference indexes outside its bucket vector otherwise), many per bucket,
        # ages spanning the >= 10 cut-off, optionally more ages than points (B3)
        pts = np.c_[rng.uniform(0, cols - 1e-3, n), rng.uniform(0, rows - 1e-3, n)].astype(np.float32)
        pts[: n // 10, 0] = cols - 1 - rng.uniform(0, bucket, n // 10).astype(np.float32) * 0.99  # last column: aliased buckets
        ages = rng.integers(0, 14, n + extra).astype(np.int32)
        p1, a1 = ref.bucketing_features(rows, cols, pts, ages, bucket, fpb)
        p2, a2 = ref.ref_bucketing_features(rows, cols, pts, ages, bucket, fpb)
        assert np.array_equal(bits(p1), bits(p2)) and np.array_equal(a1, a2)


def test_append_new_features_equals_the_reference_source(ref, small_seq):
    img = small_seq["L"][0]
    rng = np.random.default_rng(4)
    carried = np.c_[rng.uniform(0, img.shape[1], 30), rng.uniform(0, img.shape[0], 30)].astype(np.float32)
    ages = rng.integers(0, 9, 30).astype(np.int32)
    p2, a2 = ref.ref_append_new_features(img, carried, ages)
    fast = ref.fast_detect(img, 20, True)
    assert len(fast) > 50
    assert np.array_equal(bits(p2), bits(np.vstack([carried, fast])))
    assert np.array_equal(a2, np.concatenate([ages, np.zeros(len(fast), np.int32)]))


def _oracle_chain_step(orc, state, l0, r0, l1, r1, P_l, P_r, K, mono=False):
    """the oracle's restated frame step (the chain tests/test_gpu_parity.py holds the product to)"""
    h, w = l0.shape
    if len(state["pts"]) < 2000:
        fast = orc.fast_detect(l0, 20, True)
        state["pts"] = np.vstack([state["pts"], fast])
        state["ages"] = np.concatenate([state["ages"], np.zeros(len(fast), np.int32)])
    bp, ba = orc.bucketing_features(h, w, state["pts"], state["ages"], h // 10, 1)
    cm = orc.circular_matching(l0, r0, l1, r1, bp, ages=ba)
    (pl0, pr0, pl1, pr1), _ = orc.check_valid_and_remove(cm["l0"], cm["r0"], cm["l1"], cm["r1"], cm["l0_ret"])
    state["pts"], state["ages"] = pl1, cm["ages"]
    xyz = orc.triangulate(P_l, P_r, pl0, pr0)
    rc, rv, tv, inl, _ = orc.solve_pnp_ransac(xyz, pl1, K, tvec=state["t"])
    state["t"] = tv
    Rm = orc.rodrigues(rv)
    if mono:
        focal, pp = float(P_l[0, 0]), (float(P_l[0, 2]), float(P_l[1, 2]))
        ok, E, mask, _ = orc.find_essential_mat(pl0, pl1, focal, pp)
        assert ok == 1
        _, Rm, _, _ = orc.recover_pose(E, pl0, pl1, focal, pp, mask)
    e = orc.rotation_matrix_to_euler(Rm)
    integrated = False
    if abs(e[1]) < 0.1 and abs(e[0]) < 0.1 and abs(e[2]) < 0.1:
        state["pose"], integrated = orc.integrate_odometry_stereo(state["pose"], Rm, tv)
    return dict(l0=pl0, r0=pr0, l1=pl1, r1=pr1, tvec=tv, R=Rm, integrated=integrated)


@pytest.mark.parametrize("mono", [False, True])
def test_frame_loop_restatement_equals_the_reference_sources(ref, small_world, mono):
    """matchingFeatures + triangulation + trackingFrame2Frame + euler gates + integrateOdometryStereo: the
    reference's own functions (oracle/_ref) and the oracle's restated chain, frame after frame on one sequence"""
    n = 5
    L, R, poses, _ = small_world.render_sequence(n)
    P_l, P_r = small_world.proj_matrices()
    K = small_world.K()
    loop = ref.RefFrameLoop(P_l[0, 0], P_l[0, 2], P_l[1, 2], P_r[0, 3], mono_rotation=mono)
    state = dict(pts=np.zeros((0, 2), np.float32), ages=np.zeros(0, np.int32), t=np.zeros(3), pose=np.eye(4))
    loop.process(L[0], R[0])
    for k in range(1, n):
        a = loop.process(L[k], R[k])
        b = _oracle_chain_step(ref, state, L[k - 1], R[k - 1], L[k], R[k], P_l, P_r, K, mono)
        for name in ("l0", "r0", "l1", "r1"):
            assert np.array_equal(bits(a[name]), bits(b[name])), (k, name)
        assert np.array_equal(bits(loop.points), bits(state["pts"])) and np.array_equal(loop.ages, state["ages"])
        assert np.array_equal(a["tvec"], b["tvec"]) and np.abs(a["R"] - b["R"]).max() == 0
        assert a["integrated"] == b["integrated"] and a["integrated"]
        assert np.abs(loop.frame_pose - state["pose"]).max() <= 1e-12   # different 4x4 inverse routines
    assert len(a["l1"]) > 20


def test_euler_and_integration_equal_the_reference_sources(ref):
    rng = np.random.default_rng(6)
    pose = np.eye(4)
    for _ in range(50):
        rv = rng.normal(0, 0.05, 3)
        Rm = ref.rodrigues(rv)
        t = rng.normal(0, 1, 3) * rng.choice([0.01, 1.0, 20.0])     # below / inside / above the scale gate
        assert np.array_equal(ref.rotation_matrix_to_euler(Rm), ref.ref_rotation_matrix_to_euler(Rm))
        p_ref = ref.ref_integrate_odometry_stereo(pose, Rm, t)
        p_orc, applied = ref.integrate_odometry_stereo(pose, Rm, t)
        assert applied == (0.05 < np.linalg.norm(t) < 10)
        assert np.abs(p_ref - p_orc).max() <= 1e-12 * max(1.0, np.abs(p_orc).max())
        pose = p_orc
