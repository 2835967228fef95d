"""Oracle pinning, part 3: the reference's own glue (exact restatements of its sources)."""
import numpy as np


def test_check_valid_match_threshold_is_int_truncated(orc):
    # visualOdometry.cpp:46-59: int offset = max(|dx|,|dy|); reject iff offset > 0  (quirk B4)
    l0 = np.array([[10, 10], [10, 10], [10, 10], [10, 10]], np.float32)
    ret = np.array([[10.99, 10], [11.0, 10], [10, 9.01], [10.5, 8.9]], np.float32)
    z = np.zeros_like(l0)
    (a, b, c, d), valid = orc.check_valid_and_remove(l0, z, z, z, ret)
    assert valid.tolist() == [True, False, True, False] and len(a) == 2


def test_circular_matching_compaction_rules(orc, small_seq):
    s = small_seq
    pts = np.vstack([s["pts"][0], [[-3, 20], [20, -2], [1000, 50]]]).astype(np.float32)
    ages = np.arange(len(pts), dtype=np.int32)
    r = orc.circular_matching(s["L"][0], s["R"][0], s["L"][1], s["R"][1], pts, ages)
    st = r["status4"]
    raw = [orc.calc_optical_flow_pyr_lk(s["L"][0], s["R"][0], pts)]
    raw.append(orc.calc_optical_flow_pyr_lk(s["R"][0], s["R"][1], raw[0][0]))
    raw.append(orc.calc_optical_flow_pyr_lk(s["R"][1], s["L"][1], raw[1][0]))
    raw.append(orc.calc_optical_flow_pyr_lk(s["L"][1], s["L"][0], raw[2][0]))
    assert all(np.array_equal(st[k], raw[k][1]) for k in range(4))
    ok = st.all(0) & (pts >= 0).all(1)
    for k in range(3):  # pt1..pt3 sign test, NOT points0_return (feature.cpp:96-99, quirk B5)
        ok &= (raw[k][0] >= 0).all(1)
    assert np.array_equal(r["keep_idx"], np.where(ok)[0])
    assert np.array_equal(r["l0"], pts[ok]) and np.array_equal(r["l0_ret"], raw[3][0][ok])
    assert np.array_equal(r["ages"], ages[ok] + 1)  # ages += 1 first, then erased with the points
    assert r["n_out"] > 20


def test_bucketing_quirks(orc):
    rows, cols, bs = 376, 1241, 37
    # one feature in every pixel-bucket centre -> 374 emitted, 364 unique (B1, B2)
    pts = np.array([[w * bs + 5, h * bs + 5] for h in range(11) for w in range(34)
                    if w * bs + 5 < cols and h * bs + 5 < rows], np.float32)
    out, ages = orc.bucketing_features(rows, cols, pts, np.zeros(len(pts), np.int32), bs, 1)
    assert len(out) == len(ages) == 374
    assert len({tuple(p) for p in out}) == 364
    # B2': the LAST feature mapped to a full bucket replaces slot 0, whatever its age (< 10)
    pts = np.array([[5, 5], [6, 6], [7, 7]], np.float32)
    out, ages = orc.bucketing_features(rows, cols, pts, np.array([0, 5, 3], np.int32), bs, 1)
    assert np.array_equal(out[0], [7, 7]) and ages[0] == 3
    # age >= 10 is dropped (bucket.cpp:16-17)
    out, ages = orc.bucketing_features(rows, cols, pts, np.array([10, 11, 12], np.int32), bs, 1)
    assert len(out) == 0
    # features_per_bucket = 6: first six kept in order, 7th overwrites slot 0
    pts = np.array([[1 + i, 2] for i in range(7)], np.float32)
    out, ages = orc.bucketing_features(rows, cols, pts, np.arange(7, dtype=np.int32), bs, 6)
    assert np.array_equal(out[:6, 0], [7, 2, 3, 4, 5, 6])


def np_fast(img, thr):
    """brute-force FAST-9/16 + score (definition-level restatement, independent of the trick code)"""
    offs = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3),
            (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    h, w = img.shape
    score = np.zeros((h, w), np.int32)
    im = img.astype(np.int32)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            d = np.array([im[y, x] - im[y + dy, x + dx] for dx, dy in offs])
            best = -1
            for s in range(16):
                arc = np.take(d, range(s, s + 9), mode="wrap")
                best = max(best, arc.min(), (-arc).min())
            if best > thr:
                score[y, x] = best - 1
            else:
                score[y, x] = -1
    pts = []
    sc = np.where(score < 0, 0, score)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if score[y, x] < 0:
                continue
            nb = sc[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if (sc[y, x] > nb).all():
                pts.append((x, y))
    return np.array(pts, np.float32).reshape(-1, 2)


def test_fast_matches_bruteforce(orc):
    rng = np.random.default_rng(0)
    base = rng.integers(0, 256, (12, 16)).astype(np.float32)
    img = np.kron(base, np.ones((5, 5), np.float32))
    img = np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)
    got = orc.fast_detect(img, 20, True)
    ref = np_fast(img, 20)
    assert len(ref) > 10
    assert np.array_equal(got, ref)  # same set AND row-major order


def test_euler_and_integration(orc):
    from scipy.spatial.transform import Rotation
    R = Rotation.from_euler("xyz", [0.02, -0.05, 0.01]).as_matrix()
    e = orc.rotation_matrix_to_euler(R)
    assert np.allclose(e, [0.02, -0.05, 0.01], atol=1e-6)
    pose = np.eye(4)
    t = np.array([0.01, 0.0, -0.9])
    pose2, ok = orc.integrate_odometry_stereo(pose, R, t)
    assert ok
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    assert np.allclose(pose2, np.linalg.inv(T), atol=1e-14)
    _, ok = orc.integrate_odometry_stereo(pose, R, np.array([0, 0, 0.01]))
    assert not ok  # scale <= 0.05 -> pose left unchanged (utils.cpp:80-90)
