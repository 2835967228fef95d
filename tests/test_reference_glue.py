"""oracle/_ref: the reference's own src/feature.cpp + src/bucket.cpp, compiled where they lie against a type-only
OpenCV stand-in (oracle/ref_shim), with cv::FAST and cv::calcOpticalFlowPyrLK forwarding to the oracle's
restatement.  These tests pin the oracle's RESTATED glue (oracle/orc_glue.c: call order of the four LK hops,
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
