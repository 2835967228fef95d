"""Known-answer vectors (tests/golden/oracle_kat.npz, written by tools/make_golden.py).  Provenance: the repository's
own CPU oracle -- NOT OpenCV, which is unavailable here.  The CPU test keeps the oracle from
drifting (every parity claim is relative to it); the GPU test runs the product on the committed inputs."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def kat():
    return np.load(os.path.join(ROOT, "tests", "golden", "oracle_kat.npz"))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_oracle_reproduces_its_known_answers(orc, kat):
    L0, R0, L1, R1, pts = kat["l0"], kat["r0"], kat["l1"], kat["r1"], kat["pts"]
    for l, lvl in enumerate(orc.build_pyramid(L0, 3)[1:], 1):
        assert np.array_equal(lvl, kat["pyr_l0_level%d" % l])
    assert np.array_equal(orc.scharr(L0), kat["scharr_l0"])
    p = pts
    for hop, (a, b) in enumerate([(L0, R0), (R0, R1), (R1, L1), (L1, L0)]):
        p, st, _ = orc.calc_optical_flow_pyr_lk(a, b, p)
        assert np.array_equal(bits(p), bits(kat["lk_hop%d" % hop])) and np.array_equal(st, kat["lk_status%d" % hop])
    cm = orc.circular_matching(L0, R0, L1, R1, pts)
    (l0, r0, l1, r1), _ = orc.check_valid_and_remove(cm["l0"], cm["r0"], cm["l1"], cm["r1"], cm["l0_ret"])
    assert np.array_equal(cm["keep_idx"], kat["keep_idx"]) and np.array_equal(bits(l1), bits(kat["f_l1"]))
    xyz = orc.triangulate(kat["P_l"], kat["P_r"], l0, r0)
    assert np.array_equal(bits(xyz), bits(kat["xyz"]))
    rc, rv, tv, inl, _ = orc.solve_pnp_ransac(xyz, l1, kat["K"])
    assert np.array_equal(inl, kat["inliers"]) and np.abs(rv - kat["rvec"]).max() <= 1e-12 and np.abs(tv - kat["tvec"]).max() <= 1e-12
    focal, pp = float(kat["P_l"][0, 0]), (float(kat["P_l"][0, 2]), float(kat["P_l"][1, 2]))
    ok, E, mask, _ = orc.find_essential_mat(l0, l1, focal, pp)
    good, Rm, tm, m2 = orc.recover_pose(E, l0, l1, focal, pp, mask)
    assert np.abs(E - kat["E"]).max() <= 1e-12 and np.array_equal(m2, kat["em_mask"]) and np.abs(Rm - kat["R_mono"]).max() <= 1e-12
    fast = orc.fast_detect(L0, 20, True)
    assert np.array_equal(bits(fast), bits(kat["fast_l0"]))
    bp, ba = orc.bucketing_features(L0.shape[0], L0.shape[1], fast, np.zeros(len(fast), np.int32), L0.shape[0] // 10, 1)
    assert np.array_equal(bits(bp), bits(kat["bucket_pts"])) and np.array_equal(ba, kat["bucket_ages"])


@pytest.mark.gpu
def test_product_reproduces_the_known_answers(gpu_ctx, kat):
    L0, R0, L1, R1, pts = kat["l0"], kat["r0"], kat["l1"], kat["r1"], kat["pts"]
    got = gpu_ctx.track_frame(L0, R0, L1, R1, pts, kat["P_l"], kat["P_r"])
    assert np.array_equal(got["keep_idx_circ"], kat["keep_idx"])
    for name in ("l0", "r0", "l1", "r1"):
        assert np.array_equal(bits(got[name]), bits(kat["f_" + name])), name
    assert np.max(np.abs(got["xyz"] - kat["xyz"]) / np.abs(kat["xyz"]).max(1, keepdims=True)) <= 1e-5
    assert np.array_equal(got["inliers"], kat["inliers"])
    assert np.abs(got["rvec"] - kat["rvec"]).max() <= 1e-6 and np.abs(got["tvec"] - kat["tvec"]).max() <= 1e-6
    focal, pp = float(kat["P_l"][0, 0]), (float(kat["P_l"][0, 2]), float(kat["P_l"][1, 2]))
    found, E, R, t, mask, good = gpu_ctx.essential_pose(kat["f_l0"], kat["f_l1"], focal, pp)
    assert found and np.abs(E - kat["E"]).max() <= 1e-9 and np.array_equal(mask, kat["em_mask"])
    assert np.abs(R - kat["R_mono"]).max() <= 1e-9
    fast = gpu_ctx.fast_detect(L0, 20, True)
    assert np.array_equal(bits(fast), bits(kat["fast_l0"]))
    bp, ba = gpu_ctx.detect_bucket(L0, np.zeros((0, 2), np.float32), np.zeros(0, np.int32))
    assert np.array_equal(bits(bp), bits(kat["bucket_pts"])) and np.array_equal(ba, kat["bucket_ages"])


def test_oracle_against_opencv_golden_if_present(orc):
    """tools/opencv_crosscheck.py --write-golden, run on a machine that has cv2, leaves tests/golden/opencv_<version>.npz:
    cv2's own outputs on the seeded synthetic pair.  When such a file is committed the oracle is held to it here (the bars
    of DESIGN.md section 6); this container has no OpenCV, so without the file the test skips and parity stays unpinned."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "opencv_*.npz")))
    if not files:
        pytest.skip("no OpenCV-provenance golden file: tools/opencv_crosscheck.py --write-golden has not been run where cv2 exists")
    from visual_odom_amd import synth
    for path in files:
        g = np.load(path)
        world = synth.StereoWorld(seed=int(g["seed"]), width=1241, height=376, fx=718.856, cx=607.1928, cy=185.2157, bf=-386.1448)
        L, R, _, _ = world.render_sequence(2)
        pts = synth.select_keypoints(L[0], bucket=37, per_bucket=6)
        assert np.array_equal(pts, g["pts"]), "the synthetic inputs changed since %s was written" % path
        assert np.array_equal(orc.fast_detect(L[0], 20, True), g["fast"])
        p = pts
        for hop, (a, b) in enumerate([(L[0], R[0]), (R[0], R[1]), (R[1], L[1]), (L[1], L[0])]):
            q, st, _ = orc.calc_optical_flow_pyr_lk(a, b, p)
            both = (st == 1) & (g["lk_status%d" % hop] == 1)
            assert (st != g["lk_status%d" % hop]).mean() <= 0.002, hop
            assert np.abs(q[both] - g["lk_hop%d" % hop][both]).max() <= 1e-3, hop   # f32 SIMD accumulation in x86 builds
            p = q
        P_l, P_r = world.proj_matrices()
        cm = orc.circular_matching(L[0], R[0], L[1], R[1], pts)
        (l0, r0, l1, r1), _ = orc.check_valid_and_remove(cm["l0"], cm["r0"], cm["l1"], cm["r1"], cm["l0_ret"])
        xyz = orc.triangulate(P_l, P_r, l0, r0)
        assert (np.abs(xyz - g["xyz"]) / np.abs(xyz).max(1, keepdims=True)).max() <= 1e-5
        rc, rv, tv, _, _ = orc.solve_pnp_ransac(xyz, l1, world.K())
        assert rc == 1 and np.abs(rv - g["rvec"]).max() <= 1e-6 and np.abs(tv - g["tvec"]).max() <= 1e-6
