#!/usr/bin/env python3
"""Close the loop on "parity unpinned": diff the CPU oracle (oracle/*.c, a restatement of OpenCV 4.5.x written without
OpenCV at hand) against a REAL OpenCV, on machines that have one (`import cv2`).  Neither the authoring container nor the
GPU boxes have one by any route -- import under every interpreter, pkg-config, a filesystem search for libopencv* / cv2*.so
/ wheels, pip against a package index (PIP_NO_INDEX=1, no DNS, connections refused): tools/opencv_probe.sh, log of the
round-5 GPU-box run in profiles/r05_opencv_probe.txt -- so this tool is shipped unexecuted; it exits with code 2 and a
message when cv2 is missing.

    python tools/opencv_crosscheck.py                 # diff every call of the hot path, print a table, exit 0 / 1
    python tools/opencv_crosscheck.py --write-golden  # also dump cv2's outputs to tests/golden/opencv_<version>.npz

Calls diffed (the reference's call sites): cv2.pyrDown (inside buildOpticalFlowPyramid), cv2.calcOpticalFlowPyrLK as
feature.cpp:136-139 calls it (win 21, maxLevel 3, 30 / 0.01, minEig 1e-3), cv2.FAST(20, nonmax), cv2.triangulatePoints +
convertPointsFromHomogeneous (main.cpp:170-171), cv2.solvePnPRansac + Rodrigues (visualOdometry.cpp:176,188; also with exactly 4 points = the P3P switch) and
cv2.findEssentialMat + recoverPose (visualOdometry.cpp:152-153).  Expected against x86 OpenCV: pyramids and FAST
bit-exact, LK positions within ~1e-3 px with identical status (OpenCV accumulates the 2x2 system in f32 SIMD lanes, the
oracle exactly), triangulation 1e-5 relative, poses 1e-6 when the inlier sets match."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write-golden", action="store_true")
    ap.add_argument("--seed", type=int, default=20260925)
    args = ap.parse_args()
    try:
        import cv2
    except ImportError:
        print("opencv_crosscheck: no cv2 in this environment -- nothing to diff against (parity stays unpinned here)")
        return 2
    from oracle import oracle as orc
    from visual_odom_amd import synth

    world = synth.StereoWorld(seed=args.seed, width=1241, height=376, fx=718.856, cx=607.1928, cy=185.2157, bf=-386.1448)
    L, R, poses, _ = world.render_sequence(2)
    pts = synth.select_keypoints(L[0], bucket=37, per_bucket=6)
    P_l, P_r = world.proj_matrices()
    K = world.K()
    rows, golden, bad = [], {}, 0

    def report(name, diff, tol, extra=""):
        nonlocal bad
        ok = diff <= tol
        bad += not ok
        rows.append("%-44s max |diff| %-12.4g tol %-8g %s %s" % (name, diff, tol, "ok" if ok else "MISMATCH", extra))

    # ---- pyramid
    lvl = L[0]
    for l in range(1, 4):
        cv = cv2.pyrDown(lvl)
        mine = orc.pyr_down(lvl)
        report("pyrDown level %d" % l, float(np.abs(cv.astype(int) - mine.astype(int)).max()), 0)
        lvl = mine
    # ---- FAST
    kp = cv2.FastFeatureDetector_create(20, True, cv2.FAST_FEATURE_DETECTOR_TYPE_9_16).detect(L[0])
    cv_pts = np.array([k.pt for k in kp], np.float32).reshape(-1, 2)
    my_pts = orc.fast_detect(L[0], 20, True)
    same = cv_pts.shape == my_pts.shape and np.array_equal(cv_pts, my_pts)
    report("FAST(20, nonmax) corners + order", 0.0 if same else 1.0, 0, "cv2 %d / oracle %d" % (len(cv_pts), len(my_pts)))
    golden["fast"] = cv_pts
    kp0 = cv2.FastFeatureDetector_create(20, False, cv2.FAST_FEATURE_DETECTOR_TYPE_9_16).detect(L[0])
    cv0 = np.array([k.pt for k in kp0], np.float32).reshape(-1, 2)
    my0 = orc.fast_detect(L[0], 20, False, cap=1 << 20)
    report("FAST(20, no nonmax) corners + order", 0.0 if cv0.shape == my0.shape and np.array_equal(cv0, my0) else 1.0, 0,
           "cv2 %d / oracle %d" % (len(cv0), len(my0)))
    # ---- the four LK hops
    lk = dict(winSize=(21, 21), maxLevel=3, criteria=(cv2.TERM_CRITERIA_COUNT + cv2.TERM_CRITERIA_EPS, 30, 0.01), flags=0,
              minEigThreshold=0.001)
    chain = [(L[0], R[0]), (R[0], R[1]), (R[1], L[1]), (L[1], L[0])]
    p_cv = p_my = pts
    # oracle accum_mode (oracle/vo_oracle.h): 0 exact integer sums (the HIP kernel), 1 sequential f32, 2 OpenCV's 128-bit
    # universal-intrinsics order without FMA (default x86-64 baseline), 3 the same with fused v_muladd -- which one is THIS cv2?
    MODES = {0: "exact integer sums", 1: "sequential f32", 2: "x86 SIMD128 order, no FMA", 3: "x86 SIMD128 order, FMA"}
    identical = {m: 0 for m in MODES}
    total = 0
    for hop, (a, b) in enumerate(chain):
        q_cv, st_cv, _ = cv2.calcOpticalFlowPyrLK(a, b, p_cv.reshape(-1, 1, 2), None, **lk)
        q_cv, st_cv = q_cv.reshape(-1, 2), st_cv.reshape(-1)
        q_my, st_my, _ = orc.calc_optical_flow_pyr_lk(a, b, p_my)
        both = (st_cv == 1) & (st_my == 1)
        report("calcOpticalFlowPyrLK hop %d positions" % hop, float(np.abs(q_cv[both] - q_my[both]).max()), 1e-3,
               "status differs at %d of %d" % (int((st_cv != st_my).sum()), len(st_cv)))
        # every accumulation mode from cv2's OWN input points of this hop: bit-identical tracks per mode
        for m in MODES:
            q_m, st_m, _ = orc.calc_optical_flow_pyr_lk(a, b, p_cv, accum_mode=m)
            ok = (st_cv == 1) & (st_m == 1)
            identical[m] += int((np.ascontiguousarray(q_m[ok]).view(np.uint32) == np.ascontiguousarray(q_cv[ok]).view(np.uint32)).all(1).sum())
        total += int((st_cv == 1).sum())
        golden["lk_hop%d" % hop], golden["lk_status%d" % hop] = q_cv, st_cv
        p_cv, p_my = q_cv, q_my
    best = max(MODES, key=lambda m: identical[m])
    print("LK accumulation order of this cv2 build: " + "; ".join("mode %d (%s) %d of %d tracks bit-identical" % (m, MODES[m], identical[m], total)
                                                                   for m in MODES))
    print("  -> matches accum_mode %d%s" % (best, " BIT FOR BIT on every track" if identical[best] == total else " best (not exactly: the "
          "restatement of the accumulation order, [upstream-memory] in oracle/vo_oracle.h, needs another look for this version)"))
    golden["lk_accum_mode"] = np.array([best, identical[best], total])
    # ---- triangulation
    ref = orc.circular_matching(L[0], R[0], L[1], R[1], pts)
    (l0, r0, l1, r1), _ = orc.check_valid_and_remove(ref["l0"], ref["r0"], ref["l1"], ref["r1"], ref["l0_ret"])
    X4 = cv2.triangulatePoints(P_l, P_r, l0.T.copy(), r0.T.copy())
    xyz_cv = cv2.convertPointsFromHomogeneous(X4.T).reshape(-1, 3)
    xyz_my = orc.triangulate(P_l, P_r, l0, r0)
    report("triangulatePoints + convertFromHomogeneous", float((np.abs(xyz_cv - xyz_my) / np.abs(xyz_my).max(1, keepdims=True)).max()), 1e-5)
    golden["xyz"] = xyz_cv
    # ---- PnP / RANSAC
    rvec, tvec = np.zeros((3, 1)), np.zeros((3, 1))
    ok, rvec, tvec, inl = cv2.solvePnPRansac(xyz_my.reshape(-1, 1, 3), l1.reshape(-1, 1, 2), K.astype(np.float32), np.zeros((4, 1)),
                                             rvec, tvec, True, 500, 0.5, float(np.float32(0.999)), None, cv2.SOLVEPNP_ITERATIVE)
    rc, rv, tv, inl_my, _ = orc.solve_pnp_ransac(xyz_my, l1, K)
    same_inl = inl is not None and np.array_equal(inl.reshape(-1), inl_my)
    report("solvePnPRansac rvec | tvec", float(max(np.abs(rvec.reshape(3) - rv).max(), np.abs(tvec.reshape(3) - tv).max())), 1e-6,
           "inlier sets %s" % ("identical" if same_inl else "DIFFER"))
    report("Rodrigues", float(np.abs(cv2.Rodrigues(rv.reshape(3, 1))[0] - orc.rodrigues(rv)).max()), 1e-12)
    # ---- exactly four correspondences: OpenCV's `npoints == 4 -> SOLVEPNP_P3P` switch (oracle/orc_p3p.c, round 3).  The
    # closed-form quartic loses digits near double roots, so several well-spread quadruples and a median
    worst4, n4 = [], 0
    rng = np.random.default_rng(args.seed)
    for _ in range(50):
        idx = rng.choice(len(xyz_my), 4, replace=False)
        X4p, u4 = xyz_my[idx].astype(np.float32), l1[idx].astype(np.float32)
        r0v, t0v = np.zeros((3, 1)), np.zeros((3, 1))
        ok4, r4, t4, inl4 = cv2.solvePnPRansac(X4p.reshape(-1, 1, 3), u4.reshape(-1, 1, 2), K.astype(np.float32), np.zeros((4, 1)),
                                               r0v, t0v, True, 500, 0.5, float(np.float32(0.999)), None, cv2.SOLVEPNP_ITERATIVE)
        rc4, rv4, tv4, inl4_my, _ = orc.solve_pnp_ransac(X4p, u4, K)
        if bool(ok4) != (rc4 == 1):
            worst4.append(1.0)
        elif ok4:
            worst4.append(float(max(np.abs(r4.reshape(3) - rv4).max(), np.abs(t4.reshape(3) - tv4).max())))
            n4 += 1
    report("solvePnPRansac with 4 points (P3P), median", float(np.median(worst4)) if worst4 else 1.0, 1e-6,
           "%d of 50 quadruples solved by cv2; 90 %% quantile %.3g" % (n4, float(np.percentile(worst4, 90)) if worst4 else 1.0))
    golden["rvec"], golden["tvec"] = rvec.reshape(3), tvec.reshape(3)
    # ---- essential matrix + recoverPose
    focal, pp = float(P_l[0, 0]), (float(P_l[0, 2]), float(P_l[1, 2]))
    E, mask = cv2.findEssentialMat(l0, l1, focal, pp, cv2.RANSAC, 0.999, 1.0)
    okE, E_my, mask_my, _ = orc.find_essential_mat(l0, l1, focal, pp)
    dE = min(np.abs(E - E_my).max(), np.abs(E + E_my).max()) if E is not None and E.shape == (3, 3) else 1.0
    report("findEssentialMat E (up to sign)", float(dE), 1e-6, "mask differs at %d" % int((mask.reshape(-1) != mask_my).sum()))
    _, Rcv, tcv, _ = cv2.recoverPose(E, l0, l1, focal=focal, pp=pp, mask=mask.copy())
    _, Rmy, tmy, _ = orc.recover_pose(E_my, l0, l1, focal, pp, mask_my)
    report("recoverPose R", float(np.abs(Rcv - Rmy).max()), 1e-6)
    golden["E"], golden["R_mono"] = E, Rcv

    print("OpenCV %s vs oracle (seed %d)" % (cv2.__version__, args.seed))
    print("\n".join(rows))
    if args.write_golden:
        out = os.path.join(ROOT, "tests", "golden", "opencv_%s.npz" % cv2.__version__.replace(".", "_"))
        os.makedirs(os.path.dirname(out), exist_ok=True)
        np.savez_compressed(out, seed=args.seed, pts=pts, **golden)
        print("wrote", out)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
