#!/usr/bin/env python3
"""Write tests/golden/oracle_kat.npz: known-answer vectors of the hot path on one small stereo quadruple.

Provenance, stated plainly: these vectors come from THIS REPOSITORY'S CPU ORACLE (oracle/*.c) and from the reference's
own glue sources compiled over it (oracle/_ref), not from OpenCV -- there is no OpenCV in the authoring container
(DESIGN.md section 6).  They pin the oracle against drift (every parity claim is relative to it) and give the GPU suite
fixed inputs whose answers do not depend on rebuilding anything.  tools/opencv_crosscheck.py --write-golden is the tool
that produces real-OpenCV vectors on a machine that has cv2.

    python tools/make_golden.py        # rewrites the file; commit it together with the change that required it
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from oracle import oracle as orc
    from visual_odom_amd import synth
    orc.build()
    w, h = 480, 160
    world = synth.StereoWorld(seed=5, width=w, height=h, fx=300.0, cx=239.5, cy=79.5, bf=-160.0, tex_size=1024)
    L, R, poses, _ = world.render_sequence(2)
    pts = synth.select_keypoints(L[0], bucket=h // 10, per_bucket=2)
    border = np.array([[0, 0], [479, 159], [2.5, 80.25], [-5, 50], [100, -3], [520, 100], [240, 185]], np.float32)
    pts = np.vstack([pts, border]).astype(np.float32)
    P_l, P_r = world.proj_matrices()
    K = world.K()
    out = dict(l0=L[0], r0=R[0], l1=L[1], r1=R[1], pts=pts, P_l=P_l, P_r=P_r, K=K)
    for l, lvl in enumerate(orc.build_pyramid(L[0], 3)[1:], 1):
        out["pyr_l0_level%d" % l] = lvl
    out["scharr_l0"] = orc.scharr(L[0])
    p = pts
    for hop, (a, b) in enumerate([(L[0], R[0]), (R[0], R[1]), (R[1], L[1]), (L[1], L[0])]):
        p, st, _ = orc.calc_optical_flow_pyr_lk(a, b, p)
        out["lk_hop%d" % hop], out["lk_status%d" % hop] = p, st
    cm = orc.circular_matching(L[0], R[0], L[1], R[1], pts)
    (l0, r0, l1, r1), valid = orc.check_valid_and_remove(cm["l0"], cm["r0"], cm["l1"], cm["r1"], cm["l0_ret"])
    out.update(keep_idx=cm["keep_idx"], f_l0=l0, f_r0=r0, f_l1=l1, f_r1=r1)
    xyz = orc.triangulate(P_l, P_r, l0, r0)
    rc, rv, tv, inl, dbg = orc.solve_pnp_ransac(xyz, l1, K)
    out.update(xyz=xyz, rvec=rv, tvec=tv, inliers=inl, pnp_dbg=dbg)
    focal, pp = float(P_l[0, 0]), (float(P_l[0, 2]), float(P_l[1, 2]))
    ok, E, mask, _ = orc.find_essential_mat(l0, l1, focal, pp)
    good, Rm, tm, m2 = orc.recover_pose(E, l0, l1, focal, pp, mask)
    out.update(E=E, em_mask=m2, R_mono=Rm, t_mono=tm)
    out["fast_l0"] = orc.fast_detect(L[0], 20, True)
    bp, ba = orc.bucketing_features(h, w, out["fast_l0"], np.zeros(len(out["fast_l0"]), np.int32), h // 10, 1)
    out.update(bucket_pts=bp, bucket_ages=ba)
    path = os.path.join(ROOT, "tests", "golden", "oracle_kat.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%d arrays, %.0f KB)" % (path, len(out), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
