"""Developer first-contact script: run every stage on the GPU and print diffs vs the oracle."""
import sys, time, os, faulthandler, functools
print = functools.partial(print, flush=True)
faulthandler.dump_traceback_later(int(os.environ.get("VO_WATCHDOG", "100")), exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visual_odom_amd import synth, _lib
from oracle import oracle as orc

t0 = time.time()
W = synth.StereoWorld()
L, R, poses, D = W.render_sequence(2)
print("render", time.time() - t0, flush=True)
kp = synth.select_keypoints(L[0])
# add border / out-of-range points to hit the slow paths
extra = np.array([[0, 0], [1240, 375], [3.5, 200.25], [1238.2, 10.7], [600, 2.1], [620.4, 374.9],
                  [-5, 50], [100, -3], [1300, 100], [15.5, 15.5]], np.float32)
kp = np.vstack([kp, extra]).astype(np.float32)
print("kp", kp.shape)
ctx = _lib.Context(0, 1241, 376, 4096, 2)
ctx.set_params(lk_full_chain=1)  # raw per-hop diffs below need all four hops of every feature
Pl, Pr = W.proj_matrices()

# ---- pyramid
ctx.batch_configure(4, 1241, 376, 1)
for i, im in enumerate((L[0], R[0], L[1], R[1])):
    ctx.batch_upload_image(i, im)
ctx.batch_set_quads([[0, 1, 2, 3]])
ctx.batch_set_points(0, kp)
ctx.batch_set_projection(Pl, Pr)
STG = _lib.STAGE_ALL if os.environ.get("VO_PNP") == "1" else 15
ms = ctx.batch_run_timed(STG)
print("stage ms (first, cold):", ms)
ms = ctx.batch_run_timed(STG)
print("stage ms (warm):", ms)
for i, im in enumerate((L[0], R[0])):
    pyr = orc.build_pyramid(im, 3)
    for l in range(4):
        g = ctx.batch_get_pyramid_level(i, l)
        print("pyr img", i, "lvl", l, g.shape, "mismatch px:", int((g != pyr[l]).sum()))

# ---- LK
t = time.time()
o = orc.circular_matching(L[0], R[0], L[1], R[1], kp)
print("oracle circ time", time.time() - t)
n = kp.shape[0]
g = ctx.batch_get_tracks(0, n)
st_o = o["status4"]
print("status equal:", [(g["status4"][h] == st_o[h]).all() for h in range(4)], "ok counts", st_o.sum(1), g["status4"].sum(1))
# oracle raw (uncompacted) tracks: recompute hops
p1, s1, _ = orc.calc_optical_flow_pyr_lk(L[0], R[0], kp)
p2, s2, _ = orc.calc_optical_flow_pyr_lk(R[0], R[1], p1)
p3, s3, _ = orc.calc_optical_flow_pyr_lk(R[1], L[1], p2)
p4, s4, _ = orc.calc_optical_flow_pyr_lk(L[1], L[0], p3)
for name, a, b in (("r0", g["r0"], p1), ("r1", g["r1"], p2), ("l1", g["l1"], p3), ("l0_ret", g["l0_ret"], p4)):
    bit = (a.view(np.uint32) == b.view(np.uint32)).all(1)
    with np.errstate(invalid="ignore"):
        d = np.abs(a - b)
    print(name, "bit-equal", int(bit.sum()), "/", n, "max abs diff", np.nanmax(d) if n else 0)
    if not bit.all():
        bad = np.where(~bit)[0][:8]
        print("   first bad idx", bad, "\n   in", kp[bad], "\n   gpu", a[bad], "\n   orc", b[bad])
# ---- filter
f = ctx.batch_get_filtered(0)
(l0, r0, l1, r1), valid = orc.check_valid_and_remove(o["l0"], o["r0"], o["l1"], o["r1"], o["l0_ret"])
print("circ survivors gpu/orc", len(f["keep_idx_circ"]), o["n_out"], "equal idx", np.array_equal(f["keep_idx_circ"], o["keep_idx"]))
print("consistency survivors gpu/orc", len(f["l0"]), len(l0), "equal", all(np.array_equal(a, b) for a, b in ((f["l0"], l0), (f["r0"], r0), (f["l1"], l1), (f["r1"], r1))))
# ---- triangulation
xyz_o = orc.triangulate(Pl, Pr, l0, r0)
if len(f["xyz"]) == len(xyz_o):
    print("tri bit-equal", int((f["xyz"] == xyz_o).all(1).sum()), "/", len(xyz_o), "max rel", np.max(np.abs(f["xyz"] - xyz_o) / np.abs(xyz_o).max(1, keepdims=True)))
if os.environ.get("VO_PNP") != "1":
    print("NO-PNP DONE"); sys.exit(0)
# ---- pnp
pose = ctx.batch_get_pose(0)
rc, rv, tv, inl, dbg = orc.solve_pnp_ransac(xyz_o, l1, W.K())
print("gpu pose", pose["rvec"], pose["tvec"], pose["status"], len(pose["inliers"]), pose["niters"], pose["best_iter"], pose["max_good"], pose["lm_iters"])
print("orc pose", rv, tv, rc, len(inl), dbg[:4])
print("pose diff r", np.abs(pose["rvec"] - rv).max(), "t", np.abs(pose["tvec"] - tv).max(), "inliers equal", np.array_equal(pose["inliers"], inl))
# ---- drop-in calls
d = ctx.circular_match(L[0], R[0], L[1], R[1], kp)
print("drop-in circular_match equal:", d["n_out"] == o["n_out"] and all(np.array_equal(d[k], o[k]) for k in ("l0", "r0", "r1", "l1", "l0_ret", "keep_idx")))
x = ctx.triangulate(Pl, Pr, l0, r0)
print("drop-in triangulate equal:", np.array_equal(x, xyz_o))
ok, rv2, tv2, R2, inl2 = ctx.pnp_ransac(xyz_o, l1, W.K())
print("drop-in pnp diff", ok, np.abs(rv2 - rv).max(), np.abs(tv2 - tv).max(), np.array_equal(inl2, inl))
tf = ctx.track_frame(L[0], R[0], L[1], R[1], kp, Pl, Pr)
print("track_frame rc", tf["rc"], "pose diff", np.abs(tf["rvec"] - rv).max(), np.abs(tf["tvec"] - tv).max())

# ---- batch of 2 frames (same quad twice, different point subsets)
ctx.batch_configure(4, 1241, 376, 2)
for i, im in enumerate((L[0], R[0], L[1], R[1])):
    ctx.batch_upload_image(i, im)
ctx.batch_set_quads([[0, 1, 2, 3], [2, 3, 0, 1]])
ctx.batch_set_points(0, kp[:1000])
ctx.batch_set_points(1, kp[500:])
ctx.batch_run(_lib.STAGE_ALL)
ctx.batch_sync()
g0 = ctx.batch_get_tracks(0, 1000)
print("batch frame0 r0 equal", np.array_equal(g0["r0"].view(np.uint32), p1[:1000].view(np.uint32)))
print("batch poses", ctx.batch_get_pose(0)["tvec"], ctx.batch_get_pose(1)["tvec"])
print("DONE")
