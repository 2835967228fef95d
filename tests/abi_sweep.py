"""Argument sweep over every export of libvo_hip.so: NULL pointers, n = 0, n > capacity, w / h beyond
the context's maximum, bad indices, calls in the wrong state.  Every such call must come back with the documented error code
-- never a fault, never a silent success.  Run as a SCRIPT in a child process by tests/test_gpu_round4.py (a fault would
otherwise take the test session down with it); prints one JSON object {"checked": n, "exports_covered": [...], "failures":
[...]} and exits 0 iff there is no failure.  Needs a GPU (vo_create)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visual_odom_amd import _lib  # noqa: E402

OK, ARG, HIP, STATE, TOO_FEW, OVERFLOW = 0, -1, -2, -3, -4, -5
W, H, CAP, FRAMES = 320, 96, 256, 2


def main():
    lib = _lib.load()
    lib.vo_create.restype = C.c_void_p
    lib.vo_last_error.restype = C.c_char_p
    fails, covered, checked = [], set(), [0]
    NULL = C.c_void_p(None)

    def vp(a):
        return a.ctypes.data_as(C.c_void_p)

    def expect(name, want, *args):
        """call lib.<name>(*args): the return code must be one of `want`"""
        covered.add(name)
        checked[0] += 1
        fn = getattr(lib, name)
        fn.restype = C.c_int
        rc = fn(*args)
        want = want if isinstance(want, (tuple, list)) else (want,)
        if rc not in want:
            fails.append("%s%r -> %d, expected %s" % (name, tuple(str(a)[:24] for a in args[1:]), rc, list(want)))
        return rc

    # ---- vo_create / vo_destroy / vo_last_error ----
    for bad in ((0, 16, H, CAP, 1), (0, W, 16, CAP, 1), (0, W, H, 0, 1), (0, W, H, CAP, 0), (-1, W, H, CAP, 1), (4096, W, H, CAP, 1)):
        covered.add("vo_create")
        checked[0] += 1
        h = lib.vo_create(*bad)
        if h:
            fails.append("vo_create%r returned a context" % (bad,))
    lib.vo_destroy.restype = None
    lib.vo_destroy(NULL)  # must be a no-op
    covered.add("vo_destroy")
    covered.add("vo_last_error")
    assert lib.vo_last_error(NULL) is not None
    ctx = C.c_void_p(lib.vo_create(0, W, H, CAP, FRAMES))
    assert ctx, "vo_create failed"

    img = np.zeros((H, W), np.uint8)
    big = np.zeros((H + 8, W + 8), np.uint8)
    pts = np.full((CAP + 8, 2), 50.0, np.float32)
    out = [np.zeros((CAP + 8, 2), np.float32) for _ in range(5)]
    xyz = np.zeros((CAP + 8, 3), np.float32)
    idx = np.zeros(CAP + 8, np.int32)
    st4 = np.zeros(4 * (CAP + 8), np.uint8)
    n_out, n2 = C.c_int(0), C.c_int(0)
    P = np.array([[300, 0, 160, 0], [0, 300, 48, 0], [0, 0, 1, 0]], np.float32)
    Pr = P.copy()
    Pr[0, 3] = -150
    K = np.ascontiguousarray(P[:, :3])
    rv, tv, R9 = np.zeros(3), np.zeros(3), np.zeros(9)
    fbuf = np.zeros(64, np.float32)
    dbuf = np.zeros(64, np.float64)

    # ---- parameters / schedule ----
    lib.vo_default_params.restype = None
    lib.vo_default_detect_params.restype = None
    prm = _lib.VoParams()
    lib.vo_default_params(C.byref(prm))
    dprm = _lib.VoDetectParams()
    lib.vo_default_detect_params(C.byref(dprm))
    covered.update(["vo_default_params", "vo_default_detect_params"])
    expect("vo_set_params", ARG, NULL, C.byref(prm))
    expect("vo_set_params", ARG, ctx, NULL)
    for field, val in (("lk_max_level", -1), ("lk_max_level", 5), ("ransac_iterations", 0), ("ransac_iterations", 100000),
                       ("ransac_confidence", 0.0), ("ransac_confidence", 1.0)):
        p2 = _lib.VoParams()
        lib.vo_default_params(C.byref(p2))
        setattr(p2, field, val)
        expect("vo_set_params", ARG, ctx, C.byref(p2))
    expect("vo_set_params", OK, ctx, C.byref(prm))
    expect("vo_get_params", ARG, NULL, C.byref(prm))
    expect("vo_get_params", ARG, ctx, NULL)
    # vo_kept_pair_id: an int64 id, 0 = no kept pair (NULL context, fresh context)
    lib.vo_kept_pair_id.restype = C.c_int64
    covered.add("vo_kept_pair_id")
    checked[0] += 2
    if lib.vo_kept_pair_id(NULL) != 0 or lib.vo_kept_pair_id(ctx) != 0:
        fails.append("vo_kept_pair_id: expected 0 for a NULL / fresh context")
    sch = _lib.VoSchedule(0, 0, -1, 0)
    expect("vo_set_schedule", ARG, NULL, C.byref(sch))
    for bad in ((3, 0, -1), (4, 0, -1), (5, 0, -1), (-1, 0, -1), (0, 3, -1), (0, -1, -1), (0, 0, 2), (0, 0, -2), (0, 0, -1, 8), (0, 0, -1, -4),
                (0, 0, -1, 32)):
        expect("vo_set_schedule", ARG, ctx, C.byref(_lib.VoSchedule(*bad)))
    expect("vo_set_schedule", OK, ctx, NULL)
    expect("vo_get_schedule", ARG, NULL, C.byref(sch), NULL)
    expect("vo_get_schedule", ARG, ctx, NULL, NULL)
    expect("vo_get_probe_log", ARG, NULL, NULL, NULL, NULL, C.byref(n_out))
    expect("vo_get_probe_log", ARG, ctx, NULL, NULL, NULL, NULL)
    rec = _lib.VoScheduleRecord()
    expect("vo_export_schedule", ARG, NULL, 0, NULL)
    expect("vo_export_schedule", ARG, NULL, 4, C.byref(n_out))
    expect("vo_export_schedule", ARG, C.byref(rec), -1, C.byref(n_out))
    expect("vo_export_schedule", OK, NULL, 0, C.byref(n_out))
    expect("vo_import_schedule", ARG, NULL, 1)
    expect("vo_import_schedule", ARG, C.byref(rec), -1)
    for key, sc in (((0, 0, 640, 480, 4, 1, 20, 0), (3, 1, 0, 4)), ((0, 0, 640, 480, 4, 1, 20, 0), (1, 0, 0, 4)), ((0, 0, 640, 480, 4, 1, 20, 0), (1, 1, 2, 4)),
                    ((0, 0, 640, 480, 4, 1, 20, 0), (1, 1, 0, 0)), ((0, 0, 640, 480, 4, 1, 20, 0), (1, 1, 0, 8)),
                    ((-1, 0, 640, 480, 4, 1, 20, 0), (1, 1, 0, 4)), ((0, 0, 8, 480, 4, 1, 20, 0), (1, 1, 0, 4)), ((0, 0, 640, 480, 9, 1, 20, 0), (1, 1, 0, 4)),
                    ((0, 0, 640, 480, 4, 0, 20, 0), (1, 1, 0, 4))):
        r2 = _lib.VoScheduleRecord()
        for i in range(8):
            r2.key[i] = key[i]
        r2.schedule = _lib.VoSchedule(*sc)
        expect("vo_import_schedule", ARG, C.byref(r2), 1)
    expect("vo_import_schedule", OK, NULL, 0)
    expect("vo_batch_set_detect_params", ARG, NULL, C.byref(dprm))
    for field, val in (("features_per_bucket", 0), ("features_per_bucket", 9), ("bucket_size", -1)):
        d2 = _lib.VoDetectParams()
        lib.vo_default_detect_params(C.byref(d2))
        setattr(d2, field, val)
        expect("vo_batch_set_detect_params", ARG, ctx, C.byref(d2))
    expect("vo_batch_set_detect_params", OK, ctx, NULL)

    # ---- host-only helper ----
    pose = np.eye(4)
    expect("vo_integrate_odometry", ARG, NULL, vp(R9), vp(tv), NULL)
    expect("vo_integrate_odometry", ARG, vp(pose), NULL, vp(tv), NULL)
    expect("vo_integrate_odometry", ARG, vp(pose), vp(R9), NULL, NULL)

    # ---- batch API before vo_batch_configure: state errors, not faults ----
    expect("vo_batch_run", (STATE, ARG), ctx, 31)
    expect("vo_batch_run_timed", (STATE, ARG), ctx, 31, vp(fbuf))
    expect("vo_batch_upload_image", (STATE, ARG), ctx, 0, vp(img), W)
    expect("vo_batch_set_pyramid_range", (STATE, ARG), ctx, 0, 1)
    expect("vo_batch_set_quads", (STATE, ARG), ctx, vp(idx), 1)
    expect("vo_batch_set_features", (STATE, ARG), ctx, 0, vp(pts), 1, vp(idx), 1)
    expect("vo_batch_get_essential", (STATE, ARG), ctx, 0, vp(dbuf), vp(dbuf), vp(dbuf), NULL, 0, NULL, NULL, NULL, NULL)
    expect("vo_seq_step", (STATE, ARG), ctx)
    expect("vo_seq_push_pair", (STATE, ARG), ctx, 0, vp(img), vp(img), W, 0)
    expect("vo_seq_get_state", (STATE, ARG), ctx, 0, NULL, NULL, NULL, NULL, NULL)
    expect("vo_seq_get_trajectory", (STATE, ARG), ctx, 0, 0, 1, vp(dbuf), NULL, C.byref(n_out))
    expect("vo_seq_reset", (STATE, ARG), ctx, 0)
    expect("vo_seq_sync", (STATE, ARG, OK), ctx)

    # ---- vo_batch_configure ----
    expect("vo_batch_configure", ARG, NULL, 4, W, H, 1)
    for bad in ((0, W, H, 1), (4, W + 1, H, 1), (4, W, H + 1, 1), (4, 16, H, 1), (4, W, 16, 1), (4, W, H, 0), (4, W, H, FRAMES + 1),
                (6 * FRAMES + 1, W, H, 1), (-1, W, H, 1)):
        expect("vo_batch_configure", ARG, ctx, *bad)
    expect("vo_batch_configure", OK, ctx, 4, W, H, 1)

    # ---- uploads / tables / points ----
    expect("vo_batch_upload_image", ARG, NULL, 0, vp(img), W)
    for bad in ((-1, vp(img), W), (4, vp(img), W), (0, NULL, W), (0, vp(img), W - 1), (0, vp(img), 0), (0, vp(img), -W)):
        expect("vo_batch_upload_image", ARG, ctx, *bad)
    for bad in ((-1, vp(img), W), (4, vp(img), W), (0, NULL, W), (0, vp(img), W - 1)):
        expect("vo_batch_upload_image_dev", ARG, ctx, *bad)
    expect("vo_batch_upload_image_dev", ARG, NULL, 0, vp(img), W)
    expect("vo_batch_set_quads", ARG, NULL, vp(idx), 1)
    expect("vo_batch_set_quads", ARG, ctx, NULL, 1)
    expect("vo_batch_set_quads", (ARG, STATE), ctx, vp(idx), 2)
    expect("vo_batch_set_quads", (ARG, STATE), ctx, vp(idx), 0)
    expect("vo_batch_set_quads", ARG, ctx, vp(np.array([0, 1, 2, 4], np.int32)), 1)
    expect("vo_batch_set_quads", ARG, ctx, vp(np.array([0, -1, 2, 3], np.int32)), 1)
    expect("vo_batch_set_pyramid_range", ARG, NULL, 0, 1)
    for bad in ((-1, 1), (0, -1), (3, 2), (0, 5)):
        expect("vo_batch_set_pyramid_range", ARG, ctx, *bad)
    expect("vo_batch_set_points", ARG, NULL, 0, vp(pts), 1)
    for bad in ((-1, vp(pts), 1), (1, vp(pts), 1), (0, vp(pts), -1), (0, vp(pts), CAP + 1), (0, NULL, 1)):
        expect("vo_batch_set_points", ARG, ctx, *bad)
    expect("vo_batch_set_points", OK, ctx, 0, NULL, 0)
    expect("vo_batch_set_features", ARG, NULL, 0, vp(pts), 1, vp(idx), 1)
    for bad in ((-1, vp(pts), 1, vp(idx), 1), (1, vp(pts), 1, vp(idx), 1), (0, vp(pts), -1, vp(idx), 1), (0, vp(pts), 2, vp(idx), 1),
                (0, NULL, 1, vp(idx), 1), (0, vp(pts), 1, NULL, 1), (0, vp(pts), 1, vp(idx), 1 << 28)):
        expect("vo_batch_set_features", ARG, ctx, *bad)
    expect("vo_batch_set_projection", ARG, NULL, vp(P), vp(Pr))
    expect("vo_batch_set_projection", ARG, ctx, NULL, vp(Pr))
    expect("vo_batch_set_projection", ARG, ctx, vp(P), NULL)

    # ---- runs ----
    expect("vo_batch_run", ARG, NULL, 31)
    expect("vo_batch_run", STATE, ctx, 31)  # projection not set
    expect("vo_batch_set_projection", OK, ctx, vp(P), vp(Pr))
    expect("vo_batch_run_timed", ARG, NULL, 31, vp(fbuf))
    expect("vo_batch_run_timed", ARG, ctx, 31, NULL)
    expect("vo_batch_run_slot", ARG, NULL, 31, 0)
    expect("vo_batch_run_slot", ARG, ctx, 31, -1)
    expect("vo_batch_run_slot", ARG, ctx, 31, 256)
    expect("vo_batch_slot_times", ARG, NULL, 0, vp(fbuf))
    expect("vo_batch_slot_times", ARG, ctx, 0, NULL)
    expect("vo_batch_slot_times", ARG, ctx, -1, vp(fbuf))
    expect("vo_batch_slot_times", ARG, ctx, 256, vp(fbuf))
    expect("vo_batch_sync", ARG, NULL)
    for i in range(4):
        expect("vo_batch_upload_image", OK, ctx, i, vp(img), W)
    expect("vo_batch_run", OK, ctx, 31)  # zero points, blank images: a valid (empty) run
    expect("vo_batch_sync", OK, ctx)

    # ---- getters ----
    expect("vo_batch_get_tracks", ARG, NULL, 0, NULL, NULL, NULL, NULL, NULL, 0)
    for bad in ((-1, 0), (1, 0), (0, -1), (0, CAP + 1)):
        expect("vo_batch_get_tracks", ARG, ctx, bad[0], vp(out[0]), NULL, NULL, NULL, NULL, bad[1])
    expect("vo_batch_get_tracks", OK, ctx, 0, NULL, NULL, NULL, NULL, NULL, 0)
    expect("vo_batch_get_filtered", ARG, NULL, 0, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL)
    expect("vo_batch_get_filtered", ARG, ctx, -1, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL)
    expect("vo_batch_get_filtered", ARG, ctx, 1, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL)
    expect("vo_batch_get_filtered", OK, ctx, 0, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL)
    expect("vo_batch_get_pose", ARG, NULL, 0, NULL, NULL, NULL, NULL, NULL, NULL, NULL)
    expect("vo_batch_get_pose", ARG, ctx, -1, NULL, NULL, NULL, NULL, NULL, NULL, NULL)
    expect("vo_batch_get_pose", ARG, ctx, 1, NULL, NULL, NULL, NULL, NULL, NULL, NULL)
    expect("vo_batch_get_pose", OK, ctx, 0, NULL, NULL, NULL, NULL, NULL, NULL, NULL)
    expect("vo_batch_get_features", ARG, NULL, 0, NULL, NULL, C.byref(n_out))
    expect("vo_batch_get_features", ARG, ctx, 0, NULL, NULL, NULL)
    expect("vo_batch_get_features", ARG, ctx, -1, NULL, NULL, C.byref(n_out))
    expect("vo_batch_get_features", ARG, ctx, 1, NULL, NULL, C.byref(n_out))
    expect("vo_batch_get_essential", ARG, NULL, 0, NULL, NULL, NULL, NULL, 0, NULL, NULL, NULL, NULL)
    expect("vo_batch_get_essential", (ARG, STATE), ctx, -1, NULL, NULL, NULL, NULL, 0, NULL, NULL, NULL, NULL)
    expect("vo_batch_get_essential", (ARG, STATE), ctx, 0, NULL, NULL, NULL, NULL, CAP + 1, NULL, NULL, NULL, NULL)
    lvl = np.zeros(W * H, np.uint8)
    expect("vo_batch_get_pyramid_level", ARG, NULL, 0, 0, vp(lvl), C.byref(n_out), C.byref(n2))
    for bad in ((-1, 0), (4, 0), (0, -1), (0, 5)):
        expect("vo_batch_get_pyramid_level", ARG, ctx, bad[0], bad[1], vp(lvl), C.byref(n_out), C.byref(n2))
    expect("vo_batch_get_pyramid_level", OK, ctx, 0, 0, NULL, C.byref(n_out), C.byref(n2))  # sizes only
    if (n_out.value, n2.value) != (W, H):
        fails.append("vo_batch_get_pyramid_level(out = NULL) did not return the level's size")
    expect("vo_model_bytes", ARG, NULL, W, H, 10, vp(dbuf))
    expect("vo_model_bytes", ARG, ctx, W, H, 10, NULL)
    expect("vo_model_bytes", ARG, ctx, 0, H, 10, vp(dbuf))
    expect("vo_model_bytes", ARG, ctx, W, H, -1, vp(dbuf))

    # ---- drop-in calls ----
    def cm(ctx_, l0, w, h, stride, p, n, nout=C.byref(n_out)):
        return (ctx_, l0, vp(img), vp(img), vp(img), w, h, stride, p, n, vp(out[0]), vp(out[1]), vp(out[2]), vp(out[3]),
                vp(out[4]), vp(st4), vp(idx), nout, 0)
    expect("vo_circular_match", ARG, *cm(NULL, vp(img), W, H, W, vp(pts), 4))
    expect("vo_circular_match", ARG, *cm(ctx, vp(img), W, H, W, vp(pts), 4, NULL))
    expect("vo_circular_match", ARG, *cm(ctx, NULL, W, H, W, vp(pts), 4))
    expect("vo_circular_match", ARG, *cm(ctx, vp(img), W + 1, H, W + 1, vp(pts), 4))
    expect("vo_circular_match", ARG, *cm(ctx, vp(img), W, H + 1, W, vp(pts), 4))
    expect("vo_circular_match", ARG, *cm(ctx, vp(img), 16, H, 16, vp(pts), 4))
    expect("vo_circular_match", ARG, *cm(ctx, vp(img), W, H, W - 1, vp(pts), 4))
    expect("vo_circular_match", ARG, *cm(ctx, vp(img), W, H, W, vp(pts), -1))
    expect("vo_circular_match", ARG, *cm(ctx, vp(img), W, H, W, vp(pts), CAP + 1))
    expect("vo_circular_match", ARG, *cm(ctx, vp(img), W, H, W, NULL, 4))
    expect("vo_circular_match", OK, *cm(ctx, vp(img), W, H, W, NULL, 0))
    expect("vo_circular_match", OK, *cm(ctx, vp(img), W, H, W, vp(pts), 4))
    expect("vo_triangulate", ARG, NULL, vp(P), vp(Pr), vp(pts), vp(pts), 4, vp(xyz))
    expect("vo_triangulate", ARG, ctx, NULL, vp(Pr), vp(pts), vp(pts), 4, vp(xyz))
    expect("vo_triangulate", ARG, ctx, vp(P), NULL, vp(pts), vp(pts), 4, vp(xyz))
    expect("vo_triangulate", ARG, ctx, vp(P), vp(Pr), NULL, vp(pts), 4, vp(xyz))
    expect("vo_triangulate", ARG, ctx, vp(P), vp(Pr), vp(pts), NULL, 4, vp(xyz))
    expect("vo_triangulate", ARG, ctx, vp(P), vp(Pr), vp(pts), vp(pts), 4, NULL)
    expect("vo_triangulate", ARG, ctx, vp(P), vp(Pr), vp(pts), vp(pts), -1, vp(xyz))
    expect("vo_triangulate", ARG, ctx, vp(P), vp(Pr), vp(pts), vp(pts), CAP + 1, vp(xyz))
    expect("vo_triangulate", OK, ctx, vp(P), vp(Pr), NULL, NULL, 0, NULL)
    expect("vo_pnp_ransac", ARG, NULL, vp(xyz), vp(pts), 10, vp(K), vp(rv), vp(tv), vp(R9), vp(idx), C.byref(n_out))
    expect("vo_pnp_ransac", ARG, ctx, NULL, vp(pts), 10, vp(K), vp(rv), vp(tv), vp(R9), vp(idx), C.byref(n_out))
    expect("vo_pnp_ransac", ARG, ctx, vp(xyz), NULL, 10, vp(K), vp(rv), vp(tv), vp(R9), vp(idx), C.byref(n_out))
    expect("vo_pnp_ransac", ARG, ctx, vp(xyz), vp(pts), 10, NULL, vp(rv), vp(tv), vp(R9), vp(idx), C.byref(n_out))
    expect("vo_pnp_ransac", ARG, ctx, vp(xyz), vp(pts), -1, vp(K), vp(rv), vp(tv), vp(R9), vp(idx), C.byref(n_out))
    expect("vo_pnp_ransac", ARG, ctx, vp(xyz), vp(pts), CAP + 1, vp(K), vp(rv), vp(tv), vp(R9), vp(idx), C.byref(n_out))
    expect("vo_pnp_ransac", TOO_FEW, ctx, vp(xyz), vp(pts), 3, vp(K), vp(rv), vp(tv), vp(R9), vp(idx), C.byref(n_out))
    expect("vo_pnp_ransac", TOO_FEW, ctx, NULL, NULL, 0, vp(K), NULL, NULL, NULL, NULL, NULL)
    expect("vo_essential_pose", ARG, NULL, vp(pts), vp(pts), 10, C.c_double(300), C.c_double(160), C.c_double(48), C.c_double(0.999),
           C.c_double(1.0), vp(dbuf), vp(dbuf), vp(dbuf), NULL, NULL)
    for bad_n, p0, p1 in ((-1, vp(pts), vp(pts)), (CAP + 1, vp(pts), vp(pts)), (10, NULL, vp(pts)), (10, vp(pts), NULL)):
        expect("vo_essential_pose", ARG, ctx, p0, p1, bad_n, C.c_double(300), C.c_double(160), C.c_double(48), C.c_double(0.999),
               C.c_double(1.0), vp(dbuf), vp(dbuf), vp(dbuf), NULL, NULL)
    for prob, thr in ((0.0, 1.0), (1.0, 1.0), (0.999, 0.0), (0.999, -1.0)):
        expect("vo_essential_pose", ARG, ctx, vp(pts), vp(pts), 10, C.c_double(300), C.c_double(160), C.c_double(48), C.c_double(prob),
               C.c_double(thr), vp(dbuf), vp(dbuf), vp(dbuf), NULL, NULL)
    expect("vo_essential_pose", TOO_FEW, ctx, vp(pts), vp(pts), 4, C.c_double(300), C.c_double(160), C.c_double(48), C.c_double(0.999),
           C.c_double(1.0), vp(dbuf), vp(dbuf), vp(dbuf), NULL, NULL)
    corners = np.zeros((4096, 2), np.float32)
    expect("vo_fast_detect", ARG, NULL, vp(img), W, H, W, 20, 1, vp(corners), 4096, C.byref(n_out))
    # img == NULL: the left image of the pair the last vo_circular_match / vo_track_frame kept (there is one: the call above);
    # no such pair of another size
    expect("vo_fast_detect", OK, ctx, NULL, W, H, W, 20, 1, vp(corners), 4096, C.byref(n_out))
    expect("vo_fast_detect", STATE, ctx, NULL, W - 8, H, W - 8, 20, 1, vp(corners), 4096, C.byref(n_out))
    for bad in ((vp(img), W + 1, H, W + 1, vp(corners), 4096, C.byref(n_out)),
                (vp(img), W, H + 1, W, vp(corners), 4096, C.byref(n_out)), (vp(img), W, H, W - 1, vp(corners), 4096, C.byref(n_out)),
                (vp(img), 8, H, 8, vp(corners), 4096, C.byref(n_out)), (vp(img), W, H, W, vp(corners), -1, C.byref(n_out)),
                (vp(img), W, H, W, NULL, 4096, C.byref(n_out)), (vp(img), W, H, W, vp(corners), 4096, NULL)):
        expect("vo_fast_detect", ARG, ctx, bad[0], bad[1], bad[2], bad[3], 20, 1, bad[4], bad[5], bad[6])
    expect("vo_fast_detect", OK, ctx, vp(img), W, H, W, 20, 1, NULL, 0, C.byref(n_out))
    ages = np.zeros(CAP + 8, np.int32)
    np_, na_ = C.c_int(0), C.c_int(0)
    expect("vo_detect_bucket", ARG, NULL, vp(img), W, H, W, NULL, vp(pts), C.byref(np_), vp(ages), C.byref(na_), CAP)
    expect("vo_detect_bucket", OK, ctx, NULL, W, H, W, NULL, vp(pts), C.byref(np_), vp(ages), C.byref(na_), CAP)   # (the kept pair)
    expect("vo_detect_bucket", STATE, ctx, NULL, W - 8, H, W - 8, NULL, vp(pts), C.byref(np_), vp(ages), C.byref(na_), CAP)
    np_.value = na_.value = 0
    for bad in ((vp(img), W + 1, H, W + 1, vp(pts), C.byref(np_), vp(ages), C.byref(na_), CAP),
                (vp(img), W, H, W - 1, vp(pts), C.byref(np_), vp(ages), C.byref(na_), CAP),
                (vp(img), W, H, W, NULL, C.byref(np_), vp(ages), C.byref(na_), CAP),
                (vp(img), W, H, W, vp(pts), NULL, vp(ages), C.byref(na_), CAP),
                (vp(img), W, H, W, vp(pts), C.byref(np_), NULL, C.byref(na_), CAP),
                (vp(img), W, H, W, vp(pts), C.byref(np_), vp(ages), NULL, CAP),
                (vp(img), W, H, W, vp(pts), C.byref(np_), vp(ages), C.byref(na_), -1)):
        expect("vo_detect_bucket", ARG, ctx, bad[0], bad[1], bad[2], bad[3], NULL, *bad[4:])
    np_.value = na_.value = 0
    expect("vo_detect_bucket", ARG, ctx, vp(img), W, H, W, NULL, vp(pts), C.byref(np_), vp(ages), C.byref(na_), 0)
    expect("vo_detect_bucket", OK, ctx, vp(img), W, H, W, NULL, vp(pts), C.byref(np_), vp(ages), C.byref(na_), CAP)
    for npts, nages in ((-1, 0), (2, 1), (CAP + 1, CAP + 1)):
        np_.value, na_.value = npts, nages
        expect("vo_detect_bucket", ARG, ctx, vp(img), W, H, W, NULL, vp(pts), C.byref(np_), vp(ages), C.byref(na_), CAP)

    def tf(ctx_, l0, w, h, stride, p, n, pl, pr, nout=C.byref(n_out)):
        return (ctx_, l0, vp(img), vp(img), vp(img), w, h, stride, p, n, pl, pr, vp(out[0]), vp(out[1]), vp(out[2]), vp(out[3]),
                vp(xyz), vp(idx), nout, NULL, NULL, vp(rv), vp(tv), vp(R9), NULL, NULL)
    expect("vo_track_frame", ARG, *tf(NULL, vp(img), W, H, W, vp(pts), 4, vp(P), vp(Pr)))
    expect("vo_track_frame", ARG, *tf(ctx, NULL, W, H, W, vp(pts), 4, vp(P), vp(Pr)))
    expect("vo_track_frame", ARG, *tf(ctx, vp(img), W + 1, H, W + 1, vp(pts), 4, vp(P), vp(Pr)))
    expect("vo_track_frame", ARG, *tf(ctx, vp(img), W, H, W - 1, vp(pts), 4, vp(P), vp(Pr)))
    expect("vo_track_frame", ARG, *tf(ctx, vp(img), W, H, W, vp(pts), CAP + 1, vp(P), vp(Pr)))
    expect("vo_track_frame", ARG, *tf(ctx, vp(img), W, H, W, vp(pts), -1, vp(P), vp(Pr)))
    expect("vo_track_frame", ARG, *tf(ctx, vp(img), W, H, W, NULL, 4, vp(P), vp(Pr)))
    expect("vo_track_frame", ARG, *tf(ctx, vp(img), W, H, W, vp(pts), 4, NULL, vp(Pr)))
    expect("vo_track_frame", ARG, *tf(ctx, vp(img), W, H, W, vp(pts), 4, vp(P), NULL))
    expect("vo_track_frame", (OK, TOO_FEW, 1), *tf(ctx, vp(img), W, H, W, vp(pts), 4, vp(P), vp(Pr), NULL))  # n_out is optional
    expect("vo_track_frame", (OK, TOO_FEW, 1), *tf(ctx, vp(img), W, H, W, vp(pts), 4, vp(P), vp(Pr)))  # blank images: nothing tracks
    # both t0 images NULL: the kept pair (the call above left one); refused calls (a bad stride, another size) leave it alone
    def tfk(w, h, stride):
        t = list(tf(ctx, NULL, w, h, stride, vp(pts), 4, vp(P), vp(Pr)))
        t[2] = NULL
        return t
    expect("vo_track_frame", ARG, *tfk(W, H, W - 1))
    expect("vo_track_frame", STATE, *tfk(W - 8, H, W - 8))
    expect("vo_track_frame", (OK, TOO_FEW, 1), *tfk(W, H, W))
    expect("vo_batch_upload_image", OK, ctx, 0, vp(img), W)          # the batch API takes the image table over
    expect("vo_track_frame", STATE, *tfk(W, H, W))
    expect("vo_fast_detect", STATE, ctx, NULL, W, H, W, 20, 1, vp(corners), 4096, C.byref(n_out))
    expect("vo_track_frame", (OK, TOO_FEW, 1), *tf(ctx, vp(img), W, H, W, vp(pts), 4, vp(P), vp(Pr)))

    # ---- lock-step sequence loop ----
    expect("vo_seq_configure", ARG, NULL, 2, W, H, 3, 8)
    for bad in ((0, W, H, 3, 8), (FRAMES + 1, W, H, 3, 8), (2, W + 1, H, 3, 8), (2, W, H + 1, 3, 8), (2, 16, H, 3, 8), (2, W, H, 1, 8),
                (2, W, H, 4, 8), (2, W, H, 3, 0), (-1, W, H, 3, 8)):
        expect("vo_seq_configure", ARG, ctx, *bad)
    expect("vo_seq_configure", OK, ctx, 2, W, H, 3, 8)
    expect("vo_batch_run", STATE, ctx, 31)
    expect("vo_batch_set_points", STATE, ctx, 0, vp(pts), 1)
    expect("vo_batch_upload_image", STATE, ctx, 0, vp(img), W)
    expect("vo_seq_push_pair", ARG, NULL, 0, vp(img), vp(img), W, 0)
    for bad in ((-1, vp(img), vp(img), W), (2, vp(img), vp(img), W), (0, NULL, vp(img), W), (0, vp(img), NULL, W), (0, vp(img), vp(img), W - 1)):
        expect("vo_seq_push_pair", ARG, ctx, bad[0], bad[1], bad[2], bad[3], 0)
        expect("vo_seq_push_pair_dev", ARG, ctx, bad[0], bad[1], bad[2], bad[3])
    expect("vo_seq_push_pair", ARG, ctx, 0, vp(img), vp(img), W, 1)  # "page-locked" memory that is not
    expect("vo_seq_push_pair_dev", ARG, NULL, 0, vp(img), vp(img), W)
    ids = (C.c_int32 * 2)(0, 1)
    ptrs = (C.c_void_p * 2)(img.ctypes.data, img.ctypes.data)
    expect("vo_seq_push_pairs", ARG, NULL, 2, ids, ptrs, ptrs, W, 0)
    expect("vo_seq_push_pairs", ARG, ctx, -1, ids, ptrs, ptrs, W, 0)
    expect("vo_seq_push_pairs", ARG, ctx, 2, NULL, ptrs, ptrs, W, 0)
    expect("vo_seq_push_pairs", ARG, ctx, 2, ids, NULL, ptrs, W, 0)
    expect("vo_seq_push_pairs", ARG, ctx, 2, ids, ptrs, NULL, W, 0)
    expect("vo_seq_push_pairs", ARG, ctx, 2, ids, ptrs, ptrs, W, 3)
    expect("vo_seq_push_pairs", ARG, ctx, 2, ids, ptrs, ptrs, W, -1)
    expect("vo_seq_push_pairs", OK, ctx, 0, NULL, NULL, NULL, W, 0)
    expect("vo_seq_push_pair", OK, ctx, 0, vp(img), vp(img), W, 0)
    expect("vo_seq_push_pair", STATE, ctx, 0, vp(img), vp(img), W, 0)  # twice for one step
    expect("vo_seq_step", ARG, NULL)
    expect("vo_seq_step", OK, ctx)
    expect("vo_seq_sync", ARG, NULL)
    expect("vo_seq_sync", OK, ctx)
    expect("vo_seq_reset", ARG, NULL, 0)
    expect("vo_seq_reset", ARG, ctx, 2)
    expect("vo_seq_get_state", ARG, NULL, 0, NULL, NULL, NULL, NULL, NULL)
    expect("vo_seq_get_state", ARG, ctx, -1, NULL, NULL, NULL, NULL, NULL)
    expect("vo_seq_get_state", ARG, ctx, 2, NULL, NULL, NULL, NULL, NULL)
    expect("vo_seq_get_state", OK, ctx, 0, NULL, NULL, NULL, NULL, NULL)
    expect("vo_seq_get_trajectory", ARG, NULL, 0, 0, 1, vp(dbuf), NULL, C.byref(n_out))
    expect("vo_seq_get_trajectory", ARG, ctx, -1, 0, 1, vp(dbuf), NULL, C.byref(n_out))
    expect("vo_seq_get_trajectory", ARG, ctx, 2, 0, 1, vp(dbuf), NULL, C.byref(n_out))
    expect("vo_seq_get_trajectory", ARG, ctx, 0, -1, 1, vp(dbuf), NULL, C.byref(n_out))
    expect("vo_seq_get_trajectory", ARG, ctx, 0, 0, -1, vp(dbuf), NULL, C.byref(n_out))
    expect("vo_seq_get_trajectory", OK, ctx, 0, 0, 0, NULL, NULL, C.byref(n_out))
    expect("vo_seq_reset", OK, ctx, -1)

    lib.vo_destroy(ctx)
    missing = sorted(set(_lib.EXPORTS) - covered)
    print(json.dumps({"checked": checked[0], "exports_covered": len(covered), "not_covered": missing, "failures": fails}))
    return 1 if (fails or missing) else 0


if __name__ == "__main__":
    sys.exit(main())
