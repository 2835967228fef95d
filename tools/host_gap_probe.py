"""Where the host side of one synchronous drop-in call goes (PCIe-inclusive, one MI355X): each ABI call of single_frame_setup
timed alone in a loop, next to the whole vo_track_frame.  python tools/host_gap_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from visual_odom_amd import _lib, synth

world = synth.StereoWorld(seed=20260925)
L, R, _, _ = world.render_sequence(3)
P_l, P_r = world.proj_matrices()
pts = synth.select_keypoints(L[0], bucket=37, per_bucket=6)
ctx = _lib.Context(0, world.w, world.h, 4096, 1)
for _ in range(12):
    ctx.track_frame(L[0], R[0], L[1], R[1], pts, P_l, P_r)


def timed(f, n=300):
    for _ in range(10):
        f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return 1e6 * (time.perf_counter() - t0) / n


print("vo_track_frame                      %7.1f us" % timed(lambda: ctx.track_frame(L[0], R[0], L[1], R[1], pts, P_l, P_r)))
if hasattr(ctx.lib, "vo_dev_host_stamps"):  # developer build: where the call's host side goes (steady-clock stamps inside vo_track_frame)
    import ctypes as C
    acc, prev_exit, n = np.zeros(11), None, 0
    for i in range(220):
        ctx.track_frame(L[0], R[0], L[1], R[1], pts, P_l, P_r)
        buf = (C.c_longlong * 16)()
        ctx.lib.vo_dev_host_stamps(buf)
        t = np.array(buf[:10], dtype=np.float64) / 1e3
        if i >= 20:
            acc[:9] += np.diff(t)
            acc[9] += t[9] - t[0]
            acc[10] += t[0] - prev_exit
            n += 1
        prev_exit = t[9]
    names = ["configure + sync_all", "image 0 staged + copy enqueued", "image 1", "image 2", "image 3", "points enqueued",
             "set_projection + run_stages (all kernels enqueued)", "final stream synchronisation", "results copied out",
             "inside vo_track_frame", "between two calls (python harness)"]
    for k, v in zip(names, acc / n):
        print("    %-52s %7.1f us" % (k, v))
ctx.batch_configure(4, world.w, world.h, 1)
print("vo_batch_configure (same shape)     %7.1f us" % timed(lambda: ctx.batch_configure(4, world.w, world.h, 1)))
QUAD = (L[0], R[0], L[1], R[1])
print("vo_batch_upload_image x 4 (+ sync)  %7.1f us" % timed(lambda: [ctx.batch_upload_image(i, QUAD[i]) for i in range(4)]))
print("vo_batch_set_quads (cached)         %7.1f us" % timed(lambda: ctx.batch_set_quads([[0, 1, 2, 3]])))
print("vo_batch_set_points (2039 points)   %7.1f us" % timed(lambda: ctx.batch_set_points(0, pts)))
print("vo_batch_set_projection (cached)    %7.1f us" % timed(lambda: ctx.batch_set_projection(P_l, P_r)))
print("vo_batch_sync (idle)                %7.1f us" % timed(lambda: ctx.batch_sync()))
print("vo_get_params (a bare ABI call from python) %5.1f us" % timed(lambda: ctx.get_params()))
a = np.ascontiguousarray(L[0])
b = np.empty_like(a)
print("numpy copy of one 467 KB image      %7.1f us" % timed(lambda: np.copyto(b, a)))
ctx.batch_set_quads([[0, 1, 2, 3]])
ctx.batch_set_points(0, pts)
ctx.batch_set_projection(P_l, P_r)
ctx.batch_run(_lib.STAGE_ALL)
ctx.batch_sync()
print("vo_batch_run(all) + sync, 1 frame   %7.1f us" % timed(lambda: (ctx.batch_run(_lib.STAGE_ALL), ctx.batch_sync())))
print("vo_batch_get_filtered + get_pose    %7.1f us" % timed(lambda: (ctx.batch_get_filtered(0), ctx.batch_get_pose(0))))
ctx.close()
