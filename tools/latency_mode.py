"""Latency mode of the drop-in boundary: vo_track_frame per call with HOST images (4 uploads of 466 KB
over PCIe, one download, synchronous), one frame in flight -- the honest "switch the reference over"
number.  Never bench.py's `value` (that one has inputs resident in HBM); quoted in DESIGN.md section 5.
    python tools/latency_mode.py [n_frames]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visual_odom_amd import _lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for name, per_bucket in (("~2000 points (6 per bucket)", 6), ("reference default (1 per bucket)", 1)):
    world = synth.StereoWorld(seed=20260925)
    L, R, poses, _ = world.render_sequence(5)
    P_l, P_r = world.proj_matrices()
    pts = [synth.select_keypoints(L[k], bucket=37, per_bucket=per_bucket) for k in range(4)]
    ctx = _lib.Context(0, world.w, world.h, 4096, 1)
    for k in range(4):                       # warm-up
        ctx.track_frame(L[k], R[k], L[k + 1], R[k + 1], pts[k], P_l, P_r)
    t0 = time.perf_counter()
    for i in range(n):
        k = i % 4
        ctx.track_frame(L[k], R[k], L[k + 1], R[k + 1], pts[k], P_l, P_r)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for i in range(n):
        k = i % 4
        ctx.detect_bucket(L[k], np.zeros((0, 2), np.float32), np.zeros(0, np.int32), features_per_bucket=per_bucket)
    dt2 = time.perf_counter() - t1
    print("%s: track_frame %.2f ms/frame = %.0f frames/s (PCIe-inclusive, %d points); detect_bucket %.2f ms/frame"
          % (name, 1e3 * dt / n, n / dt, len(pts[0]), 1e3 * dt2 / n))
    # the whole frame loop (detect/bucket -> track -> pose -> integrate) as a stream: one upload and two
    # pyramids per frame (visual_odom_amd.odometry.StereoOdometry, streaming ring)
    from visual_odom_amd import odometry
    order = [0, 1, 2, 3, 4, 3, 2, 1]           # ping-pong over the rendered pairs: always adjacent frames
    for streaming in (True, False):
        vo = odometry.StereoOdometry(P_l, P_r, ctx=ctx, streaming=streaming, features_per_bucket=per_bucket)
        for i in range(9):
            vo.process(L[order[i % 8]], R[order[i % 8]])
        t2 = time.perf_counter()
        for i in range(9, 9 + n):
            vo.process(L[order[i % 8]], R[order[i % 8]])
        dt3 = time.perf_counter() - t2
        print("    frame loop incl. FAST + bucketing + pose integration, %s: %.2f ms/frame = %.0f frames/s"
              % ("streaming ring" if streaming else "stateless drop-in calls", 1e3 * dt3 / n, n / dt3))
    ctx.close()
    # the same loop PIPELINED (vo_seq_* with one sequence, host images): the pose solve of frame k runs under detection
    # and tracking of frame k + 1, nothing comes back until the trajectory is asked for
    vo = odometry.MultiSequenceOdometry(P_l, P_r, 1, world.w, world.h, ring=3, max_steps=n + 32, features_per_bucket=per_bucket)
    for i in range(9):
        vo.push(0, L[order[i % 8]], R[order[i % 8]])
        vo.step()
    vo.sync()
    t4 = time.perf_counter()
    for i in range(9, 9 + n):
        vo.push(0, L[order[i % 8]], R[order[i % 8]])
        vo.step()
    vo.sync()
    dt4 = time.perf_counter() - t4
    print("    frame loop, lock-step sequence API with ONE sequence (pipelined, host images): %.2f ms/frame = %.0f frames/s"
          % (1e3 * dt4 / n, n / dt4))
    vo.close()
