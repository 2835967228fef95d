"""Latency mode of the drop-in boundary: vo_track_frame per call with HOST images (4 uploads of 466 KB
over PCIe, one download, synchronous), one frame in flight -- the honest "switch the reference over"
number -- and with the t0 pair kept on the device from the previous call (2 uploads), next to the whole frame loop through
the streaming ring, through the stateless calls, through the calls on the kept pair, and PIPELINED
through the lock-step sequence API with one sequence.  Never bench.py's `value` (that one has inputs resident
in HBM); quoted in DESIGN.md section 5.

Every measurement runs in its own process: the HIP runtime multiplexes a process's streams onto a few hardware
queues, and a second vo_ctx created in the same process gets a different (measurably slower: 0.8 instead of 0.63 ms
per lock-step step) mapping than the first.

    python tools/latency_mode.py [n_frames]"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def inputs(per_bucket):
    from visual_odom_amd import synth
    world = synth.StereoWorld(seed=20260925)
    L, R, poses, _ = world.render_sequence(5)
    P_l, P_r = world.proj_matrices()
    pts = [synth.select_keypoints(L[k], bucket=37, per_bucket=per_bucket) for k in range(5)]
    return world, L, R, P_l, P_r, pts


ORDER = [0, 1, 2, 3, 4, 3, 2, 1]  # ping-pong over the rendered pairs: always adjacent frames


def run(what, per_bucket, n):
    from visual_odom_amd import _lib, odometry
    world, L, R, P_l, P_r, pts = inputs(per_bucket)
    if what == "trackonly":  # for a kernel trace of vo_track_frame alone (tools/kernel_timeline.py)
        ctx = _lib.Context(0, world.w, world.h, 4096, 1)
        for i in range(n):
            k = i % 4
            ctx.track_frame(L[k], R[k], L[k + 1], R[k + 1], pts[k], P_l, P_r)
        print("  schedule", ctx.get_schedule(), ctx.get_probe_log())
        return
    if what == "trackkeptonly":  # the same on the kept pair (two new images per call)
        ctx = _lib.Context(0, world.w, world.h, 4096, 1)
        ctx.track_frame(L[0], R[0], L[1], R[1], pts[0], P_l, P_r)
        for i in range(1, n):
            a, b = ORDER[i % 8], ORDER[(i + 1) % 8]
            ctx.detect_bucket(None, np.zeros((0, 2), np.float32), np.zeros(0, np.int32), features_per_bucket=per_bucket)
            ctx.track_frame(None, None, L[b], R[b], pts[a], P_l, P_r)
        return
    if what == "track":
        ctx = _lib.Context(0, world.w, world.h, 4096, 1)
        for k in range(4):
            ctx.track_frame(L[k], R[k], L[k + 1], R[k + 1], pts[k], P_l, P_r)
        t0 = time.perf_counter()
        for i in range(n):
            k = i % 4
            ctx.track_frame(L[k], R[k], L[k + 1], R[k + 1], pts[k], P_l, P_r)
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        for i in range(n):
            ctx.detect_bucket(L[i % 4], np.zeros((0, 2), np.float32), np.zeros(0, np.int32), features_per_bucket=per_bucket)
        dt2 = time.perf_counter() - t1
        print("  track_frame %.2f ms/frame = %.0f frames/s (PCIe-inclusive, %d points); detect_bucket %.2f ms/frame"
              % (1e3 * dt / n, n / dt, len(pts[0]), 1e3 * dt2 / n))
    elif what == "adapter":  # the three calls at the reference's own function boundaries (INTEGRATION.md's adapter)
        ctx = _lib.Context(0, world.w, world.h, 4096, 1)
        K = world.K()

        def frame(k):
            cm = ctx.circular_match(L[k], R[k], L[k + 1], R[k + 1], pts[k], apply_consistency=True)
            xyz = ctx.triangulate(P_l, P_r, cm["l0"], cm["r0"])
            return ctx.pnp_ransac(xyz, cm["l1"], K)
        for k in range(4):
            frame(k)
        tt = [0.0, 0.0, 0.0]
        t0 = time.perf_counter()
        for i in range(n):
            k = i % 4
            a = time.perf_counter()
            cm = ctx.circular_match(L[k], R[k], L[k + 1], R[k + 1], pts[k], apply_consistency=True)
            b = time.perf_counter()
            xyz = ctx.triangulate(P_l, P_r, cm["l0"], cm["r0"])
            c_ = time.perf_counter()
            ctx.pnp_ransac(xyz, cm["l1"], K)
            d = time.perf_counter()
            tt[0] += b - a; tt[1] += c_ - b; tt[2] += d - c_
        dt = time.perf_counter() - t0
        print("  adapter calls vo_circular_match + vo_triangulate + vo_pnp_ransac: %.2f ms/frame = %.0f frames/s (%.2f + %.2f + %.2f)"
              % (1e3 * dt / n, n / dt, 1e3 * tt[0] / n, 1e3 * tt[1] / n, 1e3 * tt[2] / n))
    elif what == "adapterkept":  # the same three calls with the adapter's opt-in (vo_adapter_keep_pair): circularMatching on the kept pair
        ctx = _lib.Context(0, world.w, world.h, 4096, 1)
        K = world.K()
        ctx.circular_match(L[0], R[0], L[1], R[1], pts[0], apply_consistency=True)
        tt = [0.0, 0.0, 0.0]
        t0 = None
        for i in range(1, 9 + n):
            if i == 9:
                t0 = time.perf_counter()
                tt = [0.0, 0.0, 0.0]
            a_, b_ = ORDER[i % 8], ORDER[(i + 1) % 8]
            a = time.perf_counter()
            cm = ctx.circular_match(None, None, L[b_], R[b_], pts[a_], apply_consistency=True)
            b = time.perf_counter()
            xyz = ctx.triangulate(P_l, P_r, cm["l0"], cm["r0"])
            c_ = time.perf_counter()
            ctx.pnp_ransac(xyz, cm["l1"], K)
            d = time.perf_counter()
            tt[0] += b - a; tt[1] += c_ - b; tt[2] += d - c_
        dt = time.perf_counter() - t0
        print("  adapter calls, circularMatching on the kept pair (vo_adapter_keep_pair): %.2f ms/frame = %.0f frames/s (%.2f + %.2f + %.2f)"
              % (1e3 * dt / n, n / dt, 1e3 * tt[0] / n, 1e3 * tt[1] / n, 1e3 * tt[2] / n))
    elif what == "trackkept":  # the t0 pair = the previous call's t1 pair, kept on the device: two images per call
        ctx = _lib.Context(0, world.w, world.h, 4096, 1)
        ctx.track_frame(L[0], R[0], L[1], R[1], pts[0], P_l, P_r)
        for i in range(1, 9):
            a, b = ORDER[i % 8], ORDER[(i + 1) % 8]
            ctx.track_frame(None, None, L[b], R[b], pts[a], P_l, P_r)
        t0 = time.perf_counter()
        for i in range(9, 9 + n):
            a, b = ORDER[i % 8], ORDER[(i + 1) % 8]
            ctx.track_frame(None, None, L[b], R[b], pts[a], P_l, P_r)
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        for i in range(n):
            ctx.detect_bucket(None, np.zeros((0, 2), np.float32), np.zeros(0, np.int32), features_per_bucket=per_bucket)
        dt2 = time.perf_counter() - t1
        print("  track_frame on the kept pair (2 new images per call) %.2f ms/frame = %.0f frames/s (PCIe-inclusive, ~%d points); "
              "detect_bucket on the kept image %.2f ms/frame" % (1e3 * dt / n, n / dt, len(pts[0]), 1e3 * dt2 / n))
    elif what in ("ring", "stateless", "kept"):
        vo = odometry.StereoOdometry(P_l, P_r, streaming=what == "ring", keep_pair=what == "kept", features_per_bucket=per_bucket)
        for i in range(9):
            vo.process(L[ORDER[i % 8]], R[ORDER[i % 8]])
        t2 = time.perf_counter()
        for i in range(9, 9 + n):
            vo.process(L[ORDER[i % 8]], R[ORDER[i % 8]])
        dt3 = time.perf_counter() - t2
        print("  frame loop incl. FAST + bucketing + pose integration, %s: %.2f ms/frame = %.0f frames/s"
              % ({"ring": "streaming ring (synchronous)", "stateless": "stateless drop-in calls (4 images per frame)",
                 "kept": "drop-in calls on the kept pair (2 images per frame)"}[what], 1e3 * dt3 / n, n / dt3))
    else:  # pipelined: vo_seq_* with one sequence, host images; nothing comes back until the trajectory is asked for
        vo = odometry.MultiSequenceOdometry(P_l, P_r, 1, world.w, world.h, ring=3, max_steps=5 * n + 32,
                                            features_per_bucket=per_bucket)
        for i in range(9):
            vo.push(0, L[ORDER[i % 8]], R[ORDER[i % 8]])
            vo.step()
        vo.sync()
        t4 = time.perf_counter()
        for i in range(9, 9 + 5 * n):
            vo.push(0, L[ORDER[i % 8]], R[ORDER[i % 8]])
            vo.step()
        vo.sync()
        dt4 = time.perf_counter() - t4
        print("  frame loop, lock-step sequence API with ONE sequence (pipelined: pose solve of frame k under detection + "
              "tracking of k + 1; host images): %.2f ms/frame = %.0f frames/s" % (1e3 * dt4 / (5 * n), 5 * n / dt4))


if __name__ == "__main__":
    if len(sys.argv) >= 4:
        run(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
    else:
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
        for name, per_bucket in (("~2000 points (6 per bucket)", 6), ("reference default (1 per bucket)", 1)):
            print(name + ":", flush=True)
            for what in ("track", "trackkept", "adapter", "adapterkept", "ring", "stateless", "kept", "pipelined"):
                subprocess.run([sys.executable, os.path.abspath(__file__), what, str(per_bucket), str(n)], check=False)
