"""Where the time of ONE pose solve goes (developer build): 100 MHz device time stamps that hypothesis 0 of frame 0 leaves
in epnp_kernel (vo_epnp.h, VO_EPNP_STAMP) and thread 0 of frame 0 in select_refine_kernel, read after vo_track_frame calls.
The ONE-kernel EPnP is what is stamped (VO_EPNP_SPLIT_MAX=0 is set here); the four-kernel form used for small launches shows
its parts as kernels in tools/kernel_timeline.py.

    VO_HIP_LIB=visual_odom_amd/libvo_hip_dev.so python tools/pose_phases.py [per_bucket] [calls]"""
import ctypes as C
import os
import sys

os.environ.setdefault("VO_EPNP_SPLIT_MAX", "0")

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tools.latency_mode import inputs


def main():
    per_bucket = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    from visual_odom_amd import _lib
    if not hasattr(_lib.load(), "vo_dev_pose_prof"):
        raise SystemExit("needs the developer build: VO_HIP_LIB=.../libvo_hip_dev.so (python -m visual_odom_amd.build --dev)")
    world, L, R, P_l, P_r, pts = inputs(per_bucket)
    ctx = _lib.Context(0, world.w, world.h, 4096, 1)
    ctx.set_schedule(pose_waves=1, pose_streams=1, prepare=0)  # what the probe picks for the synchronous call
    names = ["normalise, control points, alphas", "M^T M (12 x 12)", "Jacobi SVD 12 x 12", "L_6x10, rho",
             "approximation 1 (+ Gauss-Newton, R, t)", "approximation 2", "approximation 3", "best of three + Rodrigues"]
    acc = np.zeros(8)
    ref = np.zeros(6)
    for i in range(calls):
        k = i % 4
        ctx.track_frame(L[k], R[k], L[k + 1], R[k + 1], pts[k], P_l, P_r)
        buf = (C.c_longlong * 64)()
        ctx._chk(ctx.lib.vo_dev_pose_prof(ctx.h, buf))
        t = np.array(buf[:32], dtype=np.float64) / 100.0  # us
        if i >= 2:
            acc += np.diff(t[:9])
            ref += [t[17] - t[16], buf[18] / 100.0, buf[19], t[20] - t[17], buf[21], 1]
    n = ref[5]
    print("passes of the last call (us per pass): without J: rodrigues %.1f, points %.1f, reduction + barriers %.1f (%d passes); "
          "with J: %.1f, %.1f, %.1f (%d passes)" % tuple(
              [buf[22 + k] / 100.0 / max(buf[25], 1) for k in range(3)] + [buf[25]] +
              [buf[26 + k] / 100.0 / max(buf[29], 1) for k in range(3)] + [buf[29]]))
    print("EPnP, hypothesis 0 of the frame (us, mean of %d calls, %d points per frame):" % (n, len(pts[0])))
    for i in range(8):
        print("  %-40s %7.1f" % (names[i], acc[i] / n))
    print("  %-40s %7.1f" % ("total", acc.sum() / n))
    print("refinement kernel, frame 0: inlier mask + compaction %.1f us; LM loop %.1f us of which %.1f in %.1f SVD solves "
          "(%.1f us each); %d inliers" % (ref[0] / n, ref[3] / n, ref[1] / n, ref[2] / n, ref[1] / max(ref[2], 1), ref[4] / n))


if __name__ == "__main__":
    main()
