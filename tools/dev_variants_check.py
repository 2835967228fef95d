"""Developer check (GPU, libvo_hip_dev.so): the kernel variants that were built, measured and lost stay CORRECT in the
developer build, so that the A/B measurements quoted in DESIGN.md can be repeated.  Not part of the product test-suite --
the product library does not contain these kernels (python -m visual_odom_amd.build --dev builds the library this loads).

    VO_HIP_LIB=$PWD/visual_odom_amd/libvo_hip_dev.so python tools/dev_variants_check.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VO_HIP_LIB", os.path.join(ROOT, "visual_odom_amd", "libvo_hip_dev.so"))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def check_two_features_per_wave_lk_kernel_is_bit_identical(volib, small_seq):
    """VERDICT r01 item 3 asked for the 2-features-per-wavefront LK variant to be built and measured: lk_circular_pair_kernel
    (VO_LK_PAIR=1).  Same tracks and status bytes as lk_circular_kernel on odd and even numbers of points (a wavefront with
    an empty half), points off the image and a batch of several frames (forward, backward and a static quadruple), in both
    retirement modes."""
    s = small_seq
    h, w = s["L"][0].shape
    rng = np.random.default_rng(11)
    extra = np.stack([rng.uniform(-30, w + 30, 40), rng.uniform(-30, h + 30, 40)], 1).astype(np.float32)
    pts = [np.vstack([s["pts"][k % 2], extra])[:len(s["pts"][k % 2]) + 40 - (k % 2)] for k in range(3)]
    out = {}
    for pair in ("0", "1"):
        os.environ["VO_LK_PAIR"] = pair
        ctx = volib.Context(0, w, h, 4096, 3)
        try:
            res = []
            for full in (1, 0):
                ctx.set_params(lk_full_chain=full)
                ctx.batch_configure(6, w, h, 3)
                for k in range(3):
                    ctx.batch_upload_image(2 * k, s["L"][k])
                    ctx.batch_upload_image(2 * k + 1, s["R"][k])
                ctx.batch_set_quads([[0, 1, 2, 3], [4, 5, 2, 3], [4, 5, 4, 5]])
                for k in range(3):
                    ctx.batch_set_points(k, pts[k])
                ctx.batch_run(volib.STAGE_PYRAMID | volib.STAGE_LK)
                ctx.batch_sync()
                res.append([ctx.batch_get_tracks(k, len(pts[k])) for k in range(3)])
            out[pair] = res
        finally:
            ctx.close()
    for a, b in zip(out["0"], out["1"]):
        for fa, fb in zip(a, b):
            assert np.array_equal(fa["status4"], fb["status4"])
            for key in ("r0", "r1", "l1", "l0_ret"):
                assert np.array_equal(bits(fa[key]), bits(fb[key])), key          # bit for bit
            assert fa["status4"].all(0).sum() > 100


def main():
    from visual_odom_amd import _lib, synth
    world = synth.StereoWorld(seed=11, width=480, height=160, fx=300.0, cx=239.5, cy=79.5, bf=-160.0, tex_size=1024)
    lefts, rights, poses, depths = world.render_sequence(3)
    pts = [synth.select_keypoints(lefts[k], bucket=16, per_bucket=2) for k in range(2)]
    small_seq = dict(L=lefts, R=rights, poses=poses, depths=depths, pts=pts)
    check_two_features_per_wave_lk_kernel_is_bit_identical(_lib, small_seq)
    print("lk_circular_pair_kernel: bit-identical to lk_circular_kernel")
    # the 128-register pose kernels (vo_set_schedule accepts pose_waves = 4 in the developer build only)
    ctx = _lib.Context(0, 480, 160, 4096, 1)
    P_l, P_r = world.proj_matrices()
    ref = None
    for waves in (1, 2, 4):
        ctx.set_schedule(pose_waves=waves, pose_streams=1, prepare=0)
        got = ctx.track_frame(lefts[0], rights[0], lefts[1], rights[1], pts[0], P_l, P_r)
        assert got["rc"] == 0
        if ref is None:
            ref = got
        assert np.array_equal(got["rvec"], ref["rvec"]) and np.array_equal(got["inliers"], ref["inliers"]), waves
    ctx.close()
    print("epnp / select_refine <1>, <2>, <4>: identical poses and inlier sets")


if __name__ == "__main__":
    main()
