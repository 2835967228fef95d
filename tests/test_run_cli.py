"""visual_odom_amd.run -- the reference's `./run <sequence_dir> <calibration.yaml> [gt_poses]` front end: the host-side
pieces on CPU (calibration YAML, image decoding, BGR2GRAY), the whole command on the GPU."""
import os

import numpy as np
import pytest


def test_calibration_yaml_and_projection(tmp_path):
    from visual_odom_amd import run, synth
    p = tmp_path / "kitti00.yaml"
    p.write_text("%YAML:1.0\n\n# Camera calibration\nCamera.fx: 718.8560\nCamera.fy: 718.8560\nCamera.cx: 607.1928\n"
                 "Camera.cy: 185.2157\n\n# stereo baseline times fx\nCamera.bf: -386.1448\n\nThDepth: 35\n")
    cal = run.read_calibration(str(p))
    P_l, P_r = run.projection_matrices(cal)
    ref_l, ref_r = synth.proj_matrices()          # the same numbers (calibration/kitti00.yaml), main.cpp:73-74
    assert np.array_equal(P_l, ref_l) and np.array_equal(P_r, ref_r)
    (tmp_path / "bad.yaml").write_text("Camera.fx: 1.0\n")
    with pytest.raises(ValueError):
        run.read_calibration(str(tmp_path / "bad.yaml"))


def test_image_reading_and_gray_conversion(tmp_path):
    from PIL import Image
    from visual_odom_amd import run
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    Image.fromarray(img, "L").save(tmp_path / "g.png")
    Image.fromarray(np.stack([img] * 3, -1), "RGB").save(tmp_path / "c.png")   # gray stored as colour: identity (B10)
    with open(tmp_path / "g.pgm", "wb") as f:
        f.write(b"P5\n# a comment\n53 37\n255\n" + img.tobytes())
    for name in ("g.png", "c.png", "g.pgm"):
        assert np.array_equal(run.read_gray(str(tmp_path / name)), img), name
    assert run.read_gray(str(tmp_path / "missing.png")) is None
    # cv::cvtColor(BGR2GRAY) on 8-bit: the well-known values of pure blue / green / red, white stays white
    bgr = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [12, 200, 99]]], np.uint8)
    assert list(run.bgr_to_gray(bgr)[0]) == [29, 150, 76, 255, (12 * 1868 + 200 * 9617 + 99 * 4899 + 8192) >> 14]
    d = tmp_path / "seq"
    for cam in (0, 1):
        (d / ("image_%d" % cam)).mkdir(parents=True)
        Image.fromarray(img, "L").save(d / ("image_%d" % cam) / "000000.png")
    pair = run.read_pair(str(d), 0)
    assert np.array_equal(pair[0], img) and run.read_pair(str(d), 1) is None


@pytest.mark.gpu
def test_run_command_on_png_sequences(volib, tmp_path):
    """two rendered sequences of different lengths written as PNG + a calibration file + ground truth: the command's
    trajectories equal the lock-step loop fed directly, and the evaluation against ground truth is reported"""
    from PIL import Image
    from visual_odom_amd import run, odometry, synth
    kw = dict(width=480, height=160, fx=300.0, cx=239.5, cy=79.5, bf=-160.0, tex_size=1024)
    worlds = [synth.StereoWorld(seed=31 + s, **kw) for s in range(2)]
    lengths = [8, 6]
    seqs = [w.render_sequence(n) for w, n in zip(worlds, lengths)]
    dirs, gts = [], []
    for s, (L, R, poses, _) in enumerate(seqs):
        d = tmp_path / ("%02d" % s)
        for cam, imgs in ((0, L), (1, R)):
            (d / ("image_%d" % cam)).mkdir(parents=True)
            for k, img in enumerate(imgs):
                Image.fromarray(img, "L").save(d / ("image_%d" % cam) / ("%06d.png" % k))
        T0inv = np.linalg.inv(poses[0])
        g = tmp_path / ("%02d_gt.txt" % s)
        with open(g, "w") as f:
            for T in poses:
                f.write(" ".join("%.9e" % v for v in (T0inv @ T)[:3].reshape(-1)) + "\n")
        dirs.append(str(d))
        gts.append(str(g))
    cal = tmp_path / "cal.yaml"
    cal.write_text("%YAML:1.0\nCamera.fx: 300.0\nCamera.fy: 300.0\nCamera.cx: 239.5\nCamera.cy: 79.5\nCamera.bf: -160.0\n")
    res = run.main([",".join(dirs), str(cal), ",".join(gts), "--out", str(tmp_path / "poses"), "--max-frames", "20",
                    "--features-per-bucket", "2"])
    P_l, P_r = worlds[0].proj_matrices()
    ctx = volib.Context(0, 480, 160, 4096, 2)
    try:
        vo = odometry.MultiSequenceOdometry(P_l, P_r, 2, 480, 160, ctx=ctx, ring=3, max_steps=32, features_per_bucket=2)
        for k in range(max(lengths)):
            for s in range(2):
                if k < lengths[s]:
                    vo.push(s, seqs[s][0][k], seqs[s][1][k])
            vo.step()
        for s in range(2):
            got = odometry.load_poses(res[s]["trajectory"])
            assert got.shape == (lengths[s], 3, 4) and res[s]["frames"] == lengths[s]
            assert np.abs(got - np.asarray(vo.trajectory(s))).max() < 1e-8
            assert res[s]["ate_rmse_m"] < 0.3 and res[s]["integrated"] >= lengths[s] - 2
            assert "kitti_segment_errors" in res[s]          # None here: the sequences are shorter than 100 m
    finally:
        ctx.close()
