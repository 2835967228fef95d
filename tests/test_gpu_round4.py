"""Round-4 GPU tests (VERDICT r03 "next round" item 1):
  * every image-taking entry point with a byte stride != width -- a padded buffer (stride = w + 37) and an ROI sub-view of
    a bigger image (the cv::Mat the reference would pass after `img(cv::Rect(...))`) -- bit for bit the contiguous call:
    vo_circular_match, vo_track_frame, vo_fast_detect, vo_detect_bucket, vo_batch_upload_image(_dev), and vo_seq_push_pair
    from all three kinds of memory (pageable, page-locked, device);
  * an argument sweep over all exports (tests/abi_sweep.py, in a child process): NULL, n = 0, n > capacity, sizes beyond the
    context's maximum, bad indices, wrong state -> the documented error code, never a fault;
  * two real ranks of bench.py sharing GPU 0 (the N > 1 launch line of the driver, gloo for the barrier).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def padded(img, extra=37, fill=0xA5):
    """the same pixels in a buffer whose rows are `extra` bytes longer (the padding holds garbage that must never be read
    as pixels)"""
    h, w = img.shape
    buf = np.full((h, w + extra), fill, np.uint8)
    buf[:, :w] = img
    return buf[:, :w]


def roi(img, x0=19, y0=7, fill=0x3C):
    """the same pixels as a sub-view of a bigger image (row stride = the big image's width)"""
    h, w = img.shape
    big = np.full((h + 2 * y0 + 3, w + 2 * x0 + 5), fill, np.uint8)
    big[y0:y0 + h, x0:x0 + w] = img
    return big[y0:y0 + h, x0:x0 + w]


VIEWS = [("stride w + 37", padded), ("ROI of a bigger image", roi)]


def same(a, b, keys):
    for k in keys:
        assert np.array_equal(a[k], b[k]), k


def test_strided_views_are_what_they_claim():
    img = np.arange(20 * 30, dtype=np.uint8).reshape(20, 30)
    for name, view in VIEWS:
        v = view(img)
        assert np.array_equal(v, img) and v.strides[0] > 30 and v.strides[1] == 1 and not v.flags["C_CONTIGUOUS"], name
    from visual_odom_amd import _lib
    arrs, stride = _lib._imgs(padded(img), padded(img))
    assert stride == 67 and arrs[0].ctypes.data == arrs[0].__array_interface__["data"][0]  # passed as they are, no copy
    arrs, stride = _lib._imgs(padded(img), roi(img))                                       # different strides: repacked
    assert stride == 30


@pytest.mark.parametrize("name,view", VIEWS)
def test_drop_in_calls_with_stride_not_width(volib, small_world, small_seq, name, view):
    """vo_circular_match / vo_track_frame / vo_fast_detect / vo_detect_bucket on strided inputs == on contiguous ones"""
    from visual_odom_amd import synth
    lefts, rights = small_seq["L"], small_seq["R"]
    h, w = lefts[0].shape
    P_l, P_r = small_world.proj_matrices()
    pts = synth.select_keypoints(lefts[0], bucket=h // 10, per_bucket=4)
    ctx = volib.Context(0, w, h, 4096, 1)
    try:
        quad = (lefts[0], rights[0], lefts[1], rights[1])
        ref = ctx.circular_match(*quad, pts)
        got = ctx.circular_match(*[view(a) for a in quad], pts)
        assert ref["n_out"] > 50
        same(ref, got, ("l0", "r0", "r1", "l1", "l0_ret", "status4", "keep_idx"))
        ref = ctx.track_frame(*quad, pts, P_l, P_r)
        got = ctx.track_frame(*[view(a) for a in quad], pts, P_l, P_r)
        assert ref["rc"] == 0 and len(ref["inliers"]) > 20
        same(ref, got, ("l0", "r0", "l1", "r1", "xyz", "keep_idx", "keep_idx_circ", "rvec", "tvec", "R", "inliers"))
        for nonmax in (True, False):
            a = ctx.fast_detect(lefts[0], 20, nonmax)
            b = ctx.fast_detect(view(lefts[0]), 20, nonmax)
            assert len(a) > 100 and np.array_equal(a, b)
        carried = pts[:40]
        ages = np.arange(len(carried), dtype=np.int32)
        pa, aa = ctx.detect_bucket(lefts[0], carried, ages, features_per_bucket=2)
        pb, ab = ctx.detect_bucket(view(lefts[0]), carried, ages, features_per_bucket=2)
        assert len(pa) > 20 and np.array_equal(pa, pb) and np.array_equal(aa, ab)
    finally:
        ctx.close()


@pytest.mark.parametrize("name,view", VIEWS)
def test_batch_uploads_with_stride_not_width(volib, small_world, small_seq, name, view):
    """vo_batch_upload_image from a strided host buffer and vo_batch_upload_image_dev from a strided device buffer: the
    pyramids and the whole frame equal those of the contiguous upload"""
    from test_gpu_sequences import _DeviceImages
    from visual_odom_amd import synth
    lefts, rights = small_seq["L"], small_seq["R"]
    h, w = lefts[0].shape
    P_l, P_r = small_world.proj_matrices()
    pts = synth.select_keypoints(lefts[0], bucket=h // 10, per_bucket=4)
    quad = (lefts[0], rights[0], lefts[1], rights[1])
    di = _DeviceImages()
    ctx = volib.Context(0, w, h, 4096, 1)
    try:
        res = []
        for mode in ("contiguous", "host view", "device view"):
            ctx.batch_configure(4, w, h, 1)
            for i, a in enumerate(quad):
                if mode == "contiguous":
                    ctx.batch_upload_image(i, a)
                elif mode == "host view":
                    ctx.batch_upload_image(i, view(a))
                else:
                    v = view(a)
                    base = v.base if v.base is not None else v
                    dev = di.upload(base)  # the whole padded / big buffer on the device
                    off = v.ctypes.data - base.ctypes.data
                    ctx.batch_upload_image_dev(i, dev + off, v.strides[0])
            ctx.batch_set_quads([[0, 1, 2, 3]])
            ctx.batch_set_points(0, pts)
            ctx.batch_set_projection(P_l, P_r)
            ctx.batch_run(volib.STAGE_ALL)
            ctx.batch_sync()
            lv = [ctx.batch_get_pyramid_level(i, l) for i in range(4) for l in range(3)]
            res.append((lv, ctx.batch_get_filtered(0), ctx.batch_get_pose(0)))
        for lv, filt, pose in res[1:]:
            assert all(np.array_equal(a, b) for a, b in zip(lv, res[0][0]))
            same(filt, res[0][1], ("l0", "r0", "l1", "r1", "xyz", "keep_idx", "keep_idx_circ"))
            same(pose, res[0][2], ("rvec", "tvec", "inliers"))
        assert len(res[0][2]["inliers"]) > 20
    finally:
        ctx.close()
        di.free()


@pytest.mark.parametrize("name,view", VIEWS)
def test_sequence_loop_pairs_with_stride_not_width(volib, small_world, name, view):
    """vo_seq_push_pair from pageable memory (staged), from page-locked memory (read over PCIe by the ingest kernel) and
    vo_seq_push_pair_dev from device memory, each with a stride != width: four sequences fed the same 6 frames, sequence 0
    from contiguous pageable arrays -- identical feature state after every step and identical trajectories"""
    from test_gpu_sequences import _DeviceImages
    L, R, _, _ = small_world.render_sequence(6)
    h, w = L[0].shape
    P_l, P_r = small_world.proj_matrices()
    di = _DeviceImages()
    ctx = volib.Context(0, w, h, 4096, 4)
    try:
        ctx.batch_set_projection(P_l, P_r)
        ctx.batch_set_detect_params(features_per_bucket=3)
        ctx.seq_configure(4, w, h, 3, 16)
        keep = []
        for k in range(6):
            ctx.seq_push_pair(0, L[k], R[k])
            ctx.seq_push_pair(1, view(L[k]), view(R[k]))                      # pageable, strided
            pl, pr = view(L[k]), view(R[k])
            bl, br = pl.base, pr.base
            hl, hr = di.pinned(bl), di.pinned(br)                             # page-locked copies of the WHOLE buffers
            offl, offr = pl.ctypes.data - bl.ctypes.data, pr.ctypes.data - br.ctypes.data
            vl = hl.reshape(-1)[offl:].view()
            # views into the page-locked buffers with the same geometry
            vl = np.lib.stride_tricks.as_strided(hl.reshape(-1)[offl:], shape=pl.shape, strides=pl.strides)
            vr = np.lib.stride_tricks.as_strided(hr.reshape(-1)[offr:], shape=pr.shape, strides=pr.strides)
            assert np.array_equal(vl, L[k]) and np.array_equal(vr, R[k])
            ctx.seq_push_pair(2, vl, vr, pinned=True)
            dl, dr = di.upload(bl), di.upload(br)
            ctx.seq_push_pair_dev(3, dl + offl, dr + offr, pl.strides[0])
            keep.append((hl, hr, vl, vr))
            ctx.seq_step()
            ctx.seq_sync()
            st = [ctx.seq_get_state(s) for s in range(4)]
            for s in (1, 2, 3):
                assert all(np.array_equal(a, b) for a, b in zip(st[s], st[0])), (k, s)
        t0 = ctx.seq_get_trajectory(0)
        assert len(t0[0]) == 5
        for s in (1, 2, 3):
            ts = ctx.seq_get_trajectory(s)
            assert np.array_equal(ts[0], t0[0]) and np.array_equal(ts[1], t0[1]), s
    finally:
        ctx.close()
        di.free()


def test_argument_sweep_over_every_export():
    """tests/abi_sweep.py in a child process (a fault must fail THIS test, not end the session): ~300 bad calls over all
    exports of include/vo_hip.h, each answered with the documented error code"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "abi_sweep.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode in (0, 1), "abi_sweep died (rc %d): %s" % (r.returncode, r.stderr[-2000:])
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    print("argument sweep: %d calls over %d exports" % (rep["checked"], rep["exports_covered"]))
    assert rep["failures"] == [], "\n".join(rep["failures"])
    assert rep["not_covered"] == []
    assert rep["checked"] >= 250


def test_two_ranks_of_bench_share_gpu_zero():
    """the N > 1 path of bench.py with REAL GPU work in both ranks (tests/test_replicas_gloo.py only
    fabricates timings): the driver's launch line with two ranks, both on GPU 0 (VO_ALLOW_SHARED_GPU=1, gloo for the barrier
    and the reductions since RCCL refuses duplicate devices).  The line must say ranks = 2 on ONE GPU, sum the frames of both
    ranks over the max-over-ranks time, have both ranks' frames validated against the oracle, and carry BASELINE config 5
    (one sequence per GPU, exact replay) with a per-GPU figure for each rank."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, VO_ALLOW_SHARED_GPU="1", VO_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "16", "--steps", "3",
           "--warmup", "1", "--no-cpu-baseline", "--sustain", "0", "--no-replay-leg", "--validate", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE JSON line, from rank 0
    b = json.loads(lines[0])
    assert b["ranks"] == 2 and b["n_gpus"] == 1 and b["scaling"] == "weak"
    assert b["validated_frames"] == 2                 # the smaller of the two ranks' counts: both validated
    frames = b["value"] * b["ms_per_step"] * 1e-3 * b["steps"]
    assert abs(frames - 2 * 16 * 3) < 1e-6 * frames   # frames of both ranks / max-over-ranks time
    c5 = [c for c in b["configs"] if c["name"] == "config5_one_sequence_per_gpu"]
    assert len(c5) == 1 and c5[0]["baseline_config"] == 5 and c5[0]["ranks"] == 2 and c5[0]["sequences_per_gpu"] == 1
    assert len(c5[0]["per_gpu_value"]) == 2 and all(v > 0 for v in c5[0]["per_gpu_value"])
    assert c5[0]["value"] <= sum(c5[0]["per_gpu_value"]) * 1.0001 and c5[0]["validated_frames"] >= 1
    print("two ranks on GPU 0: batch %.0f frames/s; config 5: %.0f frames/s aggregate, per rank %s"
          % (b["value"], c5[0]["value"], ["%.0f" % v for v in c5[0]["per_gpu_value"]]))


def test_schedule_export_import_skips_the_probe(volib, small_world, small_seq):
    """vo_export_schedule / vo_import_schedule: a settled schedule leaves the process as plain records
    and comes back in; a context whose key is in the table does not probe -- its first run takes one run's time -- and
    reports the imported schedule with probed = 1.  Here: probe an 8-frame batch, export, re-key the record to a 6-frame
    batch with the OTHER pose_streams value (a schedule no probe of this process produced), import, run 6 frames."""
    import time
    from visual_odom_amd import synth
    lefts, rights = small_seq["L"], small_seq["R"]
    h, w = lefts[0].shape
    P_l, P_r = small_world.proj_matrices()
    pts = synth.select_keypoints(lefts[0], bucket=h // 10, per_bucket=4)

    def batch(B):
        ctx = volib.Context(0, w, h, 4096, B)
        ctx.batch_configure(4, w, h, B)
        for i, a in enumerate((lefts[0], rights[0], lefts[1], rights[1])):
            ctx.batch_upload_image(i, a)
        ctx.batch_set_quads([[0, 1, 2, 3]] * B)
        for b in range(B):
            ctx.batch_set_points(b, pts)
        ctx.batch_set_projection(P_l, P_r)
        return ctx

    ctx = batch(8)
    try:
        ctx.batch_run(volib.STAGE_ALL)   # probes
        ctx.batch_sync()
        s8 = ctx.get_schedule()
        assert s8["probed"]
        ref_pose = ctx.batch_get_pose(0)
    finally:
        ctx.close()
    table = volib.export_schedules()
    mine = [r for r in table if r["key"][2] == w and r["key"][3] == h and r["key"][5] == 8 and r["key"][1] == 0]
    assert len(mine) == 1 and (mine[0]["pose_waves"], mine[0]["pose_streams"], mine[0]["epnp_wide_frames"]) == \
        (s8["pose_waves"], s8["pose_streams"], s8["epnp_wide_frames"])
    assert s8["epnp_wide_frames"] in (4, 16)          # 8 frames per run: the four-kernel EPnP is one of the probed knobs
    rec = dict(mine[0], key=list(mine[0]["key"]))
    rec["key"][5] = 6
    rec["pose_streams"] = 3 - s8["pose_streams"]
    rec["pose_waves"] = 3 - s8["pose_waves"]
    rec["epnp_wide_frames"] = 20 - s8["epnp_wide_frames"]
    volib.import_schedules([rec])
    assert rec in volib.export_schedules()
    ctx = batch(6)
    try:
        t0 = time.perf_counter()
        ctx.batch_run(volib.STAGE_ALL)
        ctx.batch_sync()
        first = time.perf_counter() - t0
        s6 = ctx.get_schedule()
        assert (s6["pose_waves"], s6["pose_streams"], s6["epnp_wide_frames"], s6["probed"]) == \
            (rec["pose_waves"], rec["pose_streams"], rec["epnp_wide_frames"], True)
        assert ctx.get_probe_log() == {} or len(ctx.get_probe_log()) == 0   # no probe ran in this context
        t0 = time.perf_counter()
        ctx.batch_run(volib.STAGE_ALL)
        ctx.batch_sync()
        second = time.perf_counter() - t0
        assert first < 5 * second + 2e-3, (first, second)                   # one run's time, not a probe's 25-100
        p = ctx.batch_get_pose(0)
        assert np.array_equal(p["rvec"], ref_pose["rvec"]) and np.array_equal(p["inliers"], ref_pose["inliers"])
    finally:
        ctx.close()
    with pytest.raises(volib.VoError):
        volib.import_schedules([dict(rec, pose_waves=3)])


@pytest.mark.parametrize("h,w", [(32, 32), (33, 47), (150, 70), (64, 257), (121, 1023), (480, 642)])
def test_fused_pyramid_pass_on_small_and_odd_shapes(volib, orc, h, w):
    """the fused pyramid pass (one launch per level: Scharr image + next level + border, 4 columns x 8 rows per lane) on the
    shapes the KITTI / camera tests do not reach: the smallest image vo_create takes (one level only), widths of every residue
    mod 4, levels narrower than a wavefront's 256-column span, fewer rows than a row block -- every level bit-exact against
    the oracle's buildOpticalFlowPyramid, and the four LK hops on top of it (which read the Scharr images and the borders)
    bit-exact for points next to all four image edges"""
    rng = np.random.default_rng(h * 1000 + w)
    base = rng.integers(0, 256, (h // 4 + 2, w // 4 + 2)).astype(np.float32)
    img = np.kron(base, np.ones((4, 4), np.float32))[:h, :w]
    img = np.clip(img + rng.normal(0, 6, (h, w)), 0, 255).astype(np.uint8)
    shifted = np.roll(img, (1, -2), (0, 1))
    ref = orc.build_pyramid(img, 3)
    n_levels, cw, ch = 1, w, h                        # buildOpticalFlowPyramid stops before a level of 21 pixels or fewer
    while n_levels < 4 and (cw + 1) // 2 > 21 and (ch + 1) // 2 > 21:
        cw, ch, n_levels = (cw + 1) // 2, (ch + 1) // 2, n_levels + 1
    ctx = volib.Context(0, max(w, 32), max(h, 32), 512, 1)
    ctx.set_params(lk_full_chain=1)                   # raw per-hop status like four independent calcOpticalFlowPyrLK calls
    try:
        xs = rng.uniform(0, w - 1, 60).astype(np.float32)
        ys = rng.uniform(0, h - 1, 60).astype(np.float32)
        xs[:8] = [0, 0.4, w - 1, w - 1.3, 2.5, w / 2, w / 2, 7.25]
        ys[:8] = [0, h - 1, 0.6, h - 1, h / 2, 0, h - 1, 3.75]
        pts = np.stack([xs, ys], 1)
        got = ctx.circular_match(img, shifted, shifted, img, pts)
        for l in range(n_levels):
            lv = ctx.batch_get_pyramid_level(0, l)
            assert lv.shape == ref[l].shape and np.array_equal(lv, ref[l]), (l, lv.shape)
        with pytest.raises(volib.VoError):
            ctx.batch_get_pyramid_level(0, n_levels)
        want = orc.circular_matching(img, shifted, shifted, img, pts, max_level=n_levels - 1)
        assert np.array_equal(got["status4"], want["status4"]) and np.array_equal(got["keep_idx"], want["keep_idx"])
        for k in ("l0", "r0", "r1", "l1", "l0_ret"):
            assert np.array_equal(got[k], want[k]), k
    finally:
        ctx.close()


def test_pyramid_stage_over_more_images_than_one_launch(volib, orc):
    """the pyramid pass takes at most 4096 images per launch (pyramid.hip, pass_images_per_launch) and decodes its workgroup id
    per launch: 4104 small images -- the last launch holds 8 -- with distinct pictures either side of the boundary, every
    level bit-exact against the oracle"""
    w, h, n = 64, 44, 4104
    rng = np.random.default_rng(4104)
    ctx = volib.Context(0, w, h, 8, n // 6)
    try:
        ctx.batch_configure(n, w, h, 1)
        picks = [0, 1, 7, 8, 4087, 4095, 4096, 4097, 4103]
        imgs = {i: rng.integers(0, 256, (h, w), dtype=np.uint8) for i in picks}
        blank = np.zeros((h, w), np.uint8)
        for i in range(n):
            ctx.batch_upload_image(i, imgs.get(i, blank))
        ctx.batch_set_pyramid_range(0, n)
        ctx.batch_run(volib.STAGE_PYRAMID)
        ctx.batch_sync()
        for i in picks:
            ref = orc.build_pyramid(imgs[i], 3)
            for l in range(2):  # 64 x 44 -> 32 x 22, then buildOpticalFlowPyramid stops
                assert np.array_equal(ctx.batch_get_pyramid_level(i, l), ref[l]), (i, l)
        assert not ctx.batch_get_pyramid_level(4094, 1).any() and not ctx.batch_get_pyramid_level(4098, 1).any()
    finally:
        ctx.close()
