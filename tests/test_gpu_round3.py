"""Round-3 items on the MI355X (-m gpu):
  * multi-GPU host in the language north_star names: examples/vo_multi_gpu.cpp (one host thread + one vo_ctx per worker,
    sequence s -> worker s % n, no collective).  On the one GPU of the test box two workers share device 0: two host
    threads x two contexts give the trajectories one context gives, byte for byte -- which also pins the header's
    "one vo_ctx per host thread per GPU" contract (include/vo_hip.h);
  * the C++ hosts read PNG files (zlib-only decoder, examples/vo_io.h) through a decoder pool: same trajectories as from
    PGM, end-to-end frames/s reported;
  * the sequence loop's life-cycle fixes: vo_seq_reset(-1) rewinds the step counter, a resumed sequence
    flags the dropped transition (VO_SEQ_F_GAP), capacity exhaustion refuses the step without wedging the loop;
  * a second context of a process gets the first one's streams back (per-device stream pool) and runs at its speed.
"""
import json
import os
import subprocess
import sys
import time
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SMALL = dict(width=480, height=160, fx=300.0, cx=239.5, cy=79.5, bf=-160.0, tex_size=1024)


def write_pgm(path, img):
    h, w = img.shape
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (w, h) + np.ascontiguousarray(img).tobytes())


def write_png_gray(path, img, filters=(0, 1, 2, 3, 4)):
    """a valid 8-bit gray PNG written with zlib alone; scanline y uses filter type filters[y % len(filters)] so that
    the reader's five un-filter paths are all exercised (PIL is only used elsewhere, to cross-check this writer)"""
    import struct
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    a = img.astype(np.int32)
    left = np.zeros_like(a)
    left[:, 1:] = a[:, :-1]
    up = np.zeros_like(a)
    up[1:] = a[:-1]
    ul = np.zeros_like(a)
    ul[1:, 1:] = a[:-1, :-1]
    p = left + up - ul
    pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - ul)
    paeth = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, ul))
    pred = {0: np.zeros_like(a), 1: left, 2: up, 3: (left + up) >> 1, 4: paeth}
    raw = bytearray()
    for y in range(h):
        ft = filters[y % len(filters)]
        raw.append(ft)
        raw += ((a[y] - pred[ft][y]) & 255).astype(np.uint8).tobytes()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(bytes(raw), 6)) + chunk(b"IEND", b""))


def make_dirs(tmp_path, seqs, fmt="pgm"):
    dirs = []
    for s, (L, R) in enumerate(seqs):
        d = tmp_path / ("%s_seq%d" % (fmt, s))
        for cam, imgs in ((0, L), (1, R)):
            (d / ("image_%d" % cam)).mkdir(parents=True)
            for k, img in enumerate(imgs):
                path = str(d / ("image_%d" % cam) / ("%06d.%s" % (k, fmt)))
                if fmt == "pgm":
                    write_pgm(path, img)
                else:
                    write_png_gray(path, img)
        dirs.append(str(d))
    return dirs


def test_png_writer_of_this_test_is_a_png(tmp_path):
    """the hand-written PNG fixtures decode to the same pixels with PIL (so the C++ reader is tested on real PNGs)"""
    from PIL import Image
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    write_png_gray(str(tmp_path / "a.png"), img)
    with Image.open(str(tmp_path / "a.png")) as im:
        assert np.array_equal(np.asarray(im), img)


def test_two_host_threads_two_contexts_equal_one_context(volib, vo_multi_gpu_binary, vo_seq_run_binary, tmp_path):
    from visual_odom_amd import synth, odometry
    worlds = [synth.StereoWorld(seed=300 + 7 * s, **SMALL) for s in range(5)]
    lengths = [9, 7, 9, 8, 6]
    seqs = []
    for wd, n in zip(worlds, lengths):
        L, R, _, _ = wd.render_sequence(n)
        seqs.append((L, R))
    P_l, P_r = worlds[0].proj_matrices()
    cal = [repr(float(v)) for v in (P_l[0, 0], P_l[0, 2], P_l[1, 2], P_r[0, 3])]
    pgm = make_dirs(tmp_path, seqs, "pgm")
    png = make_dirs(tmp_path, seqs, "png")

    def run(exe, pre, prefix, dirs):
        r = subprocess.run([exe] + pre + cal + ["12", "2", str(tmp_path / prefix)] + dirs, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return r

    run(vo_seq_run_binary, ["--device", "0"], "one", pgm)                      # one context, five sequences
    r2 = run(vo_multi_gpu_binary, ["--devices", "0,0"], "two", pgm)           # two threads x two contexts on GPU 0
    r3 = run(vo_multi_gpu_binary, ["--devices", "0,0,0", "--decode-threads", "3"], "three", png)  # three workers, PNG input
    rep = json.loads(r2.stdout.strip().splitlines()[-1])
    assert len(rep["workers"]) == 2 and rep["frames"] == sum(n - 1 for n in lengths)
    assert [w["sequences"] for w in rep["workers"]] == [3, 2] and rep["fps"] > 0
    assert json.loads(r3.stdout.strip().splitlines()[-1])["frames"] == rep["frames"]
    for s in range(5):
        one = open(str(tmp_path / ("one_%d.txt" % s))).read()
        assert one == open(str(tmp_path / ("two_%d.txt" % s))).read(), s        # byte for byte
        assert one == open(str(tmp_path / ("three_%d.txt" % s))).read(), s
        assert len(one.strip().splitlines()) == lengths[s]
    # ... and they are what the python mirror of the same loop computes
    ctx = volib.Context(0, 480, 160, 4096, 5)
    try:
        vo = odometry.MultiSequenceOdometry(P_l, P_r, 5, 480, 160, ctx=ctx, ring=3, max_steps=16, features_per_bucket=2)
        for k in range(max(lengths)):
            for s in range(5):
                if k < lengths[s]:
                    vo.push(s, seqs[s][0][k], seqs[s][1][k])
            vo.step()
        for s in range(5):
            got = odometry.load_poses(str(tmp_path / ("two_%d.txt" % s)))
            assert np.abs(got - np.asarray(vo.trajectory(s))).max() < 1e-8
    finally:
        ctx.close()
    # a worker on a device that does not exist fails loudly
    bad = subprocess.run([vo_multi_gpu_binary, "--devices", "0,63"] + cal + ["4", "2", str(tmp_path / "bad")] + pgm,
                         capture_output=True, text=True)
    assert bad.returncode != 0 and "vo_create failed" in bad.stderr


def test_two_live_contexts_in_two_python_threads(volib, orc, small_world):
    """two contexts alive at once, driven from two host threads (ctypes releases the GIL in every call): each gives
    what the checker gives"""
    import threading
    L, R, _, _ = small_world.render_sequence(3)
    P_l, P_r = small_world.proj_matrices()
    from visual_odom_amd import synth
    pts = [synth.select_keypoints(L[k], bucket=16, per_bucket=2) for k in (0, 1)]
    refs = []
    for k in (0, 1):
        ref = orc.circular_matching(L[k], R[k], L[k + 1], R[k + 1], pts[k])
        refs.append(orc.check_valid_and_remove(ref["l0"], ref["r0"], ref["l1"], ref["r1"], ref["l0_ret"])[0])
    errors = []

    def worker(k):
        try:
            ctx = volib.Context(0, 480, 160, 4096, 1)
            for _ in range(25):
                got = ctx.track_frame(L[k], R[k], L[k + 1], R[k + 1], pts[k], P_l, P_r)
                for name, arr in zip(("l0", "r0", "l1", "r1"), refs[k]):
                    assert np.array_equal(got[name].view(np.uint32), arr.view(np.uint32)), (k, name)
                assert got["rc"] == 0 and len(got["inliers"]) > 10
            ctx.close()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(k,)) for k in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_sequence_loop_reset_rewinds_gap_is_flagged_capacity_refuses(volib, small_world):
    L, R, _, _ = small_world.render_sequence(7)
    P_l, P_r = small_world.proj_matrices()
    h, w = L[0].shape
    ctx = volib.Context(0, w, h, 2048, 2)
    try:
        ctx.seq_configure(2, w, h, 3, 4)           # 4 trajectory rows per sequence
        ctx.batch_set_projection(P_l, P_r)
        first = None
        for rnd in range(4):                        # 4 x 5 steps on a loop configured for 4 rows: only reset(-1) makes room
            for k in range(5):
                for s in (0, 1):
                    ctx.seq_push_pair(s, L[k], R[k])
                ctx.seq_step()
            rows = [ctx.seq_get_trajectory(s)[0] for s in (0, 1)]
            assert all(len(r) == 4 for r in rows)
            if first is None:
                first = rows
            else:                                   # and every round replays the first one exactly
                assert all(np.array_equal(a, b) for a, b in zip(first, rows))
            ctx.seq_reset(-1)
        # capacity exhaustion: the step is refused, its pairs are dropped, the loop stays usable
        for k in range(5):
            for s in (0, 1):
                ctx.seq_push_pair(s, L[k], R[k])
            ctx.seq_step()
        ctx.seq_push_pair(0, L[5], R[5])
        with pytest.raises(volib.VoError) as e:
            ctx.seq_step()
        assert e.value.code == volib.VO_ERR_STATE and "capacity" in str(e.value)
        assert len(ctx.seq_get_trajectory(0)[0]) == 4
        ctx.seq_reset(0)                            # sequence 0 gets its rows back and starts over; sequence 1 is full
        ctx.seq_push_pair(0, L[0], R[0])
        ctx.seq_step()
        ctx.seq_push_pair(0, L[1], R[1])
        ctx.seq_step()
        rows0, info0 = ctx.seq_get_trajectory(0)
        assert len(rows0) == 1 and np.array_equal(rows0[0], first[0][0])
        # a pause: sequence 0 gets no pair for one step, then resumes -> its next processed frame carries VO_SEQ_F_GAP
        ctx.seq_configure(2, w, h, 3, 16)
        for k, feed in enumerate([True, True, True, False, True, True]):
            if feed:
                ctx.seq_push_pair(0, L[k], R[k])
            ctx.seq_push_pair(1, L[k], R[k])
            ctx.seq_step()
        rows0, info0 = ctx.seq_get_trajectory(0)
        rows1, info1 = ctx.seq_get_trajectory(1)
        flags0 = [int(i[5]) for i in info0]
        assert len(rows0) == 3 and [bool(f & volib.SEQ_F_GAP) for f in flags0] == [False, False, True]
        assert not any(int(i[5]) & volib.SEQ_F_GAP for i in info1) and len(rows1) == 5
        # a REFUSED step is a pause for every sequence whose pair it dropped: sequence 1 is not exhausted, loses
        # its pair of step 4 with the refusal, and when the caller simply goes on its pair 5 restarts the image pair -- it is
        # not matched against pair 3 -- and the frame (5, 6) carries VO_SEQ_F_GAP
        ctx.seq_configure(2, w, h, 3, 3)
        for k in range(4):
            ctx.seq_push_pair(0, L[k], R[k])
            if k >= 1:
                ctx.seq_push_pair(1, L[k], R[k])
            ctx.seq_step()
        for s in (0, 1):
            ctx.seq_push_pair(s, L[4], R[4])
        with pytest.raises(volib.VoError) as e:
            ctx.seq_step()                          # sequence 0 has used its 3 rows
        assert e.value.code == volib.VO_ERR_STATE
        ctx.seq_reset(0)
        for k in (5, 0):                            # (a 6-frame fixture: "pair 6" = pair 0 again)
            for s in (0, 1):
                ctx.seq_push_pair(s, L[k], R[k])
            ctx.seq_step()
        rows0, info0 = ctx.seq_get_trajectory(0)
        rows1, info1 = ctx.seq_get_trajectory(1)
        assert len(rows0) == 1 and not int(info0[0][5]) & volib.SEQ_F_GAP
        assert len(rows1) == 3 and [bool(int(i[5]) & volib.SEQ_F_GAP) for i in info1] == [False, False, True]
        # the other path vo_hip.h documents: the SAME pairs pushed again after the reset.  The library cannot tell a
        # re-push from a later pair, so sequence 1 is paused all the same: pair 4 restarts its image pair (no row), the frame
        # (4, 5) is processed and carries VO_SEQ_F_GAP; sequence 0 starts over with pair 4
        ctx.seq_configure(2, w, h, 3, 3)
        for k in range(4):
            ctx.seq_push_pair(0, L[k], R[k])
            if k >= 1:
                ctx.seq_push_pair(1, L[k], R[k])
            ctx.seq_step()
        for s in (0, 1):
            ctx.seq_push_pair(s, L[4], R[4])
        with pytest.raises(volib.VoError):
            ctx.seq_step()
        ctx.seq_reset(0)
        for k in (4, 5):
            for s in (0, 1):
                ctx.seq_push_pair(s, L[k], R[k])
            ctx.seq_step()
        rows0, info0 = ctx.seq_get_trajectory(0)
        rows1, info1 = ctx.seq_get_trajectory(1)
        assert len(rows0) == 1 and not int(info0[0][5]) & volib.SEQ_F_GAP
        assert len(rows1) == 3 and [bool(int(i[5]) & volib.SEQ_F_GAP) for i in info1] == [False, False, True]
    finally:
        ctx.close()


def test_second_context_runs_at_the_speed_of_the_first(volib, small_world):
    """per-device stream pool: a context created after another one was destroyed gets the same HIP streams (same
    hardware-queue mapping).  One-sequence lock-step loop, the most mapping-sensitive mode (round 2: 0.63 vs 0.80 ms)."""
    L, R, _, _ = small_world.render_sequence(6)
    P_l, P_r = small_world.proj_matrices()
    h, w = L[0].shape

    def step_ms():
        ctx = volib.Context(0, w, h, 2048, 1)
        try:
            ctx.seq_configure(1, w, h, 3, 512)
            ctx.batch_set_projection(P_l, P_r)
            best = 1e9
            for rep in range(3):
                ctx.seq_reset(-1)
                for k in range(40):
                    ctx.seq_push_pair(0, L[k % 6], R[k % 6])
                    ctx.seq_step()
                ctx.seq_sync()
                t0 = time.perf_counter()
                for k in range(40, 240):
                    ctx.seq_push_pair(0, L[k % 6], R[k % 6])
                    ctx.seq_step()
                ctx.seq_sync()
                best = min(best, (time.perf_counter() - t0) / 200 * 1e3)
            return best
        finally:
            ctx.close()

    a = step_ms()
    b = step_ms()
    c = step_ms()
    print("one-sequence step: first context %.3f ms, second %.3f ms, third %.3f ms" % (a, b, c))
    assert min(b, c) <= 1.15 * a


def test_pnp_ransac_with_exactly_four_points_is_opencv_p3p_switch(volib, orc, host_check):
    """visualOdometry.cpp:176 with K = 4 survivors: OpenCV's `npoints == 4 -> SOLVEPNP_P3P`, solvePnP's answer as is.
    p3p_frame (thread 0 of the refinement workgroup of a four-point frame) (a) against the HOST build of the same header (tests/host_check): bit for bit -- the cubic's cube root / acos /
    cos are vo_math.h's, IEEE operations only, so nothing platform-specific is left in the path; (b) against oracle/orc_p3p.c
    (glibc's pow / acos / cos, like OpenCV): same solution count, inliers 0..3, no refinement, WORST case <= 1e-6 like every
    other pose test (round 3 only bounded the median and the 90th percentile); a quadruple without a P3P
    solution leaves rvec / tvec untouched and reports VO_NO_MODEL"""
    import ctypes as C
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_p3p import planted, KM
    host_check.hc_p3p4.restype = C.c_int
    ctx = volib.Context(0, 1241, 376, 1024, 4)
    try:
        rng = np.random.default_rng(5)
        worst, solved, exact = 0.0, 0, 0
        for k in range(400):
            X, uv, rv, t = planted(rng, noise=0.3 if k % 2 else 0.0)
            Xf, uvf = np.ascontiguousarray(X, np.float32), np.ascontiguousarray(uv, np.float32)
            rc, r_o, t_o, inl, dbg = orc.solve_pnp_ransac(Xf, uvf, KM, rvec=[0.5, 0.5, 0.5], tvec=[3, 3, 3])
            found, r_g, t_g, R_g, inl_g = ctx.pnp_ransac(Xf, uvf, KM, rvec=[0.5, 0.5, 0.5], tvec=[3, 3, 3])
            r_h, t_h = np.full(3, 0.5), np.full(3, 3.0)
            n_h = host_check.hc_p3p4(Xf.ctypes.data_as(C.c_void_p), uvf.ctypes.data_as(C.c_void_p), KM.ctypes.data_as(C.c_void_p),
                                     r_h.ctypes.data_as(C.c_void_p), t_h.ctypes.data_as(C.c_void_p))
            assert found == (rc == 1) == (n_h > 0), k
            assert np.array_equal(inl_g, inl)
            if rc == 1:
                assert list(inl) == [0, 1, 2, 3]
                assert np.array_equal(r_g, r_h) and np.array_equal(t_g, t_h), k  # device == host build of vo_p3p.h, bit for bit
                d = max(np.abs(r_g - r_o).max(), np.abs(t_g - t_o).max())
                worst = max(worst, d)
                solved += 1
                exact += d == 0
                assert np.allclose(R_g, orc.rodrigues(r_g), atol=1e-12)
            else:  # untouched
                assert np.array_equal(r_g, [0.5, 0.5, 0.5]) and np.array_equal(t_g, [3, 3, 3])
                assert np.allclose(R_g, orc.rodrigues(np.array([0.5, 0.5, 0.5])), atol=1e-12)
        print("P3P on the device: bit-identical to the host build on %d quadruples; vs the checker %d identical, worst %.3g"
              % (solved, exact, worst))
        assert solved > 350 and worst <= 1e-6
    finally:
        ctx.close()
