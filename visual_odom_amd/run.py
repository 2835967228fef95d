"""Command-line front end: the reference's `./run <sequence_dir> <calibration.yaml> [ground_truth_poses]`
(README.md:37, src/main.cpp:28-227) on the MI355X, for one or several sequences at once.

    python -m visual_odom_amd.run <sequence_dir>[,<sequence_dir>...] <calibration.yaml> [<gt_poses.txt>[,...]]
                                  [--out PREFIX] [--max-frames N] [--features-per-bucket K] [--mono-rotation]

What the reference does per frame (matchingFeatures -> triangulation -> trackingFrame2Frame -> gates ->
integrateOdometryStereo) runs inside libvo_hip through the lock-step sequence API (visual_odom_amd.odometry.
MultiSequenceOdometry): several sequences advance together, one frame each per step, state on the device.  What the
reference only draws (utils.cpp:19-48) is written instead: PREFIX_<k>.txt in the KITTI pose format (12 doubles per
line, what loadPoses reads, evaluate_odometry.cpp:24-27), and, when ground truth is given, the ATE and the KITTI
segment errors of the run (evaluate_odometry.cpp:35-116).  GUI calls are dropped.

Images: <sequence_dir>/image_0/%06d.png and image_1/%06d.png exactly like loadImageLeft / loadImageRight
(utils.cpp:172-190: imread(IMREAD_COLOR) + cvtColor(BGR2GRAY)); .pgm is accepted too.  Calibration: the OpenCV-YAML
keys Camera.fx / fy / cx / cy / bf (main.cpp:64-74, calibration/kitti00.yaml).
"""
import argparse
import os
import re
import sys

import numpy as np


def read_calibration(path):
    """Camera.fx, fy, cx, cy, bf of an OpenCV FileStorage YAML (main.cpp:64-71) -> dict of floats"""
    out = {}
    with open(path) as f:
        for line in f:
            m = re.match(r"\s*Camera\.(fx|fy|cx|cy|bf)\s*:\s*([-+0-9.eE]+)", line)
            if m:
                out[m.group(1)] = float(np.float32(m.group(2)))  # `float fx = fSettings[...]`
    missing = [k for k in ("fx", "fy", "cx", "cy", "bf") if k not in out]
    if missing:
        raise ValueError("%s: missing Camera.%s" % (path, ", Camera.".join(missing)))
    return out


def projection_matrices(cal):
    """projMatrl / projMatrr as main.cpp:73-74 builds them (3x4 float32)"""
    fx, fy, cx, cy, bf = (cal[k] for k in ("fx", "fy", "cx", "cy", "bf"))
    P_l = np.array([[fx, 0, cx, 0], [0, fy, cy, 0], [0, 0, 1, 0]], np.float32)
    P_r = np.array([[fx, 0, cx, bf], [0, fy, cy, 0], [0, 0, 1, 0]], np.float32)
    return P_l, P_r


def bgr_to_gray(bgr):
    """cv::cvtColor(BGR2GRAY) for 8-bit images: fixed-point Y = (B*1868 + G*9617 + R*4899 + 2^13) >> 14 (identity for
    R = G = B, which is what imread(IMREAD_COLOR) makes of KITTI's gray PNGs, quirk B10 of SURVEY.md)"""
    b = bgr[..., 0].astype(np.int32)
    g = bgr[..., 1].astype(np.int32)
    r = bgr[..., 2].astype(np.int32)
    return ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)


def read_gray(path):
    """one 8-bit gray image (H, W) uint8 from a binary PGM or anything PIL decodes; None if the file does not exist"""
    if not os.path.exists(path):
        return None
    if path.lower().endswith(".pgm"):
        with open(path, "rb") as f:
            data = f.read()
        m = re.match(rb"P5\s+(?:#[^\n]*\n\s*)*(\d+)\s+(\d+)\s+(\d+)\s", data)
        if not m or int(m.group(3)) != 255:
            raise ValueError("%s: not an 8-bit binary PGM" % path)
        w, h = int(m.group(1)), int(m.group(2))
        return np.frombuffer(data, np.uint8, w * h, m.end()).reshape(h, w).copy()
    from PIL import Image
    im = Image.open(path)
    if im.mode == "L":  # 8-bit gray: imread(IMREAD_COLOR) replicates it, BGR2GRAY gives it back unchanged
        return np.asarray(im, np.uint8).copy()
    rgb = np.asarray(im.convert("RGB"), np.uint8)
    return bgr_to_gray(rgb[..., ::-1])


def frame_paths(seq_dir, frame_id):
    """image_0/%06d.png, image_1/%06d.png (utils.cpp:174,184); .pgm as a fall-back"""
    out = []
    for cam in (0, 1):
        base = os.path.join(seq_dir, "image_%d" % cam, "%06d" % frame_id)
        out.append(base + ".png" if os.path.exists(base + ".png") or not os.path.exists(base + ".pgm") else base + ".pgm")
    return out


def read_pair(seq_dir, frame_id):
    left_path, right_path = frame_paths(seq_dir, frame_id)
    left, right = read_gray(left_path), read_gray(right_path)
    return (left, right) if left is not None and right is not None else None


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("sequences", help="path_to_sequence (several, comma-separated, run in lock step)")
    ap.add_argument("calibration", help="path_to_calibration (OpenCV YAML with Camera.fx/fy/cx/cy/bf)")
    ap.add_argument("ground_truth", nargs="?", default=None, help="[optional] path_to_ground_truth_pose (comma-separated, one per sequence)")
    ap.add_argument("--out", default="vo_poses", help="trajectories are written to OUT_<k>.txt")
    ap.add_argument("--max-frames", type=int, default=9000, help="the reference loops to frame 8999 (main.cpp:123)")
    ap.add_argument("--features-per-bucket", type=int, default=1, help="visualOdometry.cpp:107")
    ap.add_argument("--mono-rotation", action="store_true", help="trackingFrame2Frame(..., mono_rotation = true); main.cpp:181 passes false")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--decode-threads", type=int, default=8,
                    help="image files of frame k + 1 are read and decoded by this many threads while step k is pushed and runs "
                         "(PIL's decoders release the GIL); 0 = decode on the pushing thread like round 2")
    args = ap.parse_args(argv)

    from . import odometry
    dirs = [d for d in args.sequences.split(",") if d]
    gts = args.ground_truth.split(",") if args.ground_truth else []
    cal = read_calibration(args.calibration)
    P_l, P_r = projection_matrices(cal)
    first = [read_pair(d, 0) for d in dirs]
    if any(p is None for p in first):
        raise SystemExit("cannot read frame 0 of %s" % dirs[[p is None for p in first].index(True)])
    h, w = first[0][0].shape
    if any(p[0].shape != (h, w) or p[1].shape != (h, w) for p in first):
        raise SystemExit("all sequences run in one lock-step loop must have the same image size")
    S = len(dirs)
    vo = odometry.MultiSequenceOdometry(P_l, P_r, S, w, h, device=args.device, ring=3, max_steps=args.max_frames + 1,
                                        mono_rotation=args.mono_rotation, features_per_bucket=args.features_per_bucket)
    live = [True] * S
    n_read = [0] * S
    import time
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(args.decode_threads) if args.decode_threads > 0 else None

    def prefetch(frame_id):
        if pool is None or frame_id >= args.max_frames:
            return {}
        return {s: pool.submit(read_pair, d, frame_id) for s, d in enumerate(dirs) if live[s]}

    t_start = time.perf_counter()
    t_wait = 0.0
    # look-ahead: as many frames as keep every decoder thread busy (one sequence: a frame is only two files)
    depth = max(1, min(16, -(-max(args.decode_threads, 1) // (2 * S))))
    from collections import deque
    queue = deque(prefetch(k) for k in range(1, depth + 1))
    for frame_id in range(args.max_frames):
        pushed = 0
        cur = queue.popleft() if frame_id > 0 and queue else {}
        pairs = {}
        for s, d in enumerate(dirs):
            if not live[s]:
                continue
            if frame_id == 0:
                pairs[s] = first[s]
            elif s in cur:
                t0 = time.perf_counter()
                pairs[s] = cur[s].result()
                t_wait += time.perf_counter() - t0
            else:
                pairs[s] = read_pair(d, frame_id)
        if frame_id > 0:  # the frames ahead are decoded while this step is pushed and runs
            queue.append(prefetch(frame_id + depth))
        for s, d in enumerate(dirs):
            if not live[s]:
                continue
            pair = pairs.get(s)
            if pair is None or pair[0].shape != (h, w):
                live[s] = False   # the reference runs until imread fails (main.cpp:123, utils.cpp:178)
                continue
            vo.push(s, pair[0], pair[1])
            n_read[s] += 1
            pushed += 1
        if not pushed:
            break
        vo.step()                 # asynchronous: the next pairs are decoded while this step runs
    vo.sync()
    elapsed = time.perf_counter() - t_start
    if pool is not None:
        pool.shutdown(wait=True, cancel_futures=True)
    frames_done = sum(max(0, n - 1) for n in n_read)
    print(dict(end_to_end_frames_per_s=frames_done / elapsed if elapsed > 0 else 0.0, frames=frames_done, seconds=elapsed,
               sequences=S, decode_threads=args.decode_threads, seconds_waiting_for_decoders=t_wait,
               note="from the image files: read + decode + upload + compute"), flush=True)
    results = []
    for s, d in enumerate(dirs):
        traj = vo.trajectory(s)
        path = "%s_%d.txt" % (args.out, s)
        vo.save_trajectory(s, path)
        log = vo.log(s)
        rec = dict(sequence=d, frames=n_read[s], poses=len(traj), trajectory=path,
                   integrated=sum(r["integrated"] for r in log),
                   mean_tracked=float(np.mean([r["n_tracked"] for r in log])) if log else 0.0,
                   mean_inliers=float(np.mean([r["n_inliers"] for r in log])) if log else 0.0)
        if s < len(gts) and gts[s]:
            gt = odometry.load_poses(gts[s])[:len(traj)]
            rec["ate_rmse_m"] = odometry.ate_rmse(traj, gt)
            rec["kitti_segment_errors"] = odometry.sequence_error_summary(gt, traj[:len(gt)])
        results.append(rec)
        print(rec, flush=True)
    vo.close()
    return results


if __name__ == "__main__":
    main()
