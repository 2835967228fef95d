"""Frame loop of the reference (src/main.cpp:123-224) on top of the C ABI -- the host-side mirror a
user of ZhenghaoFei/visual_odom switches to: the same state (FeatureSet points/ages, frame_pose,
translation), the same order of operations, every arithmetic step inside libvo_hip:

    matchingFeatures        visualOdometry.cpp:81-129   -> vo_detect_bucket + vo_track_frame
    triangulatePoints ...   main.cpp:169-171            -> (inside vo_track_frame)
    trackingFrame2Frame     visualOdometry.cpp:132-193  -> (inside vo_track_frame)
    euler gate + integrateOdometryStereo  main.cpp:196-208, utils.cpp:57-131 -> vo_integrate_odometry

GUI calls (displayTracking, display) and image decoding are out of scope; the trajectory the
reference only draws (utils.cpp:19-48) is written in the KITTI pose format instead (12 doubles per
line, row-major 3x4 -- what loadPoses reads, evaluate_odometry.cpp:24-27).  There is no CPU fallback.
"""
import numpy as np

from . import _lib


class StereoOdometry:
    """The frame loop of main.cpp:123-224 over the synchronous drop-in calls (vo_detect_bucket + vo_track_frame), the way the
    INTEGRATION.md adapter runs it.  keep_pair=True (default): from the second frame on the calls name the previous call's t1
    pair as their t0 pair (two uploads and two pyramids per frame, as the reference keeps imageLeft_t0 / imageRight_t0,
    main.cpp:157-158); keep_pair=False hands all four images over every frame (the stateless form).  streaming=True runs the
    same loop through the batch API instead (a device-resident ring of two pairs, detection .. pose solve as one batch of one
    frame) -- the slowest of the three since round 5 (tools/latency_mode.py: 1.10 / 0.81 / 0.75 ms per frame).  Same results."""

    def __init__(self, P_l, P_r, device=0, max_w=1241, max_h=376, max_pts=4096, ctx=None, streaming=False,
                 mono_rotation=False, keep_pair=True, **detect_kw):
        self.P_l = np.ascontiguousarray(P_l, np.float32).reshape(3, 4)
        self.P_r = np.ascontiguousarray(P_r, np.float32).reshape(3, 4)
        self.ctx = ctx if ctx is not None else _lib.Context(device, max_w, max_h, max_pts, 1)
        self._own = ctx is None
        self.detect_kw = detect_kw
        self.streaming = streaming
        self.keep_pair, self._kept, self._kept_id = bool(keep_pair), False, 0
        # trackingFrame2Frame's `mono_rotation` (visualOdometry.h:42; main.cpp:181 passes false)
        self.mono_rotation = bool(mono_rotation)
        self.ctx.set_params(mono_rotation=int(self.mono_rotation))
        self._n_pairs = 0
        # main.cpp:81-94
        self.points = np.zeros((0, 2), np.float32)   # currentVOFeatures.points
        self.ages = np.zeros(0, np.int32)            # currentVOFeatures.ages (may be longer than points)
        self.frame_pose = np.eye(4)
        self.translation = np.zeros(3)
        self.rotation = np.eye(3)
        self.prev = None
        self.trajectory = [self.frame_pose[:3].copy()]
        self.log = []

    def close(self):
        if self._own:
            self.ctx.close()

    def process(self, left, right):
        """feed the next stereo pair; returns the per-frame record (None for the very first pair)"""
        cur = (np.ascontiguousarray(left, np.uint8), np.ascontiguousarray(right, np.uint8))
        if self.streaming:
            pts, ages, out = self._stream_step(cur)
            if out is None:
                return None
        else:
            if self.prev is None:
                self.prev = cur
                return None
            (l0, r0), (l1, r1) = self.prev, cur
            # matchingFeatures: appendNewFeatures + bucketingFeatures (visualOdometry.cpp:95-108)
            pts = out = None
            # the previous call's t1 pair is this call's t0 pair (main.cpp:157-158) -- still on the device IF nobody else has
            # used this context since: the pair's id must be the one this object saw after its own call (another frame loop
            # or a direct call on a shared context leaves another id: its images would silently become our t0 pair)
            if self._kept and self.ctx.kept_pair_id() == self._kept_id:
                try:
                    pts, ages = self.ctx.detect_bucket(None, self.points, self.ages, **self.detect_kw)
                    out = self.ctx.track_frame(None, None, l1, r1, pts, self.P_l, self.P_r, tvec=self.translation)
                except _lib.VoError as e:   # the context lost the pair in between (batch / sequence API, a failed call): all four again
                    if e.code != _lib.VO_ERR_STATE:
                        raise
                    pts = out = None
            if out is None:
                pts, ages = self.ctx.detect_bucket(l0, self.points, self.ages, **self.detect_kw)
                # circularMatching + consistency filter + triangulation + PnP (visualOdometry.cpp:110-127, main.cpp:169-181)
                out = self.ctx.track_frame(l0, r0, l1, r1, pts, self.P_l, self.P_r, tvec=self.translation)
            self._kept, self._kept_id = self.keep_pair, self.ctx.kept_pair_id()
        # deleteUnmatchFeaturesCircle: ages += 1, compacted with the circular-matching survivors only
        # (feature.cpp:83-86,111); the consistency filter does not touch ages (quirk B3)
        self.ages = (ages + 1)[out["keep_idx_circ"]]
        self.points = out["l1"]                       # currentVOFeatures.points = pointsLeft_t1
        self.prev = None if self.streaming else cur   # main.cpp:157-158 (streaming: the pair stays on the device)
        rec = dict(n_bucketed=len(pts), n_tracked=len(out["l1"]), n_inliers=len(out["inliers"]), rc=out["rc"],
                   rvec=out["rvec"].copy(), tvec=out["tvec"].copy(), integrated=False)
        if out["rc"] == _lib.VO_ERR_TOO_FEW:
            raise _lib.VoError(out["rc"], "fewer than 4 correspondences reached solvePnPRansac (the reference asserts here)")
        if self.mono_rotation and self.ctx.batch_get_essential(0, 0)["status"] != 1:
            raise _lib.VoError(1, "findEssentialMat found no model (the reference's recoverPose throws on the empty E)")
        self.rotation, self.translation = out["R"], out["tvec"]
        self.frame_pose, rec["integrated"], rec["euler"] = _lib.integrate_odometry(self.frame_pose, self.rotation,
                                                                                   self.translation)
        self.trajectory.append(self.frame_pose[:3].copy())
        self.log.append(rec)
        return rec

    def _stream_step(self, cur):
        """device-resident ring of two stereo pairs: slots (0, 1) and (2, 3) of the image table"""
        ctx = self.ctx
        h, w = cur[0].shape
        slot = 2 * (self._n_pairs % 2)
        if self._n_pairs == 0:
            ctx.batch_configure(4, w, h, 1)
            ctx.batch_set_projection(self.P_l, self.P_r)
            ctx.batch_set_detect_params(**self.detect_kw)
        ctx.batch_upload_image(slot, cur[0])
        ctx.batch_upload_image(slot + 1, cur[1])
        ctx.batch_set_pyramid_range(slot, 2)           # only the new pair's pyramids are built
        self._n_pairs += 1
        if self._n_pairs == 1:
            ctx.batch_run(_lib.STAGE_PYRAMID)
            ctx.batch_sync()
            return None, None, None
        old = 2 - slot
        ctx.batch_set_quads([[old, old + 1, slot, slot + 1]])
        ctx.batch_set_features(0, self.points, self.ages)
        ctx.batch_run(_lib.STAGE_ALL | _lib.STAGE_DETECT)
        ctx.batch_sync()
        pts, ages = ctx.batch_get_features(0)
        f = ctx.batch_get_filtered(0)
        p = ctx.batch_get_pose(0)
        rc = _lib.VO_ERR_TOO_FEW if p["status"] < 0 else (0 if p["status"] == 1 else 1)
        tvec = p["tvec"] if p["status"] >= 0 else self.translation
        out = dict(rc=rc, l1=f["l1"], keep_idx_circ=f["keep_idx_circ"], inliers=p["inliers"], rvec=p["rvec"], tvec=tvec,
                   R=p["R"])
        return pts, ages, out

    def save_trajectory(self, path):
        with open(path, "w") as f:
            for T in self.trajectory:
                f.write(" ".join("%.9e" % v for v in T.reshape(-1)) + "\n")


class MultiSequenceOdometry:
    """The same frame loop for S independent sequences in lock step (vo_seq_* of the C ABI): every step consumes one
    new stereo pair per sequence, and everything that chains frame k to frame k + 1 -- currentVOFeatures (points and
    the longer ages array, quirk B3), the previous pair, frame_pose -- stays on the device.  Within a sequence frames
    are serial (visualOdometry.cpp:127, main.cpp:157-158), so S sequences x 1 frame per step is the exact-replay way
    to fill the GPU; nothing crosses PCIe but the new images (on a copy stream, under the previous step's kernels)
    and, when asked for, the trajectories."""

    def __init__(self, P_l, P_r, n_seq, width, height, device=0, max_pts=4096, ring=3, max_steps=1024, ctx=None,
                 mono_rotation=False, **detect_kw):
        self.ctx = ctx if ctx is not None else _lib.Context(device, width, height, max_pts, n_seq)
        self._own = ctx is None
        self.n_seq = n_seq
        self.ctx.set_params(mono_rotation=int(bool(mono_rotation)))
        self.ctx.batch_set_detect_params(**detect_kw)   # before configure: they decide how a step is scheduled
        self.ctx.seq_configure(n_seq, width, height, ring, max_steps)
        self.ctx.batch_set_projection(P_l, P_r)

    def close(self):
        if self._own:
            self.ctx.close()

    def push(self, seq, left, right, pinned=False):
        """next stereo pair of sequence `seq` (a sequence that gets no pair in a step pauses)"""
        self.ctx.seq_push_pair(seq, left, right, pinned)

    def step(self):
        """enqueue one step over all sequences (asynchronous)"""
        self.ctx.seq_step()

    def sync(self):
        self.ctx.seq_sync()

    def state(self, seq):
        """currentVOFeatures.points, .ages and frame_pose of one sequence (blocks)"""
        return self.ctx.seq_get_state(seq)

    def trajectory(self, seq):
        """list of 3x4 poses like StereoOdometry.trajectory: identity, then one per processed frame"""
        rows, _ = self.ctx.seq_get_trajectory(seq)
        return [np.eye(4)[:3]] + [r[:12].reshape(3, 4) for r in rows]

    def log(self, seq):
        rows, info = self.ctx.seq_get_trajectory(seq)
        out = []
        for r, i in zip(rows, info):
            rec = dict(zip(_lib.SEQ_INFO_NAMES, (int(v) for v in i)))
            rec.update(rvec=r[12:15].copy(), tvec=r[15:18].copy(), R=r[18:27].reshape(3, 3).copy(),
                       integrated=bool(rec["flags"] & _lib.SEQ_F_INTEGRATED))
            out.append(rec)
        return out

    def save_trajectory(self, seq, path):
        with open(path, "w") as f:
            for T in self.trajectory(seq):
                f.write(" ".join("%.9e" % v for v in np.asarray(T).reshape(-1)) + "\n")


def load_poses(path):
    """KITTI pose file -> (n, 3, 4) array (evaluate_odometry.cpp:17-33 loadPoses)"""
    return np.loadtxt(path).reshape(-1, 3, 4)


KITTI_LENGTHS = (100, 200, 300, 400, 500, 600, 700, 800)  # evaluate_odometry.cpp:14


def _as4x4(T):
    T = np.asarray(T, np.float64)
    if T.shape == (4, 4):
        return T
    M = np.eye(4)
    M[:3] = T.reshape(3, 4)
    return M


def trajectory_distances(poses):
    """evaluate_odometry.cpp:35-47 trajectoryDistances: cumulative path length, float32 like the reference"""
    dist = [np.float32(0)]
    for i in range(1, len(poses)):
        d = (np.asarray(poses[i - 1])[:3, 3] - np.asarray(poses[i])[:3, 3]).astype(np.float32)
        dist.append(np.float32(dist[i - 1] + np.float32(np.sqrt(np.float32(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])))))
    return dist


def calc_sequence_errors(poses_gt, poses_result, lengths=KITTI_LENGTHS, step_size=10):
    """evaluate_odometry.cpp:71-116 calcSequenceErrors: for every 10th start frame and every segment length, the
    rotation error [rad/m] and translation error [fraction] of the result's relative motion against ground truth.
    Returns a list of (first_frame, r_err, t_err, len, speed) like the reference's `errors` records."""
    gt = [_as4x4(T) for T in poses_gt]
    res = [_as4x4(T) for T in poses_result]
    dist = trajectory_distances(gt)
    err = []
    for first in range(0, len(gt), step_size):
        for length in lengths:
            last = -1
            for i in range(first, len(dist)):                       # lastFrameFromSegmentLength (:49-54)
                if dist[i] > dist[first] + np.float32(length):
                    last = i
                    break
            if last == -1 or last >= len(res):
                continue
            delta_gt = np.linalg.inv(gt[first]) @ gt[last]
            delta_res = np.linalg.inv(res[first]) @ res[last]
            E = np.linalg.inv(delta_res) @ delta_gt
            d = np.float32(0.5) * (np.float32(E[0, 0]) + np.float32(E[1, 1]) + np.float32(E[2, 2]) - np.float32(1.0))
            r_err = float(np.arccos(max(min(d, np.float32(1.0)), np.float32(-1.0))))        # rotationError (:56-62)
            t_err = float(np.sqrt(np.sum(E[:3, 3].astype(np.float32) ** 2, dtype=np.float32)))  # translationError
            num_frames = float(last - first + 1)
            err.append((first, r_err / length, t_err / length, float(length), length / (0.1 * num_frames)))
    return err


def sequence_error_summary(poses_gt, poses_result, lengths=KITTI_LENGTHS):
    """mean t_err [%] and r_err [deg/m] over all segments -- the two numbers the KITTI benchmark quotes; None when
    the sequence is shorter than the shortest segment length"""
    err = calc_sequence_errors(poses_gt, poses_result, lengths)
    if not err:
        return None
    e = np.asarray(err)
    return dict(t_err_percent=float(100.0 * e[:, 2].mean()), r_err_deg_per_m=float(np.degrees(e[:, 1].mean())),
                segments=len(err))


def ate_rmse(traj, ref):
    """root-mean-square translation difference of two trajectories expressed in the same frame"""
    a = np.asarray(traj)[:, :3, 3]
    b = np.asarray(ref)[:, :3, 3]
    n = min(len(a), len(b))
    return float(np.sqrt(np.mean(np.sum((a[:n] - b[:n]) ** 2, axis=1))))
