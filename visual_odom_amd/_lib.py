"""ctypes binding of libvo_hip.so (the C ABI in include/vo_hip.h).

There is no CPU fallback: if the HIP library is missing, fails to load, or no GPU is present,
`load()` / `Context()` raise -- the product path never routes through the oracle.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# VO_HIP_LIB: a developer build of the same library (python -m visual_odom_amd.build --dev -> libvo_hip_dev.so, the
# measured-slower kernel variants and their environment switches compiled in) for tools/ -- never a different backend
SO_PATH = os.environ.get("VO_HIP_LIB") or os.path.join(HERE, "libvo_hip.so")

VO_OK, VO_ERR_ARG, VO_ERR_HIP, VO_ERR_STATE, VO_ERR_TOO_FEW, VO_ERR_OVERFLOW = 0, -1, -2, -3, -4, -5
VO_NO_MODEL, VO_NO_ESSENTIAL = 1, 2
SEQ_ROW, SEQ_INFO = 27, 8
SEQ_F_ACTIVE, SEQ_F_INTEGRATED, SEQ_F_TOO_FEW, SEQ_F_NO_ESSENTIAL, SEQ_F_GAP = 1, 2, 4, 8, 16
SEQ_INFO_NAMES = ("n_bucketed", "n_circ", "n_tracked", "n_inliers", "pnp_status", "flags", "ransac_iters", "overflow")
STAGE_PYRAMID, STAGE_LK, STAGE_FILTER, STAGE_TRIANGULATE, STAGE_PNP, STAGE_ALL = 1, 2, 4, 8, 16, 31
STAGE_DETECT = 32
EVENT_SLOTS = 256
STAGE_NAMES = ("pyramid", "detect", "lk", "filter", "triangulate", "pnp")
NUM_STAGES = len(STAGE_NAMES)

# every symbol include/vo_hip.h declares (checked by the CPU test-suite against the built .so)
EXPORTS = (
    "vo_default_params", "vo_default_detect_params", "vo_integrate_odometry", "vo_fast_detect", "vo_detect_bucket",
    "vo_batch_set_features", "vo_batch_set_pyramid_range", "vo_batch_set_detect_params", "vo_batch_get_features", "vo_create", "vo_destroy", "vo_last_error", "vo_set_params", "vo_get_params",
    "vo_circular_match", "vo_triangulate", "vo_pnp_ransac", "vo_track_frame",
    "vo_batch_configure", "vo_batch_upload_image", "vo_batch_upload_image_dev", "vo_batch_set_quads",
    "vo_batch_set_points", "vo_batch_set_projection", "vo_batch_run", "vo_batch_run_timed", "vo_batch_run_slot", "vo_batch_slot_times",
    "vo_batch_sync", "vo_batch_get_tracks", "vo_batch_get_filtered", "vo_batch_get_pose",
    "vo_batch_get_pyramid_level", "vo_model_bytes", "vo_essential_pose", "vo_batch_get_essential",
    "vo_seq_configure", "vo_seq_reset", "vo_seq_push_pair", "vo_seq_push_pair_dev", "vo_seq_push_pairs", "vo_seq_step", "vo_seq_sync",
    "vo_seq_get_state", "vo_seq_get_trajectory", "vo_set_schedule", "vo_get_schedule", "vo_get_probe_log",
    "vo_export_schedule", "vo_import_schedule", "vo_kept_pair_id",
)


class VoParams(C.Structure):
    _fields_ = [("lk_max_level", C.c_int), ("lk_max_count", C.c_int), ("lk_epsilon", C.c_double),
                ("lk_min_eig_threshold", C.c_double), ("lk_full_chain", C.c_int), ("consistency_threshold", C.c_int),
                ("ransac_iterations", C.c_int), ("ransac_reproj_error", C.c_float),
                ("ransac_confidence", C.c_double), ("mono_rotation", C.c_int), ("em_prob", C.c_double),
                ("em_threshold", C.c_double)]


class VoDetectParams(C.Structure):
    _fields_ = [("fast_threshold", C.c_int), ("fast_nonmax", C.c_int), ("redetect_below", C.c_int),
                ("bucket_size", C.c_int), ("features_per_bucket", C.c_int)]


class VoSchedule(C.Structure):
    """vo_schedule (include/vo_hip.h): pose_waves 0 = probe / 1 / 2, pose_streams 0 = probe / 1 / 2, prepare -1 = probe / 0 / 1,
    epnp_wide_frames 0 = probe / 4 / 16"""
    _fields_ = [("pose_waves", C.c_int), ("pose_streams", C.c_int), ("prepare", C.c_int), ("epnp_wide_frames", C.c_int)]


class VoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libvo_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    """dlopen libvo_hip.so; raises if it has not been built (python -m visual_odom_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError("libvo_hip.so not built: run `python -m visual_odom_amd.build` "
                           "(there is no CPU fallback)")
    lib = C.CDLL(SO_PATH)
    lib.vo_create.restype = C.c_void_p
    lib.vo_create.argtypes = [C.c_int] * 5
    lib.vo_destroy.argtypes = [C.c_void_p]
    lib.vo_destroy.restype = None
    lib.vo_last_error.restype = C.c_char_p
    lib.vo_last_error.argtypes = [C.c_void_p]
    lib.vo_default_params.argtypes = [C.POINTER(VoParams)]
    lib.vo_default_params.restype = None
    lib.vo_default_detect_params.argtypes = [C.POINTER(VoDetectParams)]
    lib.vo_default_detect_params.restype = None
    lib.vo_kept_pair_id.restype = C.c_int64
    lib.vo_kept_pair_id.argtypes = [C.c_void_p]
    _lib = lib
    return lib


class VoScheduleRecord(C.Structure):
    """vo_schedule_record: the key a schedule was settled for + the schedule"""
    _fields_ = [("key", C.c_int64 * 8), ("schedule", VoSchedule)]


def export_schedules():
    """the process-wide table of settled schedules as a list of dicts (JSON-able): vo_export_schedule"""
    lib = load()
    n = C.c_int(0)
    rc = lib.vo_export_schedule(None, 0, C.byref(n))
    if rc != VO_OK:
        raise VoError(rc, "vo_export_schedule")
    recs = (VoScheduleRecord * max(n.value, 1))()
    rc = lib.vo_export_schedule(recs, n.value, C.byref(n))
    if rc != VO_OK:
        raise VoError(rc, "vo_export_schedule")
    return [dict(key=[int(v) for v in r.key], pose_waves=r.schedule.pose_waves, pose_streams=r.schedule.pose_streams,
                 prepare=r.schedule.prepare, epnp_wide_frames=r.schedule.epnp_wide_frames) for r in recs[:n.value]]


def import_schedules(records):
    """vo_import_schedule: records as export_schedules() returned them (e.g. read back from a JSON file)"""
    lib = load()
    recs = (VoScheduleRecord * max(len(records), 1))()
    for r, d in zip(recs, records):
        for i in range(8):
            r.key[i] = int(d["key"][i])
        r.schedule = VoSchedule(int(d["pose_waves"]), int(d["pose_streams"]), int(d["prepare"]), int(d.get("epnp_wide_frames", 4)))
    rc = lib.vo_import_schedule(recs, len(records))
    if rc != VO_OK:
        raise VoError(rc, "vo_import_schedule: bad record")


_ZERO_BYTES = C.c_char * 0


def _p(a):
    """the array's base address as a foreign-function argument.  A zero-length ctypes array mapped onto the buffer (0.7 us; it keeps
    the array alive and is passed as its address) where the buffer protocol allows it -- writable, contiguous -- and numpy's
    `.ctypes` helper (3.8 us: eleven of them were 40 us of every vo_track_frame call) for read-only or strided views"""
    if a is None:
        return None
    try:
        return _ZERO_BYTES.from_buffer(a)
    except (TypeError, ValueError, BufferError):
        return a.ctypes.data_as(C.c_void_p)


def _imgs(*arrays):
    """8-bit gray images as the ABI takes them: a base pointer + one byte stride for all of them.  Row-contiguous views (a
    padded buffer, an ROI of a bigger image: strides (stride, 1)) are passed AS THEY ARE with their stride -- no copy --
    when they all share it; anything else is made contiguous (stride = width)."""
    arrs = [np.asarray(a) for a in arrays]
    ok = all(a.dtype == np.uint8 and a.ndim == 2 and a.strides[1] == 1 and a.strides[0] >= a.shape[1] for a in arrs)
    if not ok or len({a.strides[0] for a in arrs}) != 1 or len({a.shape for a in arrs}) != 1:
        arrs = [np.ascontiguousarray(a, np.uint8) for a in arrs]
    return arrs, int(arrs[0].strides[0])


def _imgs_opt(*arrays):
    """_imgs over the images that are given; None stays None (the C ABI's NULL: vo_hip.h, THE KEPT PAIR)"""
    arrs, stride = _imgs(*[a for a in arrays if a is not None])
    it = iter(arrs)
    return [None if a is None else next(it) for a in arrays], stride


def _pn(a):
    """_p, or NULL for None"""
    return None if a is None else _p(a)


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, np.float32)
    return a if shape is None else a.reshape(shape)


def integrate_odometry(pose, R, t):
    """main.cpp:196-208 + utils.cpp:57-131: returns (new 4x4 pose, integrated?, euler xyz)"""
    lib = load()
    pose = np.array(pose, np.float64).reshape(4, 4).copy()
    R = np.ascontiguousarray(R, np.float64).reshape(3, 3)
    t = np.ascontiguousarray(t, np.float64).reshape(3)
    e = np.zeros(3, np.float32)
    rc = lib.vo_integrate_odometry(_p(pose), _p(R), _p(t), _p(e))
    if rc < 0:
        raise VoError(rc, "vo_integrate_odometry: bad argument")
    return pose, bool(rc), e


class Context:
    """One vo_ctx: owns device buffers + a HIP stream on `device`."""

    def __init__(self, device=0, max_w=1241, max_h=376, max_pts=4096, max_frames=1):
        self.lib = load()
        self.h = self.lib.vo_create(device, max_w, max_h, max_pts, max_frames)
        if not self.h:
            raise RuntimeError("vo_create failed: no HIP device %d or out of memory" % device)
        self.h = C.c_void_p(self.h)
        self.max_pts, self.max_frames = max_pts, max_frames
        self.n_frames = 0
        self._kept_shape = (0, 0)   # (h, w) of the pair the last track_frame kept on the device

    def close(self):
        if getattr(self, "h", None):
            self.lib.vo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, allow=()):
        if rc != VO_OK and rc not in allow:
            raise VoError(rc, (self.lib.vo_last_error(self.h) or b"").decode())
        return rc

    # ---- parameters ---------------------------------------------------------------------
    def get_params(self):
        p = VoParams()
        self._chk(self.lib.vo_get_params(self.h, C.byref(p)))
        return p

    def set_params(self, **kw):
        p = self.get_params()
        for k, v in kw.items():
            if not hasattr(p, k):
                raise KeyError(k)
            setattr(p, k, v)
        self._chk(self.lib.vo_set_params(self.h, C.byref(p)))

    def set_schedule(self, pose_waves=0, pose_streams=0, prepare=-1, epnp_wide_frames=0):
        """pin knobs of the pose-chain schedule (0 / 0 / -1 / 0 = probe, the default); see vo_schedule in vo_hip.h"""
        s = VoSchedule(int(pose_waves), int(pose_streams), int(prepare), int(epnp_wide_frames))
        self._chk(self.lib.vo_set_schedule(self.h, C.byref(s)))

    def get_schedule(self):
        """the schedule the next run uses: dict(pose_waves, pose_streams, prepare, probed)"""
        s, probed = VoSchedule(), C.c_int(0)
        self._chk(self.lib.vo_get_schedule(self.h, C.byref(s), C.byref(probed)))
        return dict(pose_waves=s.pose_waves, pose_streams=s.pose_streams, prepare=s.prepare, epnp_wide_frames=s.epnp_wide_frames,
                    probed=bool(probed.value), settling=probed.value == 2)

    def get_probe_log(self):
        """{"w,s,p": steady-state ms per run} as measured by the last schedule probe of this context ({} if none ran)"""
        cands, ms, real, n = (VoSchedule * 16)(), (C.c_float * 16)(), (C.c_int * 16)(), C.c_int(0)
        self._chk(self.lib.vo_get_probe_log(self.h, cands, ms, real, C.byref(n)))
        return {"%d,%d,%d%s%s" % (cands[i].pose_waves, cands[i].pose_streams, cands[i].prepare,
                                  ",wide%d" % cands[i].epnp_wide_frames if cands[i].epnp_wide_frames != 4 else "",
                                  " (real steps)" if real[i] else ""): float(ms[i]) for i in range(n.value)}

    def kept_pair_id(self):
        """identity of the stereo pair the last track_frame / circular_match left on the device (0: none) -- a caller that
        shares this context passes l0 = r0 = None only while the id is the one it saw after its own call (vo_hip.h)"""
        return int(self.lib.vo_kept_pair_id(self.h))

    # ---- drop-in calls ------------------------------------------------------------------
    def circular_match(self, l0, r0, l1, r1, pts_l0, apply_consistency=False):
        imgs, stride = _imgs_opt(l0, r0, l1, r1)   # (l0 = r0 = None: the pair the previous call kept, see track_frame)
        h, w = imgs[2].shape
        self._kept_shape = (h, w)
        pts = _f32(pts_l0, (-1, 2))
        n = pts.shape[0]
        outs = [np.zeros((max(n, 1), 2), np.float32) for _ in range(5)]
        st = np.zeros((4, max(n, 1)), np.uint8)
        keep = np.zeros(max(n, 1), np.int32)
        n_out = C.c_int(0)
        st_flat = np.zeros(4 * max(n, 1), np.uint8)
        self._chk(self.lib.vo_circular_match(self.h, _pn(imgs[0]), _pn(imgs[1]), _p(imgs[2]), _p(imgs[3]),
                                             w, h, stride, _p(pts), n, _p(outs[0]), _p(outs[1]), _p(outs[2]),
                                             _p(outs[3]), _p(outs[4]), _p(st_flat), _p(keep),
                                             C.byref(n_out), int(apply_consistency)))
        m = n_out.value
        st = st_flat[:4 * n].reshape(4, n) if n else st[:, :0]
        return dict(l0=outs[0][:m].copy(), r0=outs[1][:m].copy(), r1=outs[2][:m].copy(),
                    l1=outs[3][:m].copy(), l0_ret=outs[4][:m].copy(), status4=st.copy(),
                    keep_idx=keep[:m].copy(), n_out=m)

    def triangulate(self, P_l, P_r, pts_l, pts_r):
        P_l, P_r = _f32(P_l, (3, 4)), _f32(P_r, (3, 4))
        pl, pr = _f32(pts_l, (-1, 2)), _f32(pts_r, (-1, 2))
        n = pl.shape[0]
        xyz = np.zeros((max(n, 1), 3), np.float32)
        self._chk(self.lib.vo_triangulate(self.h, _p(P_l), _p(P_r), _p(pl), _p(pr), n, _p(xyz)))
        return xyz[:n].copy()

    def pnp_ransac(self, xyz, uv, K, rvec=None, tvec=None):
        """returns (found, rvec, tvec, R, inliers)"""
        xyz, uv, K = _f32(xyz, (-1, 3)), _f32(uv, (-1, 2)), _f32(K, (3, 3))
        n = xyz.shape[0]
        rv = np.zeros(3) if rvec is None else np.array(rvec, np.float64).reshape(3).copy()
        tv = np.zeros(3) if tvec is None else np.array(tvec, np.float64).reshape(3).copy()
        R = np.zeros((3, 3))
        inl = np.zeros(max(n, 1), np.int32)
        ninl = C.c_int(0)
        rc = self._chk(self.lib.vo_pnp_ransac(self.h, _p(xyz), _p(uv), n, _p(K), _p(rv), _p(tv), _p(R),
                                              _p(inl), C.byref(ninl)), allow=(1,))
        return rc == VO_OK, rv, tv, R, inl[:ninl.value].copy()

    def essential_pose(self, pts0, pts1, focal, pp, prob=0.999, threshold=1.0):
        """findEssentialMat(RANSAC) + recoverPose (visualOdometry.cpp:152-153).
        Returns (found, E, R, t, mask, n_good)."""
        p0, p1 = _f32(pts0, (-1, 2)), _f32(pts1, (-1, 2))
        n = p0.shape[0]
        E, R, t = np.zeros((3, 3)), np.zeros((3, 3)), np.zeros(3)
        mask = np.zeros(max(n, 1), np.uint8)
        good = C.c_int(0)
        rc = self._chk(self.lib.vo_essential_pose(self.h, _p(p0), _p(p1), n, C.c_double(focal), C.c_double(pp[0]),
                                                  C.c_double(pp[1]), C.c_double(prob), C.c_double(threshold), _p(E),
                                                  _p(R), _p(t), _p(mask), C.byref(good)), allow=(1,))
        return rc == VO_OK, E, R, t, mask[:n].copy(), good.value

    def detect_params(self, **kw):
        p = VoDetectParams()
        self.lib.vo_default_detect_params(C.byref(p))
        for k, v in kw.items():
            if not hasattr(p, k):
                raise KeyError(k)
            setattr(p, k, v)
        return p

    def fast_detect(self, img, threshold=20, nonmax=True, cap=65536):
        """featureDetectionFast (feature.cpp:39-47): corners in row-major order, (n, 2) float32"""
        if img is None:   # the left image of the pair the last track_frame kept
            (h, w), stride = self._kept_shape, self._kept_shape[1]
        else:
            (img,), stride = _imgs(img)
            h, w = img.shape
        pts = np.zeros((cap, 2), np.float32)
        n = C.c_int(0)
        self._chk(self.lib.vo_fast_detect(self.h, _pn(img), w, h, stride, int(threshold), int(bool(nonmax)), _p(pts), cap,
                                          C.byref(n)))
        if n.value > cap:
            raise VoError(VO_ERR_ARG, "fast_detect: %d corners exceed cap %d" % (n.value, cap))
        return pts[:n.value].copy()

    def detect_bucket(self, img, pts, ages, **detect_kw):
        """appendNewFeatures (if fewer than redetect_below points) + bucketingFeatures
        (visualOdometry.cpp:95-108); returns (points, ages) of the bucketed set.  img is None: the left image of the pair
        the last track_frame kept (its l1)"""
        if img is None:
            (h, w), stride = self._kept_shape, self._kept_shape[1]
        else:
            (img,), stride = _imgs(img)
            h, w = img.shape
        pts = _f32(pts, (-1, 2))
        ages = np.ascontiguousarray(ages, np.int32).reshape(-1)
        cap = max(self.max_pts, len(ages), 1)
        p_io = np.zeros((cap, 2), np.float32)
        a_io = np.zeros(cap, np.int32)
        p_io[:len(pts)] = pts
        a_io[:len(ages)] = ages
        n_pts, n_ages = C.c_int(len(pts)), C.c_int(len(ages))
        dp = self.detect_params(**detect_kw)
        self._chk(self.lib.vo_detect_bucket(self.h, _pn(img), w, h, stride, C.byref(dp), _p(p_io), C.byref(n_pts), _p(a_io),
                                            C.byref(n_ages), cap))
        return p_io[:n_pts.value].copy(), a_io[:n_ages.value].copy()

    def track_frame(self, l0, r0, l1, r1, pts_l0, P_l, P_r, rvec=None, tvec=None):
        """l0 is None and r0 is None: the t0 pair is the pair the previous call received as (l1, r1) -- kept on the device
        with its pyramids (vo_hip.h, THE KEPT PAIR; main.cpp:157-158)"""
        imgs, stride = _imgs_opt(l0, r0, l1, r1)
        h, w = imgs[2].shape
        self._kept_shape = (h, w)
        pts = _f32(pts_l0, (-1, 2))
        n = pts.shape[0]
        P_l, P_r = _f32(P_l, (3, 4)), _f32(P_r, (3, 4))
        # (fresh, uninitialised output arrays, returned as views of the part the call filled: zero-filling eight arrays and
        # copying eight slices was 55 us per call of python between two calls -- tools/host_gap_probe.py, round 5)
        m = max(n, 1)
        o = np.empty((4, m, 2), np.float32)
        xyz = np.empty((m, 3), np.float32)
        idx = np.empty((3, m), np.int32)
        n_out, n_circ, ninl = C.c_int(0), C.c_int(0), C.c_int(0)
        rv = np.zeros(3) if rvec is None else np.array(rvec, np.float64).reshape(3)
        tv = np.zeros(3) if tvec is None else np.array(tvec, np.float64).reshape(3)
        R = np.zeros((3, 3))
        ob, ib = C.addressof(_ZERO_BYTES.from_buffer(o)), C.addressof(_ZERO_BYTES.from_buffer(idx))
        rc = self._chk(self.lib.vo_track_frame(self.h, _pn(imgs[0]), _pn(imgs[1]), _p(imgs[2]), _p(imgs[3]),
                                               w, h, stride, _p(pts), n, _p(P_l), _p(P_r), C.c_void_p(ob), C.c_void_p(ob + 8 * m),
                                               C.c_void_p(ob + 16 * m), C.c_void_p(ob + 24 * m), _p(xyz), C.c_void_p(ib), C.byref(n_out),
                                               C.c_void_p(ib + 4 * m), C.byref(n_circ), _p(rv), _p(tv), _p(R), C.c_void_p(ib + 8 * m),
                                               C.byref(ninl)), allow=(VO_NO_MODEL, VO_NO_ESSENTIAL, VO_ERR_TOO_FEW))
        k = n_out.value
        return dict(rc=rc, l0=o[0, :k], r0=o[1, :k], l1=o[2, :k], r1=o[3, :k], xyz=xyz[:k], keep_idx=idx[0, :k],
                    keep_idx_circ=idx[1, :n_circ.value], rvec=rv, tvec=tv, R=R, inliers=idx[2, :ninl.value])

    # ---- batched device-resident API ------------------------------------------------------
    def batch_configure(self, n_images, w, h, n_frames):
        self._chk(self.lib.vo_batch_configure(self.h, n_images, w, h, n_frames))
        self.n_frames = n_frames

    def batch_upload_image(self, idx, img):
        (img,), stride = _imgs(img)
        self._chk(self.lib.vo_batch_upload_image(self.h, idx, _p(img), stride))

    def batch_upload_image_dev(self, idx, dev_ptr, stride):
        self._chk(self.lib.vo_batch_upload_image_dev(self.h, idx, C.c_void_p(dev_ptr), stride))

    def batch_set_quads(self, quads):
        q = np.ascontiguousarray(quads, np.int32).reshape(-1, 4)
        self._chk(self.lib.vo_batch_set_quads(self.h, _p(q), q.shape[0]))

    def batch_set_pyramid_range(self, first_image, n_images):
        self._chk(self.lib.vo_batch_set_pyramid_range(self.h, first_image, n_images))

    def batch_set_points(self, frame, pts):
        pts = _f32(pts, (-1, 2))
        self._chk(self.lib.vo_batch_set_points(self.h, frame, _p(pts), pts.shape[0]))

    def batch_set_features(self, frame, pts, ages):
        pts = _f32(pts, (-1, 2))
        ages = np.ascontiguousarray(ages, np.int32).reshape(-1)
        self._chk(self.lib.vo_batch_set_features(self.h, frame, _p(pts), pts.shape[0], _p(ages), ages.shape[0]))

    def batch_set_detect_params(self, **kw):
        dp = self.detect_params(**kw)
        self._chk(self.lib.vo_batch_set_detect_params(self.h, C.byref(dp)))

    def batch_get_features(self, frame):
        pts = np.zeros((self.max_pts, 2), np.float32)
        ages = np.zeros(self.max_pts, np.int32)
        n = C.c_int(0)
        self._chk(self.lib.vo_batch_get_features(self.h, frame, _p(pts), _p(ages), C.byref(n)))
        return pts[:n.value].copy(), ages[:n.value].copy()

    def batch_set_projection(self, P_l, P_r):
        self._chk(self.lib.vo_batch_set_projection(self.h, _p(_f32(P_l, (3, 4))), _p(_f32(P_r, (3, 4)))))

    def batch_run(self, stages=STAGE_ALL):
        self._chk(self.lib.vo_batch_run(self.h, stages))

    def batch_run_timed(self, stages=STAGE_ALL):
        ms = np.zeros(NUM_STAGES, np.float32)
        self._chk(self.lib.vo_batch_run_timed(self.h, stages, _p(ms)))
        return ms

    def batch_run_slot(self, stages, slot):
        self._chk(self.lib.vo_batch_run_slot(self.h, stages, slot))

    def batch_slot_times(self, slot):
        ms = np.zeros(NUM_STAGES, np.float32)
        self._chk(self.lib.vo_batch_slot_times(self.h, slot, _p(ms)))
        return ms

    def batch_sync(self):
        self._chk(self.lib.vo_batch_sync(self.h))

    def batch_get_tracks(self, frame, n):
        o = [np.zeros((max(n, 1), 2), np.float32) for _ in range(4)]
        st = np.zeros(4 * max(n, 1), np.uint8)
        self._chk(self.lib.vo_batch_get_tracks(self.h, frame, _p(o[0]), _p(o[1]), _p(o[2]), _p(o[3]), _p(st), n))
        return dict(r0=o[0][:n], r1=o[1][:n], l1=o[2][:n], l0_ret=o[3][:n], status4=st[:4 * n].reshape(4, n))

    def batch_get_filtered(self, frame):
        cap = self.max_pts
        o = [np.zeros((cap, 2), np.float32) for _ in range(4)]
        xyz = np.zeros((cap, 3), np.float32)
        keep, keepc = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        k, m = C.c_int(0), C.c_int(0)
        self._chk(self.lib.vo_batch_get_filtered(self.h, frame, _p(o[0]), _p(o[1]), _p(o[2]), _p(o[3]), _p(xyz),
                                                 _p(keep), C.byref(k), _p(keepc), C.byref(m)))
        k, m = k.value, m.value
        return dict(l0=o[0][:k].copy(), r0=o[1][:k].copy(), l1=o[2][:k].copy(), r1=o[3][:k].copy(),
                    xyz=xyz[:k].copy(), keep_idx=keep[:k].copy(), keep_idx_circ=keepc[:m].copy())

    def batch_get_pose(self, frame):
        rv, tv, R = np.zeros(3), np.zeros(3), np.zeros((3, 3))
        inl = np.zeros(self.max_pts, np.int32)
        ninl, status = C.c_int(0), C.c_int(0)
        dbg = np.zeros(4, np.int32)
        self._chk(self.lib.vo_batch_get_pose(self.h, frame, _p(rv), _p(tv), _p(R), _p(inl), C.byref(ninl),
                                             C.byref(status), _p(dbg)))
        return dict(rvec=rv, tvec=tv, R=R, inliers=inl[:ninl.value].copy(), status=status.value,
                    niters=int(dbg[0]), best_iter=int(dbg[1]), max_good=int(dbg[2]), lm_iters=int(dbg[3]))

    def batch_get_essential(self, frame, n):
        E, R, t = np.zeros((3, 3)), np.zeros((3, 3)), np.zeros(3)
        mask = np.zeros(max(n, 1), np.uint8)
        ninl, good, status = C.c_int(0), C.c_int(0), C.c_int(0)
        dbg = np.zeros(2, np.int32)
        self._chk(self.lib.vo_batch_get_essential(self.h, frame, _p(E), _p(R), _p(t), _p(mask), n, C.byref(ninl),
                                                  C.byref(good), C.byref(status), _p(dbg)))
        return dict(E=E, R=R, t=t, mask=mask[:n].copy(), n_inliers=ninl.value, n_good=good.value,
                    status=status.value, niters=int(dbg[0]), best=int(dbg[1]))

    # ---- lock-step sequence loop ------------------------------------------------------------
    def seq_configure(self, n_seq, w, h, ring=3, max_steps=1024):
        self._chk(self.lib.vo_seq_configure(self.h, n_seq, w, h, ring, max_steps))
        self.n_frames = n_seq

    def seq_reset(self, seq=-1):
        self._chk(self.lib.vo_seq_reset(self.h, seq))

    def seq_push_pair(self, seq, left, right, pinned=False):
        """left / right: uint8 (h, w) numpy arrays (pinned=True: views of page-locked memory that stay untouched
        until the step has run)"""
        (left, right), stride = _imgs(left, right)
        self._chk(self.lib.vo_seq_push_pair(self.h, seq, _p(left), _p(right), stride, int(bool(pinned))))

    def seq_push_pair_dev(self, seq, left_ptr, right_ptr, stride):
        self._chk(self.lib.vo_seq_push_pair_dev(self.h, seq, C.c_void_p(left_ptr), C.c_void_p(right_ptr), stride))

    def seq_pair_table(self, seq_ids, left_ptrs, right_ptrs):
        """ctypes arrays for seq_push_pairs (build once, reuse every step: the per-step host cost is one C call)"""
        n = len(seq_ids)
        ids = (C.c_int32 * n)(*[int(i) for i in seq_ids])
        lp = (C.c_void_p * n)(*[int(p) for p in left_ptrs])
        rp = (C.c_void_p * n)(*[int(p) for p in right_ptrs])
        return n, ids, lp, rp

    def seq_push_pairs(self, table, stride, kind):
        """kind: 0 pageable host, 1 page-locked host, 2 device; table from seq_pair_table"""
        n, ids, lp, rp = table
        self._chk(self.lib.vo_seq_push_pairs(self.h, n, ids, lp, rp, stride, kind))

    def seq_step(self):
        self._chk(self.lib.vo_seq_step(self.h))

    def seq_sync(self):
        self._chk(self.lib.vo_seq_sync(self.h))

    def seq_get_state(self, seq):
        """(points (n, 2) f32, ages (m,) i32 with m >= n, frame_pose 4x4 f64) of one sequence"""
        pts = np.zeros((self.max_pts, 2), np.float32)
        ages = np.zeros(self.max_pts, np.int32)
        pose = np.zeros((4, 4))
        n, m = C.c_int(0), C.c_int(0)
        self._chk(self.lib.vo_seq_get_state(self.h, seq, _p(pts), C.byref(n), _p(ages), C.byref(m), _p(pose)))
        return pts[:n.value].copy(), ages[:m.value].copy(), pose

    def seq_get_trajectory(self, seq, first=0, count=None):
        """rows (k, 27) f64 = frame_pose 3x4 | rvec | tvec | rotation 3x3, info (k, 8) i32 (SEQ_INFO_NAMES)"""
        n = C.c_int(0)
        self._chk(self.lib.vo_seq_get_trajectory(self.h, seq, 0, 0, None, None, C.byref(n)))
        k = max(0, n.value - first) if count is None else max(0, min(count, n.value - first))
        rows = np.zeros((max(k, 1), SEQ_ROW))
        info = np.zeros((max(k, 1), SEQ_INFO), np.int32)
        self._chk(self.lib.vo_seq_get_trajectory(self.h, seq, first, k, _p(rows), _p(info), C.byref(n)))
        return rows[:k].copy(), info[:k].copy()

    def batch_get_pyramid_level(self, idx, level):
        w, h = C.c_int(0), C.c_int(0)
        self._chk(self.lib.vo_batch_get_pyramid_level(self.h, idx, level, None, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        self._chk(self.lib.vo_batch_get_pyramid_level(self.h, idx, level, _p(out), C.byref(w), C.byref(h)))
        return out

    def model_bytes(self, w, h, n_points):
        b = np.zeros(3, np.float64)
        self._chk(self.lib.vo_model_bytes(self.h, w, h, n_points, _p(b)))
        return b
