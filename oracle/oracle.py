"""ctypes wrapper over the CPU ORACLE (oracle/_build/libvo_oracle.so).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/vo_oracle.h).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package visual_odom_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VO_SANITIZE=1 (the sanitizer tier, tests/test_sanitize.py): the ASan + UBSan build of the same sources, `make SAN=1`; the
# process must have been started with LD_PRELOAD=libasan.so
_SAN = os.environ.get("VO_SANITIZE", "0") not in ("", "0")
_MAKE = ["make", "-s", "-C", _HERE] + (["SAN=1"] if _SAN else [])
_SO = os.path.join(_HERE, "_build", *(["san"] if _SAN else []), "libvo_oracle.so")
_lib = None

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.check_call(_MAKE)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        # bounded default OpenMP width (see tests/conftest.py: 256 threads on the GPU box's host are 30x slower than 128)
        os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(os.cpu_count() or 8, 32))))
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
        _lib = C.CDLL(_SO)
        _lib.orc_lk_last_iteration_count.restype = C.c_longlong
    return _lib


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def pyr_down(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().orc_pyr_down(_vp(img), w, h, _vp(out))
    return out


def build_pyramid(img, max_level=3):
    pyr = [np.ascontiguousarray(img, np.uint8)]
    for _ in range(max_level):
        pyr.append(pyr_down(pyr[-1]))
    return pyr


def scharr(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.empty((h, w, 2), np.int16)
    lib().orc_scharr(_vp(img), w, h, _vp(out))
    return out


def calc_optical_flow_pyr_lk(prev, nxt, pts, win=21, max_level=3, max_count=30, eps=0.01,
                             min_eig=1e-3, accum_mode=0, nthreads=0):
    prev = np.ascontiguousarray(prev, np.uint8)
    nxt = np.ascontiguousarray(nxt, np.uint8)
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    n = pts.shape[0]
    h, w = prev.shape
    out = np.zeros((n, 2), np.float32)
    st = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    rc = lib().orc_calc_optical_flow_pyr_lk(_vp(prev), _vp(nxt), w, h, _vp(pts), n, _vp(out),
                                            _vp(st), _vp(err), win, max_level, max_count,
                                            C.c_double(eps), C.c_double(min_eig), accum_mode,
                                            nthreads)
    assert rc == 0
    return out, st, err


def lk_last_iteration_count():
    return int(lib().orc_lk_last_iteration_count())


def circular_matching(l0, r0, l1, r1, pts_l0, ages=None, nthreads=0, max_level=3):
    """feature.cpp:118-148.  Returns dict with compacted arrays (M survivors).  max_level: maxLevel of the four
    calcOpticalFlowPyrLK calls (the reference's literal is 3)."""
    imgs = [np.ascontiguousarray(a, np.uint8) for a in (l0, r0, l1, r1)]
    h, w = imgs[0].shape
    p0 = np.array(pts_l0, np.float32).reshape(-1, 2).copy()
    n = p0.shape[0]
    p1, p2, p3, p0r = (np.zeros((max(n, 1), 2), np.float32) for _ in range(4))
    st = np.zeros((4, max(n, 1)), np.uint8)
    keep = np.zeros(max(n, 1), np.int32)
    if ages is None:
        ages_a = np.zeros(max(n, 1), np.int32)
        na = C.c_int(n)
    else:
        ages_a = np.array(ages, np.int32).copy()
        na = C.c_int(len(ages_a))
        if len(ages_a) == 0:
            ages_a = np.zeros(1, np.int32)
    m = lib().orc_circular_matching_lvl(_vp(imgs[0]), _vp(imgs[1]), _vp(imgs[2]), _vp(imgs[3]), w, h,
                                        _vp(p0), n, _vp(p1), _vp(p2), _vp(p3), _vp(p0r), _vp(ages_a),
                                        C.byref(na), _vp(st), _vp(keep), nthreads, int(max_level))
    return dict(l0=p0[:m].copy(), r0=p1[:m].copy(), r1=p2[:m].copy(), l1=p3[:m].copy(),
                l0_ret=p0r[:m].copy(), ages=ages_a[:na.value].copy(), status4=st[:, :n].copy(),
                keep_idx=keep[:m].copy(), n_out=m)


def check_valid_and_remove(l0, r0, l1, r1, l0_ret, threshold=0):
    arrs = [np.array(a, np.float32).reshape(-1, 2).copy() for a in (l0, r0, l1, r1)]
    ret = np.ascontiguousarray(l0_ret, np.float32).reshape(-1, 2)
    m = arrs[0].shape[0]
    valid = np.zeros(max(m, 1), np.uint8)
    k = lib().orc_check_valid_and_remove(_vp(arrs[0]), _vp(arrs[1]), _vp(arrs[2]), _vp(arrs[3]),
                                         _vp(ret), m, threshold, _vp(valid))
    return [a[:k].copy() for a in arrs], valid[:m].astype(bool)


def triangulate(P_l, P_r, pts_l, pts_r):
    P_l = np.ascontiguousarray(P_l, np.float32).reshape(3, 4)
    P_r = np.ascontiguousarray(P_r, np.float32).reshape(3, 4)
    pl = np.ascontiguousarray(pts_l, np.float32).reshape(-1, 2)
    pr = np.ascontiguousarray(pts_r, np.float32).reshape(-1, 2)
    n = pl.shape[0]
    xyz = np.zeros((n, 3), np.float32)
    lib().orc_triangulate(_vp(P_l), _vp(P_r), _vp(pl), _vp(pr), n, _vp(xyz))
    return xyz


def triangulate_points4d(P_l, P_r, pts_l, pts_r):
    P_l = np.ascontiguousarray(P_l, np.float32).reshape(3, 4)
    P_r = np.ascontiguousarray(P_r, np.float32).reshape(3, 4)
    pl = np.ascontiguousarray(pts_l, np.float32).reshape(-1, 2)
    pr = np.ascontiguousarray(pts_r, np.float32).reshape(-1, 2)
    n = pl.shape[0]
    p4 = np.zeros((4, n), np.float32)
    lib().orc_triangulate_points(_vp(P_l), _vp(P_r), _vp(pl), _vp(pr), n, _vp(p4))
    return p4


def solve_pnp_ransac(xyz, uv, K, rvec=None, tvec=None, iterations=500, reproj=0.5,
                     confidence=float(np.float32(0.999))):
    """visualOdometry.cpp:161-178.  Returns (ok, rvec, tvec, inliers, dbg)."""
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
    K = np.ascontiguousarray(K, np.float32).reshape(3, 3)
    n = xyz.shape[0]
    rv = np.zeros(3, np.float64) if rvec is None else np.array(rvec, np.float64).reshape(3).copy()
    tv = np.zeros(3, np.float64) if tvec is None else np.array(tvec, np.float64).reshape(3).copy()
    inl = np.zeros(max(n, 1), np.int32)
    ninl = C.c_int(0)
    dbg = np.zeros(8, np.float64)
    rc = lib().orc_solve_pnp_ransac(_vp(xyz), _vp(uv), n, _vp(K), _vp(rv), _vp(tv), iterations,
                                    C.c_float(reproj), C.c_double(confidence), _vp(inl),
                                    C.byref(ninl), _vp(dbg))
    return rc, rv, tv, inl[:ninl.value].copy(), dbg


def solve_p3p(xyz, uv, K):
    """solvePnP(SOLVEPNP_P3P) on exactly 4 points (orc_p3p.c): (n_solutions, rvecs [n][3], tvecs [n][3]) in solveP3P's
    final order (the first one is what solvePnPRansac returns)"""
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(4, 3)
    uv = np.ascontiguousarray(uv, np.float32).reshape(4, 2)
    K = np.ascontiguousarray(K, np.float32).reshape(3, 3)
    rv, tv = np.zeros(3), np.zeros(3)
    rvs, tvs = np.zeros((4, 3)), np.zeros((4, 3))
    n = lib().orc_solve_p3p(_vp(xyz), _vp(uv), _vp(K), _vp(rv), _vp(tv), _vp(rvs), _vp(tvs))
    return n, rvs[:n].copy(), tvs[:n].copy()


def p3p_solve(K4, uv, xyz, p4p=True):
    """p3p::solve on f64 data: K4 = (fx, fy, cx, cy), uv [4][2] pixels, xyz [4][3]; returns (R [n][3][3], t [n][3])"""
    K4 = np.ascontiguousarray(K4, np.float64).reshape(4)
    uv = np.ascontiguousarray(uv, np.float64).reshape(4, 2)
    xyz = np.ascontiguousarray(xyz, np.float64).reshape(4, 3)
    R, t = np.zeros((4, 3, 3)), np.zeros((4, 3))
    n = lib().orc_p3p_solve(_vp(K4), _vp(uv), _vp(xyz), int(p4p), _vp(R), _vp(t))
    return R[:n].copy(), t[:n].copy()


def solve_deg4(a, b, c, d, e):
    x = np.zeros(4)
    lib().orc_solve_deg4.argtypes = [C.c_double] * 5 + [C.c_void_p]
    n = lib().orc_solve_deg4(a, b, c, d, e, _vp(x))
    return x[:n].copy()


def rodrigues(r):
    r = np.ascontiguousarray(r, np.float64)
    if r.size == 3:
        R = np.zeros((3, 3), np.float64)
        lib().orc_rodrigues_vec2mat(_vp(r.reshape(3).copy()), _vp(R), None)
        return R
    rv = np.zeros(3, np.float64)
    lib().orc_rodrigues_mat2vec(_vp(r.reshape(3, 3).copy()), _vp(rv))
    return rv


def rodrigues_jac(r):
    r = np.ascontiguousarray(r, np.float64).reshape(3).copy()
    R = np.zeros((3, 3), np.float64)
    J = np.zeros((3, 9), np.float64)
    lib().orc_rodrigues_vec2mat(_vp(r), _vp(R), _vp(J))
    return R, J


def epnp(xyz, uv, K):
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
    K = np.ascontiguousarray(K, np.float32).reshape(3, 3)
    R = np.zeros((3, 3), np.float64)
    t = np.zeros(3, np.float64)
    lib().orc_epnp(_vp(xyz), _vp(uv), xyz.shape[0], _vp(K), _vp(R), _vp(t))
    return R, t


def project_points(xyz, rvec, tvec, K):
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    K = np.ascontiguousarray(K, np.float32).reshape(3, 3)
    rv = np.ascontiguousarray(rvec, np.float64).reshape(3)
    tv = np.ascontiguousarray(tvec, np.float64).reshape(3)
    out = np.zeros((xyz.shape[0], 2), np.float32)
    lib().orc_project_points(_vp(xyz), xyz.shape[0], _vp(rv), _vp(tv), _vp(K), _vp(out))
    return out


def ransac_subsets(count, iters=500):
    idx = np.zeros((iters, 5), np.int32)
    lib().orc_ransac_subsets(count, iters, _vp(idx))
    return idx


def svd(A):
    A = np.ascontiguousarray(A, np.float64)
    m, n = A.shape
    w = np.zeros(n)
    u = np.zeros((m, n))
    vt = np.zeros((n, n))
    lib().orc_svd(_vp(A), m, n, _vp(w), _vp(u), _vp(vt))
    return w, u, vt


def bucketing_features(rows, cols, points, ages, bucket_size, features_per_bucket):
    pts = np.ascontiguousarray(points, np.float32).reshape(-1, 2)
    ag = np.ascontiguousarray(ages, np.int32)
    npts = pts.shape[0]
    cap = max(npts, 1) + ((rows // bucket_size + 1) * (cols // bucket_size + 1)) * features_per_bucket
    P = np.zeros((cap, 2), np.float32)
    A = np.zeros(max(cap, len(ag)), np.int32)
    P[:npts] = pts
    A[:len(ag)] = ag
    n_p, n_a = C.c_int(npts), C.c_int(len(ag))
    rc = lib().orc_bucketing_features(rows, cols, _vp(P), _vp(A), C.byref(n_p), C.byref(n_a), cap,
                                      bucket_size, features_per_bucket)
    assert rc >= 0
    return P[:n_p.value].copy(), A[:n_a.value].copy()


def fast_detect(img, threshold=20, nonmax=True, cap=200000):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    pts = np.zeros((cap, 2), np.float32)
    n = lib().orc_fast_detect(_vp(img), w, h, threshold, int(nonmax), _vp(pts), cap)
    assert n <= cap
    return pts[:n].copy()


def rotation_matrix_to_euler(R):
    R = np.ascontiguousarray(R, np.float64).reshape(3, 3)
    e = np.zeros(3, np.float32)
    lib().orc_rotation_matrix_to_euler(_vp(R), _vp(e))
    return e


def integrate_odometry_stereo(pose, R, t):
    pose = np.array(pose, np.float64).reshape(4, 4).copy()
    R = np.ascontiguousarray(R, np.float64).reshape(3, 3)
    t = np.ascontiguousarray(t, np.float64).reshape(3)
    ok = lib().orc_integrate_odometry_stereo(_vp(pose), _vp(R), _vp(t))
    return pose, bool(ok)


def five_point(q1, q2):
    """EMEstimatorCallback::runKernel on 5 normalised correspondences -> (k, 3, 3) essential matrices"""
    q1 = np.ascontiguousarray(q1, np.float64).reshape(5, 2)
    q2 = np.ascontiguousarray(q2, np.float64).reshape(5, 2)
    Es = np.zeros((10, 9), np.float64)
    k = lib().orc_five_point(_vp(q1), _vp(q2), _vp(Es))
    return Es[:k].reshape(k, 3, 3).copy()


def find_essential_mat(pts1, pts2, focal, pp, prob=0.999, threshold=1.0):
    """visualOdometry.cpp:152.  Returns (ok, E, mask, dbg)."""
    p1 = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(pts2, np.float32).reshape(-1, 2)
    n = p1.shape[0]
    E = np.zeros((3, 3), np.float64)
    mask = np.zeros(max(n, 1), np.uint8)
    dbg = np.zeros(3, np.float64)
    rc = lib().orc_find_essential_mat(_vp(p1), _vp(p2), n, C.c_double(focal), C.c_double(pp[0]), C.c_double(pp[1]),
                                      C.c_double(prob), C.c_double(threshold), _vp(E), _vp(mask), _vp(dbg))
    return rc, E, mask[:n].copy(), dbg


def decompose_essential_mat(E):
    E = np.ascontiguousarray(E, np.float64).reshape(3, 3)
    R1, R2, t = np.zeros((3, 3)), np.zeros((3, 3)), np.zeros(3)
    lib().orc_decompose_essential_mat(_vp(E), _vp(R1), _vp(R2), _vp(t))
    return R1, R2, t


def recover_pose(E, pts1, pts2, focal, pp, mask=None):
    """visualOdometry.cpp:153.  Returns (n_good, R, t, mask)."""
    E = np.ascontiguousarray(E, np.float64).reshape(3, 3)
    p1 = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(pts2, np.float32).reshape(-1, 2)
    n = p1.shape[0]
    R, t = np.zeros((3, 3)), np.zeros(3)
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8).copy()
    good = lib().orc_recover_pose(_vp(E), _vp(p1), _vp(p2), n, C.c_double(focal), C.c_double(pp[0]),
                                  C.c_double(pp[1]), _vp(R), _vp(t), _vp(m))
    return good, R, t, m


# ---- oracle/_ref: the reference's OWN glue sources (feature.cpp, bucket.cpp) compiled where they lie ----------
_REF_SO = os.path.join(_HERE, "_ref", *(["san"] if _SAN else []), "libvo_refglue.so")
_ref = None


def build_ref():
    """`make -C oracle ref`: only possible where /root/reference exists (the authoring container); elsewhere the
    prebuilt oracle/_ref/libvo_refglue.so that travelled with the snapshot is used.  Returns the path or None."""
    if os.path.exists("/root/reference/src/feature.cpp"):
        build()
        subprocess.check_call(_MAKE + ["ref"])
    return _REF_SO if os.path.exists(_REF_SO) else None


def ref_lib():
    """the reference's own circularMatching / bucketingFeatures / appendNewFeatures (over the oracle's LK and FAST);
    None when it cannot be built here and was not shipped"""
    global _ref
    if _ref is None:
        so = build_ref()
        if so is None:
            return None
        lib()  # resolves libvo_oracle.so first (rpath covers the normal layout)
        _ref = C.CDLL(so)
    return _ref


def ref_circular_matching(l0, r0, l1, r1, pts_l0, ages=None):
    imgs = [np.ascontiguousarray(a, np.uint8) for a in (l0, r0, l1, r1)]
    h, w = imgs[0].shape
    p0 = np.array(pts_l0, np.float32).reshape(-1, 2).copy()
    n = p0.shape[0]
    p1, p2, p3, p0r = (np.zeros((max(n, 1), 2), np.float32) for _ in range(4))
    ages_a = np.zeros(max(n, 1), np.int32) if ages is None else np.array(ages, np.int32).copy()
    na = C.c_int(n if ages is None else len(ages_a))
    if len(ages_a) == 0:
        ages_a = np.zeros(1, np.int32)
    m = ref_lib().ref_circular_matching(_vp(imgs[0]), _vp(imgs[1]), _vp(imgs[2]), _vp(imgs[3]), w, h, _vp(p0), n,
                                        _vp(p1), _vp(p2), _vp(p3), _vp(p0r), _vp(ages_a), C.byref(na))
    return dict(l0=p0[:m].copy(), r0=p1[:m].copy(), r1=p2[:m].copy(), l1=p3[:m].copy(), l0_ret=p0r[:m].copy(),
                ages=ages_a[:na.value].copy(), n_out=m)


def ref_bucketing_features(rows, cols, points, ages, bucket_size, features_per_bucket):
    pts = np.ascontiguousarray(points, np.float32).reshape(-1, 2)
    ag = np.ascontiguousarray(ages, np.int32)
    npts = pts.shape[0]
    cap = max(npts, len(ag), 1) + ((rows // bucket_size + 1) * (cols // bucket_size + 1)) * features_per_bucket
    P = np.zeros((cap, 2), np.float32)
    A = np.zeros(cap, np.int32)
    P[:npts] = pts
    A[:len(ag)] = ag
    n_p, n_a = C.c_int(npts), C.c_int(len(ag))
    rc = ref_lib().ref_bucketing_features(rows, cols, _vp(P), _vp(A), C.byref(n_p), C.byref(n_a), cap, bucket_size,
                                          features_per_bucket)
    assert rc == 0
    return P[:n_p.value].copy(), A[:n_a.value].copy()


def ref_append_new_features(img, points, ages, cap=200000):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    pts = np.ascontiguousarray(points, np.float32).reshape(-1, 2)
    ag = np.ascontiguousarray(ages, np.int32)
    P = np.zeros((cap, 2), np.float32)
    A = np.zeros(cap, np.int32)
    P[:len(pts)] = pts
    A[:len(ag)] = ag
    n_p, n_a = C.c_int(len(pts)), C.c_int(len(ag))
    rc = ref_lib().ref_append_new_features(_vp(img), w, h, _vp(P), _vp(A), C.byref(n_p), C.byref(n_a), cap)
    assert rc == 0
    return P[:n_p.value].copy(), A[:n_a.value].copy()


class RefFrameLoop:
    """The body of the reference's main() frame loop (main.cpp:144-208) driven through the reference's OWN functions
    (matchingFeatures, trackingFrame2Frame, rotationMatrixToEulerAngles, integrateOdometryStereo; oracle/_ref) over the
    oracle's OpenCV-algorithm restatement.  State as in main.cpp:81-94."""

    def __init__(self, fx, cx, cy, bf, mono_rotation=False, cap=65536, lib=None, entry="ref_frame_step"):
        # lib: another library with the same ref_frame_step entry point (tests/ref_dropin: the same reference sources
        # over libvo_hip instead of over this oracle; its ref_frame_step_adapter also replaces main.cpp:169-171,181 by
        # the shipped adapter's triangulate_hip / trackingFrame2Frame_hip)
        self.lib = lib if lib is not None else ref_lib()
        self.entry = getattr(self.lib, entry)
        self.fx, self.cx, self.cy, self.bf = (float(v) for v in (fx, cx, cy, bf))
        self.mono = int(bool(mono_rotation))
        self.cap = cap
        self.points = np.zeros((0, 2), np.float32)
        self.ages = np.zeros(0, np.int32)
        self.translation = np.zeros(3)
        self.rotation = np.eye(3)
        self.frame_pose = np.eye(4)
        self.prev = None
        self.trajectory = [self.frame_pose[:3].copy()]

    def process(self, left, right):
        cur = (np.ascontiguousarray(left, np.uint8), np.ascontiguousarray(right, np.uint8))
        if self.prev is None:
            self.prev = cur
            return None
        (l0, r0), (l1, r1) = self.prev, cur
        h, w = l0.shape
        cap = self.cap
        P = np.zeros((cap, 2), np.float32)
        A = np.zeros(cap, np.int32)
        P[:len(self.points)] = self.points
        A[:len(self.ages)] = self.ages
        n_p, n_a = C.c_int(len(self.points)), C.c_int(len(self.ages))
        t = self.translation.copy()
        R = np.ascontiguousarray(self.rotation, np.float64).copy()
        pose = np.ascontiguousarray(self.frame_pose, np.float64).copy()
        outs = [np.zeros((cap, 2), np.float32) for _ in range(4)]
        n_out, integrated = C.c_int(0), C.c_int(0)
        rc = self.entry(_vp(l0), _vp(r0), _vp(l1), _vp(r1), w, h, C.c_float(self.fx), C.c_float(self.cx),
                        C.c_float(self.cy), C.c_float(self.bf), _vp(P), _vp(A), C.byref(n_p), C.byref(n_a),
                        cap, _vp(t), _vp(R), _vp(pose), self.mono, _vp(outs[0]), _vp(outs[1]),
                        _vp(outs[2]), _vp(outs[3]), C.byref(n_out), C.byref(integrated))
        assert rc == 0
        self.points, self.ages = P[:n_p.value].copy(), A[:n_a.value].copy()
        self.translation, self.rotation, self.frame_pose = t, R, pose
        self.prev = cur
        self.trajectory.append(pose[:3].copy())
        k = n_out.value
        return dict(l0=outs[0][:k].copy(), r0=outs[1][:k].copy(), l1=outs[2][:k].copy(), r1=outs[3][:k].copy(),
                    integrated=bool(integrated.value), tvec=t.copy(), R=R.copy())


def ref_rotation_matrix_to_euler(R):
    R = np.ascontiguousarray(R, np.float64).reshape(3, 3)
    e = np.zeros(3, np.float32)
    ref_lib().ref_rotation_matrix_to_euler(_vp(R), _vp(e))
    return e


def ref_integrate_odometry_stereo(pose, R, t):
    pose = np.array(pose, np.float64).reshape(4, 4).copy()
    R = np.ascontiguousarray(R, np.float64).reshape(3, 3)
    t = np.ascontiguousarray(t, np.float64).reshape(3)
    ref_lib().ref_integrate_odometry_stereo(_vp(pose), _vp(R), _vp(t))
    return pose


def ref_calc_sequence_errors(poses_gt, poses_result, cap=100000):
    """the reference's own calcSequenceErrors (src/evaluate/evaluate_odometry.cpp:71-116, compiled where it lies):
    (k, 5) float32 rows (first_frame, r_err, t_err, len, speed)"""
    g = np.ascontiguousarray([np.asarray(T, np.float64)[:3].reshape(12) for T in poses_gt], np.float64)
    r = np.ascontiguousarray([np.asarray(T, np.float64)[:3].reshape(12) for T in poses_result], np.float64)
    assert len(g) == len(r)
    out = np.zeros((cap, 5), np.float32)
    k = ref_lib().ref_calc_sequence_errors(_vp(g), _vp(r), len(g), _vp(out), cap)
    assert k <= cap
    return out[:k].copy()
