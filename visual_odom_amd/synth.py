"""Seeded procedural stereo sequences (KITTI-00 shaped by default).

There is no KITTI data (and no network) in the build or GPU environment, so the tests and
bench.py run on a rendered "street canyon": textured ground plane, two side walls of finite
height, sky.  Every image is ray-cast independently from the world model, so stereo and temporal
geometry (incl. occlusion) is exact and ground-truth depth / flow / pose is available.

Intrinsics default to calibration/kitti00.yaml of the reference (fx=fy=718.856, cx=607.1928,
cy=185.2157, bf=-386.1448 -> baseline 0.5372 m; reference main.cpp:64-74).
"""
import numpy as np

KITTI_W, KITTI_H = 1241, 376
KITTI_FX, KITTI_CX, KITTI_CY, KITTI_BF = 718.856, 607.1928, 185.2157, -386.1448


def proj_matrices(fx=KITTI_FX, fy=None, cx=KITTI_CX, cy=KITTI_CY, bf=KITTI_BF):
    """projMatrl / projMatrr exactly as main.cpp:73-74 builds them (3x4 float32)."""
    fy = fx if fy is None else fy
    P_l = np.array([[fx, 0, cx, 0], [0, fy, cy, 0], [0, 0, 1, 0]], np.float32)
    P_r = np.array([[fx, 0, cx, bf], [0, fy, cy, 0], [0, 0, 1, 0]], np.float32)
    return P_l, P_r


def _octave_noise(rng, size, octaves):
    """band-limited noise: sum of bilinearly upsampled uniform noise grids"""
    out = np.zeros((size, size), np.float32)
    amp_sum = 0.0
    for cell, amp in octaves:
        g = size // cell
        grid = rng.random((g + 1, g + 1), dtype=np.float32)
        grid[-1, :] = grid[0, :]
        grid[:, -1] = grid[:, 0]
        idx = np.arange(size, dtype=np.float32) / cell
        i0 = np.floor(idx).astype(np.int32)
        f = idx - i0
        f = f * f * (3 - 2 * f)
        a = grid[i0][:, i0]
        b = grid[i0][:, i0 + 1]
        c = grid[i0 + 1][:, i0]
        d = grid[i0 + 1][:, i0 + 1]
        fx_, fy_ = f[None, :], f[:, None]
        out += amp * ((a * (1 - fx_) + b * fx_) * (1 - fy_) + (c * (1 - fx_) + d * fx_) * fy_)
        amp_sum += amp
    return out / amp_sum


def _make_texture(rng, size=2048):
    """multi-octave noise + high-contrast rectangular blobs (corners for FAST / LK)"""
    tex = _octave_noise(rng, size, [(128, 1.0), (32, 0.8), (8, 0.6), (4, 0.35)])
    tex = (tex - tex.min()) / (tex.max() - tex.min())
    tex = 40 + 175 * tex
    nblob = size * size // 900
    xs = rng.integers(0, size - 16, nblob)
    ys = rng.integers(0, size - 16, nblob)
    ws = rng.integers(3, 14, nblob)
    hs = rng.integers(3, 14, nblob)
    vals = np.where(rng.random(nblob) < 0.5, rng.uniform(5, 60, nblob), rng.uniform(190, 250, nblob))
    for x, y, w_, h_, v in zip(xs, ys, ws, hs, vals):
        tex[y:y + h_, x:x + w_] = v
    return tex.astype(np.float32)


def _mips(tex, levels=5):
    out = [tex]
    for _ in range(levels - 1):
        t = out[-1]
        out.append(0.25 * (t[0::2, 0::2] + t[1::2, 0::2] + t[0::2, 1::2] + t[1::2, 1::2]))
    return out


def _sample_wrap_bilinear(tex, u, v):
    s = tex.shape[0]
    u = np.mod(u, s)
    v = np.mod(v, s)
    u0 = np.floor(u).astype(np.int32)
    v0 = np.floor(v).astype(np.int32)
    fu = (u - u0).astype(np.float32)
    fv = (v - v0).astype(np.float32)
    u0 %= s
    v0 %= s
    u1 = (u0 + 1) % s
    v1 = (v0 + 1) % s
    return ((tex[v0, u0] * (1 - fu) + tex[v0, u1] * fu) * (1 - fv) +
            (tex[v1, u0] * (1 - fu) + tex[v1, u1] * fu) * fv)


class StereoWorld:
    """Street canyon world + pin-hole stereo rig.  Camera frame: x right, y down, z forward."""

    def __init__(self, seed=20260925, width=KITTI_W, height=KITTI_H, fx=KITTI_FX, cx=KITTI_CX,
                 cy=KITTI_CY, bf=KITTI_BF, texel=0.02, tex_size=2048):
        self.w, self.h = int(width), int(height)
        self.fx = self.fy = float(fx)
        self.cx, self.cy = float(cx), float(cy)
        self.bf = float(bf)
        self.baseline = -self.bf / self.fx
        self.texel = texel
        self.seed = seed
        rng = np.random.default_rng(seed)
        # planes: ground y=+1.65, left wall x=-6.5, right wall x=+7.5 (height 9 m)
        self.ground_y, self.left_x, self.right_x, self.wall_top = 1.65, -6.5, 7.5, -7.5
        self.tex = [_mips(_make_texture(rng, tex_size)) for _ in range(3)]
        self._rng_motion = np.random.default_rng(seed + 1)
        self._rng_noise_seed = seed + 2
        xs = (np.arange(self.w, dtype=np.float64) - self.cx) / self.fx
        ys = (np.arange(self.h, dtype=np.float64) - self.cy) / self.fy
        self._dx, self._dy = np.meshgrid(xs, ys)

    def K(self):
        return np.array([[self.fx, 0, self.cx], [0, self.fy, self.cy], [0, 0, 1]], np.float32)

    def proj_matrices(self):
        return proj_matrices(self.fx, self.fy, self.cx, self.cy, self.bf)

    # ---- trajectory -------------------------------------------------------------------
    def poses(self, n_frames, step=(0.6, 1.1), yaw_amp=0.04, yaw_period=80.0):
        """world-from-camera 4x4 poses of the LEFT camera for frames 0..n_frames-1"""
        rng = np.random.default_rng(self.seed + 1)
        poses = []
        pos = np.zeros(3)
        for k in range(n_frames):
            yaw = yaw_amp * np.sin(2 * np.pi * k / yaw_period) + rng.normal(0, 0.002)
            pitch, roll = rng.normal(0, 0.002, 2)
            cy_, sy_ = np.cos(yaw), np.sin(yaw)
            cp, sp = np.cos(pitch), np.sin(pitch)
            cr, sr = np.cos(roll), np.sin(roll)
            Ry = np.array([[cy_, 0, sy_], [0, 1, 0], [-sy_, 0, cy_]])
            Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
            Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
            R = Ry @ Rx @ Rz
            T = np.eye(4)
            T[:3, :3] = R
            T[:3, 3] = pos
            poses.append(T)
            pos = pos + R @ np.array([0, 0, rng.uniform(*step)])
        return poses

    # ---- rendering ---------------------------------------------------------------------
    def _raycast(self, T_wc, right=False):
        R, c = T_wc[:3, :3], T_wc[:3, 3].copy()
        if right:
            c = c + R @ np.array([self.baseline, 0, 0])
        d = np.stack([self._dx, self._dy, np.ones_like(self._dx)], -1) @ R.T  # world ray dirs
        big = 1e9
        with np.errstate(divide="ignore", invalid="ignore"):
            tg = (self.ground_y - c[1]) / d[..., 1]
            tl = (self.left_x - c[0]) / d[..., 0]
            tr = (self.right_x - c[0]) / d[..., 0]
        tg = np.where((tg > 0) & np.isfinite(tg), tg, big)
        yl = c[1] + tl * d[..., 1]
        tl = np.where((tl > 0) & np.isfinite(tl) & (yl > self.wall_top) & (yl < self.ground_y), tl, big)
        yr = c[1] + tr * d[..., 1]
        tr = np.where((tr > 0) & np.isfinite(tr) & (yr > self.wall_top) & (yr < self.ground_y), tr, big)
        t = np.minimum(tg, np.minimum(tl, tr))
        plane = np.where(t >= big, -1, np.where(t == tg, 0, np.where(t == tl, 1, 2)))
        t = np.where(t > 150.0, big, t)
        plane = np.where(t >= big, -1, plane)
        P = c + t[..., None] * d
        return t, plane, P

    def render(self, T_wc, right=False, noise_id=0):
        """returns (uint8 image HxW, depth-along-z HxW float32 with inf for sky)"""
        t, plane, P = self._raycast(T_wc, right)
        img = np.full((self.h, self.w), 128.0, np.float32)
        foot = np.clip(t / self.fx / self.texel, 1.0, 15.9)
        lvl = np.log2(foot)
        for pid in range(3):
            m = plane == pid
            if not m.any():
                continue
            Pm = P[m]
            if pid == 0:
                u, v = Pm[:, 0], Pm[:, 2]
            else:
                u, v = Pm[:, 2], Pm[:, 1]
            u = u / self.texel
            v = v / self.texel
            l = lvl[m]
            l0 = np.floor(l).astype(np.int32)
            fl = (l - l0).astype(np.float32)
            val = np.zeros(len(u), np.float32)
            for L in range(len(self.tex[pid])):
                for which, wgt in ((L, 1 - fl), (L - 1, fl)):
                    sel = l0 == which
                    if which < 0 or not sel.any():
                        continue
                    s = 2.0 ** L
                    val[sel] += wgt[sel] * _sample_wrap_bilinear(self.tex[pid][L],
                                                                  u[sel] / s - 0.5 + 0.5 / s,
                                                                  v[sel] / s - 0.5 + 0.5 / s)
            img[m] = val
        rng = np.random.default_rng([self._rng_noise_seed, noise_id, int(right)])
        img = img + rng.normal(0, 1.0, img.shape).astype(np.float32)
        depth = np.where(plane >= 0, t, np.inf).astype(np.float32)  # ray param == z in camera frame
        return np.clip(np.rint(img), 0, 255).astype(np.uint8), depth

    def render_sequence(self, n_frames, **pose_kw):
        """list of (left, right) uint8 images + poses + left depth maps"""
        poses = self.poses(n_frames, **pose_kw)
        lefts, rights, depths = [], [], []
        for k, T in enumerate(poses):
            l, d = self.render(T, False, k)
            r, _ = self.render(T, True, k)
            lefts.append(l)
            rights.append(r)
            depths.append(d)
        return lefts, rights, poses, depths


def relative_pose(T_w0, T_w1):
    """[R|t] mapping t0-left-camera coordinates to t1-left-camera coordinates (quirk B8)"""
    T = np.linalg.inv(T_w1) @ T_w0
    return T[:3, :3], T[:3, 3]


def select_keypoints(img, bucket=37, per_bucket=6, border=12, min_dist=5):
    """Bucketed Shi-Tomasi keypoints (numpy): input generator for boundary-injected point sets
    (N ~ 2000 on 1241x376 with bucket=rows/10, per_bucket=6).  Returns
    integer-valued float32 (x, y) in row-major bucket order."""
    f = img.astype(np.float32)
    gx = np.zeros_like(f)
    gy = np.zeros_like(f)
    gx[:, 1:-1] = f[:, 2:] - f[:, :-2]
    gy[1:-1, :] = f[2:, :] - f[:-2, :]

    def box(a, r=3):
        c = np.cumsum(np.cumsum(np.pad(a, ((r + 1, r), (r + 1, r))), 0), 1)
        k = 2 * r + 1
        return c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k]

    a, b, c = box(gx * gx), box(gx * gy), box(gy * gy)
    resp = 0.5 * (a + c - np.sqrt((a - c) ** 2 + 4 * b * b))
    h, w = img.shape
    resp[:border] = 0
    resp[-border:] = 0
    resp[:, :border] = 0
    resp[:, -border:] = 0
    pts = []
    for by in range(0, h, bucket):
        for bx in range(0, w, bucket):
            cell = resp[by:by + bucket, bx:bx + bucket]
            if cell.size == 0:
                continue
            order = np.argsort(cell, axis=None)[::-1]
            chosen = []
            for o in order[:200]:
                y, x = divmod(int(o), cell.shape[1])
                if cell[y, x] <= 1.0:
                    break
                if all(abs(x - cx_) >= min_dist or abs(y - cy_) >= min_dist for cx_, cy_ in chosen):
                    chosen.append((x, y))
                    if len(chosen) >= per_bucket:
                        break
            pts.extend((bx + x, by + y) for x, y in chosen)
    return np.array(pts, np.float32).reshape(-1, 2)
