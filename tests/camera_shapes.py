"""Inputs for the parity tests at the camera shapes the reference ships besides KITTI, and a real stereo photograph.

* `CALIBRATIONS`: the literals of reference calibration/zed.yaml:8-14 (ZED, 1280 x 720; the file's width / height
  entries are swapped, cx = 689.9 says which is which) and calibration/rgbd.yaml:8-14 (640 x 480, the sensor of
  src/rgbd_standalone.cpp:75-76,184-196).  bucket_size = rows / 10 (visualOdometry.cpp:106) gives a 72-pixel grid of
  11 x 18 cells at 720 rows and a 48-pixel grid of 11 x 14 cells at 480 rows -- a different aliasing of the bucket grid
  against the image than KITTI's 37-pixel grid (feature.cpp:212-236).
* `real_stereo_pair()`: the Middlebury-2014 "Motorcycle" pair that scikit-image installs with the base image of this
  container and of the GPU box (skimage/data/motorcycle_{left,right}.png, 741 x 500), converted to 8-bit gray with the
  integer BT.601 weights.  A photograph has what the procedural street canyon of synth.py has not: saturated and flat
  regions (min-eigenvalue rejections, status 0), specular highlights, repetitive texture, real occlusions and a
  disparity range of ~10-70 pixels.  The file is read where it lies; nothing of it is committed.
* `warp_subpixel()`: the "t1" pair of a photograph: the same images resampled after a sub-pixel shift and a slight
  zoom about the principal point (bilinear, rounded to u8) -- forward motion as far as a single view can fake it.
"""
import os

import numpy as np

CALIBRATIONS = {
    # calibration/zed.yaml:8-14,21
    "zed": dict(width=1280, height=720, fx=684.367919921875, cx=689.889404296875, cy=406.87420654296875,
                bf=-82.12415128946304),
    # calibration/rgbd.yaml:8-14,21
    "rgbd": dict(width=640, height=480, fx=581.367919921875, cx=343.889404296875, cy=203.87420654296875,
                 bf=-28.12415128946304),
}

_SKIMAGE_DATA = ["/opt/conda/lib/python3.9/site-packages/skimage/data"]


def world(name, seed):
    from visual_odom_amd import synth
    return synth.StereoWorld(seed=seed, **CALIBRATIONS[name])


def skimage_data_dir():
    for d in _SKIMAGE_DATA:
        if os.path.exists(os.path.join(d, "motorcycle_left.png")):
            return d
    import glob
    for d in glob.glob("/opt/conda/lib/python3*/site-packages/skimage/data") + \
            glob.glob("/usr/lib/python3*/site-packages/skimage/data"):
        if os.path.exists(os.path.join(d, "motorcycle_left.png")):
            return d
    return None


def to_gray(rgb):
    """8-bit gray by the integer BT.601 weights cv::cvtColor(BGR2GRAY) uses (utils.cpp:179):
    (R * 4899 + G * 9617 + B * 1868 + 8192) >> 14"""
    rgb = np.asarray(rgb)[..., :3].astype(np.int64)
    return ((rgb[..., 0] * 4899 + rgb[..., 1] * 9617 + rgb[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)


def real_stereo_pair():
    """(left, right) uint8 gray 500 x 741, or None when scikit-image's sample data is not installed"""
    d = skimage_data_dir()
    if d is None:
        return None
    from PIL import Image
    out = []
    for side in ("left", "right"):
        with Image.open(os.path.join(d, "motorcycle_%s.png" % side)) as im:
            out.append(np.ascontiguousarray(to_gray(np.asarray(im.convert("RGB")))))
    return out[0], out[1]


# Motorcycle calibration (Middlebury 2014 calib.txt, full resolution 2964 x 1988: f = 3979.911, cx = 1244.772,
# cy = 1019.507, baseline 193.001 mm) scaled to the 741 x 500 copy; the two cameras' cx offset is ignored -- the test
# compares the device with the checker on the same numbers, not with the scene.
REAL_CALIB = dict(fx=3979.911 / 4.0, cx=1244.772 / 4.0, cy=1019.507 / 4.0, bf=-3979.911 / 4.0 * 0.193001)


def warp_subpixel(img, dx, dy, zoom=1.0, center=None):
    """out(x, y) = img(cx + (x - cx) / zoom + dx, cy + (y - cy) / zoom + dy), bilinear, border replicated"""
    h, w = img.shape
    cx, cy = (w / 2.0, h / 2.0) if center is None else center
    xs = cx + (np.arange(w, dtype=np.float64) - cx) / zoom + dx
    ys = cy + (np.arange(h, dtype=np.float64) - cy) / zoom + dy
    xs = np.clip(xs, 0, w - 1.001)
    ys = np.clip(ys, 0, h - 1.001)
    x0 = np.floor(xs).astype(np.int64)
    y0 = np.floor(ys).astype(np.int64)
    fx = (xs - x0)[None, :]
    fy = (ys - y0)[:, None]
    f = img.astype(np.float64)
    a = f[y0][:, x0]
    b = f[y0][:, x0 + 1]
    c = f[y0 + 1][:, x0]
    d = f[y0 + 1][:, x0 + 1]
    out = (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def real_quadruple():
    """(l0, r0, l1, r1) from the photograph, or None"""
    pair = real_stereo_pair()
    if pair is None:
        return None
    l0, r0 = pair
    c = (REAL_CALIB["cx"], REAL_CALIB["cy"])
    return l0, r0, warp_subpixel(l0, 2.3, -1.6, 1.012, c), warp_subpixel(r0, 2.3, -1.6, 1.012, c)
