"""KITTI odometry sequence (image_0/%06d.png, image_1/%06d.png -- what the reference's loadImageLeft /
loadImageRight read, utils.cpp:172-190) -> binary PGM files for examples/vo_run.cpp, which has no PNG
decoder.  KITTI's gray PNGs are R = G = B, so the reference's IMREAD_COLOR + BGR2GRAY is the identity.

    python tools/kitti_to_pgm.py <kitti>/sequences/00 <out_dir> [n_frames]
    examples/vo_run <out_dir> 718.856 607.1928 185.2157 -386.1448 <n_frames> poses.txt      # calibration/kitti00.yaml
"""
import os
import sys

import numpy as np
from PIL import Image


def main(src, dst, n=None):
    for cam in (0, 1):
        d_in, d_out = os.path.join(src, "image_%d" % cam), os.path.join(dst, "image_%d" % cam)
        os.makedirs(d_out, exist_ok=True)
        names = sorted(f for f in os.listdir(d_in) if f.endswith(".png"))
        for k, name in enumerate(names[:n]):
            img = np.asarray(Image.open(os.path.join(d_in, name)).convert("L"), np.uint8)
            with open(os.path.join(d_out, "%06d.pgm" % k), "wb") as f:
                f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]) + img.tobytes())
    print("wrote", dst)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else None)
