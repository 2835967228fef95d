"""Run ONE stage subset on tiny inputs with a watchdog: python tools/dev_stage.py <stages-bitmask> [n_pts]"""
import sys, os, faulthandler, functools
print = functools.partial(print, flush=True)
faulthandler.dump_traceback_later(45, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visual_odom_amd import _lib
stages = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rng = np.random.default_rng(0)
w, h = 640, 200
base = rng.integers(0, 255, (h // 8 + 2, w // 8 + 2)).astype(np.float32)
img = np.kron(base, np.ones((8, 8), np.float32))[:h, :w]
imgs = [np.clip(np.roll(img, s, 1) + rng.normal(0, 2, img.shape), 0, 255).astype(np.uint8) for s in (0, 3, 1, 4)]
print("create")
ctx = _lib.Context(0, w, h, 1024, 1)
print("configure")
ctx.batch_configure(4, w, h, 1)
for i in range(4):
    ctx.batch_upload_image(i, imgs[i])
ctx.batch_set_quads([[0, 1, 2, 3]])
pts = np.stack([rng.uniform(30, w - 30, n), rng.uniform(30, h - 30, n)], 1).astype(np.float32)
ctx.batch_set_points(0, pts)
P = np.array([[500, 0, 320, 0], [0, 500, 100, 0], [0, 0, 1, 0]], np.float32); Pr = P.copy(); Pr[0, 3] = -250
ctx.batch_set_projection(P, Pr)
for bit, name in ((1, "pyramid"), (2, "lk"), (4, "filter"), (8, "tri"), (16, "pnp")):
    if stages & bit:
        print("run", name)
        ctx.batch_run(bit); ctx.batch_sync()
        print("  ok", name)
print("tracks", ctx.batch_get_tracks(0, n)["status4"].sum(1))
f = ctx.batch_get_filtered(0); print("filtered", len(f["l0"]), len(f["keep_idx_circ"]))
print("pose", ctx.batch_get_pose(0))
print("STAGE-DONE")
