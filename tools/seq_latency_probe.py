"""Where does a lock-step step with ONE sequence and host images spend its time?  Host time per call (push, step) and
the HIP-event stage times of the last steps.   python tools/seq_latency_probe.py [pinned|host|device]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visual_odom_amd import _lib, synth

mode = sys.argv[1] if len(sys.argv) > 1 else "host"
world = synth.StereoWorld(seed=20260925)
L, R, poses, _ = world.render_sequence(5)
P_l, P_r = world.proj_matrices()
order = [0, 1, 2, 3, 4, 3, 2, 1]
ctx = _lib.Context(0, world.w, world.h, 4096, 1)
ctx.seq_configure(1, world.w, world.h, 3, 400)
ctx.batch_set_projection(P_l, P_r)
if mode == "device":
    import ctypes as C
    hip = C.CDLL("libamdhip64.so.7")
    dev = []
    for k in range(5):
        pr = []
        for img in (L[k], R[k]):
            p = C.c_void_p()
            hip.hipMalloc(C.byref(p), C.c_size_t(img.size))
            hip.hipMemcpy(p, np.ascontiguousarray(img).ctypes.data_as(C.c_void_p), C.c_size_t(img.size), 1)
            pr.append(p.value)
        dev.append(pr)
tp, ts = [], []
n = 200
for i in range(n):
    k = order[i % 8]
    t0 = time.perf_counter()
    if mode == "device":
        ctx.seq_push_pair_dev(0, dev[k][0], dev[k][1], world.w)
    else:
        ctx.seq_push_pair(0, L[k], R[k], pinned=False)
    t1 = time.perf_counter()
    ctx.seq_step()
    t2 = time.perf_counter()
    tp.append(t1 - t0)
    ts.append(t2 - t1)
    if i == 20:
        ctx.seq_sync()
        tstart = time.perf_counter()
ctx.seq_sync()
total = time.perf_counter() - tstart
print("mode %s: %.3f ms per step over %d steps; host time per push %.3f ms (median), per step call %.3f ms (median), p95 %.3f / %.3f"
      % (mode, 1e3 * total / (n - 21), n - 21, 1e3 * np.median(tp[21:]), 1e3 * np.median(ts[21:]),
         1e3 * np.percentile(tp[21:], 95), 1e3 * np.percentile(ts[21:], 95)))
st = np.mean([ctx.batch_slot_times(i % _lib.EVENT_SLOTS) for i in range(n - 50, n)], axis=0)
print("stage ms:", {k: round(float(v), 3) for k, v in zip(_lib.STAGE_NAMES, st)})
ctx.close()
