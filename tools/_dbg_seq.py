import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from visual_odom_amd import _lib, synth, odometry
world = synth.StereoWorld(seed=20260925)
L, R, poses, _ = world.render_sequence(5)
P_l, P_r = world.proj_matrices()
order = [0, 1, 2, 3, 4, 3, 2, 1]
for n in (40, 200):
    for label, mk in (("class", lambda: odometry.MultiSequenceOdometry(P_l, P_r, 1, world.w, world.h, ring=3, max_steps=n + 32)),):
        vo = mk()
        for i in range(9):
            vo.push(0, L[order[i % 8]], R[order[i % 8]]); vo.step()
        vo.sync()
        t4 = time.perf_counter(); tp=[]; ts=[]
        for i in range(9, 9 + n):
            a=time.perf_counter(); vo.push(0, L[order[i % 8]], R[order[i % 8]]); b=time.perf_counter(); vo.step(); c=time.perf_counter()
            tp.append(b-a); ts.append(c-b)
        e=time.perf_counter(); vo.sync(); f=time.perf_counter()
        print(label, n, "ms/frame %.3f  loop %.3f  final sync %.3f ms; push med %.3f step med %.3f; first 10 push %s" % (1e3*(f-t4)/n, 1e3*(e-t4)/n, 1e3*(f-e), 1e3*np.median(tp), 1e3*np.median(ts), np.round(1e3*np.array(tp[:10]),2)))
        st = np.mean([vo.ctx.batch_slot_times(i % 256) for i in range(9+n-20, 9+n)], axis=0)
        print("   stage ms", {k: round(float(v),3) for k,v in zip(_lib.STAGE_NAMES, st)})
        vo.close()
