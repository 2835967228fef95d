"""Do the Gauss-Newton orbits of non-converging features repeat EXACTLY?  6 % of the level-solves of the bench frames run
into the cap of 30 iterations and account for 24 % of all iterations.  If the window corner after iteration j is
bit-identical to the one after iteration j - p, everything that follows is periodic (the arithmetic is deterministic and
the state is the corner plus the previous delta), so the remaining iterations could be skipped and the final position
read off the cycle -- bit-exactly.  This tool measures how often that happens and how many iterations it would save.

    python tools/lk_cycle_study.py [n_frames]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(n_frames=3):
    from oracle import oracle as orc
    import bench
    world, lefts, rights, pts, max_level = bench.build_inputs("kitti2000", n_frames, 20260925)
    lib = orc.lib()
    total_it = capped_it = saved = 0
    n_capped = n_cyc = 0
    periods = {}
    first = []
    for k in range(n_frames):
        p = pts[k]
        for a, b in ((lefts[k], rights[k]), (rights[k], rights[k + 1]), (rights[k + 1], lefts[k + 1]), (lefts[k + 1], lefts[k])):
            n = len(p)
            it = np.zeros((max_level + 1, n), np.int32)
            cyc = np.zeros((max_level + 1, n, 2), np.int32)
            lib.orc_lk_set_iteration_log(it.ctypes.data_as(C.c_void_p), n)
            lib.orc_lk_set_cycle_log(cyc.ctypes.data_as(C.c_void_p))
            p, st, _ = orc.calc_optical_flow_pyr_lk(a, b, p, max_level=max_level, nthreads=8)
            lib.orc_lk_set_cycle_log(None)
            lib.orc_lk_set_iteration_log(None, 0)
            total_it += int(it.sum())
            cap = it == 30
            n_capped += int(cap.sum())
            capped_it += 30 * int(cap.sum())
            has = cap & (cyc[..., 1] > 0)
            n_cyc += int(has.sum())
            # once the orbit is known to be periodic at iteration j (0-based) the iterations j + 1 .. 29 need not run
            saved += int((29 - cyc[..., 0][has]).sum())
            for pp in cyc[..., 1][has]:
                periods[int(pp)] = periods.get(int(pp), 0) + 1
            first += list(cyc[..., 0][has])
    print("iterations %d; level-solves at the cap of 30: %d (%.1f %% of the iterations)" % (total_it, n_capped, 100.0 * capped_it / total_it))
    print("capped level-solves whose orbit repeats exactly (period <= 12): %d of %d; periods %s" % (n_cyc, n_capped, dict(sorted(periods.items()))))
    if first:
        print("first exact repeat at iteration: median %d, p90 %d" % (np.median(first), np.percentile(first, 90)))
    print("iterations an exact-cycle shortcut would skip: %d = %.2f %% of all iterations" % (saved, 100.0 * saved / total_it))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
