"""What would a Lucas-Kanade kernel that tracks TWO features per wavefront in lock step pay?

VERDICT r01 asked for the 2-features-per-wave variant to be built and measured rather than estimated.  Its gain is
bounded by two measurable things: (i) the issue cost of the per-iteration instructions (tools/ubench/valu_rate.hip,
profiles/r02_valu_issue_cost.txt), which says how much of an iteration is per-lane pixel work (doubles when a lane owns
two window segments) and how much is per-feature work every lane repeats (shared by the two features for free), and
(ii) the lock-step penalty: the pair iterates max(i1, i2) times at every level instead of i1 and i2.  This tool
measures (ii) exactly on the benchmark's own frames with the oracle's per-(level, point) iteration log, for pairs of
neighbouring features of the list, and combines it with (i).

    python tools/lk_pairing_study.py [n_frames]
"""
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
    tot_single = tot_pair = tot_pair_sorted = 0
    lvl_solves = lvl_solves_pair = 0
    hist = np.zeros(32, np.int64)
    for k in range(n_frames):
        p = pts[k]
        imgs = [(lefts[k], rights[k]), (rights[k], rights[k + 1]), (rights[k + 1], lefts[k + 1]), (lefts[k + 1], lefts[k])]
        alive = np.ones(len(p), bool)
        for a, b in imgs:
            n = len(p)
            log = np.zeros((max_level + 1, n), np.int32)
            lib.orc_lk_set_iteration_log(log.ctypes.data_as(C.c_void_p), n)
            q, st, _ = orc.calc_optical_flow_pyr_lk(a, b, p, max_level=max_level, nthreads=8)
            lib.orc_lk_set_iteration_log(None, 0)
            it = log[:, alive]                      # features already rejected by an earlier hop have retired
            np.add.at(hist, np.minimum(it.reshape(-1), 31), 1)
            m = it.shape[1] // 2 * 2
            tot_single += int(it[:, :m].sum())
            tot_pair += int(np.maximum(it[:, 0:m:2], it[:, 1:m:2]).sum()) * 2
            lvl_solves += int((it[:, :m] > 0).sum())
            lvl_solves_pair += int(((it[:, 0:m:2] > 0) | (it[:, 1:m:2] > 0)).sum()) * 2
            # best case for pairing: partners with similar totals (sorted by their iteration sum at this hop)
            order = np.argsort(it.sum(0))[:m]
            its = it[:, order]
            tot_pair_sorted += int(np.maximum(its[:, 0::2], its[:, 1::2]).sum()) * 2
            alive &= (st == 1) & (q >= 0).all(1)
            p = q
    print("level-solves: %d, iterations: %d (%.2f per level-solve)" % (lvl_solves, tot_single, tot_single / max(lvl_solves, 1)))
    print("iteration histogram (per level-solve):", {i: int(v) for i, v in enumerate(hist) if v})
    r = tot_pair / tot_single
    rs = tot_pair_sorted / tot_single
    print("lock-step pairs of list neighbours execute %.3f x the iterations (pair-slots x 2 / single iterations)" % r)
    print("lock-step pairs of iteration-sorted partners (oracle knowledge, unattainable): %.3f x" % rs)
    # issue-cost model of one iteration (profiles/r02_valu_issue_cost.txt classes; counts from the kernel ISA):
    #   per-lane pixel work  : blend 14 dot2 + repack 8 + residual dots 8           = 30 x 4.4  = 132 cycles
    #   reduction tree       : 2 half-swaps 16.8 + 4 dpp adds 17.6 + adds / split 15 ~ 50 cycles
    #   per-feature work every lane repeats: weights 54, reduction tail 23, solve + update 31, tests 48 = 156 cycles
    pixel, tree, uniform = 132.0, 50.0, 156.0
    single = pixel + tree + uniform
    paired = (2 * pixel + tree + 10 + uniform + 30) / 2.0   # per feature: + wider tree, + per-half broadcasts
    print("issue cost per feature-iteration: one feature per wave %.0f cycles; two per wave %.0f cycles (x%.2f)" % (
        single, paired, paired / single))
    print("=> iteration part of the kernel (63 %% of its time): x%.3f with list neighbours, x%.3f with perfect partners" % (
        paired / single * r, paired / single * rs))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
