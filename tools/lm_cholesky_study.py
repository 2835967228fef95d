"""The gate of VERDICT r05 item 3 ("second, gated"): would an LL^T solve of CvLevMarq's 6 x 6 step system, instead of
cvSolve(..., CV_SVD), keep the refined pose within 1e-9 of the reference's and its Levenberg-Marquardt iteration counts
identical?  Measured on the CPU with the oracle's study switch (oracle/orc_pnp.c, orc_set_lm_solve_mode): the same planted
problems (tests/test_oracle_geom.py::planted_problem, every outlier rate and noise level the GPU parity tests use and more)
solved twice, reference solve vs Cholesky.

    python tools/lm_cholesky_study.py [N=10000] > profiles/r06_lm_cholesky_study.md

Build the Cholesky kernel only if: max |d pose| <= 1e-9 AND every problem has the same LM iteration count and solve count.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", "8")
import numpy as np  # noqa: E402

from oracle import oracle as orc  # noqa: E402
from test_oracle_geom import K_KITTI, planted_problem  # noqa: E402


def solves():
    a, b = C.c_longlong(0), C.c_longlong(0)
    orc.lib().orc_lm_solve_counts(C.byref(a), C.byref(b))
    return a.value, b.value


def main(n_total):
    orc.build()
    lib = orc.lib()
    rates = [0.0, 0.1, 0.2, 0.3, 0.45, 0.6, 0.75]
    noises = [0.0, 0.05, 0.15, 0.3]
    sizes = [6, 8, 12, 30, 100, 400, 2000]
    rng = np.random.default_rng(606)
    rows = []
    d_pose, d_iter, d_solve, fallbacks, total_solves = [], 0, 0, 0, 0
    worst = None
    for k in range(n_total):
        n = int(sizes[k % len(sizes)])
        out = float(rates[(k // len(sizes)) % len(rates)])
        noise = float(noises[(k // (len(sizes) * len(rates))) % len(noises)])
        X, uv, r, t, _ = planted_problem(orc, n, out, noise, 100000 + k)
        if k % 9 == 0:   # degenerate flavours: coplanar points / tiny baselines of depth / duplicated points
            X = X.copy()
            if k % 27 == 0:
                X[:, 2] = X[0, 2]
            elif k % 27 == 9:
                X[: n // 2] = X[0]
            uv = orc.project_points(X, r, t, K_KITTI).astype(np.float32) + rng.normal(0, noise, (n, 2)).astype(np.float32)
        lib.orc_set_lm_solve_mode(0)
        s0 = solves()
        rc0, rv0, tv0, inl0, dbg0 = orc.solve_pnp_ransac(X, uv, K_KITTI)
        s1 = solves()
        lib.orc_set_lm_solve_mode(1)
        rc1, rv1, tv1, inl1, dbg1 = orc.solve_pnp_ransac(X, uv, K_KITTI)
        s2 = solves()
        lib.orc_set_lm_solve_mode(0)
        assert rc0 == rc1 and np.array_equal(inl0, inl1)   # (RANSAC runs before the refinement: cannot differ)
        if rc0 != 1:
            continue
        d = max(np.abs(rv0 - rv1).max(), np.abs(tv0 - tv1).max())
        if not np.isfinite(d):
            d = 0.0 if np.array_equal(np.isnan(rv0), np.isnan(rv1)) else np.inf
        d_pose.append(d)
        di = int(dbg0[3]) != int(dbg1[3])
        ds = (s1[0] - s0[0]) != (s2[0] - s1[0])
        d_iter += di
        d_solve += ds
        fallbacks += s2[1] - s1[1]
        total_solves += s2[0] - s1[0]
        if worst is None or d > worst[0]:
            worst = (d, k, n, out, noise, int(dbg0[3]), int(dbg1[3]), len(inl0))
        rows.append((n, out, noise, d, di or ds))
    d_pose = np.array(d_pose)
    print("# Levenberg-Marquardt step: LL^T against cvSolve(CV_SVD) -- the gate of VERDICT r05 item 3\n")
    print("`python tools/lm_cholesky_study.py %d` (CPU, the oracle's study switch; reference solve = Jacobi SVD pseudo-inverse).\n" % n_total)
    print("* problems with a model: %d of %d; LM solves in the Cholesky runs: %d, of them SVD fallbacks (pivot not safely positive): %d" % (
        len(d_pose), n_total, total_solves, fallbacks))
    print("* max |d pose| = %.3e, 99.9 %% %.3e, 99 %% %.3e, median %.3e" % (d_pose.max(), np.quantile(d_pose, 0.999), np.quantile(d_pose, 0.99),
                                                                          np.median(d_pose)))
    print("* problems whose LM iteration count differs: %d; whose number of 6 x 6 solves differs: %d" % (d_iter, d_solve))
    print("* worst: |d| %.3e at problem %d (n %d, outliers %.2f, noise %.2f, LM iterations %d vs %d, %d inliers)\n" % worst)
    print("| |d pose| | problems |\n|---|---|")
    edges = [0, 1e-15, 1e-14, 1e-13, 1e-12, 1e-11, 1e-10, 1e-9, 1e-8, 1e-6, 1e-3, np.inf]
    for lo, hi in zip(edges[:-1], edges[1:]):
        print("| [%g, %g) | %d |" % (lo, hi, int(((d_pose >= lo) & (d_pose < hi)).sum())))
    print("\n| n | problems | max d pose | control flow differs |\n|---|---|---|---|")
    for n in sizes:
        sel = [r for r in rows if r[0] == n]
        print("| %d | %d | %.2e | %d |" % (n, len(sel), max(r[3] for r in sel), sum(r[4] for r in sel)))
    gate = d_pose.max() <= 1e-9 and d_iter == 0 and d_solve == 0
    print("\n**Gate (max |d pose| <= 1e-9 and identical LM iteration / solve counts on every problem): %s**" % ("PASS" if gate else "FAIL"))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 10000)
