"""f3: the KITTI segment errors (trajectoryDistances / calcSequenceErrors, reference src/evaluate/evaluate_odometry.cpp:35-116)
restated in visual_odom_amd.odometry, held to the reference's OWN evaluator compiled where it lies (oracle/_ref)."""
import numpy as np
import pytest


def _random_walk(n, seed, drift=0.0):
    rng = np.random.default_rng(seed)
    T = np.eye(4)
    out = [T.copy()]
    for k in range(1, n):
        yaw = 0.01 * np.sin(k / 37.0) + rng.normal(0, 0.002) + drift * 1e-3
        pitch = rng.normal(0, 0.001)
        cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        D = np.eye(4)
        D[:3, :3] = Ry @ Rx
        D[:3, 3] = [rng.normal(0, 0.01), rng.normal(0, 0.01), rng.uniform(0.6, 1.1) * (1.0 + drift * 0.01)]
        T = T @ D
        out.append(T.copy())
    return out


def test_segment_errors_match_the_reference_evaluator(orc):
    from visual_odom_amd import odometry
    if orc.ref_lib() is None:
        pytest.skip("oracle/_ref needs /root/reference (or the prebuilt library shipped with the snapshot)")
    gt = _random_walk(1500, 1)
    res = _random_walk(1500, 1, drift=1.0)
    ref = orc.ref_calc_sequence_errors(gt, res)
    got = np.asarray(odometry.calc_sequence_errors(gt, res), np.float64)
    assert len(ref) == len(got) > 200
    assert np.array_equal(got[:, 0], ref[:, 0]) and np.array_equal(got[:, 3], ref[:, 3])        # same segments
    assert np.allclose(got[:, 4], ref[:, 4], rtol=1e-6)                                           # speed
    # libviso2's Matrix::inv is an LU in double; numpy's differs in the last bits, and the errors are stored as float
    assert np.allclose(got[:, 1], ref[:, 1], rtol=2e-4, atol=1e-9) and np.allclose(got[:, 2], ref[:, 2], rtol=2e-4, atol=1e-9)
    d = np.asarray(odometry.trajectory_distances(gt))
    assert d.dtype == np.float32 and d[-1] > 800
    s = odometry.sequence_error_summary(gt, res)
    assert s["segments"] == len(ref) and abs(s["t_err_percent"] - 100.0 * ref[:, 2].mean()) < 1e-3


def test_segment_errors_edge_cases():
    from visual_odom_amd import odometry
    short = _random_walk(50, 2)            # < 100 m: no segment fits
    assert odometry.calc_sequence_errors(short, short) == [] and odometry.sequence_error_summary(short, short) is None
    gt = _random_walk(400, 3)
    e = np.asarray(odometry.calc_sequence_errors(gt, gt))
    assert len(e) > 0 and np.abs(e[:, 1:3]).max() < 1e-6    # identical trajectories: zero error
    assert odometry.calc_sequence_errors(gt, [T[:3] for T in gt]) == odometry.calc_sequence_errors(gt, gt)  # 3x4 accepted
