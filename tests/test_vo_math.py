"""csrc/vo_math.h -- the cube root, acos and cos the device's P3P cubic uses instead of the platform's libm (so that gfx950
and the host build of the same header compute the same bits): accuracy against numpy's extended-precision functions over the
ranges the solver asks for, and the edge cases of the library text they replace (pow(x, 1/3.) of a negative base is NaN)."""
import ctypes as C

import numpy as np


def _run(hc, what, x):
    x = np.ascontiguousarray(x, np.float64)
    y = np.zeros_like(x)
    hc.hc_math(what, x.ctypes.data_as(C.c_void_p), len(x), y.ctypes.data_as(C.c_void_p))
    return y


def _ulps(y, ref):
    return np.abs(y.astype(np.longdouble) - ref) / np.spacing(np.abs(ref.astype(np.float64))).astype(np.longdouble)


def test_cbrt_within_one_ulp(host_check):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(0, 10, 200000), 10 ** rng.uniform(-300, 300, 200000), [1e-310, 5e-324]])
    y = _run(host_check, 0, x)
    err = _ulps(y, np.cbrt(x.astype(np.longdouble)))
    assert float(err.max()) < 1.0, float(err.max())
    edge = _run(host_check, 0, [0.0, -1.0, np.inf, np.nan, 8.0, 27.0, 1e-300 ** 3 if False else 1e-300])
    assert edge[0] == 0 and np.isnan(edge[1]) and np.isinf(edge[2]) and np.isnan(edge[3]) and edge[4] == 2.0 and edge[5] == 3.0
    # like pow(x, 1 / 3.) it agrees with glibc's correctly rounded cbrt on most arguments (the rest: one ulp)
    assert np.mean(y == np.cbrt(x)) > 0.9


def test_acos_within_one_ulp(host_check):
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-1, 1, 400000), 1 - 10 ** rng.uniform(-16, 0, 50000), -1 + 10 ** rng.uniform(-16, 0, 50000),
                        [0.0, 0.5, -0.5, 1e-20, -1e-20]])
    y = _run(host_check, 1, x)
    err = _ulps(y, np.arccos(x.astype(np.longdouble)))
    err = err[np.isfinite(err)]
    assert float(err.max()) < 1.0, float(err.max())
    edge = _run(host_check, 1, [1.0, -1.0, 1.0000001, -2.0, np.nan])
    assert edge[0] == 0 and edge[1] == np.pi and np.all(np.isnan(edge[2:]))
    assert np.mean(y == np.arccos(x)) > 0.9


def test_cos_within_one_ulp(host_check):
    rng = np.random.default_rng(3)
    near = np.pi / 2 * np.arange(1, 64) + rng.uniform(-1e-9, 1e-9, 63)  # next to the multiples of pi / 2 (cancellation)
    x = np.concatenate([rng.uniform(0, 5.3, 400000), rng.uniform(-1000, 1000, 100000), near, [0.0, np.pi / 2, np.pi, 2 * np.pi / 3]])
    y = _run(host_check, 2, x)
    err = _ulps(y, np.cos(x.astype(np.longdouble)))
    assert float(err.max()) < 1.0, float(err.max())
    big = _run(host_check, 2, [np.inf, np.nan, 1e9])  # beyond the two-piece reduction: the platform's cos
    assert np.isnan(big[0]) and np.isnan(big[1]) and big[2] == np.cos(1e9)
    assert np.mean(y == np.cos(x)) > 0.9


def test_sin_within_one_ulp(host_check):
    rng = np.random.default_rng(4)
    near = np.pi / 2 * np.arange(1, 64) + rng.uniform(-1e-9, 1e-9, 63)
    x = np.concatenate([rng.uniform(-3.2, 3.2, 400000), rng.uniform(-1000, 1000, 100000), near, -near, [0.0, 1e-300, -1e-10]])
    y = _run(host_check, 3, x)
    err = _ulps(y, np.sin(x.astype(np.longdouble)))
    err = err[np.isfinite(err)]
    assert float(err.max()) < 1.0, float(err.max())
    assert np.array_equal(_run(host_check, 3, -x), -y)  # odd
    assert np.mean(y == np.sin(x)) > 0.9


def test_levenberg_marquardt_lambda_table_is_the_cpu_expression(host_check):
    """CvLevMarq's exp(lambdaLg10 * log(10.)): the device reads the 33 values from a table; they must be what the CPU path
    (the oracle, OpenCV) computes with this image's libm"""
    import math
    k = np.arange(-16, 17)
    got = _run(host_check, 4, k.astype(np.float64))
    assert np.array_equal(got, np.array([math.exp(int(i) * math.log(10.0)) for i in k]))
