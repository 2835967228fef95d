import os
import subprocess
import sys

# The CPU oracle is OpenMP code.  On a many-core host (the GPU box lists 256 CPUs) its default width makes the small
# per-level loops crawl (measured there: 3.1 s per frame at 256 threads against 0.1 s at 128 and 0.4 s at 8; a
# 20-minute test session was lost to it), so the checker runs at a bounded width unless the caller chose one.
# Must be set before the first OpenMP runtime of the process initialises (torch brings one too).
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(os.cpu_count() or 8, 32))))
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The sanitizer tier (tests/test_sanitize.py starts it): with VO_SANITIZE=1 every host library the tests
# build -- the kernel emulator (all product kernel sources), the device-math headers, the oracle, the reference glue -- is
# compiled with ASan + UBSan into .../_build/san and the process runs under LD_PRELOAD=libasan.so.
SANITIZE = os.environ.get("VO_SANITIZE", "0") not in ("", "0")
SAN_FLAGS = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"] if SANITIZE else []
BUILD_DIR = os.path.join(ROOT, "tests", "_build", *(["san"] if SANITIZE else []))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "sanitize: the full ASan + UBSan tier (minutes; `pytest -m sanitize`), beside the quick one of the CPU suite")


@pytest.fixture(scope="session")
def orc():
    """the CPU oracle (checker) -- built on demand with gcc"""
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def small_world():
    from visual_odom_amd import synth
    return synth.StereoWorld(seed=11, width=480, height=160, fx=300.0, cx=239.5, cy=79.5, bf=-160.0,
                             tex_size=1024)


@pytest.fixture(scope="session")
def small_seq(small_world):
    """3 stereo frames (2 quadruples) of the small world + keypoints of frames 0 and 1"""
    from visual_odom_amd import synth
    lefts, rights, poses, depths = small_world.render_sequence(3)
    pts = [synth.select_keypoints(lefts[k], bucket=16, per_bucket=2) for k in range(2)]
    return dict(L=lefts, R=rights, poses=poses, depths=depths, pts=pts)


@pytest.fixture(scope="session")
def kitti_world():
    from visual_odom_amd import synth
    return synth.StereoWorld(seed=20260925)


@pytest.fixture(scope="session")
def kitti_seq(kitti_world):
    from visual_odom_amd import synth
    lefts, rights, poses, depths = kitti_world.render_sequence(2)
    pts = synth.select_keypoints(lefts[0], bucket=37, per_bucket=6)
    return dict(L=lefts, R=rights, poses=poses, depths=depths, pts=pts)


@pytest.fixture(scope="session")
def host_check():
    """device-side math headers compiled for the host with g++ (unit test of the kernel code)"""
    import ctypes
    src = os.path.join(ROOT, "tests", "host_check", "host_check.cpp")
    out_dir = BUILD_DIR
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libhost_check.so")
    deps = [src] + [os.path.join(ROOT, "visual_odom_amd", "csrc", f) for f in ("vo_linalg.h", "vo_epnp.h", "vo_tri.h", "vo_lkmath.h", "vo_fivept.h", "vo_p3p.h", "vo_math.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"] + SAN_FLAGS + ["-o", so, src])
    return ctypes.CDLL(so)


@pytest.fixture(scope="session")
def volib():
    from visual_odom_amd import _lib
    return _lib


@pytest.fixture(scope="session")
def gpu_ctx(volib):
    """a vo_ctx on GPU 0; fails loudly (no skip, no fallback) when the HIP path cannot run"""
    ctx = volib.Context(0, 1920, 1080, 8192, 8)
    yield ctx
    ctx.close()


def vp(a):
    import ctypes
    return a.ctypes.data_as(ctypes.c_void_p)


def _build_example(name):
    from visual_odom_amd import build
    build.build()
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, name)
    src = os.path.join(ROOT, "examples", name + ".cpp")
    libdir = os.path.join(ROOT, "visual_odom_amd")
    deps = [src, os.path.join(ROOT, "include", "vo_hip.h"), os.path.join(libdir, "libvo_hip.so"),
            os.path.join(ROOT, "examples", "vo_io.h"), os.path.join(ROOT, "examples", "vo_seq_host.h")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", src, "-I" + os.path.join(ROOT, "include"), "-L" + libdir,
                               "-lvo_hip", "-lz", "-lpthread", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


@pytest.fixture(scope="session")
def vo_multi_gpu_binary():
    """examples/vo_multi_gpu.cpp (one host thread + one vo_ctx per GPU, sequences sharded, no collective), built with g++"""
    return _build_example("vo_multi_gpu")


@pytest.fixture(scope="session")
def vo_seq_run_binary():
    """examples/vo_seq_run.cpp (the lock-step sequence loop as a C++ host program over the C ABI), built with g++"""
    return _build_example("vo_seq_run")


@pytest.fixture(scope="session")
def vo_run_binary():
    """examples/vo_run.cpp (the reference's frame loop as a C++ host program over the C ABI), built with g++"""
    from visual_odom_amd import build
    build.build()
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "vo_run")
    src = os.path.join(ROOT, "examples", "vo_run.cpp")
    libdir = os.path.join(ROOT, "visual_odom_amd")
    deps = [src, os.path.join(ROOT, "include", "vo_hip.h"), os.path.join(libdir, "libvo_hip.so")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", src, "-I" + os.path.join(ROOT, "include"), "-L" + libdir,
                               "-lvo_hip", "-lz", "-lpthread", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe
