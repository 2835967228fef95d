"""The sanitizer tier ("-fsanitize=address,undefined on the CPU oracle"; VERDICT r04 item 2 widened it
to the kernel emulator, which runs every product kernel source on the CPU).

With VO_SANITIZE=1 the fixtures of tests/conftest.py, tests/test_kernel_emulation.py and oracle/oracle.py build their host
libraries with `-fsanitize=address,undefined -fno-sanitize-recover=all` into .../_build/san (oracle/_ref/san for the
reference's own glue sources), and the emulator gives every pyramid level of every image its own exactly-sized heap block, so
an access of a kernel outside a level's bordered allocation aborts the run.  An instrumented library can only be loaded into
a process that has the ASan runtime first in its link order, so the tier is a child pytest under LD_PRELOAD=libasan.so:

    quick  (part of the CPU suite, ~1.3 min):  FAST tiles of every form, pyramid passes + borders, the LK kernel on the small and odd
                                             shapes, the device-math headers on the host, the oracle's LK / FAST / glue tests, the
                                             reference's own glue sources
    full   (`pytest -m sanitize`, ~10 min):    all of tests/test_kernel_emulation.py + every oracle / host-math test file

detect_stack_use_after_return=0: the emulator's wavefronts are ucontext coroutines on their own stacks; detect_leaks=0: the
interpreter itself leaks by design."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

QUICK = ["tests/test_kernel_emulation.py", "-k",
         "(fast or borders or pyramid_and_scharr or decode or lk_negative or small_and_odd or nonfinite or adversarial_values_post or seq_ingest or fine_grids) and not exhaustive "
         "and not row_packing",
         ]
QUICK_ORACLE = ["tests/test_device_math_on_host.py", "tests/test_oracle_images.py", "tests/test_oracle_glue.py", "tests/test_vo_math.py",
                "tests/test_reference_glue.py"]
FULL = ["tests/test_kernel_emulation.py", "tests/test_device_math_on_host.py", "tests/test_oracle_images.py", "tests/test_oracle_glue.py",
        "tests/test_oracle_geom.py", "tests/test_oracle_accumulation_drift.py", "tests/test_p3p.py", "tests/test_essential_oracle.py",
        "tests/test_vo_math.py", "tests/test_reference_glue.py", "tests/test_camera_shapes_oracle.py", "tests/test_golden_vectors.py",
        "tests/test_kitti_eval.py"]


def _asan_runtime():
    so = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    return so if os.path.isabs(so) and os.path.exists(so) else None


def _run_child(args, tmp_path, timeout):
    rt = _asan_runtime()
    if rt is None:
        pytest.skip("this gcc has no libasan.so")
    log = str(tmp_path / "san")
    env = dict(os.environ, VO_SANITIZE="1", LD_PRELOAD=rt,
               ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=0:log_path=" + log,
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:log_path=" + log)
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    reports = ""
    for f in sorted(os.listdir(str(tmp_path))):
        if f.startswith("san."):
            text = open(os.path.join(str(tmp_path), f)).read()
            # ASan prints one warning when it sees swapcontext (the emulator's coroutines); it is not a finding
            text = "\n".join(l for l in text.splitlines() if "doesn't fully support makecontext/swapcontext" not in l).strip()
            if text:
                reports += "---- %s\n%s\n" % (f, text[:6000])
    assert not reports, "sanitizer report(s):\n" + reports
    assert p.returncode == 0, "sanitized child pytest failed (rc %d):\n%s" % (p.returncode, p.stdout[-4000:])
    return p.stdout


def test_sanitized_quick_tier(tmp_path):
    """ASan + UBSan over the kernel sources whose loads depend on the bordered layout (FAST tiles incl. the developer build's
    128 x 32, pyramid passes, LK search tiles) with exactly-tight per-level allocations, the device-math headers and the
    oracle: no report."""
    out = _run_child(QUICK, tmp_path, 1500)
    assert " passed" in out, out[-2000:]
    out = _run_child(QUICK_ORACLE, tmp_path, 1500)
    assert " passed" in out, out[-2000:]


@pytest.mark.sanitize
def test_sanitized_full_tier(request, tmp_path):
    """every emulator / oracle / host-math / reference-glue test file under ASan + UBSan (minutes): `pytest -m sanitize`"""
    if "sanitize" not in (request.config.getoption("-m") or "") and not os.environ.get("VO_SANITIZE_FULL"):
        pytest.skip("full sanitizer tier: run with -m sanitize (or VO_SANITIZE_FULL=1)")
    out = _run_child(FULL, tmp_path, 5400)
    assert " passed" in out, out[-2000:]
