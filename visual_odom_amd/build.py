"""Build libvo_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m visual_odom_amd.build [--force] [--dev]

--dev additionally builds libvo_hip_dev.so with -DVO_DEV_VARIANTS: the same library plus the kernel variants that were
built, measured and lost (two-features-per-wavefront LK, 128-register pose kernels, 128 x 32 FAST tile, ordinary-store
Scharr) and the environment switches that select them (VO_LK_PAIR, VO_POSE_WAVES via vo_set_schedule(4), VO_FAST_TILE,
VO_SCHARR_NT, VO_EPNP_LDS_KB, VO_SERIAL_POSE).  tools/ uses it (VO_HIP_LIB=.../libvo_hip_dev.so); the product library has
none of them and reads no environment variable.

Flags that matter for parity: -ffp-contract=off (no FMA contraction: the f32 2x2 LK solve and the
f64 pose math must round like the CPU path) and correctly rounded f32 divide / sqrt.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
SO = os.path.join(HERE, "libvo_hip.so")
SOURCES = ["pyramid.hip", "fast.hip", "lk.hip", "post.hip", "pnp.hip", "essential.hip", "seq.hip", "capi.hip", "capi_run.hip",
           "capi_sched.hip", "capi_seq.hip", "capi_dropin.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def _newer(target, deps):
    return os.path.exists(target) and all(os.path.getmtime(target) >= os.path.getmtime(d) for d in deps)


def build(force=False, verbose=False, dev=False):
    obj_dir = OBJ + ("_dev" if dev else "")
    so = SO.replace("libvo_hip.so", "libvo_hip_dev.so") if dev else SO
    os.makedirs(obj_dir, exist_ok=True)
    flags = list(FLAGS) + (["-DVO_DEV_VARIANTS"] if dev else [])
    if os.environ.get("VO_LK_ATTRS"):  # developer A/B of the LK kernel's register caps
        flags.append("-DVO_LK_ATTRS=" + os.environ["VO_LK_ATTRS"])
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "vo_hip.h"))
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(obj_dir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or not _newer(obj, [src] + headers):
            jobs.append([HIPCC] + flags + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-8000:]))
        return r.stderr

    with ThreadPoolExecutor(max_workers=5) as ex:
        for msg in ex.map(run, jobs):
            if verbose and msg.strip():
                print(msg)
    if force or jobs or not _newer(so, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
    return so


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--dev" in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, dev=True))
