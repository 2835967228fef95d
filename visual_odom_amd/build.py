"""Build libvo_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m visual_odom_amd.build [--force]

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
SOURCES = ["pyramid.hip", "fast.hip", "lk.hip", "post.hip", "pnp.hip", "essential.hip", "seq.hip", "capi.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def _newer(target, deps):
    return os.path.exists(target) and all(os.path.getmtime(target) >= os.path.getmtime(d) for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    flags = list(FLAGS)
    if os.environ.get("VO_LK_ATTRS"):  # developer A/B of the LK kernel's register caps
        flags.append("-DVO_LK_ATTRS=" + os.environ["VO_LK_ATTRS"])
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "vo_hip.h"))
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
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
    if force or jobs or not _newer(SO, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
