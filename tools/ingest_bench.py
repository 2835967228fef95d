"""Decode-inclusive ingest rate: end-to-end frames/s FROM IMAGE FILES --
read + decode + upload + compute -- for 1 and 8 sequences, through the C++ host (examples/vo_seq_run.cpp: zlib-only PNG
reader + decoder thread pool) and through the python front end (visual_odom_amd.run: PIL in a thread pool), PNG and PGM.

A KITTI-00-shaped sequence is rendered (synth.py) and written once -- 40 distinct frames, continued to n frames by links
that walk them forwards and backwards (consecutive files are always consecutive renders) -- and 8 "sequences" are 8 links
to that directory: the page cache serves the files, as it would for a sequence that was just downloaded.  n defaults to
600 so that the library's one-off schedule probe (~0.2 s at the start of a loop) does not dominate.  The rendered images carry sensor-like noise and
compress to ~0.8 of their raw size, i.e. they are SLOWER to inflate than real KITTI PNGs (~0.55): a conservative number.

    python tools/ingest_bench.py [n_frames] > profiles/r03_ingest.json
"""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    n_distinct = min(n, 40)
    from PIL import Image
    from visual_odom_amd import synth
    import conftest
    exe = conftest._build_example("vo_seq_run")
    world = synth.StereoWorld(seed=20260925)
    L, R, _, _ = world.render_sequence(n_distinct)

    def tri(k):  # forwards, then backwards
        m = k % (2 * (n_distinct - 1))
        return m if m < n_distinct else 2 * (n_distinct - 1) - m
    tmp = tempfile.mkdtemp(prefix="vo_ingest_")
    sizes = {}
    for fmt in ("png", "pgm"):
        d = os.path.join(tmp, fmt, "00")
        for cam, imgs in ((0, L), (1, R)):
            os.makedirs(os.path.join(d, "image_%d" % cam))
            for k, img in enumerate(imgs):
                path = os.path.join(d, "image_%d" % cam, "%06d.%s" % (k, fmt))
                if fmt == "png":
                    Image.fromarray(img, "L").save(path, compress_level=6)
                else:
                    with open(path, "wb") as f:
                        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]) + img.tobytes())
            for k in range(n_distinct, n):
                os.symlink("%06d.%s" % (tri(k), fmt), os.path.join(d, "image_%d" % cam, "%06d.%s" % (k, fmt)))
        sizes[fmt] = os.path.getsize(os.path.join(d, "image_0", "000000." + fmt))
        for s in range(1, 8):
            os.symlink(d, os.path.join(tmp, fmt, "%02d" % s))
    cal = os.path.join(tmp, "kitti00.yaml")
    with open(cal, "w") as f:
        f.write("%YAML:1.0\nCamera.fx: 718.8560\nCamera.fy: 718.8560\nCamera.cx: 607.1928\nCamera.cy: 185.2157\nCamera.bf: -386.1448\n")
    out = {"frames_per_sequence": n, "image": "1241x376 8-bit gray", "file_bytes": sizes, "host_cpus": os.cpu_count(), "runs": []}
    calib = ["718.856", "607.1928", "185.2157", "-386.1448"]
    for fmt in ("png", "pgm"):
        for S in (1, 8):
            dirs = [os.path.join(tmp, fmt, "%02d" % s) for s in range(S)]
            for threads in (1, 8, 32):
                r = subprocess.run([exe, "--decode-threads", str(threads)] + calib + [str(n), "1", os.path.join(tmp, "cpp")] + dirs,
                                   capture_output=True, text=True)
                m = re.search(r"= ([0-9.]+) frames/s end to end.*?; ([0-9.]+) s spent waiting", r.stderr)
                out["runs"].append({"host": "c++ (examples/vo_seq_run)", "format": fmt, "sequences": S, "decode_threads": threads,
                                    "frames_per_s": float(m.group(1)) if m else None,
                                    "s_waiting_for_decoders": float(m.group(2)) if m else None, "rc": r.returncode})
            for threads in (0, 8):
                t0 = time.perf_counter()
                r = subprocess.run([sys.executable, "-m", "visual_odom_amd.run", ",".join(dirs), cal, "--max-frames", str(n),
                                    "--decode-threads", str(threads), "--out", os.path.join(tmp, "py")],
                                   capture_output=True, text=True, cwd=ROOT)
                m = re.search(r"'end_to_end_frames_per_s': ([0-9.]+)", r.stdout)
                out["runs"].append({"host": "python (visual_odom_amd.run, PIL)", "format": fmt, "sequences": S,
                                    "decode_threads": threads, "frames_per_s": float(m.group(1)) if m else None,
                                    "rc": r.returncode, "process_wall_s": time.perf_counter() - t0})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
