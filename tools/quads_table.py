"""How much does the headline depend on WHICH synthetic frames the batch cycles?  

    python tools/quads_table.py [max_quads] > profiles/r05_quads_table.txt

The 256-frame batch of bench.py cycles `--quads` distinct rendered quadruples of one synthetic street (seed = the world); the LK
launch's time is set by the iteration counts those frames produce.  This runs the lean headline (bench.main, same code path,
--validate 2) for 8 / 32 / max_quads quadruples x 3 seeds in ONE process: each seed's longest sequence is rendered once (three
worker processes) and the shorter ones are its prefixes -- render_sequence is prefix-stable, so these are the very images
`bench.py --quads Q --seed S` renders itself."""
import contextlib
import io
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def render(args):
    seed, n = args
    from visual_odom_amd import synth
    world = synth.StereoWorld(seed=seed)
    lefts, rights, _, _ = world.render_sequence(n + 1)
    return seed, lefts, rights


def main():
    qmax = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    seeds = [20260925, 7, 1234]
    quads = sorted({8, 32, qmax})
    with ProcessPoolExecutor(3) as ex:
        rendered = list(ex.map(render, [(s, qmax) for s in seeds]))
    import bench
    from visual_odom_amd import synth
    rows = []
    for seed, lefts, rights in rendered:
        for q in quads:
            bench._RENDERED[(synth.KITTI_W, synth.KITTI_H, seed, q)] = (synth.StereoWorld(seed=seed), lefts[:q + 1], rights[:q + 1], {})
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                out = bench.main(["--quads", str(q), "--seed", str(seed), "--steps", "10", "--warmup", "2", "--no-cpu-baseline",
                                  "--sustain", "0", "--no-replay-leg", "--no-configs", "--validate", "2"])
            r = out["roofline"]
            rows.append((q, seed, out["value"], out["ms_per_step"], r["launch_ms"], r["lk_ns_per_feature"],
                         out["config"]["points_per_frame"], out["validated_frames"]))
            print("quads %3d  seed %-9d  %7.0f frames/s  %7.3f ms/step  lk %7.3f ms  %6.2f ns/feature  %7.1f points/frame  validated %d"
                  % rows[-1], flush=True)
    ns = [r[5] for r in rows]
    mean = sum(ns) / len(ns)
    print("lk ns/feature over the %d runs: mean %.2f, min %.2f (%+.1f %%), max %.2f (%+.1f %%)" % (
        len(ns), mean, min(ns), 100 * (min(ns) / mean - 1), max(ns), 100 * (max(ns) / mean - 1)))
    for q in quads:
        sel = [r[5] for r in rows if r[0] == q]
        print("  %3d quadruples: %.2f ns/feature (mean of %d seeds, spread %.1f %%)" % (q, sum(sel) / len(sel), len(sel),
                                                                                        100 * (max(sel) - min(sel)) / (sum(sel) / len(sel))))


if __name__ == "__main__":
    main()
