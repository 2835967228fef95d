"""Holds the library's schedule PROBE (vo_schedule, include/vo_hip.h) to every schedule it could have picked.

For each configuration (mode x frames-or-sequences per step x workload) one context runs with the default (probe) and one
per pinned candidate, all in one process (the per-device stream pool gives every context the same HIP streams); the
probe's frames/s should be within noise of the best pinned candidate everywhere.  This replaces round 2's table of fitted
constants (48 frames, the 49-96-sequence band, 65 536 point-frames), whose sweeps are the `r02` columns of DESIGN.md.

    python tools/schedule_sweep.py [--quick] > profiles/r03_schedule_sweep.jsonl
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default=None, help="substring of the configuration name")
    a = ap.parse_args()
    import torch
    import bench
    from visual_odom_amd import replicas
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)

    def barrier(ctx):
        ctx.batch_sync()
        torch.cuda.synchronize()

    base = argparse.Namespace(gpus=1, steps=30, warmup=4, mode="batch", frames=256, seqs=256, ring=3, ingest="device", quads=8,
                              workload="kitti2000", stages="full", mono_rotation=False, no_cpu_baseline=True, cpu_frames=0,
                              sustain=0.0, validate=0, no_replay_leg=True, selftest_replicas=False, no_configs=True,
                              hd_frames=128, schedule=None)
    configs = []
    for wl in ("kitti374", "kitti2000"):
        for n in ((1, 8, 32, 64, 128, 256) if not a.quick else (8, 256)):
            configs.append(("batch/%s/%d" % (wl, n), dict(mode="batch", workload=wl, frames=n)))
        for n in ((1, 8, 32, 64, 128, 256) if not a.quick else (8, 256)):
            configs.append(("seq/%s/%d" % (wl, n), dict(mode="sequences", workload=wl, seqs=n)))
    for wl in ("zed374", "rgbd374"):
        for n in ((8, 64) if not a.quick else (8,)):
            configs.append(("seq/%s/%d" % (wl, n), dict(mode="sequences", workload=wl, seqs=n)))
            configs.append(("batch/%s/%d" % (wl, n), dict(mode="batch", workload=wl, frames=n)))
    for name, over in configs:
        if a.only and a.only not in name:
            continue
        args = copy.copy(base)
        for k, v in over.items():
            setattr(args, k, v)
        n_units = args.frames if args.mode == "batch" else args.seqs
        args.steps = 60 if n_units <= 8 else 30 if n_units <= 64 else 16
        run = bench.run_batch if args.mode == "batch" else bench.run_sequences
        rec = {"config": name, "runs": {}}
        cands = [None] + ["%d,%d,%d" % (w, s, p) for w in (1, 2) for s in (1, 2) for p in ((0, 1) if args.mode == "sequences" else (0,))]
        for sched in cands:
            args.schedule = sched
            t0 = time.perf_counter()
            out = run(args, 0, 1, 0, dev, None, barrier, torch, replicas)
            rec["runs"]["probe" if sched is None else sched] = {"fps": out["value"], "ms_per_step": out["ms_per_step"],
                                                                "schedule": out["config"]["schedule"],
                                                                "wall_s": time.perf_counter() - t0}
        pinned = {k: v["fps"] for k, v in rec["runs"].items() if k != "probe"}
        best = max(pinned, key=pinned.get)
        rec["probe_fps"] = rec["runs"]["probe"]["fps"]
        rec["probe_pick"] = rec["runs"]["probe"]["schedule"]
        rec["probe_ms"] = rec["runs"]["probe"]["schedule"].get("probe_ms")
        rec["best_pinned"] = best
        rec["best_pinned_fps"] = pinned[best]
        rec["worst_pinned_fps"] = min(pinned.values())
        rec["probe_over_best"] = rec["probe_fps"] / pinned[best]
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
