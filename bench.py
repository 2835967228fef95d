#!/usr/bin/env python
"""bench.py -- stereo frames/sec of the MI355X hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

--mode batch (default, the headline): one "step" = one pass of the hot path (bordered pyramids + Scharr images of
every stereo pair -> 4-hop LK -> filter -> triangulation -> PnP/RANSAC) over one batch of B KITTI-00-shaped
(1241x376) stereo frame quadruples with ~2000 bucketed keypoints each, all inputs already resident in HBM.  The
batch is a SEQUENCE: B + 1 stereo pairs in the image table, frame b = (pair b, pair b + 1), so every step builds
2 (B + 1) pyramids -- two new images per frame, like a real sequence where the t1 pyramids of one frame are the t0
pyramids of the next.  The LK input points of every frame are given (resident in HBM), i.e. frames are independent.

--mode sequences: EXACT replay of the reference's frame loop (main.cpp:123-224) for S independent sequences in lock
step (vo_seq_*): one step = one new stereo pair per sequence -> FAST + bucketing from the features carried on the
device -> LK x4 -> filters -> triangulation -> PnP/RANSAC -> pose integration; frame k + 1 of a sequence starts
from what frame k left (visualOdometry.cpp:127).  --ingest device: the new pairs come from HBM (device-to-device);
--ingest pinned / host: from page-locked / pageable host memory over PCIe on a copy stream (PCIe-inclusive rate).

Multi-GPU = replicas: each rank runs its own batch / its own sequences on its own GPU, no data-path collective
(SURVEY.md 8e); value = frames of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0 with the `roofline` (dominant kernel = fused LK, HBM-bound model, algorithmic bytes
of SURVEY.md 8d / launch duration from HIP events on the launch stream), `cpu_baseline` (the oracle, a scalar C port
with OpenMP, on a bounded sample of the same frames), `sustained` (a >= 5 s leg of the same loop) and
`validated_frames` (frames pulled back after the timed loop and held to the oracle, outside the timer).
"""
import argparse
import json
import os
import sys

# bounded default width of the CPU checker's OpenMP loops (tests/conftest.py has the measurement); before torch's runtime
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(os.cpu_count() or 8, 32))))
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_GPUS = 1  # distinct GPUs this job runs on (set in main)
PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (width, height, per_bucket, lk_max_level, description)
    "kitti2000": (1241, 376, 6, 3, "KITTI-00-shaped 1241x376 stereo, ~2000 bucketed keypoints/frame "
                                   "(bucket=rows/10, 6 per bucket), maxLevel 3"),
    "kitti374": (1241, 376, 1, 3, "KITTI-00-shaped 1241x376 stereo, reference-default bucketing "
                                  "(1 per bucket, <=374 pts), maxLevel 3"),
    "zed374": (1280, 720, 1, 3, "calibration/zed.yaml-shaped 1280x720 stereo, reference-default bucketing (1 per bucket), maxLevel 3"),
    "rgbd374": (640, 480, 1, 3, "calibration/rgbd.yaml-shaped 640x480 stereo, reference-default bucketing (1 per bucket), maxLevel 3"),
    "hd4000": (1920, 1080, 60, 3, "synthetic 1920x1080 stereo, 4000 keypoints/frame fed at the boundary "
                                  "(60 per bucket, 3 px spacing, first 4000), maxLevel 3"),
    "hd4000l4": (1920, 1080, 60, 4, "synthetic 1920x1080 stereo, 4000 keypoints/frame fed at the boundary "
                                    "(60 per bucket, 3 px spacing, first 4000), maxLevel 4 (5 pyramid levels)"),
}


_RENDERED = {}  # (w, h, seed, n_quads) -> rendered sequence: the legs of one run share their images


def build_inputs(workload, n_quads, seed):
    from visual_odom_amd import synth
    w, h, per_bucket, max_level, _ = WORKLOADS[workload]
    key = (w, h, seed, n_quads)
    if key not in _RENDERED:
        if (w, h) == (synth.KITTI_W, synth.KITTI_H):
            world = synth.StereoWorld(seed=seed)
        else:
            world = synth.StereoWorld(seed=seed, width=w, height=h, fx=synth.KITTI_FX * w / synth.KITTI_W,
                                      cx=w / 2.0 - 0.5, cy=h / 2.0 - 0.5, bf=synth.KITTI_BF * w / synth.KITTI_W)
        lefts, rights, poses, _ = world.render_sequence(n_quads + 1)
        _RENDERED[key] = (world, lefts, rights, {})
    world, lefts, rights, pts_cache = _RENDERED[key]
    if (workload.startswith("hd4000"), per_bucket) in pts_cache:
        return world, lefts, rights, pts_cache[(workload.startswith("hd4000"), per_bucket)], max_level
    bucket = h // 10
    if workload.startswith("hd4000"):
        pts = [synth.select_keypoints(lefts[k], bucket=bucket, per_bucket=per_bucket, min_dist=3)[:4000]
               for k in range(n_quads + 1)]
    else:
        pts = [synth.select_keypoints(lefts[k], bucket=bucket, per_bucket=per_bucket) for k in range(n_quads + 1)]
    pts_cache[(workload.startswith("hd4000"), per_bucket)] = pts
    return world, lefts, rights, pts, max_level


def tri(j, S):
    """table pair j shows rendered pair tri(j): the S + 1 rendered pairs are walked forwards then backwards, so
    consecutive table pairs are always consecutive rendered frames (real motion)"""
    m = j % (2 * S)
    return m if m <= S else 2 * S - m


def setup_batch(ctx, world, lefts, rights, pts, B, S, dev_imgs=None):
    """image table of a B-frame sequence batch (pair j = rendered pair tri(j)), quads, LK input points, projection.
    dev_imgs: [(left ptr, right ptr)] device pointers per rendered pair (bench: torch tensors), else host uploads."""
    w, h = world.w, world.h
    ctx.batch_configure(2 * (B + 1), w, h, B)
    for j in range(B + 1):
        for side in (0, 1):
            if dev_imgs is not None:
                ctx.batch_upload_image_dev(2 * j + side, dev_imgs[tri(j, S)][side], w)
            else:
                ctx.batch_upload_image(2 * j + side, (lefts, rights)[side][tri(j, S)])
    ctx.batch_sync()
    ctx.batch_set_quads([[2 * b, 2 * b + 1, 2 * b + 2, 2 * b + 3] for b in range(B)])
    frame_pts = [pts[tri(b, S)] for b in range(B)]
    for b in range(B):
        ctx.batch_set_points(b, frame_pts[b])
    ctx.batch_set_projection(*world.proj_matrices())
    return frame_pts


def validate_frames(ctx, frames, lefts, rights, frame_pts, world, S, full=True, cache=None, max_level=3):
    """Pulls the results of `frames` out of the batch and holds them to the oracle (the checker -- outside any timed
    region): circular-matching survivors and the tracks that reach triangulation BIT-EXACT, triangulation <= 1e-5
    relative, RANSAC control flow and inlier set identical, rvec / tvec <= 1e-6.  Returns the number of frames
    checked; raises AssertionError on the first difference."""
    from oracle import oracle as orc
    orc.build()
    P_l, P_r = world.proj_matrices()
    K = world.K()
    cache = {} if cache is None else cache
    n = 0
    for b in frames:
        a, c = tri(b, S), tri(b + 1, S)
        if (a, c) not in cache:
            ref = orc.circular_matching(lefts[a], rights[a], lefts[c], rights[c], frame_pts[b], max_level=max_level)
            (l0, r0, l1, r1), _ = orc.check_valid_and_remove(ref["l0"], ref["r0"], ref["l1"], ref["r1"], ref["l0_ret"])
            rec = dict(keep=ref["keep_idx"], l0=l0, r0=r0, l1=l1, r1=r1)
            if full and len(l0) >= 5:
                rec["xyz"] = orc.triangulate(P_l, P_r, l0, r0)
                rec["pnp"] = orc.solve_pnp_ransac(rec["xyz"], l1, K)
            cache[(a, c)] = rec
        rec = cache[(a, c)]
        got = ctx.batch_get_filtered(b)
        assert np.array_equal(got["keep_idx_circ"], rec["keep"]), "frame %d: circular-matching survivors differ" % b
        for name in ("l0", "r0", "l1", "r1"):
            assert np.array_equal(got[name].view(np.uint32), rec[name].view(np.uint32)), "frame %d: %s differs" % (b, name)
        if full and "pnp" in rec:
            den = np.abs(rec["xyz"]).max(1, keepdims=True)
            assert np.max(np.abs(got["xyz"] - rec["xyz"]) / den) <= 1e-5, "frame %d: triangulation differs" % b
            rc, rv, tv, inl, dbg = rec["pnp"]
            pose = ctx.batch_get_pose(b)
            assert pose["status"] == rc and np.array_equal(pose["inliers"], inl), "frame %d: inlier set differs" % b
            assert (pose["niters"], pose["best_iter"], pose["max_good"]) == tuple(int(x) for x in dbg[:3]), \
                "frame %d: RANSAC control flow differs" % b
            assert np.abs(pose["rvec"] - rv).max() <= 1e-6 and np.abs(pose["tvec"] - tv).max() <= 1e-6, \
                "frame %d: pose differs" % b
        n += 1
    return n


def validate_sequences(ctx, seqs, feed, lefts, rights, world, n_frames, per_bucket):
    """sequence mode: the first `n_frames` processed frames of the given sequences against the oracle's functions
    chained like the reference's loop (matchingFeatures -> triangulation -> trackingFrame2Frame -> integration):
    per-frame counts identical, rvec / tvec / frame_pose <= 1e-6."""
    from oracle import oracle as orc
    orc.build()
    P_l, P_r = world.proj_matrices()
    K = world.K()
    h, w = lefts[0].shape
    done = 0
    for s in seqs:
        rows, info = ctx.seq_get_trajectory(s, 0, n_frames)
        o_pts, o_ages = np.zeros((0, 2), np.float32), np.zeros(0, np.int32)
        o_pose = np.eye(4)
        for k in range(len(rows)):
            a, c = feed(s, k), feed(s, k + 1)
            if len(o_pts) < 2000:
                fast = orc.fast_detect(lefts[a], 20, True)
                o_pts = np.vstack([o_pts, fast])
                o_ages = np.concatenate([o_ages, np.zeros(len(fast), np.int32)])
            bp, ba = orc.bucketing_features(h, w, o_pts, o_ages, h // 10, per_bucket)
            cm = orc.circular_matching(lefts[a], rights[a], lefts[c], rights[c], bp, ages=ba)
            (l0, r0, l1, r1), _ = orc.check_valid_and_remove(cm["l0"], cm["r0"], cm["l1"], cm["r1"], cm["l0_ret"])
            o_pts, o_ages = l1, cm["ages"]
            xyz = orc.triangulate(P_l, P_r, l0, r0)
            rc, rv, tv, inl, _ = orc.solve_pnp_ransac(xyz, l1, K)
            Rm = orc.rodrigues(rv)
            e = orc.rotation_matrix_to_euler(Rm)
            if abs(e[1]) < 0.1 and abs(e[0]) < 0.1 and abs(e[2]) < 0.1:
                o_pose, _ = orc.integrate_odometry_stereo(o_pose, Rm, tv)
            got = tuple(int(v) for v in info[k][:4])
            assert got == (len(bp), len(cm["l0"]), len(l1), len(inl)), "sequence %d frame %d: counts %s differ" % (s, k, got)
            assert np.abs(rows[k][12:15] - rv).max() <= 1e-6 and np.abs(rows[k][15:18] - tv).max() <= 1e-6, \
                "sequence %d frame %d: pose differs" % (s, k)
            assert np.abs(rows[k][:12].reshape(3, 4) - o_pose[:3]).max() <= 1e-6, "sequence %d frame %d: frame_pose" % (s, k)
            done += 1
    return done


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default="batch", choices=["batch", "sequences"])
    ap.add_argument("--frames", type=int, default=256,
                    help="batch mode: frame quadruples per step per GPU (a 256-frame sequence batch = 514 images, "
                         "1.8 GB of pyramids + Scharr images)")
    ap.add_argument("--seqs", "--S", dest="seqs", type=int, default=256,
                    help="sequence mode: independent sequences per GPU (one frame each per step)")
    ap.add_argument("--ring", type=int, default=3, choices=[2, 3], help="sequence mode: stereo pairs resident per sequence")
    ap.add_argument("--ingest", default="device", choices=["device", "pinned", "host"],
                    help="sequence mode: where the new stereo pairs come from (device = resident in HBM)")
    ap.add_argument("--quads", type=int, default=32,
                    help="distinct rendered quadruples cycled over the batch (33 rendered stereo pairs; round 4: 8 -- the LK "
                         "iteration statistics that set the headline came from 9 pairs of one street, VERDICT r04 weak 7; "
                         "profiles/r05_quads_table.txt: 8 / 32 / 128 quadruples x 3 seeds)")
    ap.add_argument("--seed", type=int, default=20260925, help="seed of the synthetic world (rank r renders seed + r)")
    ap.add_argument("--workload", default="kitti2000", choices=sorted(WORKLOADS))
    ap.add_argument("--stages", default="full", choices=["full", "lk", "detect+full"],
                    help="batch mode: full = BASELINE config 3 (LK+tri+PnP on device); lk = config 2 (circularMatching only); "
                         "detect+full = additionally FAST + bucketing on the device produce the LK input points "
                         "(SURVEY.md 8 row f1) instead of points resident in HBM")
    ap.add_argument("--mono-rotation", action="store_true",
                    help="also run findEssentialMat + recoverPose per frame (trackingFrame2Frame's mono_rotation = true; "
                         "the reference's main loop passes false)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=6)
    ap.add_argument("--sustain", type=float, default=5.0, help="seconds of the additional sustained leg (0 = off)")
    ap.add_argument("--validate", type=int, default=3, help="frames held to the oracle after the timed loop (0 = off)")
    ap.add_argument("--no-replay-leg", action="store_true",
                    help="batch mode: skip the additional exact-replay leg (`exact_replay` in the JSON line: the same workload "
                         "through --mode sequences with 256 sequences, reported next to `value`, never instead of it)")
    ap.add_argument("--selftest-replicas", action="store_true",
                    help="CPU-only self-test of the N > 1 control flow (rank discovery, process group, barrier, max-over-ranks "
                         "time / summed frames, rank-0 JSON): no GPU work, fabricated per-rank timings (tests/test_replicas_gloo.py)")
    ap.add_argument("--no-configs", action="store_true",
                    help="default run only: skip the additional legs reported in `configs` (BASELINE config 2 = LK only, the "
                         "reference-default 374-point load, config 4 = 1080p / 4000 points at maxLevel 3 and 4)")
    ap.add_argument("--hd-frames", type=int, default=128, help="frames per step of the config-4 legs")
    ap.add_argument("--schedule", default=None,
                    help="pin the pose-chain schedule instead of letting the library probe it: pose_waves,pose_streams,prepare "
                         "(vo_set_schedule: 0 / 0 / -1 = probe); tools/schedule_sweep.py uses it to hold the probe to every candidate")
    args = ap.parse_args(argv)

    # --gpus N is a promise about the line that gets printed: n_gpus = N ranks, one per GPU.  Launched by
    # torch.distributed.run the environment says the same thing; launched bare with N > 1 this process starts the N ranks
    # itself; anything else is refused -- `--gpus 8` never prints `n_gpus: 1`.
    env_ws = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus != env_ws:
        if "RANK" in os.environ or "LOCAL_RANK" in os.environ:
            raise SystemExit("bench.py: --gpus %d contradicts the launcher's WORLD_SIZE=%d" % (args.gpus, env_ws))
        return spawn_ranks(args, sys.argv[1:] if argv is None else list(argv))

    from visual_odom_amd import replicas
    rank, local_rank, world_size = replicas.rank_info()
    # One rank per GPU also means one host per GPU: each rank keeps a contiguous slice of the node's cores (its OpenMP teams --
    # the validation's oracle, the CPU baseline -- and its staging threads stay off the other ranks' cores), sized before the
    # first OpenMP runtime of the process starts; once the GPU is known the slice moves to the GPU's NUMA node (below).
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world_size)))
    pin = replicas.pin_rank(local_rank, local_world)
    import torch
    if args.selftest_replicas:
        dist = replicas.init("gloo")
        if dist is not None:
            dist.barrier()
        elapsed, frames_total = replicas.aggregate(dist, 1.0 + 0.25 * rank, args.frames * args.steps)
        out = {"metric": "stereo frames/sec on KITTI-00 1241x376 @ ~2000 features", "value": frames_total / elapsed,
               "unit": "frames/s", "n_gpus": world_size, "ranks": world_size, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "data": "SELFTEST of the multi-rank control flow: no GPU work, fabricated timings -- not a measurement"}
        # ... and of the config-5 leg's (one sequence per GPU: per-GPU figures gathered, aggregate = frames of all ranks / the
        # slowest rank's time) and of the per-rank core slices
        t5 = 0.5 + 0.01 * rank
        e5, f5 = replicas.aggregate(dist, t5, 200)
        per_gpu = replicas.gather_values(dist, 200 / t5)
        cpus = replicas.gather_values(dist, pin.get("cpus") or 0)
        first = replicas.gather_values(dist, -1 if pin.get("first_cpu") is None else pin["first_cpu"])
        out["configs"] = [{"name": "config5_one_sequence_per_gpu", "baseline_config": 5, "value": f5 / e5, "unit": "frames/s",
                           "per_gpu_value": per_gpu, "ranks": world_size, "sequences_per_gpu": 1}]
        out["host_cores_per_rank"] = {"cpus": [int(c) for c in cpus], "first_cpu": [int(c) for c in first],
                                      "omp_num_threads": os.environ.get("OMP_NUM_THREADS")}
        if rank == 0:
            print(json.dumps(out), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return out
    dist = replicas.init()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    global N_GPUS
    n_dev = torch.cuda.device_count()
    if world_size > n_dev and os.environ.get("VO_ALLOW_SHARED_GPU") != "1":
        # (VO_ALLOW_SHARED_GPU=1 + VO_DIST_BACKEND=gloo: smoke test of the N > 1 control flow on fewer GPUs; the line then
        # reports n_gpus = the GPUs actually used and `ranks` = the processes)
        raise SystemExit("bench.py: %d ranks but %d visible GPU(s) -- n_gpus would not be the number of GPUs used"
                         % (world_size, n_dev))
    N_GPUS = min(world_size, n_dev)
    local_dev = local_rank % n_dev  # one GPU per rank on a real node
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if local_world > 1:  # the slice again, now on the NUMA node each rank's GPU hangs off (sysfs; contiguous split where it is silent)
        numa = {}
        for r in range(local_world):
            p = torch.cuda.get_device_properties(r % n_dev)
            bus = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0))
            numa[r] = replicas.gpu_numa_cpus(bus)
        try:
            os.sched_setaffinity(0, range(os.cpu_count() or 1))  # (undo the first slice: plan from the node's whole core set)
        except OSError:
            pass
        pin = replicas.pin_rank(local_rank, local_world, numa)

    def barrier(ctx):
        ctx.batch_sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    default_run = args.mode == "batch" and args.workload == "kitti2000" and args.stages == "full" and \
        not args.mono_rotation and args.frames >= 256
    replay_leg = args.mode == "batch" and not args.no_replay_leg and args.workload.startswith("kitti") and \
        not args.mono_rotation and args.frames >= 256
    config_legs = default_run and not args.no_configs and dist is None
    # N > 1: BASELINE config 5 (one sequence per GPU, exact replay) beside the weak-scaled batch headline
    config5_leg = dist is not None and args.mode == "batch" and args.stages == "full" and args.workload == "kitti2000" and \
        not args.mono_rotation and not args.no_configs
    kept = [] if (replay_leg or config_legs or config5_leg) else None
    if args.mode == "sequences":
        out = run_sequences(args, rank, world_size, local_dev, dev, dist, barrier, torch, replicas)
    else:
        out = run_batch(args, rank, world_size, local_dev, dev, dist, barrier, torch, replicas, keep_ctx=kept)
    if replay_leg:
        # the same workload as an EXACT replay of the reference's frame loop (features carried from frame to frame on the
        # device, FAST + bucketing every frame): reported beside the headline, with its own validation
        import copy
        a2 = copy.copy(args)
        a2.mode, a2.seqs, a2.ring, a2.ingest, a2.no_cpu_baseline = "sequences", 256, 3, "device", True
        a2.validate = min(args.validate, 2)
        rep = run_sequences(a2, rank, world_size, local_dev, dev, dist, barrier, torch, replicas, ctx=kept[0])
        if rank == 0 and out is not None and rep is not None:
            out["exact_replay"] = {"value": rep["value"], "unit": rep["unit"], "ms_per_step": rep["ms_per_step"],
                                   "steps": rep["steps"], "validated_frames": rep["validated_frames"],
                                   "sequences_per_gpu": 256, "points_per_frame": rep["config"]["points_per_frame"],
                                   "mode": rep["config"]["mode"], "stage_ms": rep["config"]["stage_ms"],
                                   "schedule": rep["config"]["schedule"], "roofline": rep["roofline"]}
    if config_legs:
        # every other BASELINE configuration in the same driver-run line: same code path, fewer
        # steps, each leg validated against the oracle after its timed loop.  The KITTI-size legs reuse the headline's
        # context; the 1080p legs need a bigger one, created after the first is destroyed (it gets the same HIP streams
        # back from the library's per-device pool).
        import copy
        legs = []

        def leg(name, baseline_config, ctx, **over):
            a = copy.copy(args)
            a.no_cpu_baseline, a.sustain, a.steps, a.warmup, a.validate = True, 0.0, 10, 2, 4
            for k, v in over.items():
                setattr(a, k, v)
            r = run_batch(a, rank, world_size, local_dev, dev, dist, barrier, torch, replicas, ctx=ctx)
            legs.append({"name": name, "baseline_config": baseline_config, "workload": r["config"]["workload"],
                         "stages": r["config"]["stages"], "value": r["value"], "unit": r["unit"],
                         "ms_per_step": r["ms_per_step"], "steps": r["steps"], "warmup": r["warmup"],
                         "frames_per_step": r["config"]["frames_per_step_per_gpu"],
                         "points_per_frame": r["config"]["points_per_frame"], "validated_frames": r["validated_frames"],
                         "schedule": r["config"]["schedule"], "stage_ms": r["config"]["stage_ms"], "roofline": r["roofline"]})

        leg("config2_lk_only", 2, kept[0], stages="lk")
        leg("reference_default_374", 3, kept[0], workload="kitti374", steps=20)
        if rank == 0:
            legs.append(latency_leg(kept[0], quads=min(args.quads, args.frames), seed=args.seed + rank))
        # the exact replay with the new pairs coming from HOST memory every step: the honest ingest
        # numbers in the driver's own run -- PCIe-inclusive, never `value`; pcie_gb_s = frames/s x 2 images x w x h bytes
        for ing, nm in (("pinned", "host_pinned"), ("host", "host_pageable")):
            a3 = copy.copy(args)
            a3.mode, a3.seqs, a3.ring, a3.ingest, a3.no_cpu_baseline = "sequences", 256, 3, ing, True
            a3.steps, a3.warmup, a3.validate = 20, 4, 2
            r3 = run_sequences(a3, rank, world_size, local_dev, dev, dist, barrier, torch, replicas, ctx=kept[0])
            if rank == 0 and r3 is not None:
                legs.append({"name": "exact_replay_256_sequences_" + nm, "baseline_config": 3, "workload": r3["config"]["workload"],
                             "stages": "detect+full", "ingest": r3["config"]["ingest"], "value": r3["value"], "unit": r3["unit"],
                             "ms_per_step": r3["ms_per_step"], "steps": r3["steps"], "warmup": r3["warmup"], "frames_per_step": 256,
                             "points_per_frame": r3["config"]["points_per_frame"], "validated_frames": r3["validated_frames"],
                             "pcie_gb_s": r3["value"] * 2.0 * r3["config"]["image_bytes"] / 1e9,
                             "schedule": r3["config"]["schedule"], "stage_ms": r3["config"]["stage_ms"], "roofline": r3["roofline"]})
        kept[0].close()
        kept[0] = None
        from visual_odom_amd import _lib
        hd = _lib.Context(local_dev, 1920, 1080, 8192, args.hd_frames)
        try:
            leg("config4_1080p_4000_maxlevel3", 4, hd, workload="hd4000", frames=args.hd_frames, quads=4, steps=8)
            leg("config4_1080p_4000_maxlevel4", 4, hd, workload="hd4000l4", frames=args.hd_frames, quads=4, steps=8)
        finally:
            hd.close()
        if out is not None:
            if "exact_replay" in out:
                er = out["exact_replay"]
                legs.append({"name": "exact_replay_256_sequences", "baseline_config": 3, "workload": out["config"]["workload"],
                             "stages": "detect+full", "value": er["value"], "unit": er["unit"], "ms_per_step": er["ms_per_step"],
                             "steps": er["steps"], "warmup": args.warmup, "frames_per_step": er["sequences_per_gpu"],
                             "points_per_frame": er["points_per_frame"], "validated_frames": er["validated_frames"],
                             "schedule": er["schedule"], "stage_ms": er["stage_ms"], "roofline": er["roofline"]})
            out["configs"] = legs
    if config5_leg:
        # BASELINE config 5 as written -- one sequence per GPU, exact replay of the reference's frame loop -- next to the
        # weak-scaled batch headline: every rank runs ONE sequence of its own through the lock-step loop (`--mode sequences
        # --seqs 1`), per-GPU frames/s gathered, aggregate = frames of all ranks / max-over-ranks time
        import copy
        a5 = copy.copy(args)
        a5.mode, a5.seqs, a5.ring, a5.ingest, a5.no_cpu_baseline, a5.workload = "sequences", 1, 3, "device", True, "kitti374"
        a5.steps, a5.warmup, a5.validate = 200, 10, min(args.validate, 2)
        r5 = run_sequences(a5, rank, world_size, local_dev, dev, dist, barrier, torch, replicas,
                           ctx=kept[0] if kept else None, per_rank=True)
        if rank == 0 and out is not None and r5 is not None:
            out.setdefault("configs", []).append({
                "name": "config5_one_sequence_per_gpu", "baseline_config": 5, "workload": r5["config"]["workload"],
                "mode": r5["config"]["mode"], "stages": "detect+full", "value": r5["value"], "unit": r5["unit"],
                "per_gpu_value": r5["per_rank_value"], "n_gpus": N_GPUS, "ranks": world_size, "sequences_per_gpu": 1,
                "ms_per_step": r5["ms_per_step"], "steps": r5["steps"], "warmup": r5["warmup"],
                "points_per_frame": r5["config"]["points_per_frame"], "validated_frames": r5["validated_frames"],
                "schedule": r5["config"]["schedule"], "stage_ms": r5["config"]["stage_ms"], "roofline": r5["roofline"]})
    if kept and kept[0] is not None:
        kept[0].close()
    if dist is not None:
        # one host per GPU: every rank's slice of the node's cores (replicas.pin_rank), gathered -- the slices must be disjoint
        slices = {k: replicas.gather_values(dist, -1 if pin.get(k) is None else pin[k], dev)   # (dev: RCCL reduces device tensors)
                  for k in ("cpus", "first_cpu", "last_cpu", "numa_node")}
        if rank == 0 and out is not None:
            out["config"]["host_cores_rank0"] = pin
            out["host_cores_per_rank"] = {k: [int(v) for v in vals] for k, vals in slices.items()}
            out["host_cores_per_rank"]["omp_num_threads_rank0"] = os.environ.get("OMP_NUM_THREADS")
            out["dist_backend"] = dist.get_backend()   # "nccl" = RCCL on a real node; "gloo" only in the shared-GPU smoke tests
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return out


def run_batch(args, rank, world_size, local_dev, dev, dist, barrier, torch, replicas, keep_ctx=None, ctx=None):
    from visual_odom_amd import _lib
    B, S = args.frames, min(args.quads, args.frames)
    # every rank renders its own sequence (seed by rank) = independent sequences, one per GPU
    world, lefts, rights, pts, max_level = build_inputs(args.workload, S, args.seed + rank)
    w, h = world.w, world.h
    own_ctx = ctx is None
    if own_ctx:
        ctx = _lib.Context(local_dev, w, h, 8192, B)
    ctx.set_params(lk_max_level=max_level, mono_rotation=int(args.mono_rotation))
    ctx.set_schedule(*[int(v) for v in args.schedule.split(",")]) if args.schedule else ctx.set_schedule()  # w,s,p[,wide]
    n_images = 2 * (B + 1)
    # images go through torch device tensors (PyTorch = plumbing: device memory + D2D hand-off)
    dev_t = [(torch.from_numpy(np.ascontiguousarray(lefts[k])).to(dev),
              torch.from_numpy(np.ascontiguousarray(rights[k])).to(dev)) for k in range(S + 1)]
    torch.cuda.synchronize()
    frame_pts = setup_batch(ctx, world, lefts, rights, pts, B, S, [(a.data_ptr(), b.data_ptr()) for a, b in dev_t])
    stages = (_lib.STAGE_PYRAMID | _lib.STAGE_LK | _lib.STAGE_FILTER) if args.stages == "lk" else _lib.STAGE_ALL
    if args.stages == "detect+full":
        # every frame starts from an empty carried set: FAST runs on its left t0 image, bucketing keeps
        # per_bucket corners per cell -> the same ~2000-point load, produced on the device
        for b in range(B):
            ctx.batch_set_features(b, np.zeros((0, 2), np.float32), np.zeros(0, np.int32))
        ctx.batch_set_detect_params(features_per_bucket=WORKLOADS[args.workload][2])
        stages |= _lib.STAGE_DETECT
        ctx.batch_run(stages)
        ctx.batch_sync()
        frame_pts = [ctx.batch_get_features(b)[0] for b in range(B)]

    for _ in range(args.warmup):
        ctx.batch_run(stages)
    barrier(ctx)
    K = args.steps
    t0 = time.perf_counter()
    for k in range(K):
        ctx.batch_run_slot(stages, k % _lib.EVENT_SLOTS)
    ctx.batch_sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed, frames_total = replicas.aggregate(dist, elapsed, B * K, dev)
    if dist is not None:
        dist.barrier()

    # per-stage kernel time from the HIP events recorded on the launch stream during the timed steps
    stage_ms = np.mean([ctx.batch_slot_times(k % _lib.EVENT_SLOTS) for k in range(max(0, K - _lib.EVENT_SLOTS), K)],
                       axis=0)
    # the results of the LAST timed step, held to the oracle outside the timer: the first and last frame of the
    # batch (first / last XCD group of the LK grid) and one from the middle
    validated = 0
    if args.validate > 0:  # EVERY rank holds frames of its own last step to the oracle; the line reports the smallest count
        picks = sorted({int(round(x)) for x in np.linspace(0, B - 1, max(args.validate, 2))})
        validated = validate_frames(ctx, picks, lefts, rights, frame_pts, world, S, full=args.stages != "lk",
                                    max_level=max_level)
        validated = int(min(replicas.gather_values(dist, validated, dev)))
    sustained = None
    if args.sustain > 0:
        n_sus = max(K, int(np.ceil(args.sustain / max(elapsed / K, 1e-6))))
        barrier(ctx)
        t1 = time.perf_counter()
        for k in range(n_sus):
            ctx.batch_run(stages)
        ctx.batch_sync()
        torch.cuda.synchronize()
        sus_dt = time.perf_counter() - t1
        sus_dt, sus_frames = replicas.aggregate(dist, sus_dt, B * n_sus, dev)
        sustained = {"seconds": sus_dt, "steps": n_sus, "value": sus_frames / sus_dt, "unit": "frames/s"}
    fps = frames_total / elapsed
    pts_per_launch = sum(len(p) for p in frame_pts)
    lk_bytes = sum(ctx.model_bytes(w, h, len(p))[1] for p in frame_pts)
    frame_bytes = sum(ctx.model_bytes(w, h, len(p)).sum() for p in frame_pts) / B
    lk_ms = float(stage_ms[_lib.STAGE_NAMES.index("lk")])
    achieved = lk_bytes / (lk_ms * 1e-3) / 1e9 if lk_ms > 0 else 0.0

    # the committed PMC passes were taken on the default stage set: only that configuration inherits their figures
    # (the LK launch of an LK-only run is the same kernel over the same data)
    profiled_config = args.stages in ("full", "lk") and not args.mono_rotation
    out = None
    if rank == 0:
        issue = valu_issue_frac(args.workload, B, pts_per_launch, lk_ms) if profiled_config else (None, None)
        out = {
            "metric": "stereo frames/sec on KITTI-00 1241x376 @ ~2000 features",
            "value": fps, "unit": "frames/s", "n_gpus": N_GPUS, "ranks": world_size, "steps": K, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/int32 LK (f32 2x2 solve), f64 pose solve", "data": "synthetic",
            "validated_frames": validated, "sustained": sustained,
            "config": {"workload": WORKLOADS[args.workload][4], "mode": "batch (independent frames, LK input points resident in HBM)",
                       "stages": args.stages + ("+mono_rotation" if args.mono_rotation else ""),
                       "frames_per_step_per_gpu": B, "pyramids_per_step_per_gpu": n_images,
                       "points_per_frame": float(np.mean([len(p) for p in frame_pts])),
                       "parallelism": "replicas x%d (one sequence per GPU, no collective)" % world_size,
                       # (an LK-only run has no pose chain: nothing to schedule, and the context's last probe was another leg's)
                       "schedule": dict(ctx.get_schedule(), probe_ms=ctx.get_probe_log()) if args.stages != "lk" else None,
                       "stage_ms": {n: float(v) for n, v in zip(_lib.STAGE_NAMES, stage_ms)},
                       "model_bytes_per_frame": frame_bytes,
                       "hbm_roof_fps_per_gpu": PEAK_HBM_GBS * 1e9 / frame_bytes},
            # `frac` stays the contract's figure -- algorithmic bytes / launch time / HBM peak -- but what BINDS the kernel
            # is VALU issue (profiles/r02_lk_issue_bound.md: 13.2 k VALU instructions per feature at 4 SIMD-cycles each =
            # the launch time; measured HBM traffic is half the algorithmic bytes), so `bound` says that
            "roofline": {"bound": "valu_issue", "priced_against": "hbm", "kernel": "lk_circular_kernel", "achieved": achieved,
                         "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS,
                         "traffic": measured_traffic(args.workload, B) if profiled_config else None,
                         "traffic_source": ("profiles/lk_traffic.json: rocprofv3 --pmc passes of this command, imported, not "
                                            "measured in this run") if profiled_config and measured_traffic(args.workload, B)
                         else "not profiled for this configuration",
                         # what binds the kernel: VALU issue.  valu_issue_frac (<= 1 by construction, round 6) = the launch's VALU
                         # instructions -- SQ_INSTS_VALU of the imported PMC pass, per feature x this launch's features -- at the
                         # FLOOR price of their opcode classes (3.65 SIMD-cycles per wave64 instruction for this kernel's mix)
                         # / (1024 SIMDs x 2.4 GHz x THIS run's launch time): the share of the launch that issuing this instruction
                         # stream at the hardware's best possible rate accounts for.  valu_issue_model_ratio: the same at the
                         # micro-benchmark's own per-opcode costs (4.05) -- a model that may exceed 1 (rounds 2-5 called THAT the frac)
                         "valu_issue_frac": issue[0], "valu_issue_model_ratio": issue[1],
                         "valu_issue_imported": measured_issue(args.workload, B) if profiled_config else None,
                         "bytes_per_launch": lk_bytes, "launch_ms": lk_ms, "points_per_launch": pts_per_launch,
                         "lk_ns_per_feature": 1e6 * lk_ms / max(pts_per_launch, 1),
                         "other_stages": pyramid_roofline(ctx, w, h, n_images, stage_ms,
                                                          args.workload if profiled_config else None)},
        }
        if not args.no_cpu_baseline and world_size == 1:
            out["cpu_baseline"] = cpu_baseline(lefts, rights, pts, world, min(args.cpu_frames, S), args.stages)
    del dev_t
    if keep_ctx is not None:
        keep_ctx.append(ctx)  # the exact-replay leg and the KITTI-size `configs` legs reuse the context
    elif own_ctx:
        ctx.close()
    return out


def run_sequences(args, rank, world_size, local_dev, dev, dist, barrier, torch, replicas, ctx=None, per_rank=False):
    """exact replay: S sequences x 1 frame per step, feature state carried on the device"""
    from visual_odom_amd import _lib
    S, Q = args.seqs, args.quads
    world, lefts, rights, pts, max_level = build_inputs(args.workload, Q, args.seed + rank)
    per_bucket = WORKLOADS[args.workload][2]
    w, h = world.w, world.h
    K, W = args.steps, args.warmup
    own_ctx = ctx is None
    if own_ctx:
        ctx = _lib.Context(local_dev, w, h, 4096, S)
    ctx.set_params(lk_max_level=max_level, mono_rotation=int(args.mono_rotation))
    ctx.set_schedule(*[int(v) for v in args.schedule.split(",")]) if args.schedule else ctx.set_schedule()  # w,s,p[,wide]
    ctx.batch_set_detect_params(features_per_bucket=per_bucket)
    ctx.seq_configure(S, w, h, args.ring, K + W + 8 + 520)
    ctx.batch_set_projection(*world.proj_matrices())

    def feed(s, k):  # rendered pair sequence s shows at its k-th pair: same street, every sequence phase-shifted
        return tri(k + s, Q)

    # the S new pairs of a step are handed over in ONE call (pointer tables built once per phase of the cycle)
    if args.ingest == "device":
        src = [(torch.from_numpy(np.ascontiguousarray(lefts[k])).to(dev),
                torch.from_numpy(np.ascontiguousarray(rights[k])).to(dev)) for k in range(Q + 1)]
        torch.cuda.synchronize()
        ptrs = [(a.data_ptr(), b.data_ptr()) for a, b in src]
        kind = 2
    elif args.ingest == "pinned":
        src = [(torch.from_numpy(np.ascontiguousarray(lefts[k])).pin_memory(),
                torch.from_numpy(np.ascontiguousarray(rights[k])).pin_memory()) for k in range(Q + 1)]
        ptrs = [(a.data_ptr(), b.data_ptr()) for a, b in src]
        kind = 1
    else:
        src = [(np.ascontiguousarray(lefts[k]), np.ascontiguousarray(rights[k])) for k in range(Q + 1)]
        ptrs = [(a.ctypes.data, b.ctypes.data) for a, b in src]
        kind = 0
    tables = [ctx.seq_pair_table(range(S), [ptrs[feed(s, k)][0] for s in range(S)], [ptrs[feed(s, k)][1] for s in range(S)])
              for k in range(2 * Q)]  # feed() has period 2 Q in k

    def one_step(k):
        ctx.seq_push_pairs(tables[k % (2 * Q)], w, kind)
        ctx.seq_step()

    for k in range(W + 1):  # the first step of a sequence only builds pyramids (main.cpp:110-113)
        one_step(k)
    # the library settles its schedule over the first steps of a loop (probe + up to eight candidates timed over real steps,
    # 10 + 24 .. 48 steps each, vo_schedule in vo_hip.h): warm-up goes on until that is done -- the timed steps run the settled schedule
    extra = 0
    while ctx.get_schedule()["settling"] and extra < 520:
        one_step(W + 1 + extra)
        extra += 1
    W += extra
    barrier(ctx)
    t0 = time.perf_counter()
    for k in range(W + 1, W + 1 + K):
        one_step(k)
    ctx.seq_sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    rank_fps = replicas.gather_values(dist, S * K / elapsed, dev) if per_rank else None
    elapsed, frames_total = replicas.aggregate(dist, elapsed, S * K, dev)
    if dist is not None:
        dist.barrier()
    steps_done = W + 1 + K
    stage_ms = np.mean([ctx.batch_slot_times(k % _lib.EVENT_SLOTS) for k in range(max(W + 1, steps_done - _lib.EVENT_SLOTS), steps_done)],
                       axis=0)
    info = [ctx.seq_get_trajectory(s)[1] for s in range(S)]
    n_bucketed = np.array([i[-K:, 0] for i in info])          # [S][K]
    integrated = np.mean([(i[-K:, 5] & _lib.SEQ_F_INTEGRATED) != 0 for i in info])
    validated = 0
    if args.validate > 0:
        validated = validate_sequences(ctx, sorted({0, S - 1}), feed, lefts, rights, world, args.validate, per_bucket)
        validated = int(min(replicas.gather_values(dist, validated, dev)))
    pts_per_launch = float(n_bucketed.sum(0).mean())
    lk_bytes = float(ctx.model_bytes(w, h, 1)[1]) * pts_per_launch
    frame_bytes = float(ctx.model_bytes(w, h, int(round(pts_per_launch / S))).sum())
    lk_ms = float(stage_ms[_lib.STAGE_NAMES.index("lk")])
    achieved = lk_bytes / (lk_ms * 1e-3) / 1e9 if lk_ms > 0 else 0.0
    out = None
    replay_key = "replay" + args.workload.replace("kitti", "")
    if rank == 0:
        out = {
            "metric": "stereo frames/sec on KITTI-00 1241x376 @ ~2000 features",
            "value": frames_total / elapsed, "unit": "frames/s", "n_gpus": N_GPUS, "ranks": world_size, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/int32 LK (f32 2x2 solve), f64 pose solve", "data": "synthetic",
            "validated_frames": validated, "per_rank_value": rank_fps,
            "config": {"workload": WORKLOADS[args.workload][4],
                       "mode": "sequences (exact replay of the reference frame loop: FAST + bucketing from the carried "
                               "features, state on the device, %d sequences x 1 frame per step, ring %d)" % (S, args.ring),
                       "ingest": {"device": "new pairs resident in HBM (device-to-device)",
                                  "pinned": "new pairs from page-locked host memory over PCIe (copy stream)",
                                  "host": "new pairs from pageable host memory via pinned staging over PCIe"}[args.ingest],
                       "stages": "detect+full" + ("+mono_rotation" if args.mono_rotation else ""),
                       "frames_per_step_per_gpu": S, "pyramids_per_step_per_gpu": 2 * S,
                       "points_per_frame": pts_per_launch / S, "integrated_fraction": float(integrated),
                       "parallelism": "replicas x%d (%d sequences per GPU, no collective)" % (world_size, S),
                       "schedule": dict(ctx.get_schedule(), probe_ms=ctx.get_probe_log()),
                       "stage_ms": {n: float(v) for n, v in zip(_lib.STAGE_NAMES, stage_ms)},
                       "model_bytes_per_frame": frame_bytes, "image_bytes": int(w) * int(h)},
            # traffic: the PMC passes of `bench.py --mode sequences --workload W --seqs S` (tools/gpu_round.sh pmclegs, workload
            # key "replay2000" / "replay374"; averaged over ALL the loop's LK launches incl. its warm-up) -- round 6
            "roofline": {"bound": "valu_issue", "priced_against": "hbm", "kernel": "lk_circular_kernel", "achieved": achieved,
                         "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS,
                         "traffic": measured_traffic(replay_key, S) if args.ingest == "device" else None,
                         "traffic_source": ("profiles/lk_traffic.json (%s): rocprofv3 --pmc passes of the lock-step loop, imported, "
                                            "not measured in this run" % replay_key)
                         if args.ingest == "device" and measured_traffic(replay_key, S) else "not profiled for this configuration",
                         "valu_issue_frac": valu_issue_frac(replay_key, S, pts_per_launch, lk_ms)[0] if args.ingest == "device" else None,
                         "bytes_per_launch": lk_bytes, "launch_ms": lk_ms, "points_per_launch": pts_per_launch,
                         "lk_ns_per_feature": 1e6 * lk_ms / max(pts_per_launch, 1)},
        }
        if not args.no_cpu_baseline and world_size == 1:
            out["cpu_baseline"] = cpu_baseline(lefts, rights, pts, world, min(args.cpu_frames, Q), "detect+full",
                                               per_bucket=per_bucket)
    if own_ctx:
        ctx.close()
    return out


def latency_leg(ctx, n_calls=100, n_steps=400, quads=8, seed=20260925):
    """LATENCY MODE of the drop-in boundary, measured by the driver's own run: the reference's real use is
    one sequence, one synchronous call per frame (main.cpp:123-224).  HOST images in, host results out, one frame in flight --
    PCIe-inclusive, never `value`: (a) vo_track_frame per call at the ~2000-point and the reference-default load; (b) the whole
    frame loop for ONE sequence through the lock-step API (push pair -> step, nothing read back until the end: the pose solve
    of frame k runs under detection + tracking of frame k + 1)."""
    from visual_odom_amd import synth
    world, lefts, rights, pts6, _ = build_inputs("kitti2000", quads, seed)  # (the headline's rendering: cached)
    _, _, _, pts1, _ = build_inputs("kitti374", quads, seed)
    P_l, P_r = world.proj_matrices()
    w, h = world.w, world.h
    ctx.set_params(lk_max_level=3, mono_rotation=0)
    ctx.set_schedule()
    out = {"name": "latency_drop_in", "baseline_config": 3, "workload": WORKLOADS["kitti2000"][4],
           "inputs": "pageable host images in, host results out, one frame in flight (PCIe-inclusive; never `value`)",
           "unit": "ms per vo_track_frame call"}
    order = [0, 1, 2, 3, 4, 3, 2, 1]
    for tag, pts in (("2000", pts6), ("374", pts1)):
        def call(i):
            k = i % 4
            return ctx.track_frame(lefts[k], rights[k], lefts[k + 1], rights[k + 1], pts[k], P_l, P_r)
        for i in range(12):  # (the first call of a shape probes the schedule)
            call(i)
        t = []
        for i in range(n_calls):
            t0 = time.perf_counter()
            r = call(i)
            t.append(time.perf_counter() - t0)
        t = np.array(t) * 1e3
        out["track_frame_ms_%s" % tag] = {"median": float(np.median(t)), "mean": float(t.mean()), "p95": float(np.percentile(t, 95)),
                                          "points": int(len(pts[0])), "calls": n_calls, "inliers_last": int(len(r["inliers"]))}
        # the same frames the way the reference's loop hands them over (main.cpp:157-158: the t1 pair of one frame is the t0
        # pair of the next): no t0 images, the pair the previous call kept on the device -- two uploads, two pyramids per call
        def call_kept(i):
            a, b = order[i % 8], order[(i + 1) % 8]
            return ctx.track_frame(None, None, lefts[b], rights[b], pts[a], P_l, P_r)
        ctx.track_frame(lefts[0], rights[0], lefts[1], rights[1], pts[0], P_l, P_r)
        for i in range(1, 9):
            call_kept(i)
        t = []
        for i in range(9, 9 + n_calls):
            t0 = time.perf_counter()
            r = call_kept(i)
            t.append(time.perf_counter() - t0)
        t = np.array(t) * 1e3
        out["track_frame_kept_pair_ms_%s" % tag] = {"median": float(np.median(t)), "mean": float(t.mean()),
                                                    "p95": float(np.percentile(t, 95)), "calls": n_calls,
                                                    "inliers_last": int(len(r["inliers"]))}
    for tag, fpb in (("2000", 6), ("374", 1)):
        ctx.batch_set_detect_params(features_per_bucket=fpb)
        ctx.seq_configure(1, w, h, 3, n_steps + 64)
        ctx.batch_set_projection(P_l, P_r)
        for i in range(24):
            ctx.seq_push_pair(0, lefts[order[i % 8]], rights[order[i % 8]])
            ctx.seq_step()
        extra = 0
        while ctx.get_schedule()["settling"] and extra < 280:
            ctx.seq_push_pair(0, lefts[order[(24 + extra) % 8]], rights[order[(24 + extra) % 8]])
            ctx.seq_step()
            extra += 1
        ctx.seq_sync()
        ctx.seq_reset(-1)
        t0 = time.perf_counter()
        for i in range(n_steps):
            ctx.seq_push_pair(0, lefts[order[i % 8]], rights[order[i % 8]])
            ctx.seq_step()
        ctx.seq_sync()
        dt = time.perf_counter() - t0
        rows, info = ctx.seq_get_trajectory(0)
        out["one_sequence_pipelined_%s" % tag] = {"frames_per_s": n_steps / dt, "ms_per_frame": 1e3 * dt / n_steps,
                                                   "frames": int(len(rows)), "points_per_frame": float(np.mean(info[:, 0])),
                                                   "schedule": ctx.get_schedule()}
    ctx.batch_set_detect_params()
    out["value"] = out["track_frame_ms_2000"]["median"]
    return out


def pyramid_roofline(ctx, w, h, n_images, stage_ms, profiled_workload=None):
    """the pyramid stage against the HBM roof, next to the LK entry: its ALGORITHMIC bytes (every level read
    once, every level >= 1 written once) and the bytes the stage moves BY DESIGN -- it also stores a 4-byte Scharr pixel
    (2 x int16) per pyramid pixel, which 8d's model does not count"""
    from visual_odom_amd import _lib
    ms = float(stage_ms[_lib.STAGE_NAMES.index("pyramid")])
    algo = float(ctx.model_bytes(w, h, 0)[0]) / 4.0 * n_images      # model_bytes()[0] is per frame = 4 images
    lv, cw, ch, px = 0, w, h, 0
    max_level = ctx.get_params().lk_max_level
    while True:
        px += cw * ch
        nw, nh = (cw + 1) // 2, (ch + 1) // 2
        if lv == max_level or nw <= 21 or nh <= 21:
            break
        cw, ch, lv = nw, nh, lv + 1
    moved = algo + px * 4.0 * n_images   # + the 4-byte Scharr pixel per pyramid pixel (the fused pass reads a level ONCE: round
                                         # 3's separate Scharr kernel read every level a second time, + px per image)
    return {"pyramid": {"bound": "hbm", "stage_ms": ms, "algorithmic_bytes": algo, "designed_bytes": moved,
                        "achieved": algo / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                        "achieved_designed": moved / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                        "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": algo / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS if ms > 0 else 0.0,
                        "frac_designed": moved / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS if ms > 0 else 0.0,
                        # HBM bytes of the stage from the PMC passes (imported, like roofline.traffic): against designed_bytes
                        "traffic": measured_pyramid_traffic(profiled_workload, n_images) if profiled_workload else None}}


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, exactly as the driver would"""
    import socket
    import subprocess
    if not args.selftest_replicas:
        import torch
        n = torch.cuda.device_count()
        if n < args.gpus:
            raise SystemExit("bench.py: --gpus %d but %d GPU(s) visible" % (args.gpus, n))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def _pmc_record(name, workload, frames):
    """the committed PMC record of (workload, frames per step): the headline's sits at the top level of profiles/<name>, every
    leg's under "legs" (tools/pmc_legs.py)"""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    for r in [rec] + list(rec.get("legs") or []):
        if r.get("workload") == workload and r.get("frames_per_step") == frames:
            return r
    return None


def measured_issue(workload, frames):
    """VALU issue figures of the LK launch from the committed PMC pass (profiles/lk_issue.json): the bound this
    kernel actually runs at; None when no pass matches this configuration."""
    rec = _pmc_record("lk_issue.json", workload, frames)
    return {k: rec[k] for k in rec if k not in ("workload", "frames_per_step", "legs")} if rec else None


def valu_issue_frac(workload, frames, points_per_launch, launch_ms):
    """(frac, model_ratio).  frac = issue-FLOOR cycles of the launch's VALU instructions / the cycles THIS run's launch had:
    instructions per feature from the committed PMC pass (profiles/lk_issue.json), each priced at the floor of its opcode
    class (2.46 cycles for the class the micro-benchmark measures below 3, 4.0 = one wave64 pass for the rest, 8.0 for the
    half-rate class; tools/isa_histogram.py --floor on the shipped kernel: 3.65 per instruction for the cheapest of its three
    block groups), over 1024 SIMDs x clock x the HIP-event duration.  No issue schedule can beat that price, so frac <= 1 BY
    CONSTRUCTION (round 6) -- it says how much of the launch is accounted for by issuing this
    instruction stream as fast as the hardware can, NOT that the stream is minimal.  model_ratio = the same with the
    micro-benchmark's own per-opcode costs (4.05 for the hot loop's mix): a model, a few per cent pessimistic for this mix,
    which may exceed 1 and is reported for continuity with rounds 2-5 only.  (None, None) without a matching pass."""
    rec = measured_issue(workload, frames)
    if not rec or launch_ms <= 0:
        return None, None
    try:
        per_feature = float(rec["valu_instructions_per_feature"])
        floor = float(rec.get("issue_floor_cycles_per_valu_instruction", 3.65))
        model = float(rec["issue_cost_bound_cycles_per_valu_instruction"])
        clock_hz = float(rec.get("shader_clock_mhz", 2400.0)) * 1e6
    except (KeyError, TypeError, ValueError):
        return None, None
    per_cycle = per_feature * points_per_launch / (1024.0 * clock_hz) / (launch_ms * 1e-3)
    return min(1.0, floor * per_cycle), model * per_cycle


def measured_pyramid_traffic(workload, n_images):
    """HBM bytes per pyramid stage from the same committed PMC passes (profiles/lk_traffic.json, `pyramid_stage`); None when the
    passes were not taken at this configuration"""
    try:
        with open(os.path.join(ROOT, "profiles", "lk_traffic.json")) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    for r in [rec] + list(rec.get("legs") or []):
        ps = r.get("pyramid_stage") or {}
        if r.get("workload") == workload and ps.get("images_per_step") == n_images:
            return ps.get("hbm_bytes_per_step")
    return None


def measured_traffic(workload, frames):
    """HBM bytes per LK launch from the committed rocprofv3 PMC passes (profiles/lk_traffic.json,
    written by tools/profile_summary.py from separate --pmc runs of this same command, with the gfx950
    FETCH_SIZE correction of MI355X_MICROARCH.md); None when no pass matches this configuration."""
    rec = _pmc_record("lk_traffic.json", workload, frames)
    return rec.get("hbm_bytes_per_launch") if rec else None


def cpu_baseline(lefts, rights, pts, world, n_frames, stages, per_bucket=1):
    """The oracle (scalar C port of the reference's OpenCV CPU path, OpenMP over features) timed on
    this host's cores on a bounded sample of the same frames.  Checker code is only *timed* here.
    Reported: the best OpenMP width (median of 3 repeats per width decides) AND the 1-thread figure."""
    from oracle import oracle as orc
    orc.build()
    P_l, P_r = world.proj_matrices()
    K = world.K()
    h, w = lefts[0].shape
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1

    def one_frame(k, threads):
        p = pts[k]
        if stages == "detect+full":  # the reference's own head of matchingFeatures: FAST + bucketing
            fast = orc.fast_detect(lefts[k], 20, True)
            p, _ = orc.bucketing_features(h, w, fast, np.zeros(len(fast), np.int32), h // 10, per_bucket)
        r = orc.circular_matching(lefts[k], rights[k], lefts[k + 1], rights[k + 1], p, nthreads=threads)
        (l0, r0, l1, r1), _ = orc.check_valid_and_remove(r["l0"], r["r0"], r["l1"], r["r1"], r["l0_ret"])
        if stages != "lk" and len(l0) >= 5:
            xyz = orc.triangulate(P_l, P_r, l0, r0)
            orc.solve_pnp_ransac(xyz, l1, K)

    def timed(threads, frames, budget):
        t0 = time.perf_counter()
        n = 0
        while True:
            for k in range(frames):
                one_frame(k, threads)
            n += frames
            if time.perf_counter() - t0 >= budget:
                break
        return n / (time.perf_counter() - t0)

    # pick the OpenMP width that is fastest on this host (a cgroup may expose fewer cores than it lists):
    # median of three single-frame repeats per width
    widths = sorted({1, min(avail, 8), min(avail, 32), min(avail, 64), min(avail, 128), avail})
    med = {}
    for t in widths:
        one_frame(0, t)  # warm-up (first call pays library / thread-pool start-up)
        reps = []
        for _ in range(3):
            t0 = time.perf_counter()
            one_frame(0, t)
            reps.append(time.perf_counter() - t0)
        med[t] = float(np.median(reps))
    best_t = min(med, key=med.get)
    best = timed(best_t, n_frames, 10.0)
    single = timed(1, min(n_frames, 2), 6.0)
    # (rounds 3-4 also timed a `-O3 -march=native` build of the same sources in OpenCV's x86 accumulation order as "native_build":
    # it emulates SIMD lanes with scalar volatile floats, ran SLOWER than this port (43.8 vs 53.7 frames/s) and said nothing about
    # OpenCV's speed -- dropped, VERDICT r04 weak 9.  The figure below is a port's, reported, never a target.)
    return {"value": best, "unit": "frames/s", "cores": best_t, "kind": "port",
            "single_thread": {"value": single, "unit": "frames/s", "cores": 1},
            "width_sweep_s_per_frame": {str(t): med[t] for t in widths},
            "sample": "passes over %d frame quadruples of the same workload for ~10 s (best width) and ~6 s (1 thread); "
                      "oracle = scalar C restatement of the OpenCV CPU path (no SIMD), OpenMP over features, width %d "
                      "chosen by the median of 3 repeats per width (host lists %d CPUs)" % (n_frames, best_t, avail)}


if __name__ == "__main__":
    main()
