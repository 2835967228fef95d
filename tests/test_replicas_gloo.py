"""N > 1 path on CPU: two gloo ranks run the replica aggregation bench.py uses
(max-over-ranks time, summed frames) and the sequence sharding of BASELINE config 5."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from visual_odom_amd import replicas
    dist = replicas.init("gloo")
    seqs = replicas.shard_sequences(8, rank, world)
    elapsed, frames = replicas.aggregate(dist, 1.0 + rank, 100 * len(seqs))
    dist.barrier()
    q.put((rank, seqs, elapsed, frames))
    dist.destroy_process_group()


def test_two_rank_aggregation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5, 7]
    for _, _, elapsed, frames in res:
        assert elapsed == 2.0 and frames == 800  # MAX over ranks, SUM over ranks


def test_single_rank_is_passthrough(monkeypatch):
    from visual_odom_amd import replicas
    monkeypatch.setenv("WORLD_SIZE", "1")
    assert replicas.init() is None
    assert replicas.aggregate(None, 0.5, 64) == (0.5, 64)
    assert replicas.shard_sequences(8, 0, 1) == list(range(8))


def test_bench_multi_rank_launch_line():
    """bench.py's N > 1 branch exactly as the driver launches it (python -m torch.distributed.run --nnodes=1
    --nproc-per-node 2 --master-addr 127.0.0.1 ... bench.py --gpus 2 ...), on CPU through --selftest-replicas: rank
    discovery from the environment, the gloo process group, barriers, MAX-over-ranks time and summed frames, one JSON
    line from rank 0"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--frames", "16", "--selftest-replicas"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["scaling"] == "weak"
    assert abs(out["value"] - 2 * 16 * 4 / 1.25) < 1e-9      # frames of both ranks / the slower rank's time
    assert "SELFTEST" in out["data"]


def test_eight_rank_selftest_with_config5_and_core_slices():
    """world size 8 on CPU: the driver's launch line with eight gloo ranks through --selftest-replicas --
    the headline's aggregation, the config-5 leg's (one sequence per GPU: eight per-GPU figures gathered in rank order,
    aggregate = frames of all ranks / the slowest rank's time) and the per-rank core slices (disjoint, OMP width = slice)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1",
           "--frames", "16", "--selftest-replicas"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["ranks"] == 8
    assert abs(out["value"] - 8 * 16 * 4 / (1.0 + 0.25 * 7)) < 1e-9
    c5 = out["configs"][0]
    assert c5["baseline_config"] == 5 and len(c5["per_gpu_value"]) == 8
    assert all(abs(v - 200 / (0.5 + 0.01 * k)) < 1e-9 for k, v in enumerate(c5["per_gpu_value"]))
    assert abs(c5["value"] - 8 * 200 / 0.57) < 1e-9
    hc = out["host_cores_per_rank"]
    n_avail = len(os.sched_getaffinity(0))
    if n_avail >= 8:
        per = n_avail // 8
        assert hc["cpus"] == [per] * 8 and len(set(hc["first_cpu"])) == 8      # eight disjoint slices
        assert int(hc["omp_num_threads"]) == min(per, 32)


def test_plan_affinity_slices():
    from visual_odom_amd import replicas
    avail = list(range(4, 68))                                   # a cgroup that starts at core 4
    slices = [replicas.plan_affinity(avail, r, 8) for r in range(8)]
    assert all(len(s) == 8 for s in slices) and sorted(sum(slices, [])) == avail and slices[0][0] == 4 and slices[7][-1] == 67
    assert replicas.plan_affinity(avail, 0, 1) == avail
    # GPUs 0-3 on NUMA node 0 (cores 0-31), 4-7 on node 1 (32-63): ranks split THEIR node's available cores
    node0, node1 = list(range(0, 32)), list(range(32, 64))
    s5 = replicas.plan_affinity(avail, 5, 8, node1, [4, 5, 6, 7])
    assert s5 == list(range(40, 48))
    s0 = replicas.plan_affinity(avail, 0, 8, node0, [0, 1, 2, 3])
    assert s0 == list(range(4, 11))                              # node 0 has 28 available cores: 7 each
    assert replicas.plan_affinity([3, 9], 5, 8) == [9]           # more ranks than cores: shared, never empty
    assert replicas._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]


def _run_bench(args, env=None, launcher_ranks=0):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable]
    if launcher_ranks:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(launcher_ranks), "--master-addr",
                "127.0.0.1", "--master-port", str(_free_port())]
    cmd += [os.path.join(root, "bench.py")] + args
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root, env=e)


def test_bench_gpus_flag_cannot_misreport():
    """`python bench.py --gpus N` launched WITHOUT torch.distributed.run starts its N ranks itself
    (and the line says n_gpus = N); a --gpus that contradicts the launcher's WORLD_SIZE is refused; more ranks than
    GPUs is refused -- `n_gpus: 1` is never printed for `--gpus 8`"""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--frames", "16", "--selftest-replicas"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and abs(out["value"] - 2 * 16 * 4 / 1.25) < 1e-9
    # the launcher says 2 ranks, the flag says 4
    r = _run_bench(["--gpus", "4", "--selftest-replicas"], launcher_ranks=2)
    assert r.returncode != 0 and "contradicts" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    # no GPU in this container: a real (non-selftest) multi-GPU request must fail, not fall back to fewer GPUs
    if not torch.cuda.is_available():
        r = _run_bench(["--gpus", "8"])
        assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
