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
