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
