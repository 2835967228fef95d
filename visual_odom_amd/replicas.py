"""Multi-GPU = replicas.  The reference has no distributed code and the hot path has no exchange
step: frames of one sequence are chained through the feature set, sequences are independent
(SURVEY.md 8e).  So N GPUs run N independent sequences (or independent shards of a frame batch),
one process per GPU, and the only cross-rank traffic is the host-side aggregation of the timing /
frame counters.  No RCCL collective sits on the data path."""
import os


def rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_sequences(n_sequences, rank, world_size):
    """sequence s -> rank s % world_size (BASELINE config 5: KITTI 00-07 one per GPU)"""
    return [s for s in range(n_sequences) if s % world_size == rank]


def init(backend=None):
    """process-group init for the aggregation only; returns torch.distributed or None for 1 rank"""
    rank, local_rank, world_size = rank_info()
    if world_size <= 1:
        return None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        # nccl (= RCCL) when every rank owns a GPU; VO_DIST_BACKEND=gloo lets two ranks share one GPU in a
        # smoke test of this path (RCCL refuses duplicate devices)
        backend = os.environ.get("VO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if not dist.is_initialized():
        dist.init_process_group(backend)
    return dist


def aggregate(dist, elapsed_s, frames_done, device=None):
    """whole-job numbers: (max elapsed over ranks, total frames over ranks)"""
    if dist is None:
        return float(elapsed_s), int(frames_done)
    import torch
    if dist.get_backend() == "gloo":
        device = None
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    f = torch.tensor([int(frames_done)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    return float(t.item()), int(f.item())


def gather_values(dist, value, device=None):
    """[value of rank 0, value of rank 1, ...] on every rank (per-GPU frames/s next to the aggregate: BASELINE config 5)"""
    if dist is None:
        return [float(value)]
    import torch
    if dist.get_backend() == "gloo":
        device = None
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]
