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


def plan_affinity(avail_cpus, local_rank, local_world, numa_cpus=None, ranks_on_my_node=None):
    """Which host cores rank `local_rank` of `local_world` ranks on this node keeps (a pure function: tests/test_replicas_gloo.py).
    One process per GPU also means one image decoder / staging thread / CPU-side validation per GPU; left alone, eight ranks'
    OpenMP teams of 32 land on the same cores and on the wrong socket.
      * numa_cpus (the cores of the NUMA node this rank's GPU hangs off) and ranks_on_my_node (the local ranks whose GPUs share
        that node, sorted) known: the node's available cores split contiguously among those ranks;
      * else: all available cores split contiguously by local rank.
    A slice is never empty: with more ranks than cores the ranks share cores round-robin."""
    avail = sorted(set(int(c) for c in avail_cpus))
    if not avail or local_world <= 1:
        return avail
    pool, idx, n = avail, local_rank, local_world
    if numa_cpus and ranks_on_my_node and local_rank in ranks_on_my_node:
        node = sorted(set(int(c) for c in numa_cpus) & set(avail))
        if node:
            pool, idx, n = node, sorted(ranks_on_my_node).index(local_rank), len(ranks_on_my_node)
    if len(pool) < n:
        return [pool[idx % len(pool)]]
    per = len(pool) // n
    return pool[idx * per:(idx + 1) * per]


def _parse_cpulist(text):
    out = []
    for part in text.strip().split(","):
        if "-" in part:
            a, b = part.split("-")
            out.extend(range(int(a), int(b) + 1))
        elif part:
            out.append(int(part))
    return out


def gpu_numa_cpus(pci_bus_id):
    """(node, [cpus]) -- the NUMA node of the GPU at `pci_bus_id` ("0000:c1:00.0") and its cores, from sysfs; (None, None) when
    the kernel does not say (no such device file, node -1)"""
    try:
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % pci_bus_id.lower()).read())
        if node < 0:
            return None, None
        return node, _parse_cpulist(open("/sys/devices/system/node/node%d/cpulist" % node).read())
    except (OSError, ValueError):
        return None, None


def pin_rank(local_rank, local_world, numa_of_local_rank=None, omp_cap=32):
    """sched_setaffinity of this process to its slice of the node's cores (plan_affinity) and OMP_NUM_THREADS to match -- call it
    before the first OpenMP runtime of the process starts.  numa_of_local_rank: {local rank: (node, [cpus])} when known.
    Returns what was done (bench.py reports it per rank); never raises."""
    info = {"local_rank": local_rank, "local_world": local_world, "pinned": False}
    try:
        avail = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return info
    numa_cpus = ranks_here = None
    if numa_of_local_rank and numa_of_local_rank.get(local_rank, (None, None))[0] is not None:
        node, numa_cpus = numa_of_local_rank[local_rank]
        ranks_here = [r for r, (nd, _) in numa_of_local_rank.items() if nd == node]
        info["numa_node"] = node
    mine = plan_affinity(avail, local_rank, local_world, numa_cpus, ranks_here)
    if local_world > 1 and mine:
        try:
            os.sched_setaffinity(0, mine)
            info["pinned"] = True
        except OSError:
            pass
    info["cpus"] = len(mine)
    info["first_cpu"], info["last_cpu"] = (mine[0], mine[-1]) if mine else (None, None)
    if local_world > 1:
        os.environ["OMP_NUM_THREADS"] = str(max(1, min(len(mine), omp_cap)))
    return info


def init(backend=None):
    """process-group init for the aggregation only; returns torch.distributed or None for 1 rank"""
    rank, local_rank, world_size = rank_info()
    if world_size <= 1 and not (os.environ.get("VO_DIST_FORCE") == "1" and "RANK" in os.environ):
        return None   # (VO_DIST_FORCE=1 under torchrun with ONE rank: the N > 1 code path incl. its RCCL collectives on a 1-GPU box)
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
