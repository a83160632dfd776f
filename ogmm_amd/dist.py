"""One process per GPU, pairs sharded by global pair id: no data-path collective in inference; the
only collectives are the barrier around a timed region and the max-over-ranks of the elapsed time.  Backend "nccl" is
RCCL over xGMI on ROCm; "gloo" is used by the CPU tests of this logic."""
import os

import torch


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_pairs(rank, world, pairs_per_rank, first_pair=0):
    """Weak scaling: rank r owns global pair ids [first + r*P, first + (r+1)*P)."""
    lo = first_pair + rank * pairs_per_rank
    return lo, lo + pairs_per_rank


def init(backend, rank, world, device=None, force=False):
    """process group for `world` ranks; None at world 1 (no collective is needed then) unless `force`: a one-rank group, so that the collective code paths
    -- RCCL's initialisation, the flat gradient bucket's all-reduce, the buffer broadcast -- run on a single GPU too (tests/test_hip_train.py)."""
    if world <= 1 and not force:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def barrier(dist, cuda=True):
    if cuda:
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    if cuda:
        torch.cuda.synchronize()


def max_over_ranks(dist, value, device="cpu"):
    if dist is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rows(dist, local, world):
    """All-gather of equally shaped per-rank result rows (used by tests to check shard invariance)."""
    if dist is None:
        return local
    out = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(out, local)
    return torch.cat(out, 0)
