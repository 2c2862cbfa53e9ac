"""Batch sharding across the GPUs of one node.

Every FFT of a batch is independent (one workgroup tile per 4096 elements, no inter-tile
communication), so the multi-GPU path is an embarrassingly parallel contiguous batch split:
rank g of G owns FFTs [first, first + count) and runs the identical kernel on its slab; there is
no data-path collective.  The only communication is the reduction of per-rank timings (MAX) and
error counts (SUM), which bench.py does with torch.distributed (RCCL on GPUs, gloo in the CPU tests).
"""


def shard_range(nffts: int, rank: int, world: int):
    """Contiguous, balanced split: returns (first_fft, count) of `rank`'s slab.  The first
    nffts % world ranks get one extra FFT; slabs tile [0, nffts) exactly, in rank order."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world: {rank}/{world}")
    if nffts < 0:
        raise ValueError("nffts must be >= 0")
    base, extra = divmod(nffts, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def slab_bytes(fft_size: int, count: int, bytes_per_element: int = 8) -> int:
    return fft_size * count * bytes_per_element


def agree_on_fast_group(dist, try_fast):
    """ONE decision for all ranks about the fast communicator (RCCL), taken over the group that is already up (gloo: CPU, TCP to
    the launcher's store -- it needs nothing of the GPUs).  Every rank calls try_fast(), which returns the fast process group
    after a first collective on it or raises; the ranks then all-reduce (MIN) a flag over the DEFAULT group, so either every rank
    uses the fast group or none does.  Round 5 let each rank fall back on its own inside a try: RCCL up on some ranks and not on
    others left two process groups that never matched -- a hang.  (A rank whose try_fast() BLOCKS because a peer never joined is
    ended by the fast group's own timeout: the job fails, it does not hang.)  Returns (group or None, error text of this rank or None)."""
    import torch

    group, error = None, None
    try:
        group = try_fast()
    except Exception as e:      # noqa: BLE001 -- whatever went wrong, the verdict must still reach the other ranks
        error = f"{type(e).__name__}: {e}"
    flag = torch.tensor([1 if error is None else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return (group if int(flag.item()) == 1 else None), error


def reduce_stats(dist, device, wall_s: float, kernel_ms: float, errors: int = 0, group=None):
    """MAX over ranks of (wall, kernel) times and SUM of error counts.  `dist` is
    torch.distributed (initialised) or None for a single process; group: the process group (None = the default one)."""
    import torch

    t = torch.tensor([wall_s, kernel_ms], dtype=torch.float64, device=device)
    e = torch.tensor([errors], dtype=torch.int64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        dist.all_reduce(e, op=dist.ReduceOp.SUM, group=group)
    return t[0].item(), t[1].item(), int(e.item())


def gather_stats(dist, device, values, group=None):
    """Every rank's list of floats, in rank order: [[rank 0's values], [rank 1's values], ...] on every rank (one
    all_gather of a small tensor; a single process returns [values])."""
    import torch

    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if dist is None:
        return [t.tolist()]
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t, group=group)
    return [p.tolist() for p in parts]
