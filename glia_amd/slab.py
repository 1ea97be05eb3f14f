"""z-slab partition of one volume across ranks: geometry and the keyed exchange of partial records.

The accumulation pass shards by z with one halo plane per cut; every statistic is a commutative monoid, so the only
communication is ONE exchange of the compact partial records (regions keyed by label, directed pairs by (a,b)) followed
by glia_hmt_rag_merge.  The greedy loop does not shard: the merged map lives on the gathering rank(s).
Communication is torch.distributed only (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests); the
functions below are device-agnostic so the N>1 plumbing is testable without GPUs."""
import torch
import torch.distributed as dist


def slab_bounds(nz, world, rank):
    """Global planes [z0, z1) owned by `rank`: as even as possible, in rank order."""
    base, rem = divmod(nz, world)
    z0 = rank * base + min(rank, rem)
    return z0, z0 + base + (1 if rank < rem else 0)


def slab_with_halo(nz, world, rank):
    """(first plane to hand in, one past the last, z_begin, z_end) -- the owned range plus one halo plane per cut;
    z_begin / z_end are relative to the first plane handed in (the slab argument of hmt.RegionMap)."""
    z0, z1 = slab_bounds(nz, world, rank)
    lo = max(z0 - 1, 0)
    hi = min(z1 + 1, nz)
    return lo, hi, z0 - lo, z1 - lo


def all_gather_variable(t, group=None):
    """all_gather of tensors whose first dimension differs between ranks: returns the list of every rank's tensor.
    Two collectives: the sizes (one int64 per rank) and the payload padded to the largest size."""
    world = dist.get_world_size(group)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(max(sizes), 1)
    pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return [o[:s].contiguous() for o, s in zip(out, sizes)]


def exchange_and_merge(ctx, partial, group=None):
    """Every rank contributes its partial hmt.RegionMap; every rank returns the merged map of the whole volume
    (the loop owner uses it, the others may drop it).  Records cross the wire once."""
    from . import hmt
    mine = partial.to_tensors()
    parts = {k: all_gather_variable(v, group) for k, v in mine.items()}
    world = dist.get_world_size(group)
    maps = [hmt.RegionMap.from_tensors(ctx, partial, {k: parts[k][r] for k in parts}) for r in range(world)]
    merged = hmt.RegionMap.merge(ctx, maps)
    for m in maps:
        m.close()
    return merged
