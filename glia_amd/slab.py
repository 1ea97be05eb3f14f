"""z-slab partition of one volume across ranks: geometry and the exchange of partial records.

The accumulation pass shards by z with one halo plane per cut; every statistic is a commutative monoid, so partial records
with the same key (region: label; directed pair: (a, b)) are simply reduced.  BASELINE.json's north star asks for "RCCL over
xGMI exchanging only the cross-slab boundary regions", and that is the shape of the exchange here:

  1. every rank flags the records that can have a counterpart elsewhere -- their label occurs on a plane next to a cut
     (glia_hmt_rag_cut_flags);
  2. only the flagged records take a KEYED OWNER EXCHANGE: owner = hash(label) mod N (a pair goes with its first label), one
     unpadded point-to-point message per (source, destination) -- xGMI is point-to-point, all seven links carry traffic at once;
     the owner reduces what it receives (glia_hmt_rag_merge);
  3. the reduced cut records and the untouched interior records travel ONCE to the rank that runs the merge loop (the loop
     does not shard), which reduces everything by key a last time -- so a record the flags missed (a label that occurs in two
     slabs without touching a cut plane: a disconnected "region") is still combined: the flags decide the route, never the result.

At 1024^3 / S=16 / 8 slabs a rank holds ~33 k region and ~470 k pair records (~64 MB) of which ~15 % are flagged; the previous
exchange (all_gather of everything, padded to the largest rank) moved 8 x 64 MB to every rank.

Communication is torch.distributed only (backend "nccl" = RCCL on the GPU box, "gloo" in the CPU tests); everything below but
`exchange_cut_records` is device-agnostic, so the N > 1 plumbing runs under gloo on CPU tensors."""
import torch
import torch.distributed as dist

KEYS = ("rlabel", "rrec", "pa", "pb", "prec")      # channel 0; maps with several image channels add "rrec1", "prec1", ...


def _keys(t):
    """keys of a record dictionary in a fixed order: region-shaped first (they start with "r"), then pair-shaped"""
    return sorted(t.keys(), key=lambda k: (0 if k.startswith("r") else 1, len(k), k))


def _is_region(k):
    return k.startswith("r")


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


def owner_of(label, world):
    """The rank that reduces the records of a label (uint32 payload in an int32 tensor): a multiplicative hash, mod world."""
    h = (label.to(torch.int64) & 0xFFFFFFFF) * 2654435761
    return ((h >> 11) & 0x7FFFFFFF) % world


def select_records(t, rmask, pmask):
    """The sub-dictionary of a record dictionary picked by two boolean masks (regions, pairs)."""
    return {k: t[k][rmask if _is_region(k) else pmask] for k in _keys(t)}


def concat_records(parts, like):
    if not parts:
        return {k: like[k][:0] for k in _keys(like)}
    return {k: torch.cat([p[k] for p in parts]) for k in _keys(like)}


def pack_records(t):
    """One flat int32 message: [R, P, region-shaped arrays ..., pair-shaped arrays ...]."""
    head = torch.tensor([t["rlabel"].numel(), t["pa"].numel()], dtype=torch.int32, device=t["rlabel"].device)
    return torch.cat([head] + [t[k].reshape(-1).to(torch.int32) for k in _keys(t)])


def unpack_records(buf, like):
    R, P = int(buf[0].item()), int(buf[1].item())
    o = 2
    out = {}
    for k in _keys(like):
        rows = R if _is_region(k) else P
        shape = (rows,) + tuple(like[k].shape[1:])
        n = rows * (like[k].shape[1] if like[k].dim() > 1 else 1)
        out[k] = buf[o:o + n].reshape(shape).contiguous()
        o += n
    return out


def exchange_variable(send, group=None):
    """send[d] = 1-D tensor for rank d (any length, may be empty).  Returns (recv list, bytes sent to other ranks).
    One all_gather of the length matrix, then one unpadded point-to-point message per non-empty (source, destination)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev, dt = send[0].device, send[0].dtype
    lens = torch.tensor([s.numel() for s in send], dtype=torch.int64, device=dev)
    allv = [torch.zeros_like(lens) for _ in range(world)]
    dist.all_gather(allv, lens, group=group)
    n_from = [int(allv[s][rank].item()) for s in range(world)]
    recv = [torch.empty(n_from[s], dtype=dt, device=dev) for s in range(world)]
    recv[rank] = send[rank]
    ops = []
    for d in range(world):
        if d != rank and send[d].numel():
            ops.append(dist.P2POp(dist.isend, send[d].contiguous(), d, group))
    for s in range(world):
        if s != rank and n_from[s]:
            ops.append(dist.P2POp(dist.irecv, recv[s], s, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    sent = sum(send[d].numel() * send[d].element_size() for d in range(world) if d != rank)
    return recv, sent


def exchange_cut_records(t, rcut, pcut, reduce_fn, loop_owner=0, group=None):
    """Steps 2 and 3 of the module docstring on a record dictionary `t` (tensors of this rank's partial map) with boolean
    cut masks.  reduce_fn(list of record dictionaries) -> record dictionary reduced by key (glia_hmt_rag_merge on the GPU).
    Returns (whole-volume records on the loop owner | None, statistics)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    interior = select_records(t, ~rcut, ~pcut)
    cut = select_records(t, rcut, pcut)
    # 2. keyed owner exchange of the cut records
    r_owner, p_owner = owner_of(cut["rlabel"], world), owner_of(cut["pa"], world)
    send = [pack_records(select_records(cut, r_owner == d, p_owner == d)) for d in range(world)]
    recv, sent_cut = exchange_variable(send, group)
    mine = reduce_fn([unpack_records(b, t) for b in recv])
    # 3. everything once to the loop owner
    outgoing = pack_records(concat_records([interior, mine], t))
    send = [outgoing if d == loop_owner else outgoing[:0] for d in range(world)]
    recv, sent_final = exchange_variable(send, group)
    stats = dict(records=int(t["rlabel"].numel() + t["pa"].numel()), cut_records=int(cut["rlabel"].numel() + cut["pa"].numel()),
                 bytes_sent_cut_exchange=int(sent_cut), bytes_sent_to_loop_owner=int(sent_final))
    if rank != loop_owner:
        return None, stats
    return reduce_fn([unpack_records(b, t) for b in recv if b.numel()]), stats


def exchange_and_merge(ctx, partial, labels_slab=None, z_begin=None, z_end=None, loop_owner=0, group=None):
    """The slab exchange on the GPU: `partial` = this rank's hmt.RegionMap built with slab=...; labels_slab / z_begin / z_end =
    the label planes it was built from.  Returns (hmt.RegionMap of the whole volume on the loop owner | None, statistics)."""
    from . import hmt
    t = partial.to_tensors()
    rcut, pcut = partial.cut_flags(labels_slab, z_begin, z_end)

    def reduce_fn(parts):
        maps = [hmt.RegionMap.from_tensors(ctx, partial, p) for p in parts if p["rlabel"].numel() or p["pa"].numel()]
        if not maps:
            return {k: t[k][:0] for k in _keys(t)}
        merged = hmt.RegionMap.merge(ctx, maps)
        out = merged.to_tensors()
        for m in maps:
            m.close()
        merged.close()
        return out

    whole, stats = exchange_cut_records(t, rcut, pcut, reduce_fn, loop_owner, group)
    if whole is None:
        return None, stats
    return hmt.RegionMap.from_tensors(ctx, partial, whole), stats
