"""-m gpu: the HIP accumulation pass (K1-K3) vs the oracle's voxel-list RAG, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU test run without a GPU"
    return torch


@pytest.fixture(scope="module")
def ctx():
    from glia_amd import hmt
    c = hmt.Context(0)
    yield c
    c.close()


def _oracle_rag(labels, pb, bins, lo, hi, thr):
    from oracle import pyoracle as O
    rag = O.Rag(labels, only_contour=False)
    lab, npts, nborder = rag.regions()
    rs = rag.region_stats(pb)
    a, b, n = rag.pairs()
    ps = rag.pair_stats(pb)
    return dict(lab=lab, npts=npts, nborder=nborder, rsum=rs[0], rsq=rs[1], rmin=rs[2], rmax=rs[3], lo=rs[4], hi=rs[5],
                a=a, b=b, n=n, psum=ps[0], psq=ps[1], pmin=ps[2], pmax=ps[3])


def _np_hist(vals, bins, lo, hi):
    """util/image_stats.hxx:12-37 in numpy (float32 values against double bounds)."""
    interval = (hi - lo) / bins
    bounds = np.empty(bins)
    bounds[0] = interval
    for i in range(1, bins):
        bounds[i] = bounds[i - 1] + interval
    v = vals.astype(np.float64)
    h = np.zeros(bins, np.int64)
    inside = (v > lo) & (v < hi)
    c = (v[inside][:, None] >= bounds[None, :]).sum(1)
    for k in range(bins):
        h[k] += (c == k).sum()
    h[0] += (~inside & (v <= lo)).sum()
    h[bins - 1] += (~inside & ~(v <= lo)).sum()
    return h


CASES = [
    ((32, 32, 32), 8, 16, 0, 8),
    ((40, 36, 28), 6, 12, 1, 8),       # ragged: scalar load path, partial tiles
    ((64, 64, 256), 8, 32, 0, 8),      # vector load path
    ((96, 96), 8, 32, 0, 8),           # 2D
    ((33, 70, 300), 5, 20, 1, 16),     # 16 bins, ragged
    ((3, 5, 7), 2, 4, 0, 8),           # tiny
    ((128, 128, 256), 8, 32, 0, 8),    # 4 x 4 x 4 tiles: the eight in the middle are "inner" tiles (instance without validity logic)
    ((100, 128, 200), 6, 24, 1, 16),   # inner tiles next to ragged ones, 16 bins, f32 pb
]


@pytest.mark.parametrize("shape,S,G,variant,bins", CASES)
def test_rag_matches_oracle(ctx, shape, S, G, variant, bins):
    torch = _torch()
    from glia_amd import hmt
    from oracle import pyoracle as O
    labels, pb = O.synth(shape, S, G, variant=variant)
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda()
    d_pb = torch.from_numpy(pb).cuda()
    thr = (0.2, 0.5, 0.8)
    lo, hi = 0.0, 1.0
    cfg = hmt.make_config(d_pb, rb=[(d_pb, bins, lo, hi)], thresholds=thr)
    rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, cfg=cfg)
    ref = _oracle_rag(labels, pb, bins, lo, hi, thr)
    reg, par = rm.regions(), rm.pairs()
    assert (reg["label"] == ref["lab"]).all()
    assert (reg["count"] == ref["npts"]).all()
    assert (reg["border"] == ref["nborder"]).all()
    assert (reg["lo"] == ref["lo"]).all() and (reg["hi"] == ref["hi"]).all()
    assert (reg["min"] == ref["rmin"]).all() and (reg["max"] == ref["rmax"]).all()
    assert (par["a"] == ref["a"]).all() and (par["b"] == ref["b"]).all()
    assert (par["count"] == ref["n"]).all()
    assert (par["min"] == ref["pmin"]).all() and (par["max"] == ref["pmax"]).all()
    if variant == 0:   # Q8 pb: every sum is exact in double -> bit-identical whatever the order
        assert (reg["sum"] == ref["rsum"]).all() and (reg["sumsq"] == ref["rsq"]).all()
        assert (par["sum"] == ref["psum"]).all() and (par["sumsq"] == ref["psq"]).all()
    else:              # F32 pb: summation order differs -> tolerance of the north star (1e-5), far tighter here
        assert np.allclose(reg["sum"], ref["rsum"], rtol=1e-12) and np.allclose(reg["sumsq"], ref["rsq"], rtol=1e-12)
        assert np.allclose(par["sum"], ref["psum"], rtol=1e-12) and np.allclose(par["sumsq"], ref["psq"], rtol=1e-12)
    # first voxel (raster order) of every label
    flat = labels.reshape(-1)
    first = {}
    for i in range(flat.size - 1, -1, -1):
        first[int(flat[i])] = i
    assert [first[int(l)] for l in reg["label"]] == reg["first"].tolist()
    # histograms + threshold counts: recomputed with numpy from the voxel sets
    for i, l in enumerate(reg["label"][:50]):
        assert (reg["hist"][i] == _np_hist(pb[labels == l], bins, lo, hi)).all()
    assert (reg["hist"].sum(1) == reg["count"]).all()
    assert (par["hist"].sum(1) == par["count"]).all()
    # threshold counts are monotone and bounded by the pair's voxel count
    assert (par["thr"][:, 0] <= par["count"]).all() and (np.diff(par["thr"], axis=1) <= 0).all()
    rm.close()


def _check_against_oracle(rm, labels, pb, bins, thr):
    ref = _oracle_rag(labels, pb, bins, 0.0, 1.0, thr)
    reg, par = rm.regions(), rm.pairs()
    assert (reg["label"] == ref["lab"]).all() and (reg["count"] == ref["npts"]).all() and (reg["border"] == ref["nborder"]).all()
    assert (reg["lo"] == ref["lo"]).all() and (reg["hi"] == ref["hi"]).all()
    assert (reg["min"] == ref["rmin"]).all() and (reg["max"] == ref["rmax"]).all()
    assert (reg["sum"] == ref["rsum"]).all() and (reg["sumsq"] == ref["rsq"]).all()
    assert (par["a"] == ref["a"]).all() and (par["b"] == ref["b"]).all() and (par["count"] == ref["n"]).all()
    assert (par["min"] == ref["pmin"]).all() and (par["max"] == ref["pmax"]).all()
    assert (par["sum"] == ref["psum"]).all() and (par["sumsq"] == ref["psq"]).all()
    order = np.argsort(labels.ravel(), kind="stable")
    starts = np.searchsorted(labels.ravel()[order], reg["label"])
    assert (reg["first"] == order[starts]).all()
    for i in range(0, len(reg["label"]), max(1, len(reg["label"]) // 40)):
        assert (reg["hist"][i] == _np_hist(pb[labels == reg["label"][i]], bins, 0.0, 1.0)).all()
    assert (reg["hist"].sum(1) == reg["count"]).all() and (par["hist"].sum(1) == par["count"]).all()


def _cubes(nz, ny, nx, c, seed):
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    labels = (((z // c) * (ny // c) + (y // c)) * (nx // c) + (x // c) + 1).astype(np.uint32)
    pb = (np.random.default_rng(seed).integers(0, 256, labels.shape) / 256.0).astype(np.float32)
    return labels, pb


def test_cubic_supervoxels_every_lane_ends_its_runs_in_the_same_plane(ctx):
    """Walls perpendicular to the march: all 64 lanes of a wave finish a region run AND a pair run in one plane -- more entries
    than a ring takes in one batch (the split hand-over of the pass)."""
    torch = _torch()
    from glia_amd import hmt
    labels, pb = _cubes(48, 56, 128, 8, 5)
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda()
    d_pb = torch.from_numpy(pb).cuda()
    thr = (0.2, 0.5, 0.8)
    cfg = hmt.make_config(d_pb, rb=[(d_pb, 8, 0.0, 1.0)], thresholds=thr)
    rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, cfg=cfg)
    _check_against_oracle(rm, labels, pb, 8, thr)
    rm.close()


def test_tiny_supervoxels_overflow_the_tile_tables():
    """2 x 2 x 2 cells: thousands of regions per tile, far more than the workgroup's LDS tables hold -- the finished runs that find
    them full go straight to the global tables, and the context answers with shallower tiles on the next build (own context: the
    tile depth is remembered)."""
    torch = _torch()
    from glia_amd import hmt
    own = hmt.Context(0)
    labels, pb = _cubes(32, 48, 64, 2, 6)
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda()
    d_pb = torch.from_numpy(pb).cuda()
    thr = (0.2, 0.5, 0.8)
    cfg = hmt.make_config(d_pb, rb=[(d_pb, 8, 0.0, 1.0)], thresholds=thr)
    for _ in range(3):      # 32, then 16, then 8 planes per tile
        rm = hmt.RegionMap(own, d_lab, pb=d_pb, cfg=cfg)
        _check_against_oracle(rm, labels, pb, 8, thr)
        rm.close()
    own.close()


@pytest.mark.parametrize("shape,S,G,variant,nslab", [((64, 40, 64), 8, 16, 0, 2), ((70, 36, 28), 6, 12, 1, 3),
                                                    ((33, 64, 128), 8, 32, 0, 4)])
def test_slab_partials_merge_to_the_whole(ctx, shape, S, G, variant, nslab):
    """z-slab partition: partial maps of the slabs (one halo plane per cut), merged, are bit-identical
    to the single-pass map -- regions, bounding boxes, first voxels, directed pairs and every statistic."""
    torch = _torch()
    from glia_amd import hmt, slab
    from oracle import pyoracle as O
    labels, pb = O.synth(shape, S, G, variant=variant)
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda()
    d_pb = torch.from_numpy(pb).cuda()
    mk = lambda img: hmt.make_config(img, rb=[(img, 8, 0.0, 1.0)])
    whole = hmt.RegionMap(ctx, d_lab, pb=d_pb, cfg=mk(d_pb))
    parts = []
    nz = shape[0]
    for r in range(nslab):
        lo, hi, zb, ze = slab.slab_with_halo(nz, nslab, r)
        sl, sp = d_lab[lo:hi].contiguous(), d_pb[lo:hi].contiguous()
        parts.append(hmt.RegionMap(ctx, sl, pb=sp, cfg=mk(sp), slab=(lo, nz, zb, ze)))
    # through the same tensors the RCCL exchange ships
    rebuilt = [hmt.RegionMap.from_tensors(ctx, p, p.to_tensors()) for p in parts]
    merged = hmt.RegionMap.merge(ctx, rebuilt)
    a, b = whole.regions(), merged.regions()
    for k in a:
        if variant == 0 or k not in ("sum", "sumsq"):
            assert (a[k] == b[k]).all(), k
        else:
            assert np.allclose(a[k], b[k], rtol=1e-13), k
    a, b = whole.pairs(), merged.pairs()
    for k in a:
        if variant == 0 or k not in ("sum", "sumsq"):
            assert (a[k] == b[k]).all(), k
        else:
            assert np.allclose(a[k], b[k], rtol=1e-13), k
    o1, s1 = whole.merge_order_pb(type=2)
    o2, s2 = merged.merge_order_pb(type=2)
    assert (o1 == o2).all() and (variant == 1 or (s1 == s2).all())


@pytest.mark.parametrize("shape,S,G,variant,nslab", [((64, 40, 64), 8, 16, 0, 2), ((96, 36, 28), 6, 12, 0, 3), ((128, 48, 64), 8, 32, 0, 4)])
def test_cut_record_exchange_reproduces_the_whole_map(ctx, shape, S, G, variant, nslab):
    """The north-star exchange ("only the cross-slab boundary regions"), N ranks emulated on one GPU with the routing code of
    glia_amd/slab.py: glia_hmt_rag_cut_flags marks the records next to a cut, those go to their keyed owner (hash(label) mod N)
    and are reduced there, reduced cut + interior records go once to the loop owner, whose final reduction is bit-identical to
    the single-pass map.  The cut set stays within its analytic bound: no more regions than distinct labels on the planes
    next to the cuts, and a minority of all records."""
    torch = _torch()
    from glia_amd import hmt, slab
    from oracle import pyoracle as O
    labels, pb = O.synth(shape, S, G, variant=variant)
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda()
    d_pb = torch.from_numpy(pb).cuda()
    mk = lambda img: hmt.make_config(img, rb=[(img, 8, 0.0, 1.0)])
    whole = hmt.RegionMap(ctx, d_lab, pb=d_pb, cfg=mk(d_pb))
    nz = shape[0]
    parts, recs, cuts = [], [], []
    n_cut = n_all = 0
    for r in range(nslab):
        lo, hi, zb, ze = slab.slab_with_halo(nz, nslab, r)
        sl, sp = d_lab[lo:hi].contiguous(), d_pb[lo:hi].contiguous()
        part = hmt.RegionMap(ctx, sl, pb=sp, cfg=mk(sp), slab=(lo, nz, zb, ze))
        rcut, pcut = part.cut_flags(sl, zb, ze)
        t = part.to_tensors()
        # analytic bound: a flagged region's label is on a plane next to a cut
        planes = [z for z in ((zb - 1, zb) if zb > 0 else ()) + ((ze - 1, ze) if ze < hi - lo else ())]
        on_cut = np.unique(labels[lo:hi][planes]) if planes else np.zeros(0, np.uint32)
        got = t["rlabel"][rcut].cpu().numpy().view(np.uint32)
        assert set(got.tolist()) <= set(on_cut.tolist()) and len(got) <= len(on_cut)
        own = np.unique(labels[lo + zb:lo + ze])
        assert set(got.tolist()) == set(on_cut.tolist()) & set(own.tolist())          # exactly the owned labels seen next to a cut
        pa, pb_ = t["pa"].cpu().numpy().view(np.uint32), t["pb"].cpu().numpy().view(np.uint32)
        assert (pcut.cpu().numpy() == (np.isin(pa, on_cut) | np.isin(pb_, on_cut))).all()
        n_cut += int(rcut.sum() + pcut.sum()); n_all += t["rlabel"].numel() + t["pa"].numel()
        parts.append(part); recs.append(t); cuts.append((rcut, pcut))
    assert 0 < n_cut < n_all                                                    # only part of the records is exchanged

    def reduce_fn(ps):
        maps = [hmt.RegionMap.from_tensors(ctx, parts[0], p) for p in ps if p["rlabel"].numel() or p["pa"].numel()]
        if not maps:
            return {k: recs[0][k][:0] for k in slab.KEYS}
        m = hmt.RegionMap.merge(ctx, maps)
        out = m.to_tensors()
        for x in maps:
            x.close()
        m.close()
        return out

    # step 2, emulated: what every rank packs for every owner, through pack / unpack
    inbox = [[] for _ in range(nslab)]
    for r in range(nslab):
        cut = slab.select_records(recs[r], *cuts[r])
        ro, po = slab.owner_of(cut["rlabel"], nslab), slab.owner_of(cut["pa"], nslab)
        for d in range(nslab):
            inbox[d].append(slab.unpack_records(slab.pack_records(slab.select_records(cut, ro == d, po == d)), recs[r]))
    reduced = [reduce_fn(inbox[d]) for d in range(nslab)]
    # step 3: interior + reduced cut records of every rank at the loop owner
    final = []
    for r in range(nslab):
        interior = slab.select_records(recs[r], ~cuts[r][0], ~cuts[r][1])
        final.append(slab.unpack_records(slab.pack_records(slab.concat_records([interior, reduced[r]], recs[r])), recs[r]))
    merged = hmt.RegionMap.from_tensors(ctx, parts[0], reduce_fn(final))
    for get in ("regions", "pairs"):
        a, b = getattr(whole, get)(), getattr(merged, get)()
        for k in a:
            assert (a[k] == b[k]).all(), (get, k)
    o1, s1 = whole.merge_order_pb(type=2)
    o2, s2 = merged.merge_order_pb(type=2)
    assert (o1 == o2).all() and (s1 == s2).all()


@pytest.mark.parametrize("shape,S,G,world,channels", [((64, 40, 64), 8, 16, 2, 1), ((96, 36, 28), 6, 12, 3, 1), ((128, 48, 64), 8, 32, 4, 1),
                                                      ((48, 40, 36), 6, 12, 3, 2), ((40, 24, 32), 4, 8, 5, 1)])
def test_distributed_build_through_the_c_abi(ctx, shape, S, G, world, channels):
    """glia_hmt_rag_build_distributed (glia_amd/csrc/slab_dist.cpp): the whole slab route -- build, cut flags, records sorted by
    destination, keyed owner exchange, reduction at the owners, hand-over, final reduction -- behind ONE C entry point, here
    with a local communicator (all ranks in this process: the transfers are device copies, everything else is the code the
    RCCL communicator runs).  The map at the loop owner is bit-identical to the single-pass map; only part of the records took
    the owner exchange; the loop owner can be any rank."""
    torch = _torch()
    from glia_amd import hmt
    from oracle import pyoracle as O
    labels, pb = O.synth(shape, S, G)
    rng = np.random.default_rng(5)
    raw = (np.round(rng.random(shape) * 255) / 256.0).astype(np.float32)
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda()
    d_pb, d_raw = torch.from_numpy(pb).cuda(), torch.from_numpy(raw).cuda()
    mk = (lambda p_, r_: hmt.make_config(p_, rb=[(p_, 8, 0.0, 1.0)])) if channels == 1 else \
         (lambda p_, r_: hmt.make_config(p_, rb=[(r_, 8, 0.0, 1.0), (p_, 8, 0.0, 1.0)]))
    whole = hmt.RegionMap(ctx, d_lab, pb=d_pb, cfg=mk(d_pb, d_raw))
    nz = shape[0]
    comm = hmt.Comm(ctx, world)
    slabs = []
    for r in range(world):
        first, n, zb, ze = hmt.slab_range(nz, world, r)
        sl, sp, sr = d_lab[first:first + n].contiguous(), d_pb[first:first + n].contiguous(), d_raw[first:first + n].contiguous()
        slabs.append((sl, sp, first, zb, ze, mk(sp, sr)))
    for owner in (0, world - 1):
        merged, st = hmt.build_distributed(ctx, comm, slabs, nz, loop_owner=owner)
        assert merged is not None and 0 < st.cut_records < st.records
        assert st.bytes_cut_exchange > 0 and st.bytes_to_loop_owner > 0
        for get in ("regions", "pairs"):
            a, b = getattr(whole, get)(), getattr(merged, get)()
            for k in a:
                assert (a[k] == b[k]).all(), (get, k, owner)
        assert merged.num_channels() == channels
        o1, s1 = whole.merge_order_pb(type=2)
        o2, s2 = merged.merge_order_pb(type=2)
        assert (o1 == o2).all() and (s1 == s2).all()
        if channels == 2:
            clf = hmt.FeatureStubClassifier(ctx, whole.feat_dim() - 3)
            a, b = whole.merge_order_bc(clf), merged.merge_order_bc(clf)
            assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
        merged.close()
    # median linkage (the reference tool's default) on the merged map: the boundary values travel with their pairs
    with pytest.raises(hmt.HmtError):
        hmt.build_distributed(ctx, comm, slabs, nz)[0].merge_order_pb(type=1)           # without them: refused, as before
    merged, st2 = hmt.build_distributed(ctx, comm, slabs, nz, loop_owner=world // 2, with_values=True)
    assert st2.bytes_to_loop_owner > st.bytes_to_loop_owner
    for typ in (1, 3, 2):
        o1, s1 = whole.merge_order_pb(type=typ)
        o2, s2 = merged.merge_order_pb(type=typ)
        assert (o1 == o2).all() and (s1 == s2).all(), typ
    merged.close()
    comm.close()


def test_distributed_build_over_rccl_with_one_rank(ctx):
    """the RCCL communicator (ncclCommInitRank, grouped ncclSend / ncclRecv, ncclAllGather loaded with dlopen) with the one rank a
    one-GPU box allows: the transport is set up and torn down, the route degenerates to a single slab, the map is the whole map"""
    torch = _torch()
    from glia_amd import hmt
    from oracle import pyoracle as O
    shape = (40, 36, 28)
    labels, pb = O.synth(shape, 6, 12)
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda()
    d_pb = torch.from_numpy(pb).cuda()
    cfg = hmt.make_config(d_pb, rb=[(d_pb, 8, 0.0, 1.0)])
    whole = hmt.RegionMap(ctx, d_lab, pb=d_pb, only_contour=True, cfg=cfg)
    comm = hmt.Comm(ctx, 1, rank=0, unique_id=hmt.Comm.unique_id())
    first, n, zb, ze = hmt.slab_range(shape[0], 1, 0)
    merged, st = hmt.build_distributed(ctx, comm, [(d_lab, d_pb, first, zb, ze, cfg)], shape[0], only_contour=True)
    assert st.cut_records == 0 and st.bytes_cut_exchange == 0
    for get in ("regions", "pairs"):
        a, b = getattr(whole, get)(), getattr(merged, get)()
        for k in a:
            assert (a[k] == b[k]).all(), (get, k)
    comm.close()


def test_slab_partials_with_two_image_channels(ctx):
    """feature lists with two volumes (--rbi raw --rbi pb): every channel's records travel and merge with the common keys; the
    classifier-path merge order of the merged map equals the single-pass one"""
    torch = _torch()
    from glia_amd import hmt, slab
    from oracle import pyoracle as O
    shape, nslab = (48, 40, 36), 3
    labels, pb = O.synth(shape, 6, 12)
    rng = np.random.default_rng(9)
    raw = (np.round(rng.random(shape) * 255) / 256.0).astype(np.float32)
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda()
    d_pb, d_raw = torch.from_numpy(pb).cuda(), torch.from_numpy(raw).cuda()
    mk = lambda p_, r_: hmt.make_config(p_, rb=[(r_, 8, 0.0, 1.0), (p_, 8, 0.0, 1.0)])
    whole = hmt.RegionMap(ctx, d_lab, pb=d_pb, cfg=mk(d_pb, d_raw))
    parts = []
    for r in range(nslab):
        lo, hi, zb, ze = slab.slab_with_halo(shape[0], nslab, r)
        sl, sp, sr = d_lab[lo:hi].contiguous(), d_pb[lo:hi].contiguous(), d_raw[lo:hi].contiguous()
        parts.append(hmt.RegionMap(ctx, sl, pb=sp, cfg=mk(sp, sr), slab=(lo, shape[0], zb, ze)))
    tens = [p.to_tensors() for p in parts]
    assert "rrec1" in tens[0] and "prec1" in tens[0]
    # through pack / unpack (what crosses the wire) and from_tensors
    rebuilt = [hmt.RegionMap.from_tensors(ctx, parts[0], slab.unpack_records(slab.pack_records(t), t)) for t in tens]
    merged = hmt.RegionMap.merge(ctx, rebuilt)
    mt, wt = merged.to_tensors(), whole.to_tensors()
    for k in wt:
        assert torch.equal(mt[k], wt[k]), k
    clf = hmt.FeatureStubClassifier(ctx, 40)
    o1, s1 = whole.merge_order_bc(clf)
    o2, s2 = merged.merge_order_bc(clf)
    assert (o1 == o2).all() and (s1 == s2).all()


def _mask_for(shape, seed=3):
    """a mask with holes, a masked face and a few fully masked supervoxels-worth of space"""
    rng = np.random.default_rng(seed)
    m = (rng.random(shape) > 0.15).astype(np.uint32) * 5
    m[..., :2] = 0
    sl = tuple(slice(s // 3, s // 3 + max(2, s // 5)) for s in shape)
    m[sl] = 0
    return m


@pytest.mark.parametrize("shape,S,G", [((32, 32, 32), 8, 16), ((40, 36, 28), 6, 12), ((64, 64), 4, 16), ((16, 32, 128), 8, 16)])
@pytest.mark.parametrize("only_contour", [False, True])
def test_rag_with_mask_matches_oracle(ctx, shape, S, G, only_contour):
    """type/neighbor.hxx:80-86 (masked neighbours are invalid), util/struct.hxx:86-91 (point-map mode skips masked
    voxels) and :133-143 (contour-only mode does not test the centre voxel)."""
    torch = _torch()
    from glia_amd import hmt
    from oracle import pyoracle as O
    labels, pb = O.synth(shape, S, G)
    mask = _mask_for(shape)
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda()
    d_pb = torch.from_numpy(pb).cuda()
    d_mask = torch.from_numpy(mask.view(np.int32)).cuda()
    rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, mask=d_mask, only_contour=only_contour)
    rag = O.Rag(labels, mask=mask, only_contour=only_contour)
    a, b, n = rag.pairs()
    ps = rag.pair_stats(pb)
    par = rm.pairs()
    assert (par["a"] == a).all() and (par["b"] == b).all() and (par["count"] == n).all()
    assert (par["sum"] == ps[0]).all() and (par["sumsq"] == ps[1]).all()
    assert (par["min"] == ps[2]).all() and (par["max"] == ps[3]).all()
    if not only_contour:
        lab, npts, nborder = rag.regions()
        rs = rag.region_stats(pb)
        reg = rm.regions()
        assert (reg["label"] == lab).all() and (reg["count"] == npts).all() and (reg["border"] == nborder).all()
        assert (reg["sum"] == rs[0]).all() and (reg["sumsq"] == rs[1]).all()
        assert (reg["lo"] == rs[4]).all() and (reg["hi"] == rs[5]).all()
        flat, mflat = labels.reshape(-1), mask.reshape(-1)
        first = {}
        for i in range(flat.size - 1, -1, -1):
            if mflat[i]:
                first[int(flat[i])] = i
        assert [first[int(l)] for l in reg["label"]] == reg["first"].tolist()
    rm.close()


def test_error_paths_return_status_not_exit(ctx):
    """the library reports through status codes + glia_hmt_last_error (the reference perr()s and exits)"""
    torch = _torch()
    from glia_amd import hmt
    from oracle import pyoracle as O
    labels, pb = O.synth((16, 16, 16), 4, 8)
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda()
    d_pb = torch.from_numpy(pb).cuda()
    imgs = [torch.from_numpy((pb * (i + 1) / 8).astype(np.float32)).cuda() for i in range(5)]
    with pytest.raises(hmt.HmtError) as e:      # histogram bins out of range
        hmt.RegionMap(ctx, d_lab, pb=d_pb, cfg=hmt.make_config(d_pb, rb=[(d_pb, 17, 0.0, 1.0)]))
    assert e.value.code == -1 and "bins" in str(e.value)
    with pytest.raises(hmt.HmtError) as e:      # five distinct channels
        hmt.RegionMap(ctx, d_lab, pb=d_pb, cfg=hmt.make_config(d_pb, r=[(im, 8, 0.0, 1.0) for im in imgs[:4]], b=[(imgs[4], 8, 0.0, 1.0)]))
    assert e.value.code == -3
    rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, only_contour=True)
    with pytest.raises(hmt.HmtError) as e:      # unsupported boundary stats type (main_merge_order_pb.cxx:36)
        rm.merge_order_pb(type=7)
    assert "unsupported boundary stats type" in str(e.value)
    with pytest.raises(hmt.HmtError):           # classifier linkage needs a feature configuration and region points
        rm.merge_order_bc(hmt.FeatureStubClassifier(ctx, 3))
    with pytest.raises(hmt.HmtError):           # a model file that does not exist
        hmt.RandomForest(ctx, "/nonexistent/model.bin")
    rm.close()
