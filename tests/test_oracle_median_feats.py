"""Oracle only: the GLIA_USE_MEDIAN_AS_FEATS build variant of the feature vector (type/feat.hxx:677-722, 772-808;
hmt/bc_feat.hxx:250-268).  The device path does not have this variant yet; these tests fix what it has to
reproduce: the layout (one more column at the head of every real-feature block), the median as the order statistic at n/2,
mean and standard deviation from stats::mean / stats::var over the value vector (comparable to 1e-12 relative: the
reference sums in an order that depends on rand(), see the oracle's header comment), every other column unchanged."""
import numpy as np
import pytest

from oracle import pyoracle as O


def _volume(dim, seed):
    shape = (14, 12, 10) if dim == 3 else (26, 22)
    labels, pb = O.synth(shape, 4, 8, seed=seed, variant=0)
    rng = np.random.default_rng(seed)
    raw = (np.round(rng.random(shape) * 255) / 256.0).astype(np.float32)
    return labels, pb, raw


def _layout(dim, n_thr, n_r, n_rl, n_b, med):
    """Column roles of the full vector: list of (start, kind) for every real-feature block; kind 'diff' or 'set'."""
    blocks = []
    pos = 11 + 4 * n_thr
    for _ in range(n_r):
        pos += 3; blocks.append((pos, "diff")); pos += 4 + med
    pos += 3 * n_rl
    for _ in range(n_b):
        pos += 1; blocks.append((pos, "set")); pos += 4 + med
    for _ in range(3):
        pos += 4 + dim + 2 * n_thr
        for _ in range(n_r):
            pos += 1; blocks.append((pos, "set")); pos += 4 + med
        pos += n_rl
        for _ in range(n_b):
            pos += 1; blocks.append((pos, "set")); pos += 4 + med
    return blocks, pos


@pytest.mark.parametrize("dim", [2, 3])
def test_median_variant_adds_one_column_per_real_block_and_leaves_the_rest(dim):
    labels, pb, raw = _volume(dim, 11 + dim)
    rag = O.Rag(labels)
    order, _ = rag.merge_order_pb(pb, type=2)
    kw = dict(rb=[(pb, 8, 0.0, 1.0)], r=[(raw, 4, 0.0, 1.0)], rl=[(raw, 4, 0.0, 1.0)], b=[(raw, 8, 0.0, 1.0)])
    plain = O.make_cfg(pb, **kw)
    med = O.make_cfg(pb, median_as_feats=True, **kw)
    fa, fm = rag.bc_feat(plain, order), rag.bc_feat(med, order)
    n_thr, n_r, n_rl, n_b = 3, 2, 1, 2
    blocks_a, da = _layout(dim, n_thr, n_r, n_rl, n_b, 0)
    blocks_m, dm = _layout(dim, n_thr, n_r, n_rl, n_b, 1)
    assert fa.shape[1] == da == O.feat_dim(dim, plain)
    assert fm.shape[1] == dm == O.feat_dim(dim, med) == da + len(blocks_a)
    med_cols = [s for s, _ in blocks_m]
    keep = np.ones(dm, bool); keep[med_cols] = False
    rest = fm[:, keep]
    loose = np.zeros(da, bool)
    for s, _ in blocks_a:
        loose[s] = loose[s + 1] = True                      # mean and stddev (or their differences): another summation
    assert (rest[:, ~loose] == fa[:, ~loose]).all()
    assert np.allclose(rest[:, loose], fa[:, loose], rtol=0, atol=1e-12)
    # medians lie between the block's minimum and maximum; a difference of medians is not negative
    for s, kind in blocks_m:
        if kind == "set":
            assert (fm[:, s] >= fm[:, s + 3]).all() and (fm[:, s] <= fm[:, s + 4]).all()
        else:
            assert (fm[:, s] >= 0).all()


def test_median_is_the_order_statistic_at_n_over_2_of_the_merged_region():
    labels, pb, raw = _volume(3, 5)
    rag = O.Rag(labels)
    order, _ = rag.merge_order_pb(pb, type=2)
    med = O.make_cfg(pb, r=[(raw, 4, 0.0, 1.0)], median_as_feats=True)
    fm = rag.bc_feat(med, order)
    blocks, _ = _layout(3, 3, 1, 0, 0, 1)
    col_r2 = blocks[-1][0]                                   # the region block of the merged region (third RegionFeats)
    members = {int(l): [int(l)] for l in np.unique(labels)}
    for k, (a, b, c) in enumerate(order[:40]):
        members[int(c)] = members[int(a)] + members[int(b)]
        v = np.sort(raw[np.isin(labels, members[int(c)])])
        assert fm[k, col_r2] == v[len(v) // 2]
        assert abs(fm[k, col_r2 + 1] - v.astype(np.float64).mean()) <= 1e-12
        assert abs(fm[k, col_r2 + 2] - v.astype(np.float64).std()) <= 1e-12
        assert fm[k, col_r2 + 3] == v[0] and fm[k, col_r2 + 4] == v[-1]


def test_simple_selection_carries_the_boundary_median():
    labels, pb, raw = _volume(2, 9)
    rag = O.Rag(labels)
    order, _ = rag.merge_order_pb(pb, type=2)
    kw = dict(rb=[(pb, 8, 0.0, 1.0)], b=[(raw, 8, 0.0, 1.0)], use_simple=True)
    fa = rag.bc_feat(O.make_cfg(pb, **kw), order)
    fm = rag.bc_feat(O.make_cfg(pb, median_as_feats=True, **kw), order)
    assert fa.shape[1] == 5 + 2 + 4 and fm.shape[1] == 5 + 4 + 4           # bc_feat.hxx:250-256
    assert (fm[:, :5] == fa[:, :5]).all() and (fm[:, 9:] == fa[:, 7:]).all()
    assert np.allclose(fm[:, [5, 7]], fa[:, [5, 6]], rtol=0, atol=1e-12)    # boundary means
