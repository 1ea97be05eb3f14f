"""-m gpu: device greedy merge loop (K4b+K5, pb-mean linkage) vs the oracle and the reference's known answers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "GPU test run without a GPU"
    from glia_amd import hmt
    c = hmt.Context(0)
    yield c
    c.close()


def _gpu_order(ctx, labels, pb, only_contour=True, type=2):
    import torch
    from glia_amd import hmt
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda()
    d_pb = torch.from_numpy(pb).cuda()
    rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, only_contour=only_contour)
    out = rm.merge_order_pb(type=type)
    rm.close()
    return out


@pytest.mark.parametrize("N,B,dim,n,expect", [
    (128, 8, 2, 255, [(77, 78, 257, -0.234559), (76, 92, 258, -0.245308593), (207, 208, 259, -0.259489238)]),
    (64, 8, 3, 511, [(433, 497, 513, -0.398331326), (347, 411, 514, -0.402272557), (378, 442, 515, -0.40231335)]),
    (128, 8, 3, 4095, [(3739, 3995, 4097, -0.372868222), (2181, 2437, 4098, -0.379004306),
                       (2077, 2333, 4099, -0.388795406)]),
])
def test_reference_known_answers(ctx, N, B, dim, n, expect):
    """SURVEY.md Appendix D recipe P1, answers produced by the reference's own headers."""
    from _recipes import recipe_p1
    lab, pb = recipe_p1(N, B, dim)
    order, sal = _gpu_order(ctx, lab, pb)
    assert len(order) == n
    for i, (x0, x1, x2, s) in enumerate(expect):
        assert order[i].tolist() == [x0, x1, x2]
        assert sal[i] == pytest.approx(s, rel=0, abs=5e-9)


def test_tie_case_p2(ctx):
    lab = np.array([[1 + (x // 2) + 3 * (y // 2) for x in range(6)] for y in range(6)], dtype=np.uint32)
    pb = np.full((6, 6), 0.5, np.float32)
    order, sal = _gpu_order(ctx, lab, pb)
    assert order.tolist() == [[8, 9, 10], [7, 10, 11], [6, 11, 12], [5, 12, 13], [4, 13, 14], [3, 14, 15],
                              [2, 15, 16], [1, 16, 17]]
    assert (sal == -0.5).all()


def test_non_mutual_forest_p3(ctx):
    lab = np.array([[1, 1, 4, 4], [2, 3, 4, 4], [2, 3, 5, 5]], dtype=np.uint32)
    pb = np.array([[(1 + x + 4 * y) / 16 for x in range(4)] for y in range(3)], dtype=np.float32)
    order, sal = _gpu_order(ctx, lab, pb, only_contour=False)
    assert order.tolist() == [[1, 4, 6], [2, 3, 7], [5, 6, 8]]
    assert sal.tolist() == [-0.15625, -0.46875, -0.625]


CASES = [((32, 32, 32), 8, 16, 0), ((40, 36, 28), 6, 12, 1), ((96, 96), 8, 32, 0), ((64, 64), 4, 16, 1),
         ((64, 64, 64), 8, 16, 0), ((48, 64, 256), 8, 32, 0), ((128, 128, 128), 8, 32, 0)]


@pytest.mark.parametrize("shape,S,G,variant", CASES)
def test_order_matches_oracle(ctx, shape, S, G, variant):
    from oracle import pyoracle as O
    labels, pb = O.synth(shape, S, G, variant=variant)
    order, sal = _gpu_order(ctx, labels, pb)
    o_ref, s_ref = O.Rag(labels, only_contour=True).merge_order_pb(pb, type=2)
    assert order.shape == o_ref.shape
    assert (order == o_ref).all()
    if variant == 0:
        assert (sal == s_ref).all()          # Q8 pb: bit-identical saliencies
    else:
        assert np.allclose(sal, s_ref, rtol=0, atol=1e-12)


def test_constant_pb_tie_torture(ctx):
    """Every saliency equal: the order is decided by the multimap tie rule alone."""
    from oracle import pyoracle as O
    labels, _ = O.synth((40, 40, 40), 6, 12)
    pb = np.full(labels.shape, 0.25, np.float32)
    order, sal = _gpu_order(ctx, labels, pb)
    o_ref, s_ref = O.Rag(labels, only_contour=True).merge_order_pb(pb, type=2)
    assert (order == o_ref).all() and (sal == s_ref).all()


def test_few_level_pb_many_ties(ctx):
    """pb quantised to 4 levels: massive exact ties between unrelated edges plus real ordering."""
    from oracle import pyoracle as O
    labels, pb = O.synth((48, 48, 48), 6, 12)
    pb = (np.floor(pb * 4) / 4).astype(np.float32)
    order, sal = _gpu_order(ctx, labels, pb)
    o_ref, s_ref = O.Rag(labels, only_contour=True).merge_order_pb(pb, type=2)
    assert (order == o_ref).all() and (sal == s_ref).all()


def test_full_size_properties(ctx):
    """256^3 (BASELINE config 2 size): size-independent invariants of util/struct_merge.hxx:19-31."""
    from glia_amd import hmt
    labels, pb = ctx.synth((256, 256, 256), 16, 64)
    rm = hmt.RegionMap(ctx, labels, pb=pb, only_contour=True)
    R = rm.num_regions
    order, sal = rm.merge_order_pb(type=2)
    n = len(order)
    assert n == R - 1                                     # connected mutual-edge graph
    assert (order[:, 0] < order[:, 1]).all() and (order[:, 1] < order[:, 2]).all()
    assert (order[:, 2].astype(np.int64) == R + 1 + np.arange(n)).all()     # labels are 1..R
    used = np.concatenate([order[:, 0], order[:, 1]])
    assert len(np.unique(used)) == 2 * n                  # every region is merged exactly once
    assert (np.diff(sal) <= 1e-12).all()                  # mean linkage is reducible: saliency never increases
    rm.close()


@pytest.mark.parametrize("shape,S,G,variant,sizes,rpb", [((32, 32, 32), 8, 16, 0, (300,), 0.0), ((40, 36, 28), 6, 12, 0, (150, 400), 0.30),
                                                        ((96, 96), 8, 32, 0, (40, 120), 0.25), ((48, 48, 48), 6, 12, 1, (100, 500), 0.28)])
def test_pre_merge_condition(ctx, shape, S, G, variant, sizes, rpb):
    """gadget/main_pre_merge.cxx:27-76: mean linkage, updateRegion, size / mean-pb condition."""
    import torch
    from glia_amd import hmt
    from oracle import pyoracle as O
    labels, pb = O.synth(shape, S, G, variant=variant)
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda()
    d_pb = torch.from_numpy(pb).cuda()
    rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, only_contour=False)
    order, sal = rm.pre_merge(list(sizes), rpb)
    o_ref, s_ref = O.Rag(labels).pre_merge(pb, list(sizes), rpb)
    assert 0 < len(o_ref) < rm.num_regions - 1            # the condition really stops the loop early
    assert order.shape == o_ref.shape and (order == o_ref).all()
    assert (sal == s_ref).all() if variant == 0 else np.allclose(sal, s_ref, rtol=0, atol=1e-12)


# ---- median linkage (hmt/main_merge_order_pb.cxx -t 1, the tool's default; util/struct_merge.hxx:90-136) ----
@pytest.mark.parametrize("N,B,dim,n,expect", [
    (128, 8, 2, 255, [(77, 78, 257, -0.128828347), (207, 208, 258, -0.194080412), (25, 41, 259, -0.195844293)]),
    (64, 8, 3, 511, [(303, 367, 513, -0.320634127), (28, 36, 514, -0.321784198), (378, 442, 515, -0.321870983)]),
])
def test_median_reference_known_answers(ctx, N, B, dim, n, expect):
    """SURVEY.md Appendix D recipe P1 (median), answers produced by the reference's own headers."""
    from _recipes import recipe_p1
    lab, pb = recipe_p1(N, B, dim)
    order, sal = _gpu_order(ctx, lab, pb, type=1)
    assert len(order) == n
    for i, (x0, x1, x2, s) in enumerate(expect):
        assert order[i].tolist() == [x0, x1, x2]
        assert sal[i] == pytest.approx(s, rel=0, abs=5e-9)


# the last case ends with contractions that merge several hundred thousand values: more than 512 tiles, i.e. more than
# one round of the tile loop
@pytest.mark.parametrize("shape,S,G,variant", CASES + [((33, 30, 40), 5, 20, 1), ((176, 176, 176), 16, 16, 0)])
def test_median_order_matches_oracle(ctx, shape, S, G, variant):
    """Saliency = an order statistic of the f32 values: bit-identical for Q8 and for continuous pb alike."""
    from oracle import pyoracle as O
    labels, pb = O.synth(shape, S, G, variant=variant)
    order, sal = _gpu_order(ctx, labels, pb, type=1)
    o_ref, s_ref = O.Rag(labels, only_contour=True).merge_order_pb(pb, type=1)
    assert order.shape == o_ref.shape
    assert (order == o_ref).all()
    assert (sal == s_ref).all()


def test_median_ties(ctx):
    from oracle import pyoracle as O
    labels, pb = O.synth((48, 48, 48), 6, 12)
    for q in (np.full(labels.shape, 0.25, np.float32), (np.floor(pb * 4) / 4).astype(np.float32)):
        order, sal = _gpu_order(ctx, labels, q, type=1)
        o_ref, s_ref = O.Rag(labels, only_contour=True).merge_order_pb(q, type=1)
        assert (order == o_ref).all() and (sal == s_ref).all()


def test_median_point_mode_non_mutual(ctx):
    lab = np.array([[1, 1, 4, 4], [2, 3, 4, 4], [2, 3, 5, 5]], dtype=np.uint32)
    pb = np.array([[(1 + x + 4 * y) / 16 for x in range(4)] for y in range(3)], dtype=np.float32)
    from oracle import pyoracle as O
    order, sal = _gpu_order(ctx, lab, pb, only_contour=False, type=1)
    o_ref, s_ref = O.Rag(lab, only_contour=False).merge_order_pb(pb, type=1)
    assert (order == o_ref).all() and (sal == s_ref).all()


@pytest.mark.parametrize("shape,S,G", [((32, 32, 32), 8, 16), ((40, 36, 28), 6, 12), ((64, 64), 4, 16)])
@pytest.mark.parametrize("type", [1, 2])
def test_merge_order_with_mask(ctx, shape, S, G, type):
    """merge_order_pb -m: RegionMap(seg, mask, true) (hmt/main_merge_order_pb.cxx:24-27)"""
    import torch
    from glia_amd import hmt
    from oracle import pyoracle as O
    labels, pb = O.synth(shape, S, G)
    rng = np.random.default_rng(11)
    mask = (rng.random(shape) > 0.2).astype(np.uint32)
    mask[..., :3] = 0
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda()
    d_pb = torch.from_numpy(pb).cuda()
    d_mask = torch.from_numpy(mask.view(np.int32)).cuda()
    rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, mask=d_mask, only_contour=True)
    order, sal = rm.merge_order_pb(type=type)
    o_ref, s_ref = O.Rag(labels, mask=mask, only_contour=True).merge_order_pb(pb, type=type)
    assert order.shape == o_ref.shape and (order == o_ref).all() and (sal == s_ref).all()
    rm.close()


@pytest.mark.parametrize("shape,S,G,variant", [((32, 32, 32), 8, 16, 0), ((40, 36, 28), 6, 12, 1), ((64, 64), 4, 16, 0)])
def test_median_times_min_size_linkage(ctx, shape, S, G, variant):
    """genMergeOrderGreedyUsingPbApproxMedianAndMinSize (util/struct_merge.hxx:141-185; no reference tool calls it):
    saliency = -median * min(region sizes) with the regions merged as the loop goes"""
    from oracle import pyoracle as O
    labels, pb = O.synth(shape, S, G, variant=variant)
    order, sal = _gpu_order(ctx, labels, pb, only_contour=False, type=3)
    o_ref, s_ref = O.Rag(labels, only_contour=False).merge_order_pb(pb, type=3, update_region=True)
    assert order.shape == o_ref.shape and (order == o_ref).all() and (sal == s_ref).all()
