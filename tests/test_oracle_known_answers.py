"""Pins oracle/ against the known answers SURVEY.md Appendix D records from the reference's own headers.

The reference ships no tests or fixtures; these are the only reference-produced
answers available for the path, so they are the oracle's pin (see DESIGN.md "Oracle").
"""
import numpy as np
import pytest

from oracle import pyoracle as O
from _recipes import recipe_p1


def _check(order, sal, expect):
    for i, (x0, x1, x2, s) in enumerate(expect):
        assert order[i].tolist() == [x0, x1, x2]
        assert sal[i] == pytest.approx(s, rel=0, abs=5e-9 * max(1.0, abs(s)))


@pytest.mark.parametrize("N,B,dim,type,n,expect", [
    (128, 8, 2, 2, 255, [(77, 78, 257, -0.234559), (76, 92, 258, -0.245308593), (207, 208, 259, -0.259489238)]),
    (128, 8, 2, 1, 255, [(77, 78, 257, -0.128828347), (207, 208, 258, -0.194080412), (25, 41, 259, -0.195844293)]),
    (64, 8, 3, 2, 511, [(433, 497, 513, -0.398331326), (347, 411, 514, -0.402272557), (378, 442, 515, -0.40231335)]),
    (64, 8, 3, 1, 511, [(303, 367, 513, -0.320634127), (28, 36, 514, -0.321784198), (378, 442, 515, -0.321870983)]),
    (128, 8, 3, 2, 4095, [(3739, 3995, 4097, -0.372868222), (2181, 2437, 4098, -0.379004306),
                          (2077, 2333, 4099, -0.388795406)]),
])
def test_p1_pb_linkages(N, B, dim, type, n, expect):
    lab, pb = recipe_p1(N, B, dim)
    order, sal = O.Rag(lab, only_contour=True).merge_order_pb(pb, type=type)
    assert len(order) == n
    _check(order, sal, expect)
    # invariants of util/struct_merge.hxx:19-31
    assert (order[:, 2] == lab.max() + 1 + np.arange(n)).all()
    assert (order[:, 0] < order[:, 1]).all() and (order[:, 1] < order[:, 2]).all()


def test_p2_tie_case():
    lab = np.array([[1 + (x // 2) + 3 * (y // 2) for x in range(6)] for y in range(6)], dtype=np.uint32)
    pb = np.full((6, 6), 0.5, np.float32)
    rag = O.Rag(lab, only_contour=True)
    a, b, n = rag.pairs()
    got = {}
    for x, y, c in zip(a, b, n):
        got.setdefault(int(x), {})[int(y)] = int(c)
    assert got == {1: {2: 2, 4: 1}, 2: {1: 2, 3: 2}, 3: {2: 2, 6: 1}, 4: {1: 1, 5: 2, 7: 1}, 5: {4: 2, 6: 2},
                   6: {3: 1, 5: 2, 9: 1}, 7: {4: 1, 8: 2}, 8: {7: 2, 9: 2}, 9: {6: 1, 8: 2}}
    labs, _, nborder = rag.regions()
    assert dict(zip(labs.tolist(), nborder.tolist())) == {1: 1, 2: 0, 3: 1, 4: 0, 5: 0, 6: 0, 7: 1, 8: 0, 9: 1}
    order, sal = rag.merge_order_pb(pb, type=2)
    assert order.tolist() == [[8, 9, 10], [7, 10, 11], [6, 11, 12], [5, 12, 13], [4, 13, 14], [3, 14, 15],
                              [2, 15, 16], [1, 16, 17]]
    assert (sal == -0.5).all()


def test_p3_non_mutual_forest():
    lab = np.array([[1, 1, 4, 4], [2, 3, 4, 4], [2, 3, 5, 5]], dtype=np.uint32)
    pb = np.array([[(1 + x + 4 * y) / 16 for x in range(4)] for y in range(3)], dtype=np.float32)
    order, sal = O.Rag(lab).merge_order_pb(pb, type=2, update_region=True)
    assert order.tolist() == [[1, 4, 6], [2, 3, 7], [5, 6, 8]]
    assert sal.tolist() == [-0.15625, -0.46875, -0.625]


def test_p1_feature_vector():
    lab, pb = recipe_p1(64, 8, 3)
    cfg = O.make_cfg(pb, rb=[(pb, 8, 0.0, 1.0)])
    assert O.feat_dim(3, cfg) == 104
    f = O.Rag(lab).bc_feat(cfg, np.array([[433, 497, 513]], dtype=np.uint32))[0]
    expect = [0, 0, 0, 0, 0, 0, 42, 0.0820312, 0.0820312, 0.141892, 0.141892, 30]
    assert np.allclose(f[:12], expect, rtol=1e-5, atol=1e-7)


def test_feat_dim_2d():
    pb = np.zeros((4, 4), np.float32)
    assert O.feat_dim(2, O.make_cfg(pb, rb=[(pb, 8, 0.0, 1.0)])) == 101


def test_p4_classifier_path_stub_scorer():
    lab, pb = recipe_p1(64, 8, 3, pb_shift=16)
    cfg = O.make_cfg(pb, rb=[(pb, 8, 0.0, 1.0)])
    rag = O.Rag(lab)
    order, sal = rag.merge_order_bc(cfg, None, stub_index=31)
    assert len(order) == 511
    _check(order, sal, [(433, 497, 513, 0.601676214), (347, 411, 514, 0.597735723), (378, 442, 515, 0.597694034)])
    assert rag.n_feat_evals == 9098
