"""hmt::genTree (hmt/tree_build.hxx:12-38): the library's host function vs the oracle (no GPU needed)."""
import numpy as np

from glia_amd import hmt
from oracle import pyoracle as O


def test_gen_tree_matches_oracle_and_invariants():
    labels, pb = O.synth((32, 32, 32), 8, 16)
    order, _ = O.Rag(labels, only_contour=True).merge_order_pb(pb, type=2)
    lab, par, c0, c1 = hmt.gen_tree(order)
    olab, opar, oc0, oc1 = O.gen_tree(order)
    assert (lab == olab).all() and (par == opar).all() and (c0 == oc0).all() and (c1 == oc1).all()
    n = len(order)
    assert len(lab) == 2 * n + 1 and lab[-1] == order[-1, 2] and par[-1] == -1       # root last (type/tree.hxx:100)
    inner = c0 >= 0
    assert inner.sum() == n and (c0[inner] < np.flatnonzero(inner)).all() and (c1[inner] < np.flatnonzero(inner)).all()


def test_gen_tree_forest_of_two_components():
    order = np.array([[1, 4, 6], [2, 3, 7], [5, 6, 8]], dtype=np.uint32)
    lab, par, c0, c1 = hmt.gen_tree(order)
    assert lab.tolist() == [1, 4, 6, 2, 3, 7, 5, 8]
    assert par.tolist() == [2, 2, 7, 5, 5, -1, 7, -1]
