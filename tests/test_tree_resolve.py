"""Tree resolution after the merge path (hmt/main_segment_greedy.cxx): the library's host functions vs the oracle's
restatement (full-scan picks, BFS traversals).  No GPU needed."""
import numpy as np
import pytest

from glia_amd import hmt
from oracle import pyoracle as O


def _order(shape=(32, 32, 32), S=8, G=16):
    labels, pb = O.synth(shape, S, G)
    order, sal = O.Rag(labels, only_contour=True).merge_order_pb(pb, type=2)
    return labels, order, sal


@pytest.mark.parametrize("with_region_probs", [False, True])
def test_potentials_picks_and_label_map_match_oracle(with_region_probs):
    labels, order, sal = _order()
    probs = np.clip(1.0 + sal * 2.5, 0.0, 1.0)          # merge probabilities in [0, 1], many distinct values, some ties
    rng = np.random.default_rng(2)
    rp = rng.random(2 * len(order) + 1) if with_region_probs else None
    a = hmt.tree_potentials(order, probs, rp)
    b = O.tree_potentials(order, probs, rp)
    for x, y in zip(a, b):
        assert (x == y).all()
    lab, par, c0, c1, pot = a
    picks = hmt.resolve_tree_greedy(par, c0, c1, pot)
    assert (picks == O.resolve_tree_greedy(par, c0, c1, pot)).all()
    src, dst = hmt.label_transform(lab, c0, c1, picks, 1)
    osrc, odst = O.label_transform(lab, c0, c1, picks, 1)
    assert (src == osrc).all() and (dst == odst).all()
    # the picks partition the leaves: every supervoxel gets exactly one final label 1..n_picks
    assert sorted(src.tolist()) == sorted(np.unique(labels).tolist()) and set(dst.tolist()) == set(range(1, len(picks) + 1))


def test_without_probabilities_everything_is_one_and_the_first_node_wins():
    _, order, _ = _order((24, 24, 24), 6, 12)
    lab, par, c0, c1, pot = hmt.tree_potentials(order)
    assert (pot == 1.0).all()
    picks = hmt.resolve_tree_greedy(par, c0, c1, pot)
    assert (picks == O.resolve_tree_greedy(par, c0, c1, pot)).all()
    assert picks[0] == 0                       # strict '<' in pickTreeNode keeps the first node of maximal potential


def test_constant_probabilities_tie_rule_and_forest():
    order = np.array([[1, 4, 6], [2, 3, 7], [5, 6, 8]], dtype=np.uint32)       # two components
    for probs in ([0.5, 0.5, 0.5], [0.9, 0.2, 0.7], [0.0, 1.0, 1.0]):
        a = hmt.tree_potentials(order, probs)
        b = O.tree_potentials(order, probs)
        for x, y in zip(a, b):
            assert (x == y).all()
        picks = hmt.resolve_tree_greedy(*a[1:])
        assert (picks == O.resolve_tree_greedy(*b[1:])).all()
        s1 = hmt.label_transform(a[0], a[2], a[3], picks, 5)
        s2 = O.label_transform(b[0], b[2], b[3], picks, 5)
        assert (s1[0] == s2[0]).all() and (s1[1] == s2[1]).all()


def test_several_trees_share_leaves():
    """two alternative merge orders over the same supervoxels + a third tree over a subset: picks in one tree take the
    leaves away from the others (hmt/tree_greedy.hxx:104-152)"""
    labels, pb = O.synth((24, 24, 24), 6, 12)
    rag = O.Rag(labels, only_contour=True)
    o1, s1 = rag.merge_order_pb(pb, type=2)
    o2, s2 = O.Rag(labels, only_contour=True).merge_order_pb(pb, type=1)
    sub = (labels[:12] if True else labels)
    o3, s3 = O.Rag(np.ascontiguousarray(sub), only_contour=True).merge_order_pb(np.ascontiguousarray(pb[:12]), type=2)
    trees = []
    for o, s in ((o1, s1), (o2, s2), (o3, s3)):
        lab, par, c0, c1, pot = hmt.tree_potentials(o, np.clip(1.0 + 2.5 * s, 0.0, 1.0))
        trees.append((lab, par, c0, c1, pot))
    pt, pn = hmt.resolve_trees_greedy(trees)
    ot, on = O.resolve_trees_greedy(trees)
    assert (pt == ot).all() and (pn == on).all() and len(pt) > 3
    # with one tree the joint resolver degenerates to the single-tree one (apart from leaf picks, which behave the same)
    pt1, pn1 = hmt.resolve_trees_greedy(trees[:1])
    assert (pt1 == 0).all() and (pn1 == hmt.resolve_tree_greedy(*trees[0][1:])).all()
