"""transformKeys (util/struct_merge.hxx:188-210): the library's host function vs the oracle (no GPU needed)."""
import numpy as np

from glia_amd import hmt
from oracle import pyoracle as O


def test_transform_keys_full_tree_and_forest():
    labels, pb = O.synth((32, 32, 32), 8, 16)
    order, _ = O.Rag(labels, only_contour=True).merge_order_pb(pb, type=2)
    for n in (len(order), len(order) // 2, 5, 0):
        src, dst = hmt.transform_keys(order[:n])
        osrc, odst = O.transform_keys(order[:n])
        assert (src == osrc).all() and (dst == odst).all()
        if n == len(order):
            assert set(src.tolist()) == set(np.unique(labels).tolist()) and (dst == order[-1, 2]).all()


def test_transform_keys_p3_and_unsorted_lists():
    order = np.array([[1, 4, 6], [2, 3, 7], [5, 6, 8]], dtype=np.uint32)
    src, dst = hmt.transform_keys(order)
    assert src.tolist() == [1, 2, 3, 4, 5] and dst.tolist() == [8, 7, 7, 8, 8]
    src2, dst2 = hmt.transform_keys(order[::-1].copy())       # apply_merges sorts by x2, transformKeys itself does not care
    assert (src2 == src).all() and (dst2 == dst).all()
    osrc, odst = O.transform_keys(order[::-1].copy())
    assert (osrc == src).all() and (odst == dst).all()
