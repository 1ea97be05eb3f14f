"""-m gpu: the three priority-queue implementations of the pb-mean merge loop give byte-identical merge orders.

  tournament tree (greedy_pb_kernel, round 1)  ==  window queue, one contraction at a time (greedy_window_kernel)
                                               ==  window queue, batched contractions (greedy_batch_kernel, the default)
under conditions that force every rare path of the window queue: a tiny window (spills, evictions, cells split by the bounded
heap), frequent re-baselines, massive exact ties, launches that end early.  The switches are library options (glia_hmt_set_option,
hmt.options): GLIA_HMT_PB_WINDOW=0 (tree), GLIA_HMT_PB_BATCH=0 (sequential window), GLIA_HMT_WINCAP, GLIA_HMT_REBASE, ...
No loop is ever run twice: an order that breaks the invariant of glia_hmt_check_merge_order is GLIA_HMT_ERR_INTERNAL, and the
session ends with a check that glia_hmt_internal_errors() stayed 0 (tests/conftest.py)."""
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


def _order(ctx, d_lab, d_pb, **env):
    from glia_amd import hmt
    with hmt.options(**env):
        rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, only_contour=True)
        o, s = rm.merge_order_pb(type=2)
        rm.close()
    return o, s


def _volume(ctx, shape, S, variant, levels=None):
    import torch
    labels, pb = ctx.synth(shape, S, 4 * S, variant=variant)
    if levels:
        pb = torch.floor(pb * levels) / levels
    return labels, pb.contiguous()


CASES = [((96, 96, 96), 8, 0, None), ((128, 128, 128), 8, 1, None), ((96, 80, 64), 6, 0, 4), ((160, 160), 4, 0, None)]


@pytest.mark.parametrize("shape,S,variant,levels", CASES)
def test_three_queues_agree(ctx, shape, S, variant, levels):
    d_lab, d_pb = _volume(ctx, shape, S, variant, levels)
    tree = _order(ctx, d_lab, d_pb, GLIA_HMT_PB_WINDOW=0)
    assert len(tree[0]) > 500
    for env in (dict(), dict(GLIA_HMT_PB_BATCH=0)):
        o, s = _order(ctx, d_lab, d_pb, **env)
        assert o.shape == tree[0].shape and (o == tree[0]).all() and (s == tree[1]).all(), env


@pytest.mark.parametrize("shape,S,variant,levels", CASES)
@pytest.mark.parametrize("env", [dict(GLIA_HMT_WINCAP=32), dict(GLIA_HMT_WINCAP=16, GLIA_HMT_REBASE=300), dict(GLIA_HMT_REBASE=100),
                                 dict(GLIA_HMT_WINCAP=64, GLIA_HMT_PB_BATCH=0)])
def test_window_queue_rare_paths(ctx, shape, S, variant, levels, env):
    """tiny windows: nearly every insertion spills and raises tau, every reload splits a cell; tiny re-baseline intervals: the
    launch ends every few hundred created edges and the live items are re-sorted"""
    d_lab, d_pb = _volume(ctx, shape, S, variant, levels)
    tree = _order(ctx, d_lab, d_pb, GLIA_HMT_PB_WINDOW=0)
    o, s = _order(ctx, d_lab, d_pb, **env)
    assert o.shape == tree[0].shape and (o == tree[0]).all() and (s == tree[1]).all()


@pytest.mark.parametrize("env", [dict(GLIA_HMT_HORIZON=0), dict(GLIA_HMT_HORIZON=0.25, GLIA_HMT_REBASE=5000),
                                 dict(GLIA_HMT_HORIZON=8, GLIA_HMT_REBASE=20000), dict(GLIA_HMT_HORIZON=0.05, GLIA_HMT_REBASE=2000, GLIA_HMT_WINCAP=64)])
def test_horizon_placements(ctx, env):
    """the horizon below which created edges are not queued until the next baseline (WinState::wch): switched off, placed far
    too tight (reloads run into it and end the launch early), far too loose, and tight with a tiny window"""
    d_lab, d_pb = _volume(ctx, (128, 128, 128), 8, 0, None)
    tree = _order(ctx, d_lab, d_pb, GLIA_HMT_PB_WINDOW=0)
    assert len(tree[0]) == 4095
    o, s = _order(ctx, d_lab, d_pb, **env)
    assert o.shape == tree[0].shape and (o == tree[0]).all() and (s == tree[1]).all()


@pytest.mark.parametrize("env", [dict(GLIA_HMT_FORCE_TREE=1), dict(GLIA_HMT_FORCE_TREE=1500), dict(GLIA_HMT_FORCE_TREE=700, GLIA_HMT_PB_BATCH=0),
                                 dict(GLIA_HMT_FORCE_TREE=2500, GLIA_HMT_WINCAP=64, GLIA_HMT_REBASE=400)])
def test_hand_over_to_the_tree_kernel_in_mid_run(ctx, env):
    """ST_NEED_TREE: the window kernels stop at an empty window and the tournament-tree kernel continues from the same state (edge
    records unpacked into its arrays, thin list entries).  No data set reaches that exit by itself any more -- oversized saliency
    cells are split -- so the test forces it after k merges."""
    d_lab, d_pb = _volume(ctx, (128, 128, 128), 8, 0, None)
    tree = _order(ctx, d_lab, d_pb, GLIA_HMT_PB_WINDOW=0)
    assert len(tree[0]) == 4095
    o, s = _order(ctx, d_lab, d_pb, **env)
    assert o.shape == tree[0].shape and (o == tree[0]).all() and (s == tree[1]).all()


def test_constant_pb_on_every_queue(ctx):
    """every saliency equal: one saliency cell holds the whole queue, the order is the tie rule alone"""
    import torch
    from oracle import pyoracle as O
    labels, _ = O.synth((40, 40, 40), 5, 10)
    pb = np.full(labels.shape, 0.25, np.float32)
    d_lab, d_pb = torch.from_numpy(labels.view(np.int32)).cuda(), torch.from_numpy(pb).cuda()
    o_ref, s_ref = O.Rag(labels, only_contour=True).merge_order_pb(pb, type=2)
    for env in (dict(), dict(GLIA_HMT_PB_BATCH=0), dict(GLIA_HMT_PB_WINDOW=0), dict(GLIA_HMT_WINCAP=32), dict(GLIA_HMT_WINCAP=48, GLIA_HMT_REBASE=200)):
        o, s = _order(ctx, d_lab, d_pb, **env)
        assert (o == o_ref).all() and (s == s_ref).all(), env


def test_pre_merge_on_a_tiny_window(ctx):
    """the condition kernel (greedy_window_kernel<true>) with a tiny reload budget vs the oracle"""
    import torch
    from glia_amd import hmt
    from oracle import pyoracle as O
    labels, pb = O.synth((48, 48, 48), 6, 12)
    d_lab, d_pb = torch.from_numpy(labels.view(np.int32)).cuda(), torch.from_numpy(pb).cuda()
    ro, rs = O.Rag(labels).pre_merge(pb, [150, 400], 0.3)
    for env in (dict(), dict(GLIA_HMT_WINCAP=32), dict(GLIA_HMT_REBASE=200)):
        with hmt.options(**env):
            rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, only_contour=False)
            o, s = rm.pre_merge([150, 400], 0.3)
            rm.close()
        assert o.shape == ro.shape and (o == ro).all() and (s == rs).all(), env


def _dense(order, labels):
    """keys -> dense ids: leaf i = i-th label ascending, merged key maxKey + 1 + k -> R + k"""
    lab = np.unique(labels[labels != 0]) if (labels == 0).any() else np.unique(labels)
    R, top = len(lab), int(lab.max())
    o = order.astype(np.int64)
    d = np.where(o > top, o - (top + 1) + R, np.searchsorted(lab, np.minimum(o, top)))
    return d.astype(np.uint32), R


def test_orders_satisfy_the_library_invariant_and_a_spoiled_one_does_not(ctx):
    """glia_hmt_check_merge_order is the replay the pb / pre_merge loops run on every order before they return it (a violation is
    GLIA_HMT_ERR_INTERNAL, never a second run): the device's orders pass it, an order whose last merge names a region that went at
    merge 0 does not, and neither raised the process's internal-error count."""
    import torch
    from glia_amd import hmt
    from oracle import pyoracle as O
    labels, pb = O.synth((40, 40, 24), 5, 10)
    d_lab, d_pb = torch.from_numpy(labels.view(np.int32)).cuda(), torch.from_numpy(pb).cuda()
    before = hmt.Context.internal_errors()
    o, s = _order(ctx, d_lab, d_pb)
    rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, only_contour=False)
    po, ps = rm.pre_merge([100, 300], 0.3)
    rm.close()
    for order in (o, po):
        d, R = _dense(order, labels)
        assert len(d) > 10 and hmt.check_merge_order(d, R) == -1
        bad = d.copy()
        bad[-1, 0] = bad[0, 0]
        assert hmt.check_merge_order(bad, R) == len(bad) - 1
        bad = d.copy()
        bad[5, 2] += 1
        assert hmt.check_merge_order(bad, R) == 5
    assert hmt.Context.internal_errors() == before
