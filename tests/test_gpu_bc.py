"""-m gpu: classifier-linkage merge tree (features K6 + forest K7 + greedy K5) vs the oracle."""
import os
import tempfile

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


def _gpu_rm(ctx, labels, pb, bins=8, thr=(0.2, 0.5, 0.8), **kw):
    import torch
    from glia_amd import hmt
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda()
    d_pb = torch.from_numpy(pb).cuda()
    cfg = hmt.make_config(d_pb, rb=[(d_pb, bins, 0.0, 1.0)], thresholds=thr, **kw)
    return hmt.RegionMap(ctx, d_lab, pb=d_pb, cfg=cfg)


def _feat_close(a, b):
    return np.allclose(a, b, rtol=1e-5, atol=1e-12)


def test_p4_known_answer_stub_scorer(ctx):
    """SURVEY.md Appendix D recipe P4: answers produced by the reference's own headers."""
    from glia_amd import hmt
    from _recipes import recipe_p1
    lab, pb = recipe_p1(64, 8, 3, pb_shift=16)
    rm = _gpu_rm(ctx, lab, pb)
    assert rm.feat_dim() == 104
    order, sal = rm.merge_order_bc(hmt.FeatureStubClassifier(ctx, 31))
    assert len(order) == 511
    for i, (x0, x1, x2, s) in enumerate([(433, 497, 513, 0.601676214), (347, 411, 514, 0.597735723),
                                         (378, 442, 515, 0.597694034)]):
        assert order[i].tolist() == [x0, x1, x2]
        assert sal[i] == pytest.approx(s, rel=0, abs=5e-9)
    assert rm.last_merge_timing()["n_edges_scored"] >= 9098 - 512


CASES = [((24, 24, 24), 6, 12, 0, 8), ((32, 32, 32), 8, 16, 0, 8), ((40, 36, 28), 6, 12, 0, 8), ((64, 64), 4, 16, 0, 8),
         ((33, 30, 40), 5, 20, 1, 16), ((48, 48, 48), 8, 16, 0, 8)]


@pytest.mark.parametrize("shape,S,G,variant,bins", CASES)
def test_stub_scorer_matches_oracle(ctx, shape, S, G, variant, bins):
    from glia_amd import hmt
    from oracle import pyoracle as O
    labels, pb = O.synth(shape, S, G, variant=variant)
    rm = _gpu_rm(ctx, labels, pb, bins=bins)
    stub = 11 + 4 * 3 + 7 + 1          # mean of the boundary image over the shared boundary
    order, sal, feats = rm.merge_order_bc(hmt.FeatureStubClassifier(ctx, stub), want_feats=True)
    cfg = O.make_cfg(pb, rb=[(pb, bins, 0.0, 1.0)])
    o_ref, s_ref, f_ref = O.Rag(labels).merge_order_bc(cfg, None, stub_index=stub, want_feats=True)
    assert order.shape == o_ref.shape and (order == o_ref).all()
    assert _feat_close(feats, f_ref)
    if variant == 0:
        assert (sal == s_ref).all()
    else:
        assert np.allclose(sal, s_ref, rtol=0, atol=1e-12)


@pytest.mark.parametrize("shape,S,G,ntree,depth", [((32, 32, 32), 8, 16, 31, 6), ((40, 36, 28), 6, 12, 255, 10),
                                                   ((64, 64), 4, 16, 15, 4)])
def test_random_forest_matches_oracle(ctx, shape, S, G, ntree, depth):
    from glia_amd import hmt
    from oracle import pyoracle as O
    import _rf
    labels, pb = O.synth(shape, S, G)
    cfg = O.make_cfg(pb, rb=[(pb, 8, 0.0, 1.0)])
    # feature rows to draw split thresholds from: the oracle's own vectors for a stub-scored run
    _, _, f0 = O.Rag(labels).merge_order_bc(cfg, None, stub_index=31 if len(shape) == 3 else 30, want_feats=True)
    rng = np.random.default_rng(7)
    forest = _rf.random_forest(rng, ntree, depth, f0)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "model.bin")
        _rf.write_model(path, forest)
        clf = hmt.RandomForest(ctx, path, predict_label=-1)
    rm = _gpu_rm(ctx, labels, pb)
    order, sal, feats = rm.merge_order_bc(clf, want_feats=True)
    o_ref, s_ref, f_ref = O.Rag(labels).merge_order_bc(cfg, O.make_forest(forest, -1), want_feats=True)
    assert order.shape == o_ref.shape and (order == o_ref).all()
    assert (sal == s_ref).all()              # votes / ntree: exact
    assert _feat_close(feats, f_ref)
    # the forest is walked by helper workgroups (63 by default); the result may not depend on how many there are:
    # 0 = the contraction workgroup walks the trees itself, 5 = every helper takes several records of a chunk
    for nh in ("0", "5"):
        with hmt.options(GLIA_HMT_HELPERS=nh):
            o2, s2 = rm.merge_order_bc(clf)[:2]
        assert (o2 == o_ref).all() and (s2 == s_ref).all(), "helpers=" + nh


@pytest.mark.parametrize("shape,S,G,ntrees", [((32, 32, 32), 8, 16, (31, 31, 31)), ((40, 36, 28), 6, 12, (15, 63, 7)), ((64, 64), 4, 16, (7, 7, 31))])
def test_ensemble_random_forest_matches_oracle(ctx, shape, S, G, ntrees):
    """alg::EnsembleRandomForest + opt::ThresholdModelDistributor(dim0, dim1, threshold) (alg/rf.hxx:63-98,
    type/function.hxx:71-85, hmt/main_merge_order_bc.cxx:103-109): model 0 if x[dim1] < thr, else 1 if x[dim0] < thr, else
    2.  The distributor reads the two region areas (features rfoff and rfoff + rfdim), the threshold is their median over a
    stub run, so all three members score edges; members may have different tree counts."""
    from glia_amd import hmt
    from oracle import pyoracle as O
    import _rf
    labels, pb = O.synth(shape, S, G)
    cfg = O.make_cfg(pb, rb=[(pb, 8, 0.0, 1.0)])
    dim = len(shape)
    _, _, f0 = O.Rag(labels).merge_order_bc(cfg, None, stub_index=31 if dim == 3 else 30, want_feats=True)
    bf = 11 + 4 * 3 + 7 + 5
    rf = 4 + dim + 2 * 3 + 5 + 5
    dim0, dim1 = bf, bf + rf                       # area of the smaller / of the larger region
    thr = float(np.median(np.concatenate([f0[:, dim0], f0[:, dim1]]))) + 0.5
    rng = np.random.default_rng(17)
    forests = [_rf.random_forest(rng, nt, 5 + k, f0) for k, nt in enumerate(ntrees)]
    with tempfile.TemporaryDirectory() as d:
        paths = []
        for k, f in enumerate(forests):
            paths.append(os.path.join(d, "m%d.bin" % k)); _rf.write_model(paths[-1], f)
        clf = hmt.RandomForest(ctx, paths, predict_label=-1, distributor_args=(dim0, dim1, thr))
    rm = _gpu_rm(ctx, labels, pb)
    order, sal, feats = rm.merge_order_bc(clf, want_feats=True)
    oforests = [O.make_forest(f, -1) for f in forests]
    o_ref, s_ref, f_ref = O.Rag(labels).merge_order_bc_ensemble(cfg, oforests, dim0, dim1, thr, want_feats=True)
    # every member was used
    used = np.where(f_ref[:, dim1] < thr, 0, np.where(f_ref[:, dim0] < thr, 1, 2))
    assert set(used.tolist()) == {0, 1, 2}
    assert order.shape == o_ref.shape and (order == o_ref).all()
    assert (sal == s_ref).all()
    assert _feat_close(feats, f_ref)
    for nh in ("0", "5"):
        with hmt.options(GLIA_HMT_HELPERS=nh):
            o2, s2 = rm.merge_order_bc(clf)[:2]
        assert (o2 == o_ref).all() and (s2 == s_ref).all(), "helpers=" + nh


def test_fuzz_near_tie_case_of_round_1(ctx):
    """The one classifier-path mismatch round 1's fuzz found (profiles/r01i_fuzz_summary.txt, seed 31337 case 2159): the
    scorer is a bare entropy feature, three edges with permuted histograms tie EXACTLY under glibc's log2 and came out
    1-2 ulp apart under the device libm, so the tie broke differently.  With the host libm's log2 restated on the device
    (glibc_math.hpp) order, saliencies and entropy features are bit-identical.  Fixture: tests/golden/make_fuzz_fixture.py."""
    import torch
    from glia_amd import hmt
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_seed31337_case2159.npz"))
    labels, pb, raw = g["labels"], g["pb"], g["raw"]
    assert int(g["layout"]) == 2
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda()
    d_pb, d_raw = torch.from_numpy(pb).cuda(), torch.from_numpy(raw).cuda()
    bins = int(g["bins"])
    cfg = hmt.make_config(d_pb, r=[(d_raw, bins, 0.0, 1.0)], b=[(d_pb, 8, 0.0, 1.0)], rl=[(d_raw, 4, 0.0, 1.0)],
                          use_log_shape=bool(g["use_log"]), use_simple_features=bool(g["use_simple"]))
    rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, cfg=cfg)
    o, s, f = rm.merge_order_bc(hmt.FeatureStubClassifier(ctx, int(g["stub"])), want_feats=True)
    assert o.shape == g["order"].shape and (o == g["order"]).all()
    assert (s == g["saliency"]).all()
    assert _feat_close(f, g["feats"])
    # the same through a fresh oracle run (the fixture's expected values came from it)
    from oracle import pyoracle as O
    ocfg = O.make_cfg(pb, r=[(raw, bins, 0.0, 1.0)], b=[(pb, 8, 0.0, 1.0)], rl=[(raw, 4, 0.0, 1.0)],
                      use_log=bool(g["use_log"]), use_simple=bool(g["use_simple"]))
    ro, rs = O.Rag(labels).merge_order_bc(ocfg, None, stub_index=int(g["stub"]))
    assert (ro == g["order"]).all() and (rs == g["saliency"]).all()


def test_log_and_simple_features(ctx):
    from glia_amd import hmt
    from oracle import pyoracle as O
    labels, pb = O.synth((32, 32, 32), 8, 16)
    for kw, okw in [(dict(use_log_shape=True), dict(use_log=True)),
                    (dict(use_simple_features=True), dict(use_simple=True)),
                    (dict(normalizing_area=32.0 ** 3, normalizing_length=32.0 * 3 ** 0.5),
                     dict(norm_area=32.0 ** 3, norm_len=32.0 * 3 ** 0.5))]:
        rm = _gpu_rm(ctx, labels, pb, **kw)
        cfg = O.make_cfg(pb, rb=[(pb, 8, 0.0, 1.0)], **okw)
        stub = 5 if "use_simple_features" in kw else 34
        order, sal, feats = rm.merge_order_bc(hmt.FeatureStubClassifier(ctx, stub), want_feats=True)
        o_ref, s_ref, f_ref = O.Rag(labels).merge_order_bc(cfg, None, stub_index=stub, want_feats=True)
        assert feats.shape == f_ref.shape
        assert (order == o_ref).all() and _feat_close(feats, f_ref)


@pytest.mark.parametrize("shape,S,G,layout", [((32, 32, 32), 8, 16, 0), ((40, 36, 28), 6, 12, 1), ((64, 64), 4, 16, 2), ((32, 32, 32), 8, 16, 3)])
def test_histogram_as_features_layout(ctx, shape, S, G, layout):
    """GLIA_HMT_HIST_FEAT / GLIA_USE_HISTOGRAM_AS_FEATS (CMakeLists.txt:54-58, type/feat.hxx:608-621): every image block carries
    its normalised histogram ahead of the entropy -- D_f grows by the bins of every block; order, saliencies and rows vs the oracle,
    also with --simpf (whose boundary-image means move) and --logs, and for a given order (bc_feat)."""
    import torch
    from glia_amd import hmt
    from oracle import pyoracle as O
    labels, pb = O.synth(shape, S, G)
    rng = np.random.default_rng(21)
    raw = (np.round(rng.random(shape) * 255) / 256.0).astype(np.float32)
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda()
    d_pb, d_raw = torch.from_numpy(pb).cuda(), torch.from_numpy(raw).cuda()
    if layout == 0: okw, dkw, flags = dict(rb=[(pb, 8, 0.0, 1.0)]), dict(rb=[(d_pb, 8, 0.0, 1.0)]), {}
    elif layout == 1: okw, dkw, flags = dict(rb=[(raw, 16, 0.0, 1.0), (pb, 8, 0.0, 1.0)]), dict(rb=[(d_raw, 16, 0.0, 1.0), (d_pb, 8, 0.0, 1.0)]), dict(use_log=True)
    elif layout == 2: okw, dkw, flags = dict(r=[(raw, 4, 0.0, 1.0)], b=[(pb, 8, 0.0, 1.0)], rl=[(raw, 4, 0.0, 1.0)]), dict(r=[(d_raw, 4, 0.0, 1.0)], b=[(d_pb, 8, 0.0, 1.0)], rl=[(d_raw, 4, 0.0, 1.0)]), {}
    else: okw, dkw, flags = dict(rb=[(pb, 8, 0.0, 1.0)], b=[(raw, 4, 0.0, 1.0)]), dict(rb=[(d_pb, 8, 0.0, 1.0)], b=[(d_raw, 4, 0.0, 1.0)]), dict(use_simple=True)
    ocfg = O.make_cfg(pb, hist_as_feats=True, **okw, **flags)
    plain = O.make_cfg(pb, **okw, **flags)
    cfg = hmt.make_config(d_pb, use_histogram_features=True, use_log_shape=flags.get("use_log", False),
                          use_simple_features=flags.get("use_simple", False), **dkw)
    rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, cfg=cfg)
    fd = rm.feat_dim()
    assert fd == O.feat_dim(len(shape), ocfg)
    if not flags.get("use_simple"):
        assert fd > O.feat_dim(len(shape), plain)                 # the layout really is wider
    stub = fd - 3
    order, sal, feats = rm.merge_order_bc(hmt.FeatureStubClassifier(ctx, stub), want_feats=True)
    o_ref, s_ref, f_ref = O.Rag(labels).merge_order_bc(ocfg, None, stub_index=stub, want_feats=True)
    assert feats.shape == f_ref.shape and (order == o_ref).all() and (sal == s_ref).all()
    assert _feat_close(feats, f_ref)
    # bc_feat for a given order
    po, _ = O.Rag(labels, only_contour=True).merge_order_pb(pb, type=2)
    assert _feat_close(rm.bc_feat(po), O.Rag(labels).bc_feat(ocfg, po))
    rm.close()
    # the median layout inside the greedy LOOP is refused, not silently ignored (bc_feat has it: test_median_as_features_for_a_given_order)
    rm_med = hmt.RegionMap(ctx, d_lab, pb=d_pb, cfg=hmt.make_config(d_pb, use_median_features=True, **dkw))
    with pytest.raises(hmt.HmtError, match="given merge order only"):
        rm_med.merge_order_bc(hmt.FeatureStubClassifier(ctx, stub))
    rm_med.close()


def test_full_vector_longer_than_the_kernel_buffers_is_refused(ctx):
    """with --histf the FULL vector (assembled before --simpf compacts it) of three 16-bin images on the region
    and the boundary list has 524 columns although the --simpf selection keeps 20: the per-thread buffers hold kMaxFeat = 384, so
    the call must come back with an error, not overrun them; a layout that fits still matches the oracle."""
    import torch
    from glia_amd import hmt
    from oracle import pyoracle as O
    shape = (32, 32, 32)
    labels, pb = O.synth(shape, 8, 16)
    rng = np.random.default_rng(77)
    raws = [(np.round(rng.random(shape) * 255) / 256.0).astype(np.float32) for _ in range(2)]
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda()
    d_pb = torch.from_numpy(pb).cuda()
    d_raws = [torch.from_numpy(r).cuda() for r in raws]
    big = hmt.make_config(d_pb, use_histogram_features=True, use_simple_features=True,
                          rb=[(d_pb, 16, 0.0, 1.0), (d_raws[0], 16, 0.0, 1.0), (d_raws[1], 16, 0.0, 1.0)])
    rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, cfg=big)
    assert rm.feat_dim() < 64                                    # the selection is short ...
    with pytest.raises(hmt.HmtError) as ei:
        rm.merge_order_bc(hmt.FeatureStubClassifier(ctx, 5))
    assert "too long" in str(ei.value)                           # ... the full vector is not
    with pytest.raises(hmt.HmtError):
        rm.bc_feat(O.Rag(labels, only_contour=True).merge_order_pb(pb, type=2)[0])
    rm.close()
    # five images on one list (more than the old limit of four), three distinct volumes, 4 bins: fits, and matches the oracle
    okw = dict(r=[(raws[0], 4, 0.0, 1.0), (raws[1], 4, 0.0, 1.0), (pb, 4, 0.0, 1.0), (raws[0], 4, 0.0, 1.0), (raws[1], 4, 0.0, 1.0)])
    dkw = dict(r=[(d_raws[0], 4, 0.0, 1.0), (d_raws[1], 4, 0.0, 1.0), (d_pb, 4, 0.0, 1.0), (d_raws[0], 4, 0.0, 1.0), (d_raws[1], 4, 0.0, 1.0)])
    ocfg = O.make_cfg(pb, **okw)
    rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, cfg=hmt.make_config(d_pb, **dkw))
    fd = rm.feat_dim()
    assert fd == O.feat_dim(3, ocfg)
    order, sal, feats = rm.merge_order_bc(hmt.FeatureStubClassifier(ctx, fd - 3), want_feats=True)
    o_ref, s_ref, f_ref = O.Rag(labels).merge_order_bc(ocfg, None, stub_index=fd - 3, want_feats=True)
    assert (order == o_ref).all() and (sal == s_ref).all() and _feat_close(feats, f_ref)
    rm.close()


@pytest.mark.parametrize("shape,S,G", [((32, 32, 32), 8, 16), ((40, 36, 28), 6, 12), ((64, 64), 4, 16)])
def test_bc_feat_for_a_given_order(ctx, shape, S, G):
    """hmt/main_bc_feat.cxx path: features of every merge of a GIVEN order (here: the pb-mean order)."""
    from oracle import pyoracle as O
    labels, pb = O.synth(shape, S, G)
    order, _ = O.Rag(labels, only_contour=True).merge_order_pb(pb, type=2)
    rm = _gpu_rm(ctx, labels, pb)
    feats = rm.bc_feat(order)
    ref = O.Rag(labels).bc_feat(O.make_cfg(pb, rb=[(pb, 8, 0.0, 1.0)]), order)
    assert feats.shape == ref.shape and _feat_close(feats, ref)
    # a merge order that joins non-neighbouring regions first (no shared record)
    labs = np.unique(labels)
    far = np.array([[labs[0], labs[-1], labs.max() + 1]], dtype=np.uint32)
    f2 = rm.bc_feat(far)
    r2 = O.Rag(labels).bc_feat(O.make_cfg(pb, rb=[(pb, 8, 0.0, 1.0)]), far)
    assert _feat_close(f2, r2)


def test_classifier_linkage_with_mask(ctx):
    """merge_order_bc -m: RegionMap(seg, mask, false): masked voxels belong to no region, masked neighbours are invalid"""
    import torch
    from glia_amd import hmt
    from oracle import pyoracle as O
    labels, pb = O.synth((32, 32, 32), 8, 16)
    rng = np.random.default_rng(12)
    mask = (rng.random(labels.shape) > 0.1).astype(np.uint32)
    mask[:, :, :4] = 0
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda()
    d_pb = torch.from_numpy(pb).cuda()
    d_mask = torch.from_numpy(mask.view(np.int32)).cuda()
    cfg = hmt.make_config(d_pb, rb=[(d_pb, 8, 0.0, 1.0)])
    rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, mask=d_mask, cfg=cfg)
    stub = 11 + 4 * 3 + 7 + 1
    order, sal, feats = rm.merge_order_bc(hmt.FeatureStubClassifier(ctx, stub), want_feats=True)
    ocfg = O.make_cfg(pb, rb=[(pb, 8, 0.0, 1.0)])
    o_ref, s_ref, f_ref = O.Rag(labels, mask=mask).merge_order_bc(ocfg, None, stub_index=stub, want_feats=True)
    assert order.shape == o_ref.shape and (order == o_ref).all()
    assert (sal == s_ref).all() and _feat_close(feats, f_ref)


def _aux_images(shape, seed):
    """two extra Q8 volumes (a smooth 'raw intensity' and a blocky 'texton label' image)"""
    rng = np.random.default_rng(seed)
    z = np.indices(shape).astype(np.float64)
    raw = 0.5 + 0.25 * np.sin(z[0] / 3.0) * np.cos(z[-1] / 4.0) + 0.2 * rng.random(shape)
    raw = (np.clip(np.round(raw * 255), 0, 255) / 256.0).astype(np.float32)
    tex = ((z[0] // 3 + 2 * (z[-1] // 5) + (z[1] // 4 if len(shape) == 3 else 0)) % 7).astype(np.float32)
    return raw, tex


@pytest.mark.parametrize("layout", ["rb2", "split", "pb_unlisted", "four"])
@pytest.mark.parametrize("shape,S,G", [((32, 32, 32), 8, 16), ((48, 40), 4, 16)])
def test_feature_lists_with_several_image_volumes(ctx, layout, shape, S, G):
    """prepareImages (hmt/hmt_util.hxx:17-56): --rbi images feed both lists, --ri / --bi / --rli one each; every distinct
    (volume, histogram) is a channel of its own accumulation pass.  Orders, saliencies, feature rows vs the oracle."""
    import torch
    from glia_amd import hmt
    from oracle import pyoracle as O
    labels, pb = O.synth(shape, S, G)
    raw, tex = _aux_images(shape, 21)
    d = lambda a: torch.from_numpy(a).cuda()
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda()
    d_pb, d_raw, d_tex = d(pb), d(raw), d(tex)
    if layout == "rb2":            # --rbi raw --rbi pb
        kw = dict(rb=[("raw", 8, 0.0, 1.0), ("pb", 8, 0.0, 1.0)])
    elif layout == "split":        # --ri raw --bi pb --rli textons (16 bins over the label range)
        kw = dict(r=[("raw", 8, 0.0, 1.0)], b=[("pb", 8, 0.0, 1.0)], rl=[("tex", 16, -0.5, 7.5)])
    elif layout == "pb_unlisted":  # pb only drives the shape features
        kw = dict(rb=[("raw", 16, 0.0, 1.0)])
    else:                          # the same volume with two histogram specs + everything else
        kw = dict(rb=[("pb", 8, 0.0, 1.0), ("raw", 8, 0.0, 1.0)], r=[("pb", 4, 0.0, 1.0)], rl=[("tex", 8, -0.5, 7.5)])
    host = {"raw": raw, "pb": pb, "tex": tex}
    dev = {"raw": d_raw, "pb": d_pb, "tex": d_tex}
    okw = {k: [(host[n], b, lo, hi) for n, b, lo, hi in v] for k, v in kw.items()}
    dkw = {k: [(dev[n], b, lo, hi) for n, b, lo, hi in v] for k, v in kw.items()}
    cfg = hmt.make_config(d_pb, **dkw)
    ocfg = O.make_cfg(pb, **okw)
    rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, cfg=cfg)
    dimf = rm.feat_dim()
    assert dimf == O.feat_dim(len(shape), ocfg)
    stub = 11 + 4 * 3 + 1          # a histogram distance of the first region / label image
    order, sal, feats = rm.merge_order_bc(hmt.FeatureStubClassifier(ctx, stub), want_feats=True)
    o_ref, s_ref, f_ref = O.Rag(labels).merge_order_bc(ocfg, None, stub_index=stub, want_feats=True)
    assert order.shape == o_ref.shape and (order == o_ref).all()
    assert (sal == s_ref).all() and _feat_close(feats, f_ref)
    # bc_feat for the given order goes through the forced-order mode of the same kernel
    f2 = rm.bc_feat(order)
    assert _feat_close(f2, O.Rag(labels).bc_feat(ocfg, o_ref))
    rm.close()


def test_several_image_volumes_with_a_forest_and_simple_features(ctx):
    import torch
    from glia_amd import hmt
    from oracle import pyoracle as O
    import _rf
    shape = (32, 32, 32)
    labels, pb = O.synth(shape, 8, 16)
    raw, tex = _aux_images(shape, 5)
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda()
    d_pb, d_raw, d_tex = (torch.from_numpy(a).cuda() for a in (pb, raw, tex))
    for simple in (False, True):
        ocfg = O.make_cfg(pb, rb=[(raw, 8, 0.0, 1.0), (pb, 8, 0.0, 1.0)], rl=[(tex, 8, -0.5, 7.5)], use_simple=simple)
        cfg = hmt.make_config(d_pb, rb=[(d_raw, 8, 0.0, 1.0), (d_pb, 8, 0.0, 1.0)], rl=[(d_tex, 8, -0.5, 7.5)], use_simple_features=simple)
        _, _, f0 = O.Rag(labels).merge_order_bc(ocfg, None, stub_index=4, want_feats=True)
        forest = _rf.random_forest(np.random.default_rng(9), 63, 7, f0)
        with tempfile.TemporaryDirectory() as dd:
            path = os.path.join(dd, "model.bin")
            _rf.write_model(path, forest)
            clf = hmt.RandomForest(ctx, path, predict_label=-1)
        rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, cfg=cfg)
        order, sal, feats = rm.merge_order_bc(clf, want_feats=True)
        o_ref, s_ref, f_ref = O.Rag(labels).merge_order_bc(ocfg, O.make_forest(forest, -1), want_feats=True)
        assert order.shape == o_ref.shape and (order == o_ref).all()
        assert (sal == s_ref).all() and _feat_close(feats, f_ref)
        rm.close()


@pytest.mark.parametrize("use_log", [False, True])
def test_bc_feat_with_saliency_features(ctx, use_log):
    """bc_feat -y: genSaliencyMap + the five saliency columns (hmt/bc_feat.hxx:12-26,76,163-166,208-213)"""
    import torch
    from glia_amd import hmt
    from oracle import pyoracle as O
    labels, pb = O.synth((32, 32, 32), 8, 16)
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda()
    d_pb = torch.from_numpy(pb).cuda()
    cfg = hmt.make_config(d_pb, rb=[(d_pb, 8, 0.0, 1.0)], use_log_shape=use_log)
    ocfg = O.make_cfg(pb, rb=[(pb, 8, 0.0, 1.0)], use_log=use_log)
    order, sal = O.Rag(labels, only_contour=True).merge_order_pb(pb, type=2)
    rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, cfg=cfg)
    got = rm.bc_feat(order, saliencies=sal, init_sal=0.75, sal_bias=1.5)
    ref = O.Rag(labels).bc_feat(ocfg, order, saliencies=sal, init_sal=0.75, sal_bias=1.5)
    assert got.shape == ref.shape == (len(order), 104 + 5) and _feat_close(got, ref)
    assert (got[:, 35] <= got[:, 36]).all()               # (min, max) of the two saliency differences
    rm.close()


def test_sharded_initial_edge_scoring(ctx):
    """the scores of the initial edges are independent, so ranks can own residue classes of the records"""
    import torch
    from glia_amd import hmt
    from oracle import pyoracle as O
    import _rf
    labels, pb = O.synth((32, 32, 32), 8, 16)
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda()
    d_pb = torch.from_numpy(pb).cuda()
    cfg = hmt.make_config(d_pb, rb=[(d_pb, 8, 0.0, 1.0)])
    ocfg = O.make_cfg(pb, rb=[(pb, 8, 0.0, 1.0)])
    _, _, f0 = O.Rag(labels).merge_order_bc(ocfg, None, stub_index=31, want_feats=True)
    forest = _rf.random_forest(np.random.default_rng(7), 31, 6, f0)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "model.bin")
        _rf.write_model(path, forest)
        clf = hmt.RandomForest(ctx, path, predict_label=-1)
    rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, cfg=cfg)
    full = rm.score_initial_edges_shard(clf, 0, 1)
    parts = [rm.score_initial_edges_shard(clf, r, 3) for r in range(3)]
    assert all(len(p) == len(full) for p in parts)
    assert (np.maximum.reduce(parts) == full).all()
    fin = [np.isfinite(p) for p in parts]
    assert not (fin[0] & fin[1]).any() and not (fin[1] & fin[2]).any() and np.isfinite(full).sum() == sum(f.sum() for f in fin)
    # the first merge of the classifier loop takes the best initial score
    order, sal = rm.merge_order_bc(clf)
    assert sal[0] == full[np.isfinite(full)].max()
    rm.close()


@pytest.mark.parametrize("shape,S,G", [((40, 36, 28), 6, 12), ((72, 72), 4, 16)])
def test_loop_instances_agree(ctx, shape, S, G):
    """greedy_bc.hip is compiled five times (hmt_internal.hpp): everything chosen at run time, the libm variants fixed, and
    libm + the common configuration (one image, full vector) fixed.  The dispatcher's two overrides force the less specialised
    instances; all three tiers must return the oracle's merge order and saliencies and bit-identical feature rows."""
    from glia_amd import hmt
    from oracle import pyoracle as O
    labels, pb = O.synth(shape, S, G)
    stub = 11 + 4 * 3 + 7 + 1
    results = []
    for env in ({}, {"GLIA_HMT_BC_NOCOMMON": "1"}, {"GLIA_HMT_BC_GENERIC": "1"}):
        with hmt.options(**env):
            rm = _gpu_rm(ctx, labels, pb)
            results.append(rm.merge_order_bc(hmt.FeatureStubClassifier(ctx, stub), want_feats=True))
            rm.close()
    cfg = O.make_cfg(pb, rb=[(pb, 8, 0.0, 1.0)])
    o_ref, s_ref, f_ref = O.Rag(labels).merge_order_bc(cfg, None, stub_index=stub, want_feats=True)
    for o, s, f in results:
        assert o.shape == o_ref.shape and (o == o_ref).all() and (s == s_ref).all()
        assert (f.view(np.uint64) == results[0][2].view(np.uint64)).all()
    assert (results[0][2].view(np.uint64) == f_ref.view(np.uint64)).all(), "feature rows are bit-identical to the oracle on Q8 inputs"


def _median_loose_mask(dim, n_thr, n_r, n_rl, n_b, hist_cols=(0, 0, 0)):
    """columns of the full median-layout vector that hold a mean / stddev (or their differences): another summation order than the
    reference's (which depends on rand(), util/stats.hxx:87) -- comparable to 1e-12; every other column bit for bit"""
    hr, hl, hbb = hist_cols
    loose = []
    pos = 11 + 4 * n_thr
    for _ in range(n_r):
        pos += 3; loose += [pos + 1, pos + 2]; pos += 5         # medD | meanD stdD | minD maxD
    pos += 3 * n_rl
    for _ in range(n_b):
        pos += hbb + 1; loose += [pos + 1, pos + 2]; pos += 5   # [hist] entropy | median | mean std | min max
    for _ in range(3):
        pos += 4 + dim + 2 * n_thr
        for _ in range(n_r):
            pos += hr + 1; loose += [pos + 1, pos + 2]; pos += 5
        pos += (hl + 1) * n_rl
        for _ in range(n_b):
            pos += hbb + 1; loose += [pos + 1, pos + 2]; pos += 5
    m = np.zeros(pos, bool); m[loose] = True
    return m


@pytest.mark.parametrize("shape,S,G,layout", [((32, 32, 32), 8, 16, "rb"), ((40, 36, 28), 6, 12, "four"), ((64, 64), 4, 16, "split"), ((24, 30, 22), 5, 10, "rb2")])
def test_median_as_features_for_a_given_order(ctx, shape, S, G, layout):
    """GLIA_USE_MEDIAN_AS_FEATS (CMakeLists.txt:55,62-64; type/feat.hxx:677-722, 772-808; hmt/bc_feat.hxx:252-268) through
    glia_hmt_bc_feat: D_f and the rows of the oracle's bc_feat with median_as_feats -- medians and every column the default layout
    has bit for bit, the mean / stddev columns (taken from the value vector: stats::mean, stats::var) to 1e-12 -- on four list
    layouts in 2D and 3D, with --simpf, with the saliency columns, and with a mask."""
    import torch
    from glia_amd import hmt
    from oracle import pyoracle as O
    dim = len(shape)
    labels, pb = O.synth(shape, S, G)
    rng = np.random.default_rng(5)
    raw = (np.round(rng.random(shape) * 255) / 256.0).astype(np.float32)
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda()
    d_pb, d_raw = torch.from_numpy(pb).cuda(), torch.from_numpy(raw).cuda()
    if layout == "rb": okw, dkw, dims = dict(rb=[(pb, 8, 0.0, 1.0)]), dict(rb=[(d_pb, 8, 0.0, 1.0)]), (1, 0, 1)
    elif layout == "rb2": okw, dkw, dims = dict(rb=[(raw, 16, 0.0, 1.0), (pb, 8, 0.0, 1.0)]), dict(rb=[(d_raw, 16, 0.0, 1.0), (d_pb, 8, 0.0, 1.0)]), (2, 0, 2)
    elif layout == "split": okw, dkw, dims = (dict(r=[(raw, 4, 0.0, 1.0)], b=[(pb, 8, 0.0, 1.0)], rl=[(raw, 4, 0.0, 1.0)]),
                                              dict(r=[(d_raw, 4, 0.0, 1.0)], b=[(d_pb, 8, 0.0, 1.0)], rl=[(d_raw, 4, 0.0, 1.0)]), (1, 1, 1))
    else: okw, dkw, dims = (dict(rb=[(pb, 8, 0.0, 1.0)], r=[(raw, 4, 0.0, 1.0)], rl=[(raw, 4, 0.0, 1.0)], b=[(raw, 8, 0.0, 1.0)]),
                            dict(rb=[(d_pb, 8, 0.0, 1.0)], r=[(d_raw, 4, 0.0, 1.0)], rl=[(d_raw, 4, 0.0, 1.0)], b=[(d_raw, 8, 0.0, 1.0)]), (2, 1, 2))
    n_r, n_rl, n_b = dims
    order, sal = O.Rag(labels, only_contour=True).merge_order_pb(pb, type=2)
    plain = O.make_cfg(pb, **okw)
    ocfg = O.make_cfg(pb, median_as_feats=True, **okw)
    rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, cfg=hmt.make_config(d_pb, use_median_features=True, **dkw))
    got = rm.bc_feat(order)
    ref = O.Rag(labels).bc_feat(ocfg, order)
    assert got.shape == ref.shape and got.shape[1] == O.feat_dim(dim, plain) + 4 * n_r + 4 * n_b
    loose = _median_loose_mask(dim, 3, n_r, n_rl, n_b)
    assert len(loose) == got.shape[1]
    assert (got[:, ~loose].view(np.uint64) == ref[:, ~loose].view(np.uint64)).all(), "medians and every statistic-derived column are bit-identical"
    assert np.allclose(got[:, loose], ref[:, loose], rtol=1e-12, atol=1e-13)
    # with the saliency columns (bc_feat -y) on top
    got_s = rm.bc_feat(order, saliencies=sal, init_sal=0.75, sal_bias=1.5)
    ref_s = O.Rag(labels).bc_feat(ocfg, order, saliencies=sal, init_sal=0.75, sal_bias=1.5)
    assert got_s.shape == ref_s.shape == (len(order), got.shape[1] + 5) and np.allclose(got_s, ref_s, rtol=1e-12, atol=1e-13)
    rm.close()
    # --simpf: the shared boundary's median beside its mean (bc_feat.hxx:263-268)
    rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, cfg=hmt.make_config(d_pb, use_median_features=True, use_simple_features=True, **dkw))
    got = rm.bc_feat(order)
    ref = O.Rag(labels).bc_feat(O.make_cfg(pb, median_as_feats=True, use_simple=True, **okw), order)
    assert got.shape == ref.shape and got.shape[1] == 5 + 2 * n_b + 4 * n_r + 2 * n_rl
    med_cols = [5 + 2 * j + 1 for j in range(n_b)]
    assert (got[:, med_cols] == ref[:, med_cols]).all() and np.allclose(got, ref, rtol=1e-12, atol=1e-13)
    rm.close()
    if layout == "four":
        # a mask: masked-out voxels leave every value multiset (point-map mode, util/struct.hxx:86-91); histogram columns on top
        mask = (rng.random(shape) > 0.15).astype(np.uint32)
        d_mask = torch.from_numpy(mask.view(np.int32)).cuda()
        rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, mask=d_mask, cfg=hmt.make_config(d_pb, use_median_features=True, use_histogram_features=True, **dkw))
        morder, _ = O.Rag(labels, mask=mask, only_contour=True).merge_order_pb(pb, type=2)
        got = rm.bc_feat(morder)
        ref = O.Rag(labels, mask=mask).bc_feat(O.make_cfg(pb, median_as_feats=True, hist_as_feats=True, **okw), morder)
        assert got.shape == ref.shape and np.allclose(got, ref, rtol=1e-12, atol=1e-13)
        rm.close()
