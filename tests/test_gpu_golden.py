"""-m gpu: the HIP path, through the C ABI, reproduces the committed golden fixtures (no oracle in the loop)."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = sorted(p for p in glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz"))
              if not os.path.basename(p).startswith("fuzz_"))          # fuzz fixtures have their own test (test_gpu_bc.py)


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "GPU test run without a GPU"
    from glia_amd import hmt
    c = hmt.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_device_reproduces_golden(ctx, path):
    import torch
    from glia_amd import hmt
    g = dict(np.load(path))
    labels = g["labels"].astype(np.uint32)
    pb = (g["pb_q8"].astype(np.float32) / np.float32(256.0)).astype(np.float32)
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda()
    d_pb = torch.from_numpy(pb).cuda()
    d_mask = torch.from_numpy(g["mask"].astype(np.uint32).view(np.int32)).cuda() if "mask" in g else None
    rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, mask=d_mask, only_contour=True)
    for typ, key in ((2, "mean"), (1, "median")):
        o, s = rm.merge_order_pb(type=typ)
        assert (o == g["pb_%s_order" % key]).all() and (s == g["pb_%s_sal" % key]).all()
    rm.close()
    cfg = hmt.make_config(d_pb, rb=[(d_pb, 8, 0.0, 1.0)])
    rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, mask=d_mask, cfg=cfg)
    o, s, f = rm.merge_order_bc(hmt.FeatureStubClassifier(ctx, int(g["bc_stub_index"])), want_feats=True)
    assert (o == g["bc_order"]).all() and (s == g["bc_sal"]).all()
    n = len(g["bc_feats_head"])
    assert np.allclose(f[:n], g["bc_feats_head"], rtol=1e-5, atol=1e-12)       # the north star's tolerance; measured ~1e-16
    rm.close()
    lab, par, c0, c1, pot = hmt.tree_potentials(o, np.clip(s, 0.0, 1.0))
    picks = hmt.resolve_tree_greedy(par, c0, c1, pot)
    assert (picks == g["tree_picks"]).all()
    src, dst = hmt.label_transform(lab, c0, c1, picks, 1)
    work = d_lab.clone()
    hmt.transform_image(ctx, work, src, dst, mask=d_mask, fill_missing=True)
    assert (work.cpu().numpy().view(np.uint32) == g["final_labels"]).all()
