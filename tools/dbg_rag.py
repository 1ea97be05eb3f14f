import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from glia_amd import hmt
from oracle import pyoracle as O
sys.path.insert(0, '/root/repo/tests')
import test_gpu_rag as T
ctx = hmt.Context()
for (shape, S, G, variant, bins) in T.CASES[:3]:
    labels, pb = O.synth(shape, S, G, variant=variant)
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda(); d_pb = torch.from_numpy(pb).cuda()
    cfg = hmt.make_config(d_pb, rb=[(d_pb, bins, 0.0, 1.0)], thresholds=(0.2, 0.5, 0.8))
    rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, cfg=cfg)
    ref = T._oracle_rag(labels, pb, bins, 0.0, 1.0, (0.2, 0.5, 0.8))
    reg = rm.regions()
    for k, r in (("lo", "lo"), ("hi", "hi"), ("count", "npts"), ("min", "rmin"), ("max", "rmax"), ("sum", "rsum"), ("hist", "rhist")):
        if r in ref:
            bad = np.argwhere(np.asarray(reg[k]) != np.asarray(ref[r]))
            print(shape, k, "mismatches", len(bad), bad[:5].tolist(), np.asarray(reg[k])[bad[:3, 0]].tolist() if len(bad) else "", np.asarray(ref[r])[bad[:3, 0]].tolist() if len(bad) else "")
