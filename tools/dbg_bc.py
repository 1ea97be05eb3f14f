import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, 'tests')
import numpy as np, torch
from glia_amd import hmt
from oracle import pyoracle as O
ctx = hmt.Context(0)
for shape, S, G, stub in [((64, 64), 4, 16, 30), ((32, 32, 32), 8, 16, 31), ((40, 36, 28), 6, 12, 31)]:
    labels, pb = O.synth(shape, S, G)
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda(); d_pb = torch.from_numpy(pb).cuda()
    cfg = hmt.make_config(d_pb, rb=[(d_pb, 8, 0.0, 1.0)])
    rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, cfg=cfg)
    order, sal, feats = rm.merge_order_bc(hmt.FeatureStubClassifier(ctx, stub), want_feats=True)
    ocfg = O.make_cfg(pb, rb=[(pb, 8, 0.0, 1.0)])
    o_ref, s_ref, f_ref = O.Rag(labels).merge_order_bc(ocfg, None, stub_index=stub, want_feats=True)
    print(shape, 'order equal', (order == o_ref).all(), 'feat dim', feats.shape)
    neq = feats != f_ref
    cols = np.flatnonzero(neq.any(0))
    print(' columns with bitwise differences:', cols.tolist())
    for c in cols[:12]:
        r = np.flatnonzero(neq[:, c])[0]
        print('  col', c, 'rows', neq[:, c].sum(), 'example', repr(feats[r, c]), repr(f_ref[r, c]), 'rel', abs(feats[r,c]-f_ref[r,c])/max(abs(f_ref[r,c]),1e-300))
