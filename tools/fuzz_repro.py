"""Repeats the RAG build + pre_merge of a volume dumped by tests/fuzz_gpu.py and reports how often the result differs from the
oracle's, and whether the statistics differ between runs.  usage: fuzz_repro.py file.npz sizes rpb [runs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from glia_amd import hmt
from oracle import pyoracle as O
d = np.load(sys.argv[1])
labels, pb, mask = d["labels"], d["pb"], (d["mask"] if d["mask"].size else None)
sizes, rpb = [int(x) for x in sys.argv[2].split(",")], float(sys.argv[3])
runs = int(sys.argv[4]) if len(sys.argv) > 4 else 200
ctx = hmt.Context(0)
d_lab = torch.from_numpy(labels.view(np.int32)).cuda(); d_pb = torch.from_numpy(pb).cuda()
d_mask = torch.from_numpy(mask.view(np.int32)).cuda() if mask is not None else None
rag = O.Rag(labels, mask=mask)
lab, npts, nb = rag.regions(); rs = rag.region_stats(pb); a, b, n = rag.pairs(); ps = rag.pair_stats(pb)
ro, rso = rag.pre_merge(pb, sizes, rpb)
print("shape", labels.shape, "regions", len(lab), "pairs", len(a), "oracle merges", len(ro))
bad = 0; sums = set(); cnt_bad = 0; first = None
for it in range(runs):
    rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, mask=d_mask, only_contour=False)
    reg, par = rm.regions(), rm.pairs()
    if not ((reg["count"] == npts).all() and (par["count"] == n).all() and (reg["min"] == rs[2]).all() and (reg["max"] == rs[3]).all()
            and (par["min"] == ps[2]).all() and (par["max"] == ps[3]).all() and (reg["hist"].sum(1) == npts).all()):
        cnt_bad += 1
    sums.add(hash(reg["sum"].tobytes() + par["sum"].tobytes() + reg["sumsq"].tobytes() + par["sumsq"].tobytes()))
    o, s = rm.pre_merge(sizes, rpb); rm.close()
    if o.shape != ro.shape or not (o == ro).all():
        bad += 1
        if first is None:
            k = 0
            while k < min(len(o), len(ro)) and (o[k] == ro[k]).all(): k += 1
            first = (it, k, o[k].tolist() if k < len(o) else None, ro[k].tolist() if k < len(ro) else None,
                     float(np.abs(reg["sum"] - rs[0]).max()), float(np.abs(par["sum"] - ps[0]).max()))
print("runs", runs, "pre_merge differs in", bad, "| integer statistics differ in", cnt_bad, "| distinct sum bit patterns", len(sums), "| first:", first)
