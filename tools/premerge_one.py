"""ONE pre_merge call on a dumped volume, fresh process, against the oracle.  usage: premerge_one.py file.npz sizes rpb"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from glia_amd import hmt
from oracle import pyoracle as O
d = np.load(sys.argv[1]); labels, pb = d["labels"], d["pb"]
sizes = [int(x) for x in sys.argv[2].split(",")]; rpb = float(sys.argv[3])
ro, rs = O.Rag(labels).pre_merge(pb, sizes, rpb)
ctx = hmt.Context(0)
d_lab = torch.from_numpy(labels.view(np.int32)).cuda(); d_pb = torch.from_numpy(pb).cuda()
rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, only_contour=False)
print("regions", rm.num_regions, "pairs", rm.num_pairs, "oracle merges", len(ro), flush=True)
try:
    o, s = rm.pre_merge(sizes, rpb)
    k = 0
    while k < min(len(o), len(ro)) and (o[k] == ro[k]).all() and s[k] == rs[k]: k += 1
    print("gpu merges", len(o), "first difference at", k, "gpu", o[k].tolist() if k < len(o) else None, s[k] if k < len(s) else None,
          "oracle", ro[k].tolist() if k < len(ro) else None, rs[k] if k < len(rs) else None, flush=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "premerge_one.npz"), got=o, gots=s, want=ro, wants=rs)
except hmt.HmtError as e:
    print("ERROR", repr(e), flush=True)
