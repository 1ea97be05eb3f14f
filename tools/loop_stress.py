"""Repeats every merge loop on one dumped volume until one of them fails or differs from the oracle.  usage: loop_stress.py file.npz seconds"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from glia_amd import hmt
from oracle import pyoracle as O
d = np.load(sys.argv[1]); budget = float(sys.argv[2])
labels, pb = d["labels"], d["pb"]
mask = d["mask"] if d["mask"].size else None
ctx = hmt.Context(0)
d_lab = torch.from_numpy(labels.view(np.int32)).cuda(); d_pb = torch.from_numpy(pb).cuda()
d_mask = torch.from_numpy(mask.view(np.int32)).cuda() if mask is not None else None
ref = {}
for typ in (1, 2):
    ref[typ] = O.Rag(labels, mask=mask, only_contour=True).merge_order_pb(pb, type=typ)
ref[3] = O.Rag(labels, mask=mask).merge_order_pb(pb, type=3, update_region=True)
rng = np.random.default_rng(1)
S = 3; dim = labels.ndim
t_end = time.time() + budget
n = 0; fails = {}; t_last = time.time()
while time.time() < t_end:
    for what in ("pb1", "pb2", "pb3", "pre"):
        try:
            if what in ("pb1", "pb2"):
                typ = int(what[2]); rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, mask=d_mask, only_contour=True)
                o, s = rm.merge_order_pb(type=typ); rm.close()
                ok = o.shape == ref[typ][0].shape and (o == ref[typ][0]).all()
            elif what == "pb3":
                rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, mask=d_mask, only_contour=False)
                o, s = rm.merge_order_pb(type=3); rm.close()
                ok = o.shape == ref[3][0].shape and (o == ref[3][0]).all()
            else:
                sizes = sorted(int(x) for x in rng.integers(2, 4 * S ** dim, size=int(rng.integers(1, 3)))); rpb = float(rng.uniform(0.1, 0.5))
                rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, mask=d_mask, only_contour=False)
                o, s = rm.pre_merge(sizes, rpb); rm.close()
                ro, _ = O.Rag(labels, mask=mask).pre_merge(pb, sizes, rpb)
                ok = o.shape == ro.shape and (o == ro).all()
            if not ok:
                fails[what] = fails.get(what, 0) + 1
                want = ref[int(what[2])][0] if what != "pre" else ro
                k = 0
                while k < min(len(o), len(want)) and (o[k] == want[k]).all(): k += 1
                print("DIFFERS", what, n, "merges", len(o), "wanted", len(want), "first difference at", k, o[k].tolist() if k < len(o) else None,
                      want[k].tolist() if k < len(want) else None, (sizes, rpb) if what == "pre" else "", flush=True)
                np.savez_compressed(os.path.join(ROOT, "gpurun_out", "loop_stress_fail.npz"), got=o, want=want)
                os._exit(0)          # (no further GPU work in a process whose last kernel misbehaved)
        except hmt.HmtError as e:
            fails[what + " error"] = fails.get(what + " error", 0) + 1
            print("ERROR", what, n, repr(e), flush=True)
            os._exit(0)
        n += 1
    if time.time() - t_last > 30: t_last = time.time(); print("%d calls ok" % n, flush=True)
print("loop stress: %d calls, failures %s" % (n, fails))
