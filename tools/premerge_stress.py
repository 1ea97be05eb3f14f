"""Stress of RAG build + pre_merge against the oracle on many small volumes (the one check of tests/fuzz_gpu.py that failed once in
~3400 cases).  usage: premerge_stress.py seconds seed"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from glia_amd import hmt
from oracle import pyoracle as O
budget, seed = float(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
ctx = hmt.Context(0)
t_end = time.time() + budget
n = bad = 0
while time.time() < t_end:
    shape = tuple(int(rng.integers(10, 30)) for _ in range(3))
    S = int(rng.integers(3, 6)); G = 3 * S
    labels, pb = O.synth(shape, S, G, variant=1)
    pb = (pb * np.float32(rng.uniform(0.7, 1.0))).astype(np.float32)
    d_lab = torch.from_numpy(labels.view(np.int32)).cuda(); d_pb = torch.from_numpy(pb).cuda()
    for rep in range(4):
        sizes = sorted(int(x) for x in rng.integers(2, 4 * S ** 3, size=int(rng.integers(1, 3))))
        rpb = float(rng.uniform(0.1, 0.5))
        ro, _ = O.Rag(labels).pre_merge(pb, sizes, rpb)      # (a fresh map: the oracle's pre_merge continues the key numbering of an earlier call)
        for k in range(3):
            rm = hmt.RegionMap(ctx, d_lab, pb=d_pb, only_contour=False)
            if k == 1: rm.merge_order_pb(type=3)
            o, _ = rm.pre_merge(sizes, rpb); rm.close()
            n += 1
            if not (o.shape == ro.shape and (o == ro).all()):
                bad += 1
                j = 0
                while j < min(len(o), len(ro)) and (o[j] == ro[j]).all(): j += 1
                print("MISMATCH", shape, S, sizes, repr(rpb), "variant", k, "first difference at", j, "of", len(o), len(ro), flush=True)
                if bad <= 3: np.savez_compressed(os.path.join(ROOT, "gpurun_out", "premerge_fail_%d.npz" % bad), labels=labels, pb=pb, sizes=np.asarray(sizes), rpb=np.asarray([rpb]), got=o, want=ro)
print("pre_merge stress: %d runs, %d mismatches" % (n, bad))
