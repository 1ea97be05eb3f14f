"""Generates tests/golden/headline/headline256_oracle.npz: the ORACLE's pb-mean and classifier merge orders of the 256^3 headline-shaped volume
(S = 16, G = 128, Q8 pb; 4 096 regions; the 255-tree synthetic forest of tests/test_gpu_headline.py).  The oracle re-walks voxels per
candidate edge as the reference does: 49 minutes on one core (2 959 s for the classifier order), which is why this is a fixture and not a test.
usage: python tests/golden/gen_headline256.py   (from the repo root; CPU only)"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import pyoracle as O
from glia_amd.synth_forest import synthetic_forest

sha = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()
lab, pb = O.synth((256,) * 3, 16, 128)
t = time.time()
po, ps = O.Rag(lab, only_contour=True).merge_order_pb(pb, type=2)
print("pb-mean: %d merges in %.1f s, sha1 %s %s" % (len(po), time.time() - t, sha(po), sha(ps)), flush=True)
t = time.time()
cfg = O.make_cfg(pb, rb=[(pb, 8, 0.0, 1.0)])
bo, bs = O.Rag(lab).merge_order_bc(cfg, O.make_forest(synthetic_forest(ntree=255, dim=3), -1))[:2]
print("classifier: %d merges in %.1f s, sha1 %s %s" % (len(bo), time.time() - t, sha(bo), sha(bs)), flush=True)
os.makedirs(os.path.join(ROOT, "tests", "golden", "headline"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "headline", "headline256_oracle.npz"), pb_order=po, pb_sal=ps, bc_order=bo, bc_sal=bs,
                    labels_sha1=np.array(sha(lab)), pb_sha1=np.array(sha(pb)))
print("written", flush=True)
