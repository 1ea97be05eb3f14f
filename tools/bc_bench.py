"""Times the classifier-linkage merge tree. usage: bc_bench.py [size] [S] [ntree]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import torch
from glia_amd import hmt
from glia_amd.synth_forest import synthetic_forest, write_model

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ntree = int(sys.argv[3]) if len(sys.argv) > 3 else 255
ctx = hmt.Context(0)
labels, pb = ctx.synth((size,) * 3, S, 8 * S)
cfg = hmt.make_config(pb, rb=[(pb, 8, 0.0, 1.0)])
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "m.bin")
    write_model(path, synthetic_forest(ntree=ntree, dim=3))
    clf = hmt.RandomForest(ctx, path)
for rep in range(2):
    t0 = time.time()
    rm = hmt.RegionMap(ctx, labels, pb=pb, cfg=cfg)
    t1 = time.time()
    order, sal = rm.merge_order_bc(clf)
    t2 = time.time()
    tm = rm.last_merge_timing()
    print("size=%d S=%d R=%d P=%d merges=%d  rag %.1f ms  bc total %.1f ms  (table %.1f init %.1f loop %.1f)  edges scored %d  -> %.0f merges/s, %.0f edge-features/s; sal[0..3]=%s" % (
        size, S, rm.num_regions, rm.num_pairs, len(order), (t1 - t0) * 1e3, (t2 - t1) * 1e3, tm["ms_table"], tm["ms_init"], tm["ms_loop"],
        tm["n_edges_scored"], len(order) / (tm["ms_loop"] * 1e-3), tm["n_edges_scored"] / ((tm["ms_init"] + tm["ms_loop"]) * 1e-3), sal[:3]))
    if os.environ.get("GLIA_BC_HASH"):     # kernel experiments: the whole result must not change
        import hashlib
        print("sha1 order %s sal %s" % (hashlib.sha1(np.ascontiguousarray(order).tobytes()).hexdigest(), hashlib.sha1(np.ascontiguousarray(sal).tobytes()).hexdigest()), flush=True)
    rm.close()
