"""Times the pb-linkage merge tree (type 1 = median, 2 = mean). usage: pb_bench.py [size] [S] [type] [variant: 0 Q8 | 1 f32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from glia_amd import hmt

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16
typ = int(sys.argv[3]) if len(sys.argv) > 3 else 1
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ctx = hmt.Context(0)
labels, pb = ctx.synth((size,) * 3, S, 8 * S, variant=variant)
for rep in range(2):
    rm = hmt.RegionMap(ctx, labels, pb=pb, only_contour=True)
    t1 = time.time()
    order, sal = rm.merge_order_pb(type=typ)
    t2 = time.time()
    tm = rm.last_merge_timing()
    print("size=%d S=%d type=%d variant=%d R=%d P=%d merges=%d  total %.1f ms (table %.1f loop %.1f) edges %d -> %.0f merges/s; sal[0..3]=%s" % (
        size, S, typ, variant, rm.num_regions, rm.num_pairs, len(order), (t2 - t1) * 1e3, tm["ms_table"], tm["ms_loop"], tm["n_edges_scored"],
        len(order) / (tm["ms_loop"] * 1e-3), sal[:3]), flush=True)
    if os.environ.get("GLIA_PB_HASH"):     # kernel experiments: the whole result must not change
        import hashlib
        print("sha1 order %s sal %s" % (hashlib.sha1(np.ascontiguousarray(order).tobytes()).hexdigest(), hashlib.sha1(np.ascontiguousarray(sal).tobytes()).hexdigest()), flush=True)
    rm.close()
