import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from glia_amd import hmt
ctx = hmt.Context(0)
for size, S in ((128, 8), (256, 16)):
    l2, p2 = ctx.synth((size,) * 3, S, 8 * S)
    c2 = hmt.make_config(p2, rb=[(p2, 8, 0.0, 1.0)])
    rm0 = hmt.RegionMap(ctx, l2, pb=p2, only_contour=True); o2, _ = rm0.merge_order_pb(type=2); rm0.close()
    rm2 = hmt.RegionMap(ctx, l2, pb=p2, only_contour=False, cfg=c2)
    rm2.bc_feat(o2)
    t = time.time(); f2 = rm2.bc_feat(o2); t = time.time() - t
    import hashlib
    print("bc_feat %d^3 S=%d: %d rows x %d in %.4f s = %.0f rows/s  sha1 %s" % (size, S, f2.shape[0], f2.shape[1], t, f2.shape[0] / t, hashlib.sha1(np.ascontiguousarray(f2).tobytes()).hexdigest()), flush=True)
    rm2.close()
