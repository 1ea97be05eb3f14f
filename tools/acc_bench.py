"""Times the accumulation pass alone (HIP events inside the library). usage: acc_bench.py [size] [S] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glia_amd import hmt

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = hmt.Context(0)
labels, pb = ctx.synth((size,) * 3, S, 8 * S)
cfg = hmt.make_config(pb, rb=[(pb, 8, 0.0, 1.0)])
ts, walls = [], []
for i in range(reps + 1):
    t0 = time.time()
    rm = hmt.RegionMap(ctx, labels, pb=pb, only_contour=True, cfg=cfg)
    walls.append((time.time() - t0) * 1e3)
    ms, by = rm.last_pass()
    R, P = rm.num_regions, rm.num_pairs
    rm.close()
    if i:
        ts.append(ms)
ms = min(ts)
print("debug=%s size=%d S=%d R=%d P=%d  acc %.3f ms  %.1f GB/s (%.1f%% of 8 TB/s)  whole build (pass + compaction, host clock) %.1f ms" % (
    os.environ.get("GLIA_HMT_DEBUG", "0"), size, S, R, P, ms, by / ms / 1e6, by / ms / 1e6 / 80, min(walls[1:])))
