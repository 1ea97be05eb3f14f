"""Host-clock breakdown of one bench step (bench.py's step(), call by call). usage: step_breakdown.py [size] [S]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glia_amd import hmt
from glia_amd.synth_forest import synthetic_forest, write_model

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ctx = hmt.Context(0)
labels, pb = ctx.synth((size,) * 3, S, 8 * S, seed=0x9E3779B97F4A7C15)
cfg = hmt.make_config(pb, rb=[(pb, 8, 0.0, 1.0)], thresholds=(0.2, 0.5, 0.8))
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "forest.bin")
    write_model(path, synthetic_forest(ntree=255, dim=3))
    clf = hmt.RandomForest(ctx, path, predict_label=-1)


def clock():
    torch.cuda.synchronize(); ctx.sync()
    return time.time()


for rep in range(3):
    t0 = clock()
    rm = hmt.RegionMap(ctx, labels, pb=pb, only_contour=False, cfg=cfg)
    t1 = clock()
    n_edges, ms_score = rm.score_initial_edges(clf)
    t2 = clock()
    order, sal = rm.merge_order_pb(type=2)
    t3 = clock()
    tm = rm.last_merge_timing()
    rm.close()
    t4 = clock()
    print("step %d: build %.1f ms (kernel %.2f)  score %.1f ms (kernels %.2f)  merge_order_pb %.1f ms (table %.1f loop %.1f)  close %.1f ms  total %.1f ms" % (
        rep, (t1 - t0) * 1e3, rm.last_pass()[0] if False else 0.0, (t2 - t1) * 1e3, ms_score, (t3 - t2) * 1e3, tm["ms_table"], tm["ms_loop"], (t4 - t3) * 1e3, (t4 - t0) * 1e3), flush=True)
