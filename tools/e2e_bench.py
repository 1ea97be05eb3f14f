"""BASELINE config 5: the whole HMT segmentation on one MI355X -- label volume -> RAG + statistics -> classifier-scored
greedy merge tree -> node potentials -> greedy tree resolution -> final label volume.
usage: e2e_bench.py [nz ny nx] [S] [ntree]      (default 512 2048 2048, S = 16, 255 trees)"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from glia_amd import hmt
from glia_amd.synth_forest import synthetic_forest, write_model

shape = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (512, 2048, 2048)
S = int(sys.argv[4]) if len(sys.argv) > 4 else 16
ntree = int(sys.argv[5]) if len(sys.argv) > 5 else 255
ctx = hmt.Context(0)
t = time.time()
labels, pb = ctx.synth(shape, S, 8 * S)
ctx.sync()
print("synth %s: %.2f s" % (shape, time.time() - t), flush=True)
cfg = hmt.make_config(pb, rb=[(pb, 8, 0.0, 1.0)])
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "m.bin")
    write_model(path, synthetic_forest(ntree=ntree, dim=3))
    clf = hmt.RandomForest(ctx, path)
T = {}
t0 = time.time()
rm = hmt.RegionMap(ctx, labels, pb=pb, cfg=cfg)
T["rag_build_s"] = time.time() - t0
acc_ms, acc_bytes = rm.last_pass()
print("RAG: %d regions, %d directed pairs, accumulate kernel %.2f ms (%.0f GB/s), build %.2f s" % (
    rm.num_regions, rm.num_pairs, acc_ms, acc_bytes / acc_ms / 1e6, T["rag_build_s"]), flush=True)
t0 = time.time()
order, sal = rm.merge_order_bc(clf)
T["merge_tree_s"] = time.time() - t0
tm = rm.last_merge_timing()
print("merge tree: %d merges in %.2f s (table %.1f ms, init scores %.1f ms, loop %.1f ms) -> %.0f merges/s, %d edges scored -> %.0f edge-features/s" % (
    len(order), T["merge_tree_s"], tm["ms_table"], tm["ms_init"], tm["ms_loop"], len(order) / T["merge_tree_s"], tm["n_edges_scored"],
    tm["n_edges_scored"] / T["merge_tree_s"]), flush=True)
rm.close()
t0 = time.time()
lab, par, c0, c1, pot = hmt.tree_potentials(order, sal)
picks = hmt.resolve_tree_greedy(par, c0, c1, pot)
src, dst = hmt.label_transform(lab, c0, c1, picks, 1)
T["resolve_s"] = time.time() - t0
t0 = time.time()
ms = hmt.transform_image(ctx, labels, src, dst, fill_missing=True)
nl = hmt.relabel_image(ctx, labels)
T["relabel_s"] = time.time() - t0
print("tree resolution: %d nodes, %d picks in %.2f s; final label volume: transform kernel %.3f ms (%.0f GB/s), relabel to %d labels, %.2f s" % (
    len(lab), len(picks), T["resolve_s"], ms, 8.0 * labels.numel() / ms / 1e6, nl, T["relabel_s"]), flush=True)
total = sum(T.values())
print("END TO END %.2f s for %d voxels, %d supervoxels -> %d segments  (%s)" % (
    total, labels.numel(), len(order) + 1, nl, ", ".join("%s %.2f" % kv for kv in T.items())), flush=True)
