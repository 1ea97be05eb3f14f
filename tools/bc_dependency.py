"""How sequential is the classifier-linkage merge order?  Runs the loop and reports, from order + saliencies alone,
how often a popped edge was created by the contraction just before it (a chain the loop cannot break) and how long
the runs of pops are whose edges all existed before the run started (candidates for batched contractions).
usage: bc_dependency.py [size] [S] [ntree]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
rm = hmt.RegionMap(ctx, labels, pb=pb, cfg=cfg)
order, sal = rm.merge_order_bc(clf)
order = np.asarray(order).reshape(-1, 3).astype(np.int64)
n = len(order)
first_new = order[0, 2]
# birth step of a region: leaves -1, region first_new + j was made by merge j
birth = lambda r: np.where(r >= first_new, r - first_new, -1)
made = np.maximum(birth(order[:, 0]), birth(order[:, 1]))      # the merge that created the popped edge (-1: initial edge)
k = np.arange(n)
print("size %d S %d: %d merges; distinct saliencies %d" % (size, S, n, len(np.unique(sal))))
print("popped edge created by the previous merge: %.1f %%; by one of the previous 4: %.1f %%; initial edge: %.1f %%" % (
    100.0 * np.mean(made == k - 1), 100.0 * np.mean(made >= k - 4), 100.0 * np.mean(made < 0)))
# greedy batching bound: start a batch at k; extend while the popped edge existed before the batch started and its regions are
# not regions of the batch (adjacency is not visible from the order, so this is an upper bound on batch sizes)
sizes = []
i = 0
while i < n:
    j = i + 1
    used = {order[i, 0], order[i, 1]}
    while j < n and made[j] < i and order[j, 0] not in used and order[j, 1] not in used and j - i < 64:
        used.add(order[j, 0]); used.add(order[j, 1]); j += 1
    sizes.append(j - i); i = j
sizes = np.array(sizes)
print("batches (upper bound, <= 64): %d for %d merges, mean %.2f; by tenth of the run:" % (len(sizes), n, n / len(sizes)))
pos = np.cumsum(sizes) - sizes
for t in range(10):
    m = (pos >= t * n / 10) & (pos < (t + 1) * n / 10)
    if m.any(): print("  %d0%%: mean batch %.2f  max %d" % (t, sizes[m].mean(), sizes[m].max()))
tm = rm.last_merge_timing()
print("loop %.1f ms, edges scored %d (%.1f per merge)" % (tm["ms_loop"], tm["n_edges_scored"], tm["n_edges_scored"] / max(n, 1)))
if len(sys.argv) > 4:
    np.savez_compressed(sys.argv[4], order=order, sal=sal)
