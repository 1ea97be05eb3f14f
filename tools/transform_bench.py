"""Times transform_image (util/image.hxx:227-242) three times in one process on a size^2 x depth label volume: the first call of a
process pays the kernel's code-object load inside the event bracket.  usage: transform_bench.py [xy=2048] [z=512] [S=16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from glia_amd import hmt

xy = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
z = int(sys.argv[2]) if len(sys.argv) > 2 else 512
S = int(sys.argv[3]) if len(sys.argv) > 3 else 16
ctx = hmt.Context(0)
labels, pb = ctx.synth((z, xy, xy), S, 8 * S)
del pb
R = int(labels.max().item())
rng = np.random.default_rng(1)
src = np.arange(1, R + 1, dtype=np.uint32)
dst = rng.permutation(src).astype(np.uint32)          # a bijection: every call does the same amount of work
nbytes = labels.numel() * 8
for rep in range(3):
    ms = hmt.transform_image(ctx, labels, src, dst, fill_missing=True)
    print("call %d: transform kernel %.3f ms (%.0f GB/s), map of %d labels (dense table %.1f MB)" % (rep, ms, nbytes / ms * 1e-6, R, (R + 1) * 4e-6), flush=True)
