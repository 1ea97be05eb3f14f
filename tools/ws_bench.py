"""Times glia_hmt_watershed on smoothed noise. usage: ws_bench.py [sizes ...]"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from glia_amd import hmt
ctx = hmt.Context(0)
for size in [int(a) for a in sys.argv[1:]] or [256, 512]:
    g = torch.Generator(device='cuda'); g.manual_seed(5)
    img = torch.rand((size,)*3, device='cuda', generator=g)
    for _ in range(2):
        for ax in range(3):
            img = (img + torch.roll(img, 1, ax) + torch.roll(img, -1, ax)) / 3.0
    img = ((img - img.min()) / (img.max() - img.min())).float().contiguous()
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        lab, n, sw = ctx.watershed(img, 0.02)
        torch.cuda.synchronize(); dt = time.time() - t0
    print("watershed %d^3 level 0.02: %d labels, %d launches of the fixed points, %.1f ms" % (size, n, sw, dt * 1e3))
