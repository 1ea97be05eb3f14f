import sys, time, torch
sys.path.insert(0, '/root/repo')
from glia_amd import hmt
ctx = hmt.Context(0)
shape = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (512,) * 3
labels, pb = ctx.synth(shape, 16, 128)
cfg = hmt.make_config(pb, rb=[(pb, 8, 0.0, 1.0)])
torch.cuda.synchronize()
for i in range(3):
    t0 = time.time(); rm = hmt.RegionMap(ctx, labels, pb=pb, cfg=cfg); dt = time.time() - t0
    print("build", i, "%.1f ms" % (dt * 1e3), "kernel %.2f ms" % rm.last_pass()[0]); rm.close()
