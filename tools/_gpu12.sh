cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02u
timeout -k 10 600 python -m pytest tests/test_gpu_watershed.py -x -q -m gpu > gpurun_out/r02u/pytest.log 2>&1; tail -15 gpurun_out/r02u/pytest.log
timeout -k 10 300 python - > gpurun_out/r02u/ws_time.txt 2>&1 <<'P'
import sys, time, torch
sys.path.insert(0, '.')
from glia_amd import hmt
ctx = hmt.Context(0)
for size in (256, 512):
    _, pb = ctx.synth((size,)*3, 16, 128)
    for rep in range(2):
        torch.cuda.synchronize(); t = time.time()
        lab, n, sw = ctx.watershed(pb, 0.1)
        torch.cuda.synchronize(); dt = time.time() - t
    print("watershed %d^3 level 0.1: %d labels, %d sweeps, %.1f ms (%.2f Gvoxel/s)" % (size, n, sw, dt*1e3, size**3/dt/1e9), flush=True)
P
cat gpurun_out/r02u/ws_time.txt | tail -3
