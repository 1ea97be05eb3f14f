"""Re-creates the input of the one classifier-path mismatch round 1's fuzz run found (profiles/r01i_fuzz_summary.txt:
"seed 31337: MISMATCH after 2159 cases: bc order (stub 78) ... layout 2, bins 8") as a committed fixture.
The fuzz run saved labels + pb (gpurun_out/fuzz_fail.npz) but not the random `raw` image of the feature lists, so this
script replays tests/fuzz_gpu.py's random stream on the CPU (oracle only, same draws in the same order) up to that case,
checks that labels and pb come out identical to the saved ones, and writes tests/golden/fuzz_seed31337_case2159.npz
with the expected order / saliencies / feature rows from the oracle.
usage: python tests/golden/make_fuzz_fixture.py [seed] [case]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import pyoracle as O
import _rf

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 31337
target = int(sys.argv[2]) if len(sys.argv) > 2 else 2159
rng = np.random.default_rng(seed)
saved = np.load(os.path.join(ROOT, "gpurun_out", "fuzz_fail.npz")) if os.path.exists(os.path.join(ROOT, "gpurun_out", "fuzz_fail.npz")) else None
n = 0
while True:
    dim = int(rng.choice([2, 3], p=[0.3, 0.7]))
    shape = tuple(int(rng.integers(6, 70 if dim == 2 else 44)) for _ in range(dim))
    S = int(rng.integers(3, 10)); G = int(rng.integers(2, 4)) * S
    variant = int(rng.integers(0, 2))
    labels, pb = O.synth(shape, S, G, seed=int(rng.integers(1, 1 << 60)), variant=variant)
    if rng.random() < 0.3:
        q = int(rng.choice([2, 4, 8]))
        pb = (np.floor(pb * q) / q).astype(np.float32)
    mask = None
    if rng.random() < 0.35:
        mask = (rng.random(shape) > rng.uniform(0.05, 0.4)).astype(np.uint32)
    if rng.random() < 0.5:
        pass                                   # the median x size run draws nothing
    sizes = sorted(int(x) for x in rng.integers(2, 4 * S ** dim, size=int(rng.integers(1, 3))))
    rpb = float(rng.uniform(0.1, 0.5))
    if len(np.unique(labels)) > 300:
        n += 1
        continue
    raw = (np.round(rng.random(shape) * 255) / 256.0).astype(np.float32)
    lay = int(rng.integers(0, 4))
    bins = int(rng.choice([4, 8, 16]))
    if lay == 0: okw = dict(rb=[(pb, bins, 0.0, 1.0)])
    elif lay == 1: okw = dict(rb=[(raw, bins, 0.0, 1.0), (pb, 8, 0.0, 1.0)])
    elif lay == 2: okw = dict(r=[(raw, bins, 0.0, 1.0)], b=[(pb, 8, 0.0, 1.0)], rl=[(raw, 4, 0.0, 1.0)])
    else: okw = dict(b=[(raw, bins, 0.0, 1.0)])
    flags = dict(use_log=bool(rng.random() < 0.3), use_simple=bool(rng.random() < 0.2))
    ocfg = O.make_cfg(pb, **okw, **flags)
    fd = O.feat_dim(dim, ocfg)
    stub = int(rng.integers(0, fd))
    ro, rs, rf = O.Rag(labels, mask=mask).merge_order_bc(ocfg, None, stub_index=stub, want_feats=True)
    if n == target:
        print("case %d: shape %s S %d G %d variant %d layout %d bins %d stub %d flags %s" % (n, shape, S, G, variant, lay, bins, stub, flags))
        if saved is not None:
            assert (saved["labels"] == labels).all() and (saved["pb"] == pb).all(), "replay diverged from the saved case"
            print("labels and pb identical to gpurun_out/fuzz_fail.npz")
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "fuzz_seed%d_case%d.npz" % (seed, target)), labels=labels, pb=pb,
                            raw=raw, layout=lay, bins=bins, stub=stub, use_log=flags["use_log"], use_simple=flags["use_simple"],
                            order=ro, saliency=rs, feats=rf)
        break
    if len(ro) > 4 and variant == 0:
        forest = _rf.random_forest(rng, int(rng.choice([7, 31, 63])), int(rng.integers(3, 8)), rf)
    n += 1
    if n % 100 == 0:
        print(n, flush=True)
