"""CPU, world_size 2, gloo: the N>1 plumbing of the slab exchange (geometry + variable-size all-gather)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from glia_amd import slab


def test_slab_geometry_covers_volume_once():
    for nz, world in [(1024, 8), (100, 3), (7, 4), (5, 5), (33, 2)]:
        planes = []
        for r in range(world):
            z0, z1 = slab.slab_bounds(nz, world, r)
            planes += list(range(z0, z1))
            lo, hi, zb, ze = slab.slab_with_halo(nz, world, r)
            assert lo == max(z0 - 1, 0) and hi == min(z1 + 1, nz) and lo + zb == z0 and lo + ze == z1
        assert planes == list(range(nz))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)
    n = 5 + 7 * rank                       # ragged sizes, one rank larger than the other
    keys = torch.from_numpy(rng.integers(0, 50, n).astype(np.int32))
    recs = torch.from_numpy(rng.integers(0, 1000, (n, 4)).astype(np.int32))
    gk = slab.all_gather_variable(keys)
    gr = slab.all_gather_variable(recs)
    empty = slab.all_gather_variable(torch.zeros((0, 3), dtype=torch.int32) if rank == 0 else torch.ones((2, 3), dtype=torch.int32))
    q.put((rank, [k.numpy().copy() for k in gk], [r.numpy().copy() for r in gr], [e.shape[0] for e in empty],
           keys.numpy().copy(), recs.numpy().copy()))
    dist.destroy_process_group()


def test_all_gather_variable_two_ranks_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    own_keys = [res[0][4], res[1][4]]
    own_recs = [res[0][5], res[1][5]]
    for rank, gk, gr, esz, _, _ in res:
        assert [len(k) for k in gk] == [5, 12]
        for r in range(2):
            assert (gk[r] == own_keys[r]).all() and (gr[r] == own_recs[r]).all()
        assert esz == [0, 2]
    # the keyed reduction the merge kernel performs, restated in numpy on the gathered parts
    allk = np.concatenate(res[0][1]); allr = np.concatenate(res[0][2]).astype(np.int64)
    uk = np.unique(allk)
    summed = np.stack([allr[allk == k].sum(0) for k in uk])
    assert summed.sum() == allr.sum() and len(uk) <= len(allk)
