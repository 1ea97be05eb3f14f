"""CPU, world_size 2, gloo: the N>1 plumbing of the slab exchange (glia_amd/slab.py) -- geometry, the unpadded point-to-point
exchange, and the whole cut-record route (keyed owner exchange -> reduction -> loop owner) on record-shaped tensors with a
numpy restatement of the keyed reducer standing in for glia_hmt_rag_merge."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from glia_amd import slab

RW, PW = 32, 32      # record words, as glia_hmt_rag_device_arrays reports them


def test_slab_geometry_covers_volume_once():
    for nz, world in [(1024, 8), (100, 3), (7, 4), (5, 5), (33, 2)]:
        planes = []
        for r in range(world):
            z0, z1 = slab.slab_bounds(nz, world, r)
            planes += list(range(z0, z1))
            lo, hi, zb, ze = slab.slab_with_halo(nz, world, r)
            assert lo == max(z0 - 1, 0) and hi == min(z1 + 1, nz) and lo + zb == z0 and lo + ze == z1
        assert planes == list(range(nz))


def test_owner_is_a_function_of_the_label_only():
    lab = torch.tensor([0, 1, 2, 1, 7, 2 ** 31 - 1, -5], dtype=torch.int32)      # (uint32 payloads travel as int32)
    for world in (1, 2, 3, 8):
        o = slab.owner_of(lab, world)
        assert int(o.min()) >= 0 and int(o.max()) < world
        assert o[1] == o[3]
    big = torch.arange(100000, dtype=torch.int32)
    counts = torch.bincount(slab.owner_of(big, 8), minlength=8)
    assert counts.min() > 100000 / 8 * 0.9                     # balanced


def _reduce_np(parts):
    """keyed reduction of additive int records: regions by label, pairs by (a, b); ascending keys like glia_hmt_rag_merge"""
    if not parts:
        return dict(rlabel=torch.zeros(0, dtype=torch.int32), rrec=torch.zeros((0, RW), dtype=torch.int32),
                    pa=torch.zeros(0, dtype=torch.int32), pb=torch.zeros(0, dtype=torch.int32), prec=torch.zeros((0, PW), dtype=torch.int32))
    lab = np.concatenate([p["rlabel"].numpy() for p in parts]); rec = np.concatenate([p["rrec"].numpy() for p in parts]).astype(np.int64)
    ul, inv = np.unique(lab, return_inverse=True)
    rsum = np.zeros((len(ul), RW), np.int64); np.add.at(rsum, inv, rec)
    a = np.concatenate([p["pa"].numpy() for p in parts]).astype(np.int64); b = np.concatenate([p["pb"].numpy() for p in parts]).astype(np.int64)
    prec = np.concatenate([p["prec"].numpy() for p in parts]).astype(np.int64)
    key = (a << 32) | b
    uk, inv = np.unique(key, return_inverse=True)
    psum = np.zeros((len(uk), PW), np.int64); np.add.at(psum, inv, prec)
    return dict(rlabel=torch.from_numpy(ul.astype(np.int32)), rrec=torch.from_numpy(rsum.astype(np.int32)),
                pa=torch.from_numpy((uk >> 32).astype(np.int32)), pb=torch.from_numpy((uk & 0xFFFFFFFF).astype(np.int32)),
                prec=torch.from_numpy(psum.astype(np.int32)))


def _records(rank):
    """a rank's partial records: labels 0..59 live on rank 0, 40..99 on rank 1 (40..59 on both = the cut); pairs likewise"""
    rng = np.random.default_rng(100 + rank)
    lab = np.arange(0, 60) if rank == 0 else np.arange(40, 100)
    a = rng.integers(lab.min(), lab.max() + 1, 300); b = rng.integers(lab.min(), lab.max() + 1, 300)
    keep = a != b
    ab = np.unique(np.stack([a[keep], b[keep]], 1), axis=0)
    t = dict(rlabel=torch.from_numpy(lab.astype(np.int32)), rrec=torch.from_numpy(rng.integers(0, 1000, (len(lab), RW)).astype(np.int32)),
             pa=torch.from_numpy(ab[:, 0].astype(np.int32)), pb=torch.from_numpy(ab[:, 1].astype(np.int32)),
             prec=torch.from_numpy(rng.integers(0, 1000, (len(ab), PW)).astype(np.int32)))
    cut = (lab >= 40) & (lab < 60)                        # the labels on the planes next to the cut
    # one shared label is deliberately NOT flagged on rank 1: the loop owner's final reduction must still combine it
    if rank == 1:
        cut[lab == 45] = False
    pcut = ((ab[:, 0] >= 40) & (ab[:, 0] < 60)) | ((ab[:, 1] >= 40) & (ab[:, 1] < 60))
    return t, torch.from_numpy(cut), torch.from_numpy(pcut)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # ragged point-to-point exchange, with an empty message in one direction
    send = [torch.arange(3 + 5 * rank + d, dtype=torch.int32) + 100 * rank for d in range(world)]
    if rank == 0:
        send[1] = send[1][:0]
    recv, sent = slab.exchange_variable(send)
    t, rcut, pcut = _records(rank)
    whole, stats = slab.exchange_cut_records(t, rcut, pcut, _reduce_np, loop_owner=0)
    q.put((rank, [r.numpy().copy() for r in recv], sent, None if whole is None else {k: v.numpy().copy() for k, v in whole.items()}, stats))
    dist.destroy_process_group()


def test_cut_record_exchange_two_ranks_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # exchange_variable: rank 0 got its own message and rank 1's message for it, unpadded; nothing was sent for the empty one
    r0, r1 = res[0][1], res[1][1]
    assert (r0[0] == np.arange(3)).all() and (r0[1] == np.arange(8) + 100).all()
    assert len(r1[0]) == 0 and (r1[1] == np.arange(9) + 100).all()
    assert res[0][2] == 0 and res[1][2] == 8 * 4
    # the whole route: the loop owner holds exactly the keyed reduction of both ranks' records; the other rank holds nothing
    assert res[1][3] is None
    whole = res[0][3]
    direct = _reduce_np([_records(0)[0], _records(1)[0]])
    for k in slab.KEYS:
        assert (whole[k] == direct[k].numpy()).all(), k
    assert len(whole["rlabel"]) == 100
    # only cut records took the owner exchange, and every record went to the loop owner once
    for rank in (0, 1):
        st = res[rank][4]
        t, rcut, pcut = _records(rank)
        assert st["cut_records"] == int(rcut.sum() + pcut.sum()) and st["records"] == t["rlabel"].numel() + t["pa"].numel()
        assert st["bytes_sent_cut_exchange"] < 4 * (st["cut_records"] * (RW + 2) + 4)
    assert res[0][4]["bytes_sent_to_loop_owner"] == 0 and res[1][4]["bytes_sent_to_loop_owner"] > 0
