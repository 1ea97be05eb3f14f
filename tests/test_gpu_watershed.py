"""-m gpu: watershed over-segmentation (gadget/main_watershed.cxx, util/image_alg.hxx:9-21), the step before the RAG.
ITK (MorphologicalWatershedImageFilter) is not in this image: parity with it is UNPINNED.  The device code and the oracle's
sequential restatement (worklist reconstruction, breadth-first plateaus, Dijkstra flooding) implement the same order-free tie
rules by different algorithms and must give identical label volumes."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "GPU test run without a GPU"
    from glia_amd import hmt
    c = hmt.Context(0)
    yield c
    c.close()


def _smooth(shape, seed, passes=2):
    rng = np.random.default_rng(seed)
    img = rng.random(shape).astype(np.float64)
    for _ in range(passes):
        for ax in range(len(shape)):
            img = (img + np.roll(img, 1, ax) + np.roll(img, -1, ax)) / 3.0
    img -= img.min(); img /= img.max()
    return img.astype(np.float32)


@pytest.mark.parametrize("shape,level,quant", [((40, 44, 36), 0.02, None), ((40, 44, 36), 0.0, None), ((33, 29, 31), 0.05, 32),
                                               ((96, 80), 0.03, None), ((64, 64), 0.0, 8), ((17, 1, 23), 0.01, 16)])
def test_watershed_matches_oracle(ctx, shape, level, quant):
    """smooth noise (many shallow minima the h-minima transform removes) and coarsely quantised images (large plateaus: markers that
    are plateaus, flooding ties decided by plateau distance and label)"""
    import torch
    from oracle import pyoracle as O
    img = _smooth(shape, seed=sum(shape))
    if quant:
        img = (np.round(img * quant) / quant).astype(np.float32)
    ref, n_ref = O.watershed(img, level)
    lab, n, sweeps = ctx.watershed(torch.from_numpy(img).cuda(), level)
    got = lab.cpu().numpy().view(np.uint32)
    assert n == n_ref and (got == ref).all()
    assert got.min() == 1 and got.max() == n and len(np.unique(got)) == n            # every voxel labelled, labels 1..n all used
    if level > 0:
        assert n <= O.watershed(img, 0.0)[1] and (quant or n < O.watershed(img, 0.0)[1])   # the level removes shallow minima
    assert sweeps > 0


def test_watershed_feeds_the_merge_path(ctx):
    """pb image -> watershed supervoxels -> RAG -> pb-mean merge tree: the chain of the reference's pipeline (README), on the device,
    equal to the same chain through the oracle"""
    import torch
    from glia_amd import hmt
    from oracle import pyoracle as O
    _, pb = O.synth((48, 48, 48), 8, 16)
    d_pb = torch.from_numpy(pb).cuda()
    lab, n, _ = ctx.watershed(d_pb, 0.1)
    ref, n_ref = O.watershed(pb, 0.1)
    assert n == n_ref and (lab.cpu().numpy().view(np.uint32) == ref).all() and n > 20
    rm = hmt.RegionMap(ctx, lab, pb=d_pb, only_contour=True)
    o, s = rm.merge_order_pb(type=2)
    ro, rs = O.Rag(ref, only_contour=True).merge_order_pb(pb, type=2)
    assert (o == ro).all() and (s == rs).all()


def test_watershed_at_128_cubed_matches_oracle(ctx):
    """the largest size the oracle's sequential flood finishes in seconds: 2.1 M voxels of smooth noise, thousands of basins"""
    import torch
    from oracle import pyoracle as O
    img = _smooth((128, 128, 128), seed=7, passes=3)
    ref, n_ref = O.watershed(img, 0.01)
    lab, n, _ = ctx.watershed(torch.from_numpy(img).cuda(), 0.01)
    got = lab.cpu().numpy().view(np.uint32)
    assert n == n_ref and n > 500 and (got == ref).all()


def test_config2_chain_at_256_cubed(ctx):
    """BASELINE config 2 at its own size, label bit-exact vs the CPU: 256^3 pb volume -> watershed supervoxels -> RAG -> full
    pb-mean merge tree on the device = the same chain through the oracle (its sequential flood takes ~15 s here); the classifier
    linkage (255-tree forest), which the oracle cannot finish at this size, is held to the invariant of
    util/struct_merge.hxx:19-31 (glia_hmt_check_merge_order)."""
    import torch
    from glia_amd import hmt
    from oracle import pyoracle as O
    from test_gpu_headline import _forest
    size, S = 256, 16
    _, pb = O.synth((size,) * 3, S, 8 * S)
    img = pb.astype(np.float64)                       # two box passes per axis: basins larger than single voxels
    for _ in range(2):
        for ax in range(3):
            img = (img + np.roll(img, 1, ax) + np.roll(img, -1, ax)) / 3.0
    img = img.astype(np.float32)
    ref, n_ref = O.watershed(img, 0.02)
    d_pb = torch.from_numpy(pb).cuda()
    lab, n, _ = ctx.watershed(torch.from_numpy(img).cuda(), 0.02)
    got = lab.cpu().numpy().view(np.uint32)
    assert n == n_ref and n > 1000 and (got == ref).all()
    sizes = np.bincount(got.reshape(-1), minlength=n + 1)
    assert sizes[0] == 0 and (sizes[1:] > 0).all()                                   # every voxel labelled, labels 1..n all used
    rm = hmt.RegionMap(ctx, lab, pb=d_pb, cfg=hmt.make_config(d_pb, rb=[(d_pb, 8, 0.0, 1.0)]))
    assert rm.num_regions == n
    o, s = rm.merge_order_pb(type=2)
    ro, rs = O.Rag(ref, only_contour=True).merge_order_pb(pb, type=2)
    assert o.shape == ro.shape and (o == ro).all() and (s == rs).all()
    ob, sb = rm.merge_order_bc(_forest(ctx))
    rm.close()
    d = (ob.astype(np.int64) - 1).astype(np.uint32)              # labels 1..n -> dense 0..n-1, merged key n + 1 + k -> n + k
    assert len(ob) == len(o) and hmt.check_merge_order(d, n) == -1
    assert ((sb >= 0) & (sb <= 1)).all()


def test_watershed_cli(tmp_path):
    from oracle import pyoracle as O
    from test_gpu_cli import write_mha, read_mha
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cli")
    subprocess.check_call(["make", "-C", tools, "watershed"], stdout=subprocess.DEVNULL)
    img = _smooth((30, 34, 38), seed=5)
    src, dst = str(tmp_path / "pb.mha"), str(tmp_path / "seg.mha")
    write_mha(src, img)
    subprocess.check_call([os.path.join(tools, "watershed"), "-i", src, "-l", "0.03", "-o", dst])
    ref, _ = O.watershed(img, 0.03)
    assert (read_mha(dst) == ref).all()
    # -r 1: relabelled by decreasing size (relabelImage), -u 1: 16-bit output
    subprocess.check_call([os.path.join(tools, "watershed"), "-i", src, "-l", "0.03", "-r", "1", "-u", "1", "-o", dst])
    got = read_mha(dst)
    exp, _ = O.relabel_image(ref, 0)
    assert got.dtype == np.uint16 and (got == exp).all()
    r = subprocess.run([os.path.join(tools, "watershed"), "-i", src, "-o", dst], capture_output=True)
    assert r.returncode == 1 and b"required" in r.stderr
