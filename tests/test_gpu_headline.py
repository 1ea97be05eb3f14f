"""Regression gates at BASELINE.json's headline sizes: the whole merge order and saliency arrays of the
1024^3 pb-mean loop and of the 512^3 classifier loop are compared with recorded SHA-1 digests -- the kernels that produced them
were bit-identical to the oracle wherever the oracle finishes (tests/test_gpu_merge.py, test_gpu_bc.py, the fuzz runs), and every
queue / loop variant since has to reproduce them byte for byte -- plus the size-independent invariants of
util/struct_merge.hxx:19-31.  The synthetic volume (glia_hmt_synth) and the synthetic forest (seed 1234) are deterministic."""
import hashlib
import os
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# recorded with tools/pb_bench.py 1024 16 2 (profiles/r02q_pb_hash.txt) and tools/bc_bench.py 512 16 (profiles/r03_bc_hash.txt)
PB_1024 = ("977022085e1a37a4d143841a51ea8834b6f53c59", "7923436b47d4484f1e95962ac87ff409d4f2c617")
PB_512 = ("652c84e7efe781bacc9df8c0a16675ca7e34cf17", "d78663710b1699d331a62c40471ba2d90ffc6515")
BC_512 = ("8e620b69eab2cc31991eaca662446f52ad2231df", "8e30e7ba92d7b89dc09c413260a0841df1cbf886")
BC_256 = ("51b7b5316e0d8fd648ab2b444527633d8eaf65c5", "5eadbd683d6a042f7bf950aa2352ef93bdc12956")


@pytest.fixture(scope="module")
def ctx():
    from glia_amd import hmt
    return hmt.Context(0)


def _sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def _invariants(order, R, first_new):
    n = len(order)
    assert n == R - 1                                                        # connected mutual-edge graph
    o = order.astype(np.int64)
    assert (o[:, 0] < o[:, 1]).all() and (o[:, 1] < o[:, 2]).all()
    assert (o[:, 2] == first_new + np.arange(n)).all()                       # x2 = maxKey + 1 + i
    assert len(np.unique(np.concatenate([o[:, 0], o[:, 1]]))) == 2 * n       # every region is merged exactly once


@pytest.mark.parametrize("size,expect", [(512, PB_512), (1024, PB_1024)])
def test_pb_mean_order_at_headline_size(ctx, size, expect):
    """BASELINE configs 3/4: 1024^3, S = 16, Q8 pb -- 262 143 merges, the loop bench.py times"""
    from glia_amd import hmt
    labels, pb = ctx.synth((size,) * 3, 16, 128)
    rm = hmt.RegionMap(ctx, labels, pb=pb, only_contour=True)
    R = rm.num_regions
    order, sal = rm.merge_order_pb(type=2)
    rm.close()
    _invariants(order, R, R + 1)                                             # labels are 1..R
    assert (np.diff(sal) <= 1e-12).all()                                     # mean linkage is reducible
    assert (_sha(order), _sha(sal)) == expect


@pytest.mark.parametrize("size,expect", [(256, BC_256), (512, BC_512)])
def test_classifier_order_at_headline_size(ctx, size, expect):
    """the north-star linkage (255-tree forest, D_f = 104) at 512^3: invariants + the recorded digest"""
    from glia_amd import hmt
    from glia_amd.synth_forest import synthetic_forest, write_model
    labels, pb = ctx.synth((size,) * 3, 16, 128)
    cfg = hmt.make_config(pb, rb=[(pb, 8, 0.0, 1.0)])
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "m.bin")
        write_model(path, synthetic_forest(ntree=255, dim=3))
        clf = hmt.RandomForest(ctx, path)
    rm = hmt.RegionMap(ctx, labels, pb=pb, cfg=cfg)
    R = rm.num_regions
    order, sal = rm.merge_order_bc(clf)
    rm.close()
    _invariants(order, R, R + 1)
    assert ((sal >= 0.0) & (sal <= 1.0)).all()                               # vote fractions
    assert np.all(np.abs(sal * 255.0 - np.round(sal * 255.0)) < 1e-9)        # k / ntree
    assert (_sha(order), _sha(sal)) == expect
