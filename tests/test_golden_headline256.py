"""tests/golden/headline/headline256_oracle.npz -- the ORACLE's merge orders of the 256^3 headline-shaped volume (tests/golden/gen_headline256.py) --
against the invariants of util/struct_merge.hxx:19-31 and against the classifier digest that tests/test_gpu_headline.py records for that
size: the digest chain of the headline gates (256^3 -> 512^3 -> 1024^3, same kernels, same forest, same generator) starts at an array the
CPU restatement produced.  The oracle's own synth must still produce the fixture's volume."""
import hashlib
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "golden", "headline", "headline256_oracle.npz")
BC_256 = ("51b7b5316e0d8fd648ab2b444527633d8eaf65c5", "5eadbd683d6a042f7bf950aa2352ef93bdc12956")      # = test_gpu_headline.BC_256


def _sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.skipif(not os.path.exists(PATH), reason="fixture not generated (python tests/golden/gen_headline256.py)")
def test_fixture_is_a_merge_order_and_carries_the_recorded_digest():
    from oracle import pyoracle as O
    g = np.load(PATH)
    lab, pb = O.synth((256,) * 3, 16, 128)
    assert _sha(lab) == str(g["labels_sha1"]) and _sha(pb) == str(g["pb_sha1"])
    R = int(lab.max())
    for name in ("pb", "bc"):
        o = g[name + "_order"].astype(np.int64)
        n = len(o)
        assert n == R - 1 and len(g[name + "_sal"]) == n
        assert (o[:, 0] < o[:, 1]).all() and (o[:, 1] < o[:, 2]).all()
        assert (o[:, 2] == R + 1 + np.arange(n)).all()                              # x2 = maxKey + 1 + i
        assert len(np.unique(np.concatenate([o[:, 0], o[:, 1]]))) == 2 * n          # every region is merged exactly once
    assert (np.diff(g["pb_sal"]) <= 1e-12).all()                                     # mean linkage is reducible
    assert (_sha(g["bc_order"]), _sha(g["bc_sal"])) == BC_256
