"""oracle/ restatement vs the reference's own engine headers built in place (oracle/_ref/ref_engine).

Covers TBoundaryTable (type/boundary_table.hxx), TRegion/TRegionMap merge + boundaryWith
(type/region.hxx, type/region_map.hxx) and genMergeOrderGreedy (util/struct_merge.hxx:13-33).
The binary is built by `make -C oracle ref` where /root/reference exists and travels prebuilt otherwise.
"""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import pyoracle as O

REF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "ref_engine")


def _ensure_ref():
    if not os.path.exists(REF) and os.path.isdir("/root/reference/code"):
        subprocess.check_call(["make", "-C", os.path.dirname(os.path.dirname(REF)), "ref"], stdout=subprocess.DEVNULL)
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/ref_engine not built and /root/reference absent")


def run_ref(rag, pb, type, upd):
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "dump.txt")
        rag.dump(pb, type, upd, p)
        with open(p) as f:
            out = subprocess.run([REF], stdin=f, capture_output=True, text=True, check=True).stdout
    rows = [l.split() for l in out.strip().split("\n") if l]
    order = np.array([[int(r[0]), int(r[1]), int(r[2])] for r in rows], dtype=np.uint32).reshape(-1, 3)
    return order, np.array([float(r[3]) for r in rows])


CASES = [((32, 32, 32), 8, 16, 0), ((40, 36, 28), 6, 12, 1), ((96, 96), 8, 32, 0), ((64, 64), 4, 16, 1)]


@pytest.mark.parametrize("shape,S,G,variant", CASES)
@pytest.mark.parametrize("only_contour,type,upd", [(True, 1, False), (True, 2, False), (False, 1, True),
                                                   (False, 2, True), (False, 2, False)])
def test_engine_matches_reference(shape, S, G, variant, only_contour, type, upd):
    _ensure_ref()
    lab, pb = O.synth(shape, S, G, variant=variant)
    rag = O.Rag(lab, only_contour=only_contour)
    ro, rs = run_ref(rag, pb, type, upd)
    oo, os_ = rag.merge_order_pb(pb, type, upd)
    assert ro.shape == oo.shape and (ro == oo).all()
    assert np.allclose(rs, os_, rtol=0, atol=1e-12)


def test_constant_pb_tie_torture():
    _ensure_ref()
    lab, _ = O.synth((24, 24, 24), 6, 12)
    pb = np.full(lab.shape, 0.25, np.float32)
    rag = O.Rag(lab, only_contour=True)
    ro, rs = run_ref(rag, pb, 2, False)
    oo, os_ = rag.merge_order_pb(pb, 2, False)
    assert (ro == oo).all() and (rs == os_).all()
