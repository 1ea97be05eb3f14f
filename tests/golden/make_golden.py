"""Generates the committed golden fixtures (tests/golden/*.npz).

GLIA ships no tests or sample data, so the goldens are produced HERE, where /root/reference is
mounted: inputs come from the repo's synthetic generator (or are written out by hand), expected outputs from the oracle,
and -- for the pb linkages -- additionally from the reference's own engine headers compiled in place
(oracle/_ref/ref_engine): a fixture records "ref_engine" in `pb_source` only if that binary produced the identical
order.  Fixtures are data only: label volumes (u16), Q8 pb codes (u8), masks, merge orders, saliencies, feature rows.

usage (from the repo root, with /root/reference present):  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as O          # noqa: E402
from test_oracle_vs_ref import run_ref, REF   # noqa: E402


def case(name, labels, pb, mask=None, feats_rows=40):
    assert labels.max() < 65536 and np.array_equal(np.round(pb * 256) / 256, pb)
    out = dict(labels=labels.astype(np.uint16), pb_q8=np.round(pb * 256).astype(np.uint16), shape=np.array(labels.shape))
    if mask is not None:
        out["mask"] = mask.astype(np.uint8)
    src = "oracle"
    for typ, key in ((2, "mean"), (1, "median")):
        rag = O.Rag(labels, mask=mask, only_contour=True)
        o, s = rag.merge_order_pb(pb, type=typ)
        if mask is None and os.path.exists(REF):
            ro, rs = run_ref(O.Rag(labels, only_contour=True), pb, typ, False)
            assert ro.shape == o.shape and (ro == o).all() and np.allclose(rs, s, rtol=0, atol=1e-12)
            src = "ref_engine"
        out["pb_%s_order" % key] = o
        out["pb_%s_sal" % key] = s
    out["pb_source"] = np.array(src)
    # classifier path with the diagnostic scorer P = 1 - x[stub] and the tree resolution after it
    cfg = O.make_cfg(pb, rb=[(pb, 8, 0.0, 1.0)])
    stub = (11 + 4 * 3 + 7 + 1) if labels.ndim == 3 else (11 + 4 * 3 + 7 + 1)
    o, s, f = O.Rag(labels, mask=mask).merge_order_bc(cfg, None, stub_index=stub, want_feats=True)
    out["bc_stub_index"] = np.array(stub)
    out["bc_order"] = o
    out["bc_sal"] = s
    out["bc_feats_head"] = f[:feats_rows]
    lab, par, c0, c1, pot = O.tree_potentials(o, np.clip(s, 0.0, 1.0))
    picks = O.resolve_tree_greedy(par, c0, c1, pot)
    out["tree_picks"] = picks
    src_l, dst_l = O.label_transform(lab, c0, c1, picks, 1)
    out["final_labels"] = O.transform_image(labels, src_l, dst_l, mask=mask, fill_missing=True).astype(np.uint16)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: getattr(v, "shape", None) for k, v in out.items()}, "pb orders from", src)


def main():
    # 2D 6x6 tie case: every saliency equal
    lab = np.array([[1 + (x // 2) + 3 * (y // 2) for x in range(6)] for y in range(6)], dtype=np.uint32)
    case("p2_tie_6x6", lab, np.full((6, 6), 0.5, np.float32), feats_rows=8)
    # 2D 128x128 blobs, 3D 32^3, ragged 3D with a mask
    for name, shape, S, G, masked in (("blobs_128x128", (128, 128), 8, 32, False), ("vol_32", (32, 32, 32), 8, 16, False),
                                      ("vol_40x36x28_masked", (40, 36, 28), 6, 12, True)):
        labels, pb = O.synth(shape, S, G)
        mask = None
        if masked:
            rng = np.random.default_rng(11)
            mask = (rng.random(shape) > 0.15).astype(np.uint32)
            mask[..., :3] = 0
        case(name, labels, pb, mask)


if __name__ == "__main__":
    main()
