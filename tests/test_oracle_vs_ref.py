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


def run_ref(rag, pb, type, upd, cond=None, want_keys=False):
    """cond = (size thresholds, rpb threshold): the pre_merge condition as fcond of the reference's TBoundaryTable::top"""
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "dump.txt")
        rag.dump(pb, type, upd, p)
        if cond is not None:
            with open(p, "a") as f:
                f.write("%d %s %r\n" % (len(cond[0]), " ".join(str(int(t)) for t in cond[0]), float(cond[1])))
        with open(p) as f:
            out = subprocess.run([REF], stdin=f, capture_output=True, text=True, check=True).stdout
    rows = [l.split() for l in out.strip().split("\n") if l]
    keys = [r for r in rows if r[0] == "K"]
    rows = [r for r in rows if r[0] != "K"]
    order = np.array([[int(r[0]), int(r[1]), int(r[2])] for r in rows], dtype=np.uint32).reshape(-1, 3)
    sal = np.array([float(r[3]) for r in rows])
    if want_keys:
        return order, sal, np.array([[int(r[1]), int(r[2])] for r in keys], dtype=np.uint32).reshape(-1, 2)
    return order, sal


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


@pytest.mark.parametrize("shape,S,G,variant", CASES)
@pytest.mark.parametrize("only_contour,type,upd", [(True, 2, False), (False, 1, True)])
def test_transform_keys_matches_reference(shape, S, G, variant, only_contour, type, upd):
    """transformKeys (util/struct_merge.hxx:188-210) of the reference's own order vs the oracle's restatement"""
    _ensure_ref()
    lab, pb = O.synth(shape, S, G, variant=variant)
    rag = O.Rag(lab, only_contour=only_contour)
    ro, _, keys = run_ref(rag, pb, type, upd, want_keys=True)
    src, dst = O.transform_keys(ro)
    o = np.argsort(src)
    assert len(keys) == len(src) and (keys[:, 0] == src[o]).all() and (keys[:, 1] == dst[o]).all()


def test_transform_keys_of_a_partial_order_matches_reference():
    """a pre_merge order is a forest: several final keys, untouched labels absent from the map"""
    _ensure_ref()
    lab, pb = O.synth((36, 40, 28), 6, 12)
    rag = O.Rag(lab, only_contour=False)
    ro, _, keys = run_ref(rag, pb, 2, True, cond=([150], 0.0), want_keys=True)
    assert 0 < len(ro) < rag.num_regions - 1
    src, dst = O.transform_keys(ro)
    o = np.argsort(src)
    assert len(keys) == len(src) and (keys[:, 0] == src[o]).all() and (keys[:, 1] == dst[o]).all()
    assert len(np.unique(dst)) > 1


PRE_MERGE = [((32, 32, 32), 8, 16, 0, [300], 0.0), ((32, 32, 32), 8, 16, 0, [200, 700], 0.3),
             ((40, 36, 28), 6, 12, 1, [100, 400], 0.25), ((96, 96), 8, 32, 0, [40, 90], 0.2),
             ((64, 64), 4, 16, 1, [10, 30], 0.35), ((40, 36, 28), 6, 12, 0, [1000000], 0.0)]


@pytest.mark.parametrize("shape,S,G,variant,sizes,rpb", PRE_MERGE)
def test_pre_merge_condition_matches_reference(shape, S, G, variant, sizes, rpb):
    """the fcond path of TBoundaryTable::top (type/boundary_table.hxx:46-52) in the reference's own code, with the size rule of
    gadget/main_pre_merge.cxx:27-76 restated in the driver, vs the oracle's pre_merge"""
    _ensure_ref()
    lab, pb = O.synth(shape, S, G, variant=variant)
    rag = O.Rag(lab, only_contour=False)
    ro, rs = run_ref(rag, pb, 2, True, cond=(sizes, rpb))
    oo, os_ = O.Rag(lab, only_contour=False).pre_merge(pb, sizes, rpb)
    assert ro.shape == oo.shape and (ro == oo).all()
    assert np.allclose(rs, os_, rtol=0, atol=1e-12)
    if sizes[0] < 1000000:
        assert len(ro) < rag.num_regions - 1            # the condition did reject edges


# ---- type/function.hxx ThresholdModelDistributor, util/text_io.hxx writeData / readData (oracle/_ref/ref_misc) ----
REF_MISC = os.path.join(os.path.dirname(REF), "ref_misc")
TEXT_IO_CHECK = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cli", "text_io_check")


def _ensure_misc():
    _ensure_ref()
    if not os.path.exists(REF_MISC):
        pytest.skip("oracle/_ref/ref_misc not built")


@pytest.mark.parametrize("seed", range(6))
def test_model_distributor_matches_reference(seed):
    _ensure_misc()
    rng = np.random.default_rng(500 + seed)
    D = int(rng.integers(2, 12)); n = 200
    dim0, dim1 = int(rng.integers(0, D)), int(rng.integers(0, D))
    thr = float(np.round(1 + rng.random() * 6) / 8)
    x = np.round(rng.random((n, D)) * 8) / 8                 # exact hits on the threshold included
    text = "%d %d %r %d %d\n" % (dim0, dim1, thr, n, D) + "\n".join(" ".join(repr(float(v)) for v in row) for row in x) + "\n"
    out = subprocess.run([REF_MISC, "dist"], input=text, capture_output=True, text=True, check=True).stdout.split()
    ref = np.array([int(t) for t in out])
    got = np.array([O.pick_model(dim0, dim1, thr, row) for row in x])
    assert (ref == got).all() and len(set(ref)) > 1


@pytest.mark.parametrize("seed", range(4))
def test_text_writers_match_reference_bytes(seed):
    """merge order / saliency / feature-row files of the tools (cli/text_io.hpp) vs writeData of util/text_io.hxx:103-133 at the
    call sites' precisions (default, FLT_PREC = 8), byte for byte; and the reference's readData reads them back"""
    _ensure_misc()
    if not os.path.exists(TEXT_IO_CHECK):
        subprocess.check_call(["make", "-C", os.path.dirname(TEXT_IO_CHECK), "text_io_check"], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(900 + seed)
    lab, pb = O.synth((20, 24, 16), 4, 8, variant=seed % 2)
    order, sal = O.Rag(lab, only_contour=True).merge_order_pb(pb, type=2)
    rows, cols = 37, 11
    feats = rng.standard_normal((rows, cols)) * 10.0 ** rng.integers(-12, 12, size=(rows, cols))
    feats[0, :4] = [0.0, 1.0, -1.0, 0.5]; feats[1, :3] = [1e-300, 123456789.0, 1.0 / 3.0]        # fixed / scientific / rounding cases
    sal = np.concatenate([sal, [0.0, -0.0, 1e-7, 123456.7, 1234567.0]])
    text = "%d\n" % len(order) + "".join("%d %d %d\n" % tuple(r) for r in order) + "%d\n" % len(sal) + \
           "".join("%r\n" % float(v) for v in sal) + "%d %d\n" % (rows, cols) + "".join("%r\n" % float(v) for v in feats.ravel())
    with tempfile.TemporaryDirectory() as a, tempfile.TemporaryDirectory() as b:
        out = subprocess.run([REF_MISC, "write", a], input=text, capture_output=True, text=True, check=True).stdout.split()
        subprocess.run([TEXT_IO_CHECK, b], input=text, capture_output=True, text=True, check=True)
        for name in ("order.txt", "sal.txt", "feats.txt"):
            with open(os.path.join(a, name), "rb") as fa, open(os.path.join(b, name), "rb") as fb:
                assert fa.read() == fb.read(), name
    assert out == ["roundtrip", "1", "1", str(rows), str(cols), str(rows * cols)]


# ---- merge tree: genTree / genTreeWithNodePotentials / resolveTreeGreedy (hmt/tree_build.hxx, hmt/tree_greedy.hxx) ----
REF_TREE = os.path.join(os.path.dirname(REF), "ref_tree")
REF_STATS = os.path.join(os.path.dirname(REF), "ref_stats")


def run_ref_tree(trees):
    """trees: list of (order, merge_probs | None, region_probs | None) -> per-tree node arrays, single picks, joint picks"""
    _ensure_ref()
    if not os.path.exists(REF_TREE):
        pytest.skip("oracle/_ref/ref_tree not built")
    lines = ["%d" % len(trees)]
    for order, mp, rp in trees:
        lines.append("%d %d %d" % (len(order), mp is not None, rp is not None))
        lines += ["%d %d %d" % tuple(r) for r in order]
        if mp is not None:
            lines += [repr(float(p)) for p in mp]
        if rp is not None:
            lines += [repr(float(p)) for p in rp]
    out = subprocess.run([REF_TREE], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout.split("\n")
    k = 0
    nodes = []
    for _ in trees:
        assert out[k].startswith("T ")
        n = int(out[k].split()[1]); k += 1
        rows = [out[k + i].split() for i in range(n)]; k += n
        nodes.append((np.array([int(r[0]) for r in rows], np.uint32), np.array([int(r[1]) for r in rows], np.int32),
                      np.array([int(r[2]) for r in rows], np.int32), np.array([int(r[3]) for r in rows], np.int32),
                      np.array([float(r[4]) for r in rows])))
    n = int(out[k].split()[1]); k += 1
    single = np.array([int(out[k + i]) for i in range(n)], np.int32); k += n
    n = int(out[k].split()[1]); k += 1
    joint = np.array([[int(x) for x in out[k + i].split()] for i in range(n)], np.int32).reshape(-1, 2)
    return nodes, single, joint


def _orders(seed):
    rng = np.random.default_rng(seed)
    shape = tuple(int(x) for x in rng.integers(10, 30, size=int(rng.integers(2, 4))))
    S = int(rng.integers(3, 7))
    lab, pb = O.synth(shape, S, 2 * S, seed=int(rng.integers(1, 1 << 40)), variant=int(rng.integers(0, 2)))
    o1, s1 = O.Rag(lab, only_contour=True).merge_order_pb(pb, type=2)
    o2, s2 = O.Rag(lab, only_contour=True).merge_order_pb(pb, type=1)
    return rng, (o1, s1), (o2, s2)


@pytest.mark.parametrize("seed", range(24))
def test_tree_potentials_and_greedy_picks_match_reference(seed):
    """oracle gen_tree / potentials / greedy picks (one tree and several trees) vs the reference's own headers, incl.
    equal-potential ties (constant and coarsely quantised merge probabilities) and forests (non-mutual RAGs)"""
    rng, (o1, s1), (o2, s2) = _orders(seed)
    mode = seed % 4
    if mode == 0:   p1, p2 = np.clip(1.0 + 2.5 * s1, 0, 1), np.clip(1.0 + 2.5 * s2, 0, 1)
    elif mode == 1: p1, p2 = np.full(len(o1), 0.5), np.full(len(o2), 0.5)                               # everything ties
    elif mode == 2: p1, p2 = np.round(rng.random(len(o1)) * 4) / 4, np.round(rng.random(len(o2)) * 4) / 4   # many ties, 0 and 1
    else:           p1, p2 = None, None                                                                  # no probabilities: all 1
    rp = (True, True) if seed % 3 == 0 else (None, None)
    # region probabilities are one per NODE: size them from the oracle's node count
    n1 = len(O.gen_tree(o1)[0]); n2 = len(O.gen_tree(o2)[0])
    r1 = rng.random(n1) if rp[0] is not None else None
    r2 = rng.random(n2) if rp[1] is not None else None
    if r1 is not None:
        r1[::7] = 0.0                                                      # max(p, FEPS) branch
    nodes, single, joint = run_ref_tree([(o1, p1, r1), (o2, p2, r2)])
    mine = [O.tree_potentials(o1, p1, r1), O.tree_potentials(o2, p2, r2)]
    for (lab, par, c0, c1, pot), ref in zip(mine, nodes):
        assert (lab == ref[0]).all() and (par == ref[1]).all() and (c0 == ref[2]).all() and (c1 == ref[3]).all()
        assert (pot == ref[4]).all()                                       # %.17g round-trips doubles exactly
    # genTree alone (labels / parents / children) = the same arrays
    g = O.gen_tree(o1)
    assert (g[0] == nodes[0][0]).all() and (g[1] == nodes[0][1]).all()
    assert (O.resolve_tree_greedy(*mine[0][1:]) == single).all()
    pt, pn = O.resolve_trees_greedy(mine)
    assert (pt == joint[:, 0]).all() and (pn == joint[:, 1]).all()


@pytest.mark.parametrize("seed", range(20))
def test_stats_match_reference(seed):
    """entropy / distL1 / distX2 / amedian / rescale restatements vs util/stats.hxx compiled in place"""
    _ensure_ref()
    if not os.path.exists(REF_STATS):
        pytest.skip("oracle/_ref/ref_stats not built")
    rng = np.random.default_rng(100 + seed)
    cases = []
    for _ in range(25):
        n = int(rng.integers(1, 40))
        tot = int(rng.integers(1, 5000))
        a = rng.multinomial(tot, rng.dirichlet(np.ones(n))) / tot          # histograms as the features see them (zeros included)
        b = rng.multinomial(tot + 3, rng.dirichlet(np.ones(n))) / (tot + 3)
        if rng.random() < 0.3:
            a = np.round(rng.random(n) * 4) / 4; b = np.round(rng.random(n) * 4) / 4     # ties for amedian, exact zeros
        cases.append((a, b))
    text = "%d\n" % len(cases) + "".join("%d\n%s\n%s\n" % (len(a), " ".join(repr(float(x)) for x in a), " ".join(repr(float(x)) for x in b))
                                         for a, b in cases)
    out = subprocess.run([REF_STATS], input=text, capture_output=True, text=True, check=True).stdout.strip().split("\n")
    for i, (a, b) in enumerate(cases):
        ref = np.array([float(x) for x in out[2 * i].split()])
        got = O.stats_case(a, b)
        assert (got == ref).all(), (i, got, ref)
        rs = np.array([float(x) for x in out[2 * i + 1].split()])
        assert (O.rescale(a, np.minimum(a, b), np.maximum(a, b)) == rs).all()


def test_bc_feat_under_the_reference_parfor_equals_the_serial_oracle(tmp_path):
    """oracle/_ref/ref_parfor_{st,mt}: the reference's util/mp.hxx (parfor, compiled in place without / with -DGLIA_MT -fopenmp) drives
    the oracle's feature functions over a merge order the way hmt/main_bc_feat.cxx:57-101 does (shuffled indices, two loops).  Both
    builds, and the OpenMP one on 1 and 4 threads, must write the rows of the oracle's own serial bc_feat byte for byte: the baseline
    bench.py times (cpu_baseline.bc_feat) computes what the checker computes."""
    _ensure_ref()
    st = os.path.join(os.path.dirname(REF), "ref_parfor_st")
    mt = os.path.join(os.path.dirname(REF), "ref_parfor_mt")
    if not (os.path.exists(st) and os.path.exists(mt)):
        pytest.skip("oracle/_ref/ref_parfor_* not built")
    size, S = 40, 8
    labels, pb = O.synth((size,) * 3, S, 8 * S)
    order, _ = O.Rag(labels, only_contour=True).merge_order_pb(pb, type=2)
    cfg = O.make_cfg(pb, rb=[(pb, 8, 0.0, 1.0)])
    want = O.Rag(labels).bc_feat(cfg, order)
    for exe, threads in ((st, 1), (mt, 1), (mt, 4)):
        out = str(tmp_path / "rows.bin")
        env = dict(os.environ, OMP_NUM_THREADS=str(threads))
        line = subprocess.run([exe, str(size), str(S), out], capture_output=True, text=True, check=True, env=env).stdout.split()
        kv = dict(zip(line[0::2], line[1::2]))
        assert int(kv["threads"]) == threads and int(kv["merges"]) == len(order) and int(kv["dim"]) == want.shape[1]
        got = np.fromfile(out, np.float64).reshape(want.shape)
        assert (got.view(np.uint64) == want.view(np.uint64)).all(), (exe, threads)
