"""-m gpu: the drop-in command line tools (cli/) write the reference's text formats (util/text_io.hxx:103-133)
with the oracle's values.  Images travel as MetaImage files, the format ITK writes natively."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "cli")


def write_mha(path, arr, local=True, compress=False):
    et = {np.dtype(np.uint32): "MET_UINT", np.dtype(np.float32): "MET_FLOAT", np.dtype(np.uint16): "MET_USHORT",
          np.dtype(np.uint8): "MET_UCHAR"}[arr.dtype]
    dims = " ".join(str(d) for d in arr.shape[::-1])
    data = np.ascontiguousarray(arr).tobytes()
    if compress:
        import zlib
        data = zlib.compress(data, 6)
    hdr = ("ObjectType = Image\nNDims = %d\nBinaryData = True\nBinaryDataByteOrderMSB = False\nCompressedData = %s\n%s"
           "DimSize = %s\nElementType = %s\nElementDataFile = %s\n") % (arr.ndim, "True" if compress else "False",
                                                                      "CompressedDataSize = %d\n" % len(data) if compress else "", dims, et,
                                                                      "LOCAL" if local else os.path.basename(path) + ".raw")
    with open(path, "wb") as f:
        f.write(hdr.encode())
        if local:
            f.write(data)
    if not local:
        with open(path + ".raw", "wb") as f:
            f.write(data)


@pytest.fixture(scope="module")
def tools():
    import torch
    assert torch.cuda.is_available(), "GPU test run without a GPU"
    subprocess.check_call(["make", "-C", CLI])
    return CLI


def _g(v, prec=6):
    return "%.*g" % (prec, v)


@pytest.mark.parametrize("shape,local", [((32, 32, 32), True), ((48, 40), False)])
def test_merge_order_pb_cli(tools, tmp_path, shape, local):
    from oracle import pyoracle as O
    labels, pb = O.synth(shape, 8 if len(shape) == 3 else 4, 16)
    seg, pbf = str(tmp_path / "seg.mha"), str(tmp_path / "pb.mhd")
    write_mha(seg, labels, True)
    write_mha(pbf, pb, local)
    order_f, sal_f = str(tmp_path / "order.txt"), str(tmp_path / "sal.txt")
    subprocess.check_call([os.path.join(tools, "merge_order_pb"), "-s", seg, "-p", pbf, "-t", "2", "-o", order_f, "-y", sal_f])
    o_ref, s_ref = O.Rag(labels, only_contour=True).merge_order_pb(pb, type=2)
    got = np.loadtxt(order_f, dtype=np.int64).reshape(-1, 3)
    assert (got == o_ref).all()
    lines = open(sal_f).read().split("\n")
    assert lines[-1] == "" and lines[:-1] == [_g(s) for s in s_ref]       # default ostream precision


@pytest.mark.parametrize("slabs,typ,local", [(3, 1, True), (4, 2, False), (2, 1, False)])
def test_merge_order_pb_cli_slab_route(tools, tmp_path, slabs, typ, local):
    """merge_order_pb --slabs N: every slab read from the MetaImage by its own plane range, the records (and, for the median
    linkage, the boundary values) through glia_hmt_rag_build_distributed; the same files as the single-pass tool writes"""
    from oracle import pyoracle as O
    shape = (48, 36, 40)
    labels, pb = O.synth(shape, 6, 12)
    seg, pbf = str(tmp_path / "seg.mha"), str(tmp_path / ("pb.mha" if local else "pb.mhd"))
    write_mha(seg, labels, True)
    write_mha(pbf, pb, local)
    outs = []
    for extra in ([], ["--slabs", str(slabs)]):
        order_f, sal_f = str(tmp_path / ("order%d.txt" % len(outs))), str(tmp_path / ("sal%d.txt" % len(outs)))
        subprocess.check_call([os.path.join(tools, "merge_order_pb"), "-s", seg, "-p", pbf, "-t", str(typ), "-o", order_f, "-y", sal_f] + extra)
        outs.append((open(order_f, "rb").read(), open(sal_f, "rb").read()))
    assert outs[0] == outs[1] and len(outs[0][0]) > 0
    o_ref, _ = O.Rag(labels, only_contour=True).merge_order_pb(pb, type=typ)
    assert (np.loadtxt(str(tmp_path / "order1.txt"), dtype=np.int64).reshape(-1, 3) == o_ref).all()


def test_merge_order_pb_cli_slab_route_one_rccl_rank(tools, tmp_path):
    """the one-process-per-slab form with the one rank a one-GPU box allows: unique id through the file, RCCL communicator"""
    from oracle import pyoracle as O
    labels, pb = O.synth((32, 32, 32), 8, 16)
    seg, pbf = str(tmp_path / "seg.mha"), str(tmp_path / "pb.mha")
    write_mha(seg, labels, True)
    write_mha(pbf, pb, True)
    order_f = str(tmp_path / "order.txt")
    subprocess.check_call([os.path.join(tools, "merge_order_pb"), "-s", seg, "-p", pbf, "-o", order_f, "--slabs", "1", "--rank", "0",
                           "--commId", str(tmp_path / "id.bin")])
    o_ref, _ = O.Rag(labels, only_contour=True).merge_order_pb(pb, type=1)
    assert (np.loadtxt(order_f, dtype=np.int64).reshape(-1, 3) == o_ref).all()
    r = subprocess.run([os.path.join(tools, "merge_order_pb"), "-s", seg, "-p", pbf, "--slabs", "2", "-m", seg], capture_output=True)
    assert r.returncode == 1 and b"mask" in r.stderr


def test_merge_order_bc_cli_slab_route(tools, tmp_path):
    """merge_order_bc --slabs 3 (two image volumes, --ns: the normalisers stay the whole volume's): byte-identical output files"""
    from oracle import pyoracle as O
    import _rf
    shape = (36, 32, 40)
    labels, pb = O.synth(shape, 6, 12)
    rng = np.random.default_rng(4)
    raw = (np.round(rng.random(shape) * 255) / 256.0).astype(np.float32)
    cfg = O.make_cfg(pb, rb=[(raw, 8, 0.0, 1.0), (pb, 8, 0.0, 1.0)], norm_area=float(np.prod(shape)), norm_len=float(np.sqrt(sum(d * d for d in shape))))
    _, _, f0 = O.Rag(labels).merge_order_bc(cfg, None, stub_index=31, want_feats=True)
    model = str(tmp_path / "model.bin")
    _rf.write_model(model, _rf.random_forest(np.random.default_rng(3), 31, 6, f0))
    seg, pbf, rawf = str(tmp_path / "seg.mha"), str(tmp_path / "pb.mha"), str(tmp_path / "raw.mha")
    write_mha(seg, labels); write_mha(pbf, pb); write_mha(rawf, raw)
    outs = []
    for extra in ([], ["--slabs", "3"]):
        files = [str(tmp_path / ("%s%d.txt" % (n, len(outs)))) for n in ("order", "sal", "bfeat")]
        subprocess.check_call([os.path.join(tools, "merge_order_bc"), "--bct", "1", "--bcm", model, "-s", seg, "--pb", pbf,
                               "--rbi", rawf, "--rbb", "8", "--rbl", "0.0", "--rbu", "1.0", "--rbi", pbf, "--rbb", "8", "--rbl", "0.0", "--rbu", "1.0",
                               "--bt", "0.2", "0.5", "0.8", "-n", "1", "-l", "0", "-o", files[0], "--sal", files[1], "-b", files[2]] + extra)
        outs.append([open(f_, "rb").read() for f_ in files])
    assert outs[0] == outs[1] and len(outs[0][0]) > 0


def test_merge_order_pb_cli_errors(tools, tmp_path):
    r = subprocess.run([os.path.join(tools, "merge_order_pb"), "-s", str(tmp_path / "missing.mha"), "-p", "x.mha"], capture_output=True)
    assert r.returncode == 1 and b"Error" in r.stderr
    r = subprocess.run([os.path.join(tools, "merge_order_pb"), "-p", "x.mha"], capture_output=True)
    assert r.returncode == 1 and b"required" in r.stderr


def test_merge_order_bc_cli(tools, tmp_path):
    from oracle import pyoracle as O
    import _rf
    shape = (32, 32, 32)
    labels, pb = O.synth(shape, 8, 16)
    cfg = O.make_cfg(pb, rb=[(pb, 8, 0.0, 1.0)])
    _, _, f0 = O.Rag(labels).merge_order_bc(cfg, None, stub_index=31, want_feats=True)
    forest = _rf.random_forest(np.random.default_rng(3), 31, 6, f0)
    model = str(tmp_path / "model.bin")
    _rf.write_model(model, forest)
    seg, pbf = str(tmp_path / "seg.mha"), str(tmp_path / "pb.mha")
    write_mha(seg, labels)
    write_mha(pbf, pb)
    order_f, sal_f, feat_f = (str(tmp_path / n) for n in ("order.txt", "sal.txt", "bfeat.txt"))
    subprocess.check_call([os.path.join(tools, "merge_order_bc"), "--bct", "1", "--bcm", model, "-s", seg, "--pb", pbf,
                           "--rbi", pbf, "--rbb", "8", "--rbl", "0.0", "--rbu", "1.0", "--bt", "0.2", "0.5", "0.8",
                           "-n", "0", "-l", "0", "-o", order_f, "--sal", sal_f, "-b", feat_f])
    o_ref, s_ref, f_ref = O.Rag(labels).merge_order_bc(cfg, O.make_forest(forest, -1), want_feats=True)
    assert (np.loadtxt(order_f, dtype=np.int64).reshape(-1, 3) == o_ref).all()
    assert open(sal_f).read().split("\n")[:-1] == [_g(s) for s in s_ref]
    rows = [ln.split(" ") for ln in open(feat_f).read().split("\n")[:-1]]
    assert len(rows) == len(f_ref) and all(r[-1] == "" and len(r) == f_ref.shape[1] + 1 for r in rows)   # trailing delimiter
    got = np.array([[float(x) for x in r[:-1]] for r in rows])
    assert np.allclose(got, f_ref, rtol=6e-8, atol=1e-12)            # FLT_PREC = 8 significant digits: half a unit of the 8th


def test_merge_order_bc_cli_ensemble(tools, tmp_path):
    """--bcm m0 --bcm m1 --bcm m2 --bcmd dim0 dim1 threshold (hmt/main_merge_order_bc.cxx:103-109)"""
    from oracle import pyoracle as O
    import _rf
    shape = (32, 32, 32)
    labels, pb = O.synth(shape, 8, 16)
    cfg = O.make_cfg(pb, rb=[(pb, 8, 0.0, 1.0)])
    _, _, f0 = O.Rag(labels).merge_order_bc(cfg, None, stub_index=31, want_feats=True)
    dim0, dim1 = 35, 35 + 23
    thr = float(np.median(np.concatenate([f0[:, dim0], f0[:, dim1]]))) + 0.5
    rng = np.random.default_rng(5)
    forests = [_rf.random_forest(rng, nt, 6, f0) for nt in (15, 31, 7)]
    models = []
    for k, f in enumerate(forests):
        models.append(str(tmp_path / ("m%d.bin" % k))); _rf.write_model(models[-1], f)
    seg, pbf, order_f, sal_f = (str(tmp_path / n) for n in ("seg.mha", "pb.mha", "order.txt", "sal.txt"))
    write_mha(seg, labels)
    write_mha(pbf, pb)
    subprocess.check_call([os.path.join(tools, "merge_order_bc"), "--bct", "1", "--bcm", models[0], "--bcm", models[1], "--bcm", models[2],
                           "--bcmd", str(dim0), "--bcmd", str(dim1), "--bcmd", repr(thr), "-s", seg, "--pb", pbf,
                           "--rbi", pbf, "--rbb", "8", "--rbl", "0.0", "--rbu", "1.0", "--bt", "0.2", "0.5", "0.8",
                           "-o", order_f, "--sal", sal_f])
    o_ref, s_ref = O.Rag(labels).merge_order_bc_ensemble(cfg, [O.make_forest(f, -1) for f in forests], dim0, dim1, thr)
    assert (np.loadtxt(order_f, dtype=np.int64).reshape(-1, 3) == o_ref).all()
    assert open(sal_f).read().split("\n")[:-1] == [_g(s) for s in s_ref]
    # two models, or three without the distributor arguments: the reference's message and exit code
    r = subprocess.run([os.path.join(tools, "merge_order_bc"), "--bct", "1", "--bcm", models[0], "--bcm", models[1], "--bcm", models[2],
                        "-s", seg, "--pb", pbf, "-o", order_f], capture_output=True)
    assert r.returncode == 1 and b"model distributor needs 3 arguments" in r.stderr


def read_mha(path):
    raw = open(path, "rb").read()
    i = raw.index(b"ElementDataFile = LOCAL\n") + len(b"ElementDataFile = LOCAL\n")
    hdr = dict(ln.split(" = ") for ln in raw[:i].decode().strip().split("\n"))
    dims = [int(x) for x in hdr["DimSize"].split()][::-1]
    dt = {"MET_UINT": np.uint32, "MET_USHORT": np.uint16, "MET_FLOAT": np.float32}[hdr["ElementType"]]
    body = raw[i:]
    if hdr.get("CompressedData") == "True":
        import zlib
        assert int(hdr["CompressedDataSize"]) == len(body)
        body = zlib.decompress(body)
    return np.frombuffer(body, dtype=dt).reshape(dims)


@pytest.mark.parametrize("local", [True, False])
def test_compressed_metaimage_in_and_out(tools, tmp_path, local):
    """--compress / -z (itk::ImageFileWriter::SetUseCompression, util/image_io.hxx:46-52): the tools read zlib-compressed MetaImage
    files (data in the .mha or beside it) and write them"""
    from oracle import pyoracle as O
    labels, pb = O.synth((40, 36, 28), 6, 12)
    seg, pbf, out = (str(tmp_path / n) for n in ("seg.mha", "pb.mha", "out.mha"))
    write_mha(seg, labels, local=local, compress=True)
    write_mha(pbf, pb, local=local, compress=True)
    subprocess.check_call([os.path.join(tools, "pre_merge"), "-s", seg, "-p", pbf, "-t", "100", "500", "-b", "0.28", "-r", "1", "-z", "1", "-o", out])
    assert b"CompressedData = True" in open(out, "rb").read(300)
    o_ref, _ = O.Rag(labels).pre_merge(pb, [100, 500], 0.28)
    ref, _ = O.relabel_image(O.transform_image(labels, *O.transform_keys(o_ref)))
    assert (read_mha(out) == ref).all()
    # the next tool: compressed in, compressed out
    order_f, out2 = str(tmp_path / "order.txt"), str(tmp_path / "out2.mha")
    with open(order_f, "w") as f:
        for r in o_ref:
            f.write("%d %d %d\n" % tuple(r))
    subprocess.check_call([os.path.join(tools, "apply_merges"), "-i", seg, "-g", order_f, "-z", "1", "-o", out2])
    ref = O.transform_image(labels, *O.transform_keys(o_ref))
    assert (read_mha(out2) == ref).all()


def test_bc_feat_cli(tools, tmp_path):
    from oracle import pyoracle as O
    labels, pb = O.synth((32, 32, 32), 8, 16)
    cfg = O.make_cfg(pb, rb=[(pb, 8, 0.0, 1.0)], use_log=True)
    order, _ = O.Rag(labels, only_contour=True).merge_order_pb(pb, type=2)
    seg, pbf, order_f, feat_f = (str(tmp_path / n) for n in ("seg.mha", "pb.mha", "order.txt", "bfeat.txt"))
    write_mha(seg, labels)
    write_mha(pbf, pb)
    with open(order_f, "w") as f:
        for r in order:
            f.write("%d %d %d\n" % tuple(r))
    subprocess.check_call([os.path.join(tools, "bc_feat"), "-s", seg, "-o", order_f, "--pb", pbf, "--rbi", pbf, "--rbb", "8",
                           "--rbl", "0", "--rbu", "1", "--bt", "0.2", "0.5", "0.8", "-l", "1", "-b", feat_f])
    f_ref = O.Rag(labels).bc_feat(cfg, order)
    got = np.array([[float(x) for x in ln.split(" ")[:-1]] for ln in open(feat_f).read().split("\n")[:-1]])
    assert got.shape == f_ref.shape and np.allclose(got, f_ref, rtol=6e-8, atol=1e-12)
    # -y: saliency file -> five more columns
    _, sal = O.Rag(labels, only_contour=True).merge_order_pb(pb, type=2)
    sal_f = str(tmp_path / "sal.txt")
    with open(sal_f, "w") as f:
        for v in sal:
            f.write("%.17g\n" % v)
    subprocess.check_call([os.path.join(tools, "bc_feat"), "-s", seg, "-o", order_f, "-y", sal_f, "--s0", "0.5", "--sb", "2.0", "--pb", pbf, "--rbi", pbf,
                           "--rbb", "8", "--rbl", "0", "--rbu", "1", "--bt", "0.2", "0.5", "0.8", "-l", "1", "-b", feat_f])
    f_ref = O.Rag(labels).bc_feat(cfg, order, saliencies=sal, init_sal=0.5, sal_bias=2.0)
    got = np.array([[float(x) for x in ln.split(" ")[:-1]] for ln in open(feat_f).read().split("\n")[:-1]])
    assert got.shape == f_ref.shape and np.allclose(got, f_ref, rtol=6e-8, atol=1e-12)
    # --medf 1: the GLIA_USE_MEDIAN_AS_FEATS layout (a build option of the reference, CMakeLists.txt:55,62-64): eight more columns here
    subprocess.check_call([os.path.join(tools, "bc_feat"), "-s", seg, "-o", order_f, "--pb", pbf, "--rbi", pbf, "--rbb", "8",
                           "--rbl", "0", "--rbu", "1", "--bt", "0.2", "0.5", "0.8", "-l", "1", "--medf", "1", "-b", feat_f])
    f_ref = O.Rag(labels).bc_feat(O.make_cfg(pb, rb=[(pb, 8, 0.0, 1.0)], use_log=True, median_as_feats=True), order)
    got = np.array([[float(x) for x in ln.split(" ")[:-1]] for ln in open(feat_f).read().split("\n")[:-1]])
    assert got.shape == f_ref.shape == (len(order), 112) and np.allclose(got, f_ref, rtol=6e-8, atol=1e-12)


@pytest.mark.parametrize("relabel,w16", [(0, 0), (1, 1)])
def test_pre_merge_and_apply_merges_cli(tools, tmp_path, relabel, w16):
    from oracle import pyoracle as O
    labels, pb = O.synth((48, 48, 48), 6, 12)
    seg, pbf, out, out2, order_f = (str(tmp_path / n) for n in ("seg.mha", "pb.mha", "out.mha", "out2.mha", "order.txt"))
    write_mha(seg, labels)
    write_mha(pbf, pb)
    subprocess.check_call([os.path.join(tools, "pre_merge"), "-s", seg, "-p", pbf, "-t", "100", "500", "-b", "0.28", "-r", str(relabel),
                           "-u", str(w16), "-o", out])
    o_ref, _ = O.Rag(labels).pre_merge(pb, [100, 500], 0.28)
    ref = O.transform_image(labels, *O.transform_keys(o_ref))
    if relabel:
        ref, _ = O.relabel_image(ref)
    got = read_mha(out)
    assert got.dtype == (np.uint16 if w16 else np.uint32) and (got == ref).all()
    # apply_merges with the order split over two files (they are concatenated and sorted by x2) and a mask
    with open(order_f, "w") as f:
        for r in o_ref[::2]:
            f.write("%d %d %d\n" % tuple(r))
    with open(order_f + "2", "w") as f:
        for r in o_ref[1::2]:
            f.write("%d %d %d\n" % tuple(r))
    mask = np.ones(labels.shape, np.uint32)
    mask[:, :, :10] = 0
    maskf = str(tmp_path / "mask.mha")
    write_mha(maskf, mask)
    subprocess.check_call([os.path.join(tools, "apply_merges"), "-i", seg, "-m", maskf, "-g", order_f, "-g", order_f + "2", "-o", out2])
    ref2 = O.transform_image(labels, *O.transform_keys(o_ref), mask=mask)
    assert (read_mha(out2) == ref2).all()


def test_end_to_end_classifier_tree_to_final_segmentation(tools, tmp_path):
    """seg + pb -> merge_order_bc (order + P(merge) per merge) -> segment_greedy -> final label image, against the same
    chain run through the oracle (merge order, potentials, full-scan greedy picks, label transform, fill-missing)."""
    from oracle import pyoracle as O
    import _rf
    labels, pb = O.synth((32, 32, 32), 8, 16)
    cfg = O.make_cfg(pb, rb=[(pb, 8, 0.0, 1.0)])
    _, _, f0 = O.Rag(labels).merge_order_bc(cfg, None, stub_index=31, want_feats=True)
    forest = _rf.random_forest(np.random.default_rng(3), 31, 6, f0)
    model, seg, pbf, order_f, sal_f, out = (str(tmp_path / n) for n in ("model.bin", "seg.mha", "pb.mha", "order.txt", "sal.txt", "final.mha"))
    _rf.write_model(model, forest)
    write_mha(seg, labels)
    write_mha(pbf, pb)
    subprocess.check_call([os.path.join(tools, "merge_order_bc"), "--bct", "1", "--bcm", model, "-s", seg, "--pb", pbf, "--rbi", pbf, "--rbb", "8",
                           "--rbl", "0.0", "--rbu", "1.0", "--bt", "0.2", "0.5", "0.8", "-o", order_f, "--sal", sal_f])
    subprocess.check_call([os.path.join(tools, "segment_greedy"), "-s", seg, "-o", order_f, "-p", sal_f, "-f", out])
    o_ref, s_ref = O.Rag(labels).merge_order_bc(cfg, O.make_forest(forest, -1))
    probs = np.array([float("%.6g" % v) for v in s_ref])           # the saliency file holds 6 significant digits
    lab, par, c0, c1, pot = O.tree_potentials(o_ref, probs)
    picks = O.resolve_tree_greedy(par, c0, c1, pot)
    ref = O.transform_image(labels, *O.label_transform(lab, c0, c1, picks, 1), fill_missing=True)
    got = read_mha(out)
    assert (got == ref).all() and 1 < len(np.unique(got)) <= len(np.unique(labels))


def test_merge_order_bc_cli_with_two_volumes(tools, tmp_path):
    """--rbi raw --rbi pb (both lists get both images) + --rli textons, through files"""
    from oracle import pyoracle as O
    import _rf
    shape = (32, 32, 32)
    labels, pb = O.synth(shape, 8, 16)
    rng = np.random.default_rng(4)
    raw = (np.round(rng.random(shape) * 255) / 256.0).astype(np.float32)
    tex = (np.indices(shape).sum(0) % 5).astype(np.float32)
    ocfg = O.make_cfg(pb, rb=[(raw, 8, 0.0, 1.0), (pb, 8, 0.0, 1.0)], rl=[(tex, 8, -0.5, 4.5)])
    _, _, f0 = O.Rag(labels).merge_order_bc(ocfg, None, stub_index=31, want_feats=True)
    forest = _rf.random_forest(np.random.default_rng(3), 31, 6, f0)
    files = {n: str(tmp_path / n) for n in ("model.bin", "seg.mha", "pb.mha", "raw.mha", "tex.mha", "order.txt", "sal.txt", "bfeat.txt")}
    _rf.write_model(files["model.bin"], forest)
    for n, a in (("seg.mha", labels), ("pb.mha", pb), ("raw.mha", raw), ("tex.mha", tex)):
        write_mha(files[n], a)
    subprocess.check_call([os.path.join(tools, "merge_order_bc"), "--bct", "1", "--bcm", files["model.bin"], "-s", files["seg.mha"], "--pb", files["pb.mha"],
                           "--rbi", files["raw.mha"], "--rbb", "8", "--rbl", "0", "--rbu", "1",
                           "--rbi", files["pb.mha"], "--rbb", "8", "--rbl", "0", "--rbu", "1",
                           "--rli", files["tex.mha"], "--rlb", "8", "--rll", "-0.5", "--rlu", "4.5",
                           "--bt", "0.2", "0.5", "0.8", "-o", files["order.txt"], "--sal", files["sal.txt"], "-b", files["bfeat.txt"]])
    o_ref, s_ref, f_ref = O.Rag(labels).merge_order_bc(ocfg, O.make_forest(forest, -1), want_feats=True)
    assert (np.loadtxt(files["order.txt"], dtype=np.int64).reshape(-1, 3) == o_ref).all()
    got = np.array([[float(x) for x in ln.split(" ")[:-1]] for ln in open(files["bfeat.txt"]).read().split("\n")[:-1]])
    assert got.shape == f_ref.shape and np.allclose(got, f_ref, rtol=6e-8, atol=1e-12)


def test_segment_greedy_cli_with_two_trees(tools, tmp_path):
    """two merge orders of the same supervoxels resolved jointly (-o a -o b -p pa -p pb)"""
    from oracle import pyoracle as O
    labels, pb = O.synth((32, 32, 32), 8, 16)
    seg, out = str(tmp_path / "seg.mha"), str(tmp_path / "final.mha")
    write_mha(seg, labels)
    trees, args = [], []
    for k, typ in enumerate((2, 1)):
        o, s = O.Rag(labels, only_contour=True).merge_order_pb(pb, type=typ)
        probs = np.clip(1.0 + 2.5 * s, 0.0, 1.0)
        of, pf = str(tmp_path / ("order%d.txt" % k)), str(tmp_path / ("prob%d.txt" % k))
        with open(of, "w") as f:
            for r in o:
                f.write("%d %d %d\n" % tuple(r))
        with open(pf, "w") as f:
            for v in probs:
                f.write("%.17g\n" % v)
        args += ["-o", of, "-p", pf]
        trees.append(O.tree_potentials(o, probs))
    bc = str(tmp_path / "bc.mha")
    subprocess.check_call([os.path.join(tools, "segment_greedy"), "-s", seg] + args + ["-f", out, "-b", bc])
    raw = open(bc, "rb").read()
    got_bc = np.frombuffer(raw[raw.index(b"ElementDataFile = LOCAL\n") + 24:], dtype=np.float32).reshape(labels.shape)
    orders = [np.loadtxt(args[i + 1], dtype=np.int64).reshape(-1, 3).astype(np.uint32) for i in range(0, len(args), 4)]
    assert (got_bc == O.Rag(labels, only_contour=True).boundary_confidence(orders, trees)).all()
    pt, pn = O.resolve_trees_greedy(trees)
    src, dst = [], []
    for k in range(len(pt)):
        lab, par, c0, c1, pot = trees[pt[k]]
        s1, d1 = O.label_transform(lab, c0, c1, np.array([pn[k]], np.int32), 1 + k)
        src += s1.tolist(); dst += d1.tolist()
    ref = O.transform_image(labels, np.array(src, np.uint32), np.array(dst, np.uint32), fill_missing=True)
    assert (read_mha(out) == ref).all()
