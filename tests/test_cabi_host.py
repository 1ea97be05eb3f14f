"""CPU-only checks of the C ABI: the library loads, exports every symbol include/glia_hmt.h declares (no compute
calls without a GPU), and its host-side RF model parser agrees with the reference's own reader/writer."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

import _rf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "glia_amd", "libglia_hmt.so")
REF = os.path.join(ROOT, "oracle", "_ref", "ref_rfmodel")


def _lib():
    if not os.path.exists(SO):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "glia_amd", "csrc"), "-j4"], stdout=subprocess.DEVNULL)
    return C.CDLL(SO)


def test_library_exports_every_declared_symbol():
    lib = _lib()
    hdr = open(os.path.join(ROOT, "include", "glia_hmt.h")).read()
    names = sorted(set(re.findall(r"\b(glia_hmt_[a-z_0-9]+)\s*\(", hdr)))
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    lib.glia_hmt_version.restype = C.c_char_p
    assert b"gfx950" in lib.glia_hmt_version()


def test_errors_are_status_codes_not_exits():
    lib = _lib()
    lib.glia_hmt_last_error.restype = C.c_char_p
    nt, nn, nc = C.c_int(), C.c_int(), C.c_int()
    rc = lib.glia_hmt_forest_file_parse(b"/nonexistent/model.bin", C.c_int(-1), C.byref(nt), C.byref(nn), C.byref(nc),
                                        None, None, C.c_int64(0))
    assert rc == -6 and b"model file" in lib.glia_hmt_last_error()


def test_merge_order_invariant_and_options_are_host_code():
    """glia_hmt_check_merge_order (the replay of util/struct_merge.hxx:19-31 every pb / pre_merge order passes before it is returned)
    and glia_hmt_set_option need no GPU."""
    lib = _lib()
    lib.glia_hmt_internal_errors.restype = C.c_ulonglong
    R = 6
    good = np.array([[0, 1, 6], [2, 6, 7], [3, 4, 8], [7, 8, 9], [5, 9, 10]], np.uint32)
    bad_at = C.c_int64(7)
    assert lib.glia_hmt_check_merge_order(good.ctypes.data_as(C.c_void_p), C.c_int64(5), C.c_int64(R), C.byref(bad_at)) == 0 and bad_at.value == -1
    for row, col, val, where in ((3, 0, 0, 3), (2, 2, 9, 2), (4, 1, 5, 4), (1, 1, 7, 1)):      # gone region, wrong key, a == b, not created yet
        o = good.copy(); o[row, col] = val
        assert lib.glia_hmt_check_merge_order(o.ctypes.data_as(C.c_void_p), C.c_int64(5), C.c_int64(R), C.byref(bad_at)) == -1
        assert bad_at.value == where
    six = np.array([[0, 1, 6]] * 6, np.uint32)
    assert lib.glia_hmt_check_merge_order(six.ctypes.data_as(C.c_void_p), C.c_int64(6), C.c_int64(R), C.byref(bad_at)) == -1   # more merges than R - 1
    assert lib.glia_hmt_internal_errors() == 0
    assert lib.glia_hmt_set_option(b"GLIA_HMT_WINCAP", b"32") == 0 and lib.glia_hmt_set_option(b"GLIA_HMT_WINCAP", None) == 0
    assert lib.glia_hmt_set_option(b"GLIA_HMT_NO_SUCH_SWITCH", b"1") == -1


def _parse(lib, path, label=-1):
    nt, nn, nc = C.c_int(), C.c_int(), C.c_int()
    assert lib.glia_hmt_forest_file_parse(path.encode(), C.c_int(label), C.byref(nt), C.byref(nn), C.byref(nc), None, None,
                                          C.c_int64(0)) == 0
    split = np.empty((nt.value, nn.value)); meta = np.empty((nt.value, nn.value, 4), np.int32)
    assert lib.glia_hmt_forest_file_parse(path.encode(), C.c_int(label), C.byref(nt), C.byref(nn), C.byref(nc),
                                          split.ctypes.data_as(C.c_void_p), meta.ctypes.data_as(C.c_void_p),
                                          C.c_int64(split.size)) == 0
    return split, meta


@pytest.mark.parametrize("sparse", [False, True])
def test_model_parser_matches_python_writer(sparse):
    lib = _lib()
    rng = np.random.default_rng(3)
    forest = _rf.random_forest(rng, 9, 5, rng.random((50, 12)))
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "m.bin")
        _rf.write_model(p, forest, sparse=sparse)
        split, meta = _parse(lib, p)
    term = forest["nodestatus"] == -1
    used = np.arange(split.shape[1])[None, :] < forest["ndbigtree"][:, None]
    assert (split[used] == forest["xbestsplit"][used]).all()
    inner = used & ~term
    assert (meta[..., 0][inner] == forest["bestvar"][inner] - 1).all()
    assert (meta[..., 1][inner] == forest["treemap"][..., 0][inner] - 1).all()
    assert (meta[..., 2][inner] == forest["treemap"][..., 1][inner] - 1).all()
    assert (meta[..., 3][inner] == -1).all()
    assert (meta[..., 3][used & term] == (forest["nodeclass"][used & term] == 1)).all()     # class 1 <-> label -1


def _ensure_ref():
    if not os.path.exists(REF) and os.path.isdir("/root/reference/code"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/ref_rfmodel not built and /root/reference absent")


def _ref_text(forest):
    ntree, nrnodes = forest["xbestsplit"].shape
    nclass = len(forest["orig_labels"])

    def filemat(mem, n0, n1):
        return np.asarray(mem).reshape(n1, n0).T.reshape(-1)

    lines = ["%d %d %d 3" % (nrnodes, ntree, nclass), " ".join(map(str, forest["orig_labels"])),
             " ".join(map(str, range(1, nclass + 1)))]
    for name, n1, fmt in (("xbestsplit", ntree, "%.17g"), ("treemap", 2 * ntree, "%d"), ("nodestatus", ntree, "%d"),
                          ("nodeclass", ntree, "%d"), ("bestvar", ntree, "%d")):
        lines.append("%d %d " % (nrnodes, n1) + " ".join(fmt % v for v in filemat(forest[name], nrnodes, n1)))
    lines.append("%d 1 " % ntree + " ".join(map(str, forest["ndbigtree"])))
    lines.append("%d 1 " % nclass + " ".join(["1"] * nclass))
    lines.append("%d 1 " % nclass + " ".join(["%.17g" % (1.0 / nclass)] * nclass))
    return "\n".join(lines) + "\n"


def test_parser_reads_files_written_by_the_reference_writer():
    """rf_old::writeModelToBinaryFile (built in place from /root/reference) -> our parser."""
    _ensure_ref()
    lib = _lib()
    rng = np.random.default_rng(11)
    forest = _rf.random_forest(rng, 33, 7, rng.random((80, 20)))     # > 128 nodes/array: exercises the sparse encoder
    with tempfile.TemporaryDirectory() as d:
        txt, p = os.path.join(d, "f.txt"), os.path.join(d, "m.bin")
        open(txt, "w").write(_ref_text(forest))
        subprocess.check_call([REF, "write", txt, p])
        split, meta = _parse(lib, p)
    used = np.arange(split.shape[1])[None, :] < forest["ndbigtree"][:, None]
    inner = used & (forest["nodestatus"] != -1)
    assert (split[used] == forest["xbestsplit"][used]).all()
    assert (meta[..., 1][inner] == forest["treemap"][..., 0][inner] - 1).all()
    assert (meta[..., 2][inner] == forest["treemap"][..., 1][inner] - 1).all()


def test_reference_reader_reads_files_written_by_our_test_writer():
    """tests/_rf.write_model (same layout as glia_amd.synth_forest) -> rf_old::readModelFromBinaryFile."""
    _ensure_ref()
    rng = np.random.default_rng(5)
    forest = _rf.random_forest(rng, 17, 6, rng.random((60, 10)))
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "m.bin")
        _rf.write_model(p, forest)
        out = subprocess.run([REF, "read", p], capture_output=True, text=True, check=True).stdout
    rows = {l.split()[0]: l.split()[1:] for l in out.strip().split("\n")}
    assert rows["sizeof_model"] == ["520"]
    xb = np.array(rows["xbestsplit"][2:], dtype=np.float64)
    tm = np.array(rows["treemap"][2:], dtype=np.int64)
    ns = np.array(rows["nodestatus"][2:], dtype=np.int64)
    # after the reader's transposes the memory is node-fastest per tree, exactly our in-memory layout
    assert (xb == forest["xbestsplit"].reshape(-1)).all()
    assert (tm == forest["treemap"].reshape(-1)).all()
    assert (ns == forest["nodestatus"].reshape(-1)).all()


def test_region_map_order_emulation_equals_the_libstdcxx_container():
    """The orientation of initial edges follows the iteration order of the reference's std::unordered_map region map; the
    library replays the insertions on arrays under libstdc++'s hashtable rules (rmap_order.cpp) -- it must give what the
    real container gives, for dense and sparse label sets, across many growth steps."""
    import ctypes as C
    import numpy as np
    lib = C.CDLL(SO)
    rng = np.random.default_rng(7)
    for n, sparse in ((1, 0), (2, 0), (12, 1), (13, 0), (14, 1), (1000, 1), (4099, 0), (70001, 1), (300000, 0)):
        step = rng.integers(1, 5000, size=n) if sparse else np.ones(n, dtype=np.int64)
        labels = np.cumsum(step).astype(np.uint32)
        first = rng.permutation(n * 7)[:n].astype(np.int64)
        out = []
        for mode in (1, 2, 0):
            r = np.empty(n, dtype=np.uint32)
            assert lib.glia_hmt_host_rmap_ranks(labels.ctypes.data_as(C.c_void_p), first.ctypes.data_as(C.c_void_p), C.c_int64(n), C.c_int(mode),
                                                r.ctypes.data_as(C.c_void_p)) == 0
            out.append(r)
        assert sorted(out[0].tolist()) == list(range(n))
        assert (out[0] == out[1]).all() and (out[0] == out[2]).all(), n
