"""The logarithms and the power of the feature vector must carry the HOST libm's bits (the reference calls std::log2 in
stats::entropy, util/stats.hxx:145-152, std::log in slog, glia_base.hxx:80-81, and std::pow(perim, 1.5) in
type/feat.hxx:78-79): the library restates glibc's
table-driven algorithms (glia_amd/csrc/glibc_math.hpp) and selects the variant that reproduces this host's libm.
CPU part: the host code of the restatement against the libm, bit for bit.  GPU part: the device code against both."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "glia_amd", "libglia_hmt.so")


def _inputs(n, seed=5):
    """histogram fractions p = c/n (what stats::entropy feeds log2), values around 1, and a wide sweep"""
    rng = np.random.default_rng(seed)
    tot = rng.integers(1, 4_000_000, size=n).astype(np.float64)
    cnt = np.floor(rng.random(n) * tot) + 1.0
    p = cnt / tot
    near = 1.0 + (rng.random(n // 4) - 0.5) * rng.choice([0.2, 1e-3, 1e-8], size=n // 4)
    wide = np.ldexp(1.0 + rng.random(n // 4), rng.integers(-300, 300, size=n // 4))
    small = (np.arange(1, 4097, dtype=np.float64)[:, None] / np.arange(1, 4097, dtype=np.float64)[None, :]).ravel()
    return np.concatenate([p, near, wide, small[small <= 1.0], [1.0, 0.5, 2.0, 5e-324, 2.2250738585072014e-308]])


def _host_eval(function, variant, x):
    lib = C.CDLL(SO)
    out = np.empty_like(x)
    rc = lib.glia_hmt_host_libm_eval(C.c_int(function), C.c_int(variant), x.ctypes.data_as(C.c_void_p),
                                     out.ctypes.data_as(C.c_void_p), C.c_int64(x.size))
    assert rc == 0
    return out


def test_host_libm_is_pinned_by_a_restatement():
    lib = C.CDLL(SO)
    a, b = C.c_int(-1), C.c_int(-1)
    assert lib.glia_hmt_host_libm_probe(C.byref(a), C.byref(b)) == 0
    assert a.value in (1, 2) and b.value in (1, 2), "no restatement reproduces this host's libm: entropy features unpinned"
    p = C.c_int(-1)
    assert lib.glia_hmt_host_libm_probe_pow(C.byref(p)) == 0
    assert p.value in (1, 2), "no restatement reproduces this host's pow: compactness unpinned"


def _perimeters(seed=3, dense=1 << 23):
    rng = np.random.default_rng(seed)
    return np.concatenate([np.arange(1, dense + 1, dtype=np.float64), rng.integers(1 << 23, 1 << 40, size=1 << 20).astype(np.float64)])


def test_pow_restatement_equals_host_pow_bit_for_bit():
    """pow(perim, 1.5) over every integer up to 2^23 (a 1024^3 volume's largest perimeters are ~1e6) and a sample up to 2^40"""
    from oracle import pyoracle as O
    lib = C.CDLL(SO)
    p = C.c_int(-1)
    lib.glia_hmt_host_libm_probe_pow(C.byref(p))
    x = _perimeters()
    got = _host_eval(2, p.value, x)
    ref = O.libm_eval(2, x)
    bad = got.view(np.uint64) != ref.view(np.uint64)
    assert not bad.any(), "pow variant %d: %d of %d differ, first x=%r" % (p.value, bad.sum(), x.size, x[bad][:3])
    other = _host_eval(2, 3 - p.value, x)      # the other build of pow really is a different function
    assert (other.view(np.uint64) != ref.view(np.uint64)).any()


def test_restatement_equals_host_libm_bit_for_bit():
    from oracle import pyoracle as O
    lib = C.CDLL(SO)
    a, b = C.c_int(-1), C.c_int(-1)
    lib.glia_hmt_host_libm_probe(C.byref(a), C.byref(b))
    x = _inputs(1_500_000)
    assert x.size >= 2_000_000
    for function, variant in ((0, a.value), (1, b.value)):
        got = _host_eval(function, variant, x)
        ref = O.libm_eval(function, x)
        bad = got.view(np.uint64) != ref.view(np.uint64)
        assert not bad.any(), "function %d variant %d: %d of %d differ, first x=%r" % (function, variant, bad.sum(), x.size, x[bad][:3])


@pytest.mark.gpu
def test_device_logarithms_equal_host_libm_bit_for_bit():
    import torch
    from glia_amd import hmt
    from oracle import pyoracle as O
    ctx = hmt.Context(0)
    v2, v1 = ctx.libm()
    assert v2 != 0 and v1 != 0, "host libm not pinned"
    x = _inputs(1_500_000, seed=11)
    d_x = torch.from_numpy(x).cuda()
    for function, variant in ((0, v2), (1, v1)):
        got = ctx.libm_eval(function, variant, d_x).cpu().numpy()
        ref = O.libm_eval(function, x)
        bad = got.view(np.uint64) != ref.view(np.uint64)
        assert not bad.any(), "function %d: %d of %d differ, first x=%r" % (function, bad.sum(), x.size, x[bad][:3])
    ctx.close()


@pytest.mark.gpu
def test_device_pow_of_perimeters_equals_host_pow_bit_for_bit():
    """compactness = pow(perim, 1.5) (type/feat.hxx:78-79): perim is a voxel count, so the domain is the integers.  glibc's
    pow is within 0.52 ulp but not correctly rounded (about 0.09 % of the integers are one ulp off the rounded true value), so
    the device evaluates the restatement of the host's pow the context probed -- bit for bit.  Variant 0 (the correctly rounded
    double-double value a context falls back to on an unknown libm) stays within one ulp."""
    import torch
    from glia_amd import hmt
    from oracle import pyoracle as O
    ctx = hmt.Context(0)
    vp = ctx.libm_pow()
    assert vp != 0, "host pow not pinned"
    x = _perimeters(dense=1 << 24)
    d_x = torch.from_numpy(x).cuda()
    ref = O.libm_eval(2, x)
    got = ctx.libm_eval(2, vp, d_x).cpu().numpy()
    bad = got.view(np.uint64) != ref.view(np.uint64)
    assert not bad.any(), "%d of %d differ, first x=%r" % (bad.sum(), x.size, x[bad][:3])
    ulps = np.abs(ctx.libm_eval(2, 0, d_x).cpu().numpy().view(np.int64) - ref.view(np.int64))
    assert ulps.max() <= 1 and (ulps != 0).mean() < 2e-3
    ctx.close()
