"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only to check (or time, as the CPU baseline) -- never as a
fallback of the product path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MAX_IMAGES = 8
MAX_THRESH = 8


class FeatCfg(C.Structure):
    _fields_ = [
        ("n_rimg", C.c_int), ("rimg", C.c_void_p * MAX_IMAGES), ("rbins", C.c_int * MAX_IMAGES),
        ("rlo", C.c_double * MAX_IMAGES), ("rhi", C.c_double * MAX_IMAGES),
        ("n_rlimg", C.c_int), ("rlimg", C.c_void_p * MAX_IMAGES), ("rlbins", C.c_int * MAX_IMAGES),
        ("rllo", C.c_double * MAX_IMAGES), ("rlhi", C.c_double * MAX_IMAGES),
        ("n_bimg", C.c_int), ("bimg", C.c_void_p * MAX_IMAGES), ("bbins", C.c_int * MAX_IMAGES),
        ("blo", C.c_double * MAX_IMAGES), ("bhi", C.c_double * MAX_IMAGES),
        ("pb", C.c_void_p), ("n_thr", C.c_int), ("thr", C.c_double * MAX_THRESH),
        ("norm_area", C.c_double), ("norm_len", C.c_double),
        ("use_log", C.c_int), ("use_simple", C.c_int), ("hist_as_feats", C.c_int), ("median_as_feats", C.c_int),
    ]


class Forest(C.Structure):
    _fields_ = [
        ("nrnodes", C.c_int), ("ntree", C.c_int), ("nclass", C.c_int),
        ("xbestsplit", C.c_void_p), ("treemap", C.c_void_p), ("nodestatus", C.c_void_p),
        ("nodeclass", C.c_void_p), ("bestvar", C.c_void_p), ("orig_labels", C.c_void_p),
        ("predict_label", C.c_int),
    ]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("hmt_oracle.cc", "hmt_oracle.hpp")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_rag_build.restype = C.c_void_p
        L.orc_rag_num_regions.restype = C.c_int64
        L.orc_rag_num_pairs.restype = C.c_int64
        L.orc_merge_order_pb.restype = C.c_int64
        L.orc_merge_order_bc.restype = C.c_int64
        L.orc_bc_feat.restype = C.c_int64
        L.orc_pre_merge.restype = C.c_int64
        L.orc_gen_tree.restype = C.c_int64
        L.orc_transform_keys.restype = C.c_int64
        L.orc_relabel_image.restype = C.c_int64
        L.orc_transform_image.restype = None
        L.orc_forest_predict.restype = C.c_double
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _dims(shape):
    """numpy shape (z,y,x) or (y,x) -> (dim, int64[3] as x,y,z)."""
    dim = len(shape)
    d = np.ones(3, dtype=np.int64)
    d[:dim] = shape[::-1]
    return dim, d


def synth(shape, S, G, seed=0x9E3779B97F4A7C15, variant=0):
    """Synthetic supervoxels + pb. shape is numpy order (z,y,x) / (y,x)."""
    dim, d = _dims(shape)
    labels = np.empty(shape, dtype=np.uint32)
    pb = np.empty(shape, dtype=np.float32)
    rc = lib().orc_synth(C.c_int(dim), _p(d), C.c_int(S), C.c_int(G), C.c_uint64(seed), C.c_int(variant),
                         _p(labels), _p(pb))
    assert rc == 0
    return labels, pb


def make_cfg(pb, rb=(), r=(), rl=(), b=(), thr=(0.2, 0.5, 0.8), norm_area=1.0, norm_len=1.0,
             use_log=False, use_simple=False, hist_as_feats=False, median_as_feats=False):
    """rb/r/rl/b: lists of (image, bins, lo, hi).  Mirrors prepareImages (hmt_util.hxx:17-56)."""
    cfg = FeatCfg()
    keep = [pb]
    rimgs = list(rb) + list(r)
    bimgs = list(rb) + list(b)
    for name, lst in (("r", rimgs), ("rl", list(rl)), ("b", bimgs)):
        setattr(cfg, "n_%simg" % name, len(lst))
        for i, (img, bins, lo, hi) in enumerate(lst):
            assert img.dtype == np.float32 and img.flags.c_contiguous
            keep.append(img)
            getattr(cfg, "%simg" % name)[i] = img.ctypes.data
            getattr(cfg, "%sbins" % name)[i] = bins
            getattr(cfg, "%slo" % name)[i] = lo
            getattr(cfg, "%shi" % name)[i] = hi
    cfg.pb = pb.ctypes.data
    cfg.n_thr = len(thr)
    for i, t in enumerate(thr):
        cfg.thr[i] = t
    cfg.norm_area, cfg.norm_len = norm_area, norm_len
    cfg.use_log, cfg.use_simple = int(use_log), int(use_simple)
    cfg.hist_as_feats = int(hist_as_feats)
    cfg.median_as_feats = int(median_as_feats)
    cfg._keep = keep
    return cfg


def make_forest(arrays, predict_label=-1):
    """arrays: dict with xbestsplit[ntree,nrnodes] f64, treemap[ntree,nrnodes,2] i32, nodestatus, nodeclass,
    bestvar [ntree,nrnodes] i32, orig_labels[nclass] i32."""
    f = Forest()
    xb = np.ascontiguousarray(arrays["xbestsplit"], dtype=np.float64)
    tm = np.ascontiguousarray(arrays["treemap"], dtype=np.int32)
    ns = np.ascontiguousarray(arrays["nodestatus"], dtype=np.int32)
    nc = np.ascontiguousarray(arrays["nodeclass"], dtype=np.int32)
    bv = np.ascontiguousarray(arrays["bestvar"], dtype=np.int32)
    ol = np.ascontiguousarray(arrays["orig_labels"], dtype=np.int32)
    f.ntree, f.nrnodes = xb.shape
    f.nclass = ol.size
    f.xbestsplit, f.treemap, f.nodestatus = xb.ctypes.data, tm.ctypes.data, ns.ctypes.data
    f.nodeclass, f.bestvar, f.orig_labels = nc.ctypes.data, bv.ctypes.data, ol.ctypes.data
    f.predict_label = predict_label
    f._keep = (xb, tm, ns, nc, bv, ol)
    return f


class Rag:
    def __init__(self, labels, mask=None, only_contour=False):
        labels = np.ascontiguousarray(labels, dtype=np.uint32)
        self.shape = labels.shape
        self.dim, d = _dims(labels.shape)
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint32)
        self.h = C.c_void_p(lib().orc_rag_build(C.c_int(self.dim), _p(d), _p(labels), _p(mask),
                                                C.c_int(int(only_contour))))

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:      # (module globals are gone at interpreter shutdown)
            lib().orc_rag_free(self.h)
            self.h = None

    @property
    def num_regions(self):
        return lib().orc_rag_num_regions(self.h)

    @property
    def num_pairs(self):
        return lib().orc_rag_num_pairs(self.h)

    def regions(self):
        n = self.num_regions
        lab = np.empty(n, np.uint32); npts = np.empty(n, np.int64); nb = np.empty(n, np.int64)
        lib().orc_rag_regions(self.h, _p(lab), _p(npts), _p(nb))
        return lab, npts, nb

    def pairs(self):
        n = self.num_pairs
        a = np.empty(n, np.uint32); b = np.empty(n, np.uint32); c = np.empty(n, np.int64)
        lib().orc_rag_pairs(self.h, _p(a), _p(b), _p(c))
        return a, b, c

    def region_iter_order(self):
        lab = np.empty(self.num_regions, np.uint32)
        lib().orc_rag_region_iter_order(self.h, _p(lab))
        return lab

    def pair_stats(self, img):
        n = self.num_pairs
        out = [np.empty(n, np.float64) for _ in range(4)]
        lib().orc_rag_pair_stats(self.h, _p(img), *[_p(o) for o in out])
        return out

    def region_stats(self, img):
        n = self.num_regions
        out = [np.empty(n, np.float64) for _ in range(4)]
        lo = np.empty((n, 3), np.int64); hi = np.empty((n, 3), np.int64)
        lib().orc_rag_region_stats(self.h, _p(img), *[_p(o) for o in out], _p(lo), _p(hi))
        return out + [lo, hi]

    def dump(self, pb, type, update_region, path):
        rc = lib().orc_rag_dump(self.h, _p(pb), C.c_int(type), C.c_int(int(update_region)), path.encode())
        assert rc == 0

    def merge_order_pb(self, pb, type=1, update_region=False):
        cap = max(self.num_regions, 1)
        order = np.empty((cap, 3), np.uint32); sal = np.empty(cap, np.float64)
        n = lib().orc_merge_order_pb(self.h, _p(pb), C.c_int(type), C.c_int(int(update_region)), _p(order),
                                     _p(sal), C.c_int64(cap))
        if n < 0:
            raise RuntimeError("orc_merge_order_pb failed: %d" % n)
        return order[:n].copy(), sal[:n].copy()

    def merge_order_bc(self, cfg, forest=None, stub_index=31, want_feats=False):
        cap = max(self.num_regions, 1)
        order = np.empty((cap, 3), np.uint32); sal = np.empty(cap, np.float64)
        d = lib().orc_feat_dim(C.c_int(self.dim), C.byref(cfg))
        feats = np.empty((cap, d), np.float64) if want_feats else None
        nev = C.c_int64(0)
        n = lib().orc_merge_order_bc(self.h, C.byref(cfg), C.byref(forest) if forest is not None else None,
                                     C.c_int(stub_index), _p(order), _p(sal), _p(feats), C.c_int64(cap),
                                     C.byref(nev))
        if n < 0:
            raise RuntimeError("orc_merge_order_bc failed: %d" % n)
        self.n_feat_evals = nev.value
        if want_feats:
            return order[:n].copy(), sal[:n].copy(), feats[:n].copy()
        return order[:n].copy(), sal[:n].copy()

    def merge_order_bc_ensemble(self, cfg, forests, dim0, dim1, threshold, want_feats=False):
        """alg::EnsembleRandomForest + ThresholdModelDistributor: forests = three make_forest() results."""
        cap = max(self.num_regions, 1)
        order = np.empty((cap, 3), np.uint32); sal = np.empty(cap, np.float64)
        d = lib().orc_feat_dim(C.c_int(self.dim), C.byref(cfg))
        feats = np.empty((cap, d), np.float64) if want_feats else None
        arr = (C.c_void_p * 3)(*[C.addressof(f) for f in forests])
        lib().orc_merge_order_bc_ensemble.restype = C.c_int64
        n = lib().orc_merge_order_bc_ensemble(self.h, C.byref(cfg), arr, C.c_int(dim0), C.c_int(dim1), C.c_double(threshold),
                                              _p(order), _p(sal), _p(feats), C.c_int64(cap))
        if n < 0:
            raise RuntimeError("orc_merge_order_bc_ensemble failed: %d" % n)
        if want_feats:
            return order[:n].copy(), sal[:n].copy(), feats[:n].copy()
        return order[:n].copy(), sal[:n].copy()

    def bc_feat(self, cfg, order, saliencies=None, init_sal=1.0, sal_bias=1.0):
        order = np.ascontiguousarray(order, dtype=np.uint32)
        d = lib().orc_feat_dim(C.c_int(self.dim), C.byref(cfg))
        if saliencies is not None and not cfg.use_simple:
            d += 5
        feats = np.empty((len(order), d), np.float64)
        sal = None if saliencies is None else np.ascontiguousarray(saliencies, dtype=np.float64)
        lib().orc_bc_feat_sal.restype = C.c_int64
        n = lib().orc_bc_feat_sal(self.h, C.byref(cfg), _p(order), C.c_int64(len(order)), _p(sal) if sal is not None else None,
                                  C.c_double(init_sal), C.c_double(sal_bias), _p(feats))
        if n < 0:
            raise RuntimeError("orc_bc_feat failed")
        return feats

    def boundary_confidence(self, orders, trees):
        """segment_greedy -b: orders = list of merge orders, trees = list of (label, parent, child0, child1, potential)"""
        nt = len(orders)
        ko = [np.ascontiguousarray(o, np.uint32) for o in orders]
        kl = [np.ascontiguousarray(t[0], np.uint32) for t in trees]
        kp = [np.ascontiguousarray(t[4], np.float64) for t in trees]
        arr = lambda ks: (C.c_void_p * nt)(*[k.ctypes.data for k in ks])
        nm = (C.c_int64 * nt)(*[len(o) for o in ko]); nn = (C.c_int64 * nt)(*[len(k) for k in kl])
        out = np.empty(self.shape, np.float32)
        rc = lib().orc_boundary_confidence(self.h, C.c_int(nt), arr(ko), nm, arr(kl), nn, arr(kp), _p(out))
        assert rc == 0
        return out

    def pre_merge(self, pb, size_thresholds, rpb_threshold):
        cap = max(self.num_regions, 1)
        order = np.empty((cap, 3), np.uint32); sal = np.empty(cap, np.float64)
        st = np.asarray(size_thresholds, dtype=np.int32)
        n = lib().orc_pre_merge(self.h, _p(pb), _p(st), C.c_int(len(st)), C.c_double(rpb_threshold), _p(order),
                                _p(sal), C.c_int64(cap))
        if n < 0:
            raise RuntimeError("orc_pre_merge failed")
        return order[:n].copy(), sal[:n].copy()


def feat_dim(dim, cfg):
    return lib().orc_feat_dim(C.c_int(dim), C.byref(cfg))


def forest_predict(forest, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    return lib().orc_forest_predict(C.byref(forest), _p(x), C.c_int(x.size))


def gen_tree(order):
    order = np.ascontiguousarray(order, dtype=np.uint32)
    cap = 3 * len(order) + 1      # a forest of several components has more than 2n+1 nodes
    lab = np.empty(cap, np.uint32)
    par = np.empty(cap, np.int32); c0 = np.empty(cap, np.int32); c1 = np.empty(cap, np.int32)
    n = lib().orc_gen_tree(_p(order), C.c_int64(len(order)), _p(lab), _p(par), _p(c0), _p(c1), C.c_int64(cap))
    assert n >= 0
    return lab[:n], par[:n], c0[:n], c1[:n]


def pick_model(dim0, dim1, threshold, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    return lib().orc_pick_model(C.c_int(dim0), C.c_int(dim1), C.c_double(threshold), _p(x))


def transform_keys(order):
    order = np.ascontiguousarray(order, dtype=np.uint32)
    cap = 2 * len(order) + 1
    src = np.empty(cap, np.uint32); dst = np.empty(cap, np.uint32)
    n = lib().orc_transform_keys(_p(order), C.c_int64(len(order)), _p(src), _p(dst), C.c_int64(cap))
    assert n >= 0
    return src[:n].copy(), dst[:n].copy()


def transform_image(labels, src, dst, mask=None, fill_missing=False):
    out = np.ascontiguousarray(labels, dtype=np.uint32).copy()
    src = np.ascontiguousarray(src, dtype=np.uint32); dst = np.ascontiguousarray(dst, dtype=np.uint32)
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint32)
    lib().orc_transform_image(_p(out), C.c_int64(out.size), _p(src), _p(dst), C.c_int64(len(src)),
                              _p(m) if m is not None else None, C.c_int(1 if fill_missing else 0))
    return out


def relabel_image(labels, min_size=0):
    out = np.ascontiguousarray(labels, dtype=np.uint32).copy()
    n = lib().orc_relabel_image(_p(out), C.c_int64(out.size), C.c_int64(min_size))
    return out, int(n)


def watershed(img, level):
    """morphological watershed labels (1..n) of a float32 image, see orc_watershed"""
    img = np.ascontiguousarray(img, dtype=np.float32)
    dim, d = _dims(img.shape)
    out = np.empty(img.shape, np.uint32)
    lib().orc_watershed.restype = C.c_int64
    n = lib().orc_watershed(C.c_int(dim), _p(d), _p(img), C.c_double(level), _p(out))
    return out, int(n)


def stats_case(a, b):
    """entropy(a), entropy(b), distL1, distX2, amedian(a), amedian(b) of the util/stats.hxx restatements"""
    a = np.ascontiguousarray(a, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
    out = np.empty(6)
    lib().orc_stats_case(C.c_int(a.size), _p(a), _p(b), _p(out))
    return out


def rescale(feat, mn, mx, out_min=-1.0, out_max=1.0):
    f = np.ascontiguousarray(feat, dtype=np.float64).copy()
    mn = np.ascontiguousarray(mn, dtype=np.float64); mx = np.ascontiguousarray(mx, dtype=np.float64)
    lib().orc_rescale(C.c_int(f.size), _p(f), _p(mn), _p(mx), C.c_double(out_min), C.c_double(out_max))
    return f


def libm_eval(function, x):
    """std::log2 (0) / std::log (1) / std::pow(x, 1.5) (2) of the host libm, as the reference's features call them."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    lib().orc_libm_eval(C.c_int(function), _p(x), _p(out), C.c_int64(x.size))
    return out


def _tree_potentials(L, prefix, order, merge_probs, region_probs, ptr):
    order = np.ascontiguousarray(order, dtype=np.uint32)
    cap = 3 * len(order) + 1
    lab = np.empty(cap, np.uint32); par = np.empty(cap, np.int32); c0 = np.empty(cap, np.int32); c1 = np.empty(cap, np.int32)
    pot = np.empty(cap, np.float64)
    mp = None if merge_probs is None else np.ascontiguousarray(merge_probs, dtype=np.float64)
    rp = None if region_probs is None else np.ascontiguousarray(region_probs, dtype=np.float64)
    f = getattr(L, prefix + "tree_potentials"); f.restype = C.c_int64
    n = f(ptr(order), C.c_int64(len(order)), ptr(mp) if mp is not None else None, ptr(rp) if rp is not None else None,
          ptr(lab), ptr(par), ptr(c0), ptr(c1), ptr(pot), C.c_int64(cap))
    assert n >= 0
    return lab[:n].copy(), par[:n].copy(), c0[:n].copy(), c1[:n].copy(), pot[:n].copy()


def _resolve(L, prefix, par, c0, c1, pot, ptr):
    par = np.ascontiguousarray(par, np.int32); c0 = np.ascontiguousarray(c0, np.int32); c1 = np.ascontiguousarray(c1, np.int32)
    pot = np.ascontiguousarray(pot, np.float64)
    picks = np.empty(max(len(par), 1), np.int32)
    f = getattr(L, prefix + "resolve_tree_greedy"); f.restype = C.c_int64
    n = f(ptr(par), ptr(c0), ptr(c1), ptr(pot), C.c_int64(len(par)), ptr(picks), C.c_int64(len(picks)))
    assert n >= 0
    return picks[:n].copy()


def _label_transform(L, prefix, lab, c0, c1, picks, key, ptr):
    lab = np.ascontiguousarray(lab, np.uint32); c0 = np.ascontiguousarray(c0, np.int32); c1 = np.ascontiguousarray(c1, np.int32)
    picks = np.ascontiguousarray(picks, np.int32)
    src = np.empty(max(len(lab), 1), np.uint32); dst = np.empty(max(len(lab), 1), np.uint32)
    f = getattr(L, prefix + "label_transform"); f.restype = C.c_int64
    n = f(ptr(lab), ptr(c0), ptr(c1), C.c_int64(len(lab)), ptr(picks), C.c_int64(len(picks)), C.c_uint32(key), ptr(src), ptr(dst),
          C.c_int64(len(src)))
    assert n >= 0
    o = np.argsort(src[:n], kind="stable")
    return src[:n][o].copy(), dst[:n][o].copy()


def tree_potentials(order, merge_probs=None, region_probs=None):
    return _tree_potentials(lib(), "orc_", order, merge_probs, region_probs, _p)


def resolve_tree_greedy(par, c0, c1, pot):
    return _resolve(lib(), "orc_", par, c0, c1, pot, _p)


def label_transform(lab, c0, c1, picks, key=1):
    return _label_transform(lib(), "orc_", lab, c0, c1, picks, key, _p)


def _resolve_trees(L, prefix, trees, ptr):
    """trees: list of (label, parent, child0, child1, potential)"""
    nt = len(trees)
    keep = [[np.ascontiguousarray(t[0], np.uint32), np.ascontiguousarray(t[1], np.int32), np.ascontiguousarray(t[2], np.int32),
             np.ascontiguousarray(t[3], np.int32), np.ascontiguousarray(t[4], np.float64)] for t in trees]
    nn = (C.c_int64 * nt)(*[len(k[0]) for k in keep])
    arr = lambda j: (C.c_void_p * nt)(*[k[j].ctypes.data for k in keep])
    tot = sum(len(k[0]) for k in keep)
    pt = np.empty(max(tot, 1), np.int32); pn = np.empty(max(tot, 1), np.int32)
    f = getattr(L, prefix + "resolve_trees_greedy"); f.restype = C.c_int64
    n = f(C.c_int(nt), nn, arr(0), arr(1), arr(2), arr(3), arr(4), ptr(pt), ptr(pn), C.c_int64(len(pt)))
    assert n >= 0
    return pt[:n].copy(), pn[:n].copy()


def resolve_trees_greedy(trees):
    return _resolve_trees(lib(), "orc_", trees, _p)
