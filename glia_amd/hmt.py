"""Host-side mirror of GLIA's HMT operators over the C ABI (include/glia_hmt.h).

Names follow the reference: RegionMap ~ TRegionMap(image, mask, onlyContour) (type/region_map.hxx:38-40),
merge_order_pb ~ hmt/main_merge_order_pb.cxx.  Volumes are torch CUDA tensors (device memory plumbing
only) in numpy axis order (z, y, x).  No CPU path exists here.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("GLIA_HMT_LIB", os.path.join(_HERE, "libglia_hmt.so"))   # override: kernel experiments only

MAX_IMAGES, MAX_BINS, MAX_THRESH = 8, 16, 4


class HmtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("glia_hmt error %d: %s" % (code, msg))
        self.code = code


class Image(C.Structure):
    _fields_ = [("d_image", C.c_void_p), ("bins", C.c_int), ("lo", C.c_double), ("hi", C.c_double)]


class FeatConfig(C.Structure):
    _fields_ = [
        ("n_region", C.c_int), ("region", Image * MAX_IMAGES),
        ("n_rlabel", C.c_int), ("rlabel", Image * MAX_IMAGES),
        ("n_boundary", C.c_int), ("boundary", Image * MAX_IMAGES),
        ("d_pb", C.c_void_p), ("n_thresholds", C.c_int), ("thresholds", C.c_double * MAX_THRESH),
        ("normalizing_area", C.c_double), ("normalizing_length", C.c_double),
        ("use_log_shape", C.c_int), ("use_simple_features", C.c_int),
        ("use_histogram_features", C.c_int), ("use_median_features", C.c_int),
    ]


_lib = None


def lib():
    """Loads libglia_hmt.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise ImportError("glia_amd/libglia_hmt.so is missing: run `python -c 'import __graft_entry__ as g; "
                              "g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback")
        # torch wheels bundle their own HIP runtime: when torch is used in the process (tests, bench.py) it has to
        # be loaded first so that both sides share ONE runtime; a plain C host simply gets /opt/rocm's.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(_SO)
        L.glia_hmt_last_error.restype = C.c_char_p
        L.glia_hmt_version.restype = C.c_char_p
        L.glia_hmt_rag_num_regions.restype = C.c_int64
        L.glia_hmt_rag_num_pairs.restype = C.c_int64
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise HmtError(rc, lib().glia_hmt_last_error().decode())


ERR_ARG, ERR_INTERNAL = -1, -7


def set_option(key, value):
    """glia_hmt_set_option: a process-wide tuning / test switch of the loops (value None unsets it)."""
    _check(lib().glia_hmt_set_option(key.encode(), None if value is None else str(value).encode()))


class options:
    """with options(GLIA_HMT_WINCAP=32, ...): the switches are set inside the block and unset afterwards (no result depends on
    them; the queue tests use them to drive the window queue through its rare paths)."""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        for k, v in self.kv.items():
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.kv:
            set_option(k, None)
        return False


def check_merge_order(dense_order, n_regions):
    """glia_hmt_check_merge_order on a DENSE-id order ([n][3] uint32): -1 when every merge joins two regions that still exist and
    creates region n_regions + k, else the first merge that does not."""
    o = np.ascontiguousarray(dense_order, dtype=np.uint32).reshape(-1, 3)
    bad = C.c_int64(-1)
    rc = lib().glia_hmt_check_merge_order(o.ctypes.data_as(C.c_void_p), C.c_int64(len(o)), C.c_int64(int(n_regions)), C.byref(bad))
    if rc not in (0, ERR_ARG):
        _check(rc)
    return int(bad.value)


def _dims(shape):
    dim = len(shape)
    d = (C.c_int64 * 3)(1, 1, 1)
    for i, n in enumerate(shape[::-1]):
        d[i] = n
    return dim, d


def _np(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _fence(t):
    """The library works on its own HIP stream: finish whatever torch has queued on the tensor's device first
    (a clone() or copy_ producing the input), so the kernels below see the data."""
    import torch
    torch.cuda.current_stream(t.device).synchronize()


class Context:
    def __init__(self, device=0, stream=None):
        self.h = C.c_void_p()
        _check(lib().glia_hmt_ctx_create(C.c_int(device), C.c_void_p(stream) if stream else None, C.byref(self.h)))
        self.device = device
        if lib().glia_hmt_ctx_libm_status(self.h) == 0:
            import warnings
            warnings.warn(lib().glia_hmt_last_error().decode())

    def close(self):
        if self.h:
            lib().glia_hmt_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _check(lib().glia_hmt_ctx_sync(self.h))

    def libm(self):
        """(log2 variant, log variant) the kernels use: 1 = glibc non-FMA, 2 = glibc FMA, 0 = device libm (unpinned)."""
        a, b = C.c_int(0), C.c_int(0)
        _check(lib().glia_hmt_ctx_libm(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    @staticmethod
    def release_cached_memory():
        """Scratch blocks the library parks for reuse go back to the driver (glia_hmt_release_cached_memory); returns the bytes released."""
        f = lib().glia_hmt_release_cached_memory
        f.restype = C.c_ulonglong
        return int(f())

    @staticmethod
    def internal_errors():
        """Calls of this process that ended with GLIA_HMT_ERR_INTERNAL -- a merge loop's own consistency stop or an order that
        failed the replay of glia_hmt_check_merge_order (glia_hmt_internal_errors).  0 is the only healthy answer."""
        f = lib().glia_hmt_internal_errors
        f.restype = C.c_ulonglong
        return int(f())

    def libm_pinned(self):
        """True when log2, log and pow of the host libm are all reproduced bit for bit (glia_hmt_ctx_libm_status)."""
        return lib().glia_hmt_ctx_libm_status(self.h) == 1

    def libm_pow(self):
        """variant of std::pow(perim, 1.5) the kernels use (same codes as libm())"""
        a = C.c_int(0)
        _check(lib().glia_hmt_ctx_libm_pow(self.h, C.byref(a)))
        return a.value

    def libm_eval(self, function, variant, x):
        """Device restatement of the host libm over a CUDA f64 tensor: function 0 = log2, 1 = log, 2 = pow(x, 1.5)."""
        import torch
        out = torch.empty_like(x)
        _check(lib().glia_hmt_libm_eval(self.h, C.c_int(function), C.c_int(variant), C.c_void_p(x.data_ptr()),
                                        C.c_void_p(out.data_ptr()), C.c_int64(x.numel())))
        self.sync()
        return out

    def watershed(self, image, level):
        """glia::watershed (util/image_alg.hxx:9-21) of a CUDA float32 tensor -> (labels int32 tensor holding uint32 1..n, n, sweeps)."""
        import torch
        assert image.is_cuda and image.dtype == torch.float32 and image.is_contiguous()
        dim, d = _dims(tuple(image.shape))
        out = torch.empty(image.shape, dtype=torch.int32, device=image.device)
        n, sw = C.c_uint32(0), C.c_int(0)
        _fence(image)
        _check(lib().glia_hmt_watershed(self.h, C.c_int(dim), d, C.c_void_p(image.data_ptr()), C.c_double(level), C.c_void_p(out.data_ptr()),
                                        C.byref(n), C.byref(sw)))
        return out, n.value, sw.value

    def set_table_hint(self, regions, pairs):
        _check(lib().glia_hmt_ctx_set_table_hint(self.h, C.c_int64(regions), C.c_int64(pairs)))

    def synth(self, shape, S, G, seed=0x9E3779B97F4A7C15, variant=0):
        """Synthetic supervoxels + pb on the device; returns torch tensors (labels i32 view, pb)."""
        import torch
        dim, d = _dims(shape)
        dev = torch.device("cuda", self.device)
        labels = torch.empty(shape, dtype=torch.int32, device=dev)   # uint32 payload
        pb = torch.empty(shape, dtype=torch.float32, device=dev)
        _check(lib().glia_hmt_synth(self.h, C.c_int(dim), d, C.c_int(S), C.c_int(G), C.c_uint64(seed),
                                    C.c_int(variant), C.c_void_p(labels.data_ptr()), C.c_void_p(pb.data_ptr())))
        return labels, pb


def make_config(pb, rb=(), r=(), rl=(), b=(), thresholds=(0.2, 0.5, 0.8), normalizing_area=1.0,
                normalizing_length=1.0, use_log_shape=False, use_simple_features=False, use_histogram_features=False,
                use_median_features=False):
    """Builds the image lists the way prepareImages does (hmt/hmt_util.hxx:17-56).
    rb/r/rl/b: sequences of (device tensor, bins, lo, hi)."""
    cfg = FeatConfig()
    keep = [pb]

    def fill(arr, lst):
        for i, (img, bins, lo, hi) in enumerate(lst):
            keep.append(img)
            arr[i].d_image = img.data_ptr()
            arr[i].bins, arr[i].lo, arr[i].hi = bins, lo, hi
        return len(lst)

    cfg.n_region = fill(cfg.region, list(rb) + list(r))
    cfg.n_rlabel = fill(cfg.rlabel, list(rl))
    cfg.n_boundary = fill(cfg.boundary, list(rb) + list(b))
    cfg.d_pb = pb.data_ptr()
    cfg.n_thresholds = len(thresholds)
    for i, t in enumerate(thresholds):
        cfg.thresholds[i] = t
    cfg.normalizing_area, cfg.normalizing_length = normalizing_area, normalizing_length
    cfg.use_log_shape, cfg.use_simple_features = int(use_log_shape), int(use_simple_features)
    cfg.use_histogram_features, cfg.use_median_features = int(use_histogram_features), int(use_median_features)
    cfg._keep = keep
    return cfg


def gen_tree(order):
    """hmt::genTree (hmt/tree_build.hxx:12-38) -> (label, parent, child0, child1) arrays."""
    order = np.ascontiguousarray(order, dtype=np.uint32)
    cap = 3 * len(order) + 1      # a forest of several components has more than 2n+1 nodes
    lab = np.empty(cap, np.uint32)
    par = np.empty(cap, np.int32); c0 = np.empty(cap, np.int32); c1 = np.empty(cap, np.int32)
    lib().glia_hmt_gen_tree.restype = C.c_int64
    n = lib().glia_hmt_gen_tree(_np(order), C.c_int64(len(order)), _np(lab), _np(par), _np(c0), _np(c1), C.c_int64(cap))
    if n < 0:
        raise HmtError(int(n), lib().glia_hmt_last_error().decode())
    return lab[:n], par[:n], c0[:n], c1[:n]


def transform_keys(order):
    """transformKeys (util/struct_merge.hxx:188-210): merge order -> (src, dst) label map, sorted by src."""
    order = np.ascontiguousarray(order, dtype=np.uint32)
    cap = 2 * len(order) + 1
    src = np.empty(cap, np.uint32); dst = np.empty(cap, np.uint32)
    lib().glia_hmt_transform_keys.restype = C.c_int64
    n = lib().glia_hmt_transform_keys(_np(order), C.c_int64(len(order)), _np(src), _np(dst), C.c_int64(cap))
    if n < 0:
        raise HmtError(int(n), lib().glia_hmt_last_error().decode())
    return src[:n].copy(), dst[:n].copy()


def transform_image(ctx, labels, src, dst, mask=None, fill_missing=False):
    """transformImage (util/image.hxx:227-242): rewrites the CUDA label tensor in place."""
    src = np.ascontiguousarray(src, dtype=np.uint32); dst = np.ascontiguousarray(dst, dtype=np.uint32)
    assert labels.is_cuda and labels.is_contiguous() and labels.element_size() == 4
    _fence(labels)
    _check(lib().glia_hmt_transform_image(ctx.h, C.c_void_p(labels.data_ptr()), C.c_int64(labels.numel()), _np(src), _np(dst),
                                          C.c_int64(len(src)), C.c_void_p(mask.data_ptr()) if mask is not None else None,
                                          C.c_int(1 if fill_missing else 0)))
    lib().glia_hmt_last_transform_ms.restype = C.c_double
    return lib().glia_hmt_last_transform_ms(ctx.h)


def relabel_image(ctx, labels, min_size=0):
    """relabelImage (util/image.hxx:992-1001): consecutive labels by decreasing size, in place; returns #labels."""
    assert labels.is_cuda and labels.is_contiguous() and labels.element_size() == 4
    _fence(labels)
    n = C.c_uint32(0)
    _check(lib().glia_hmt_relabel_image(ctx.h, C.c_void_p(labels.data_ptr()), C.c_int64(labels.numel()), C.c_int64(min_size),
                                        C.byref(n)))
    return n.value


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
    """genTree / genTreeWithNodePotentials (hmt/tree_build.hxx:12-63) -> (label, parent, child0, child1, potential)."""
    return _tree_potentials(lib(), "glia_hmt_", order, merge_probs, region_probs, _np)


def resolve_tree_greedy(parent, child0, child1, potential):
    """resolveTreeGreedy (hmt/tree_greedy.hxx:104-152, one tree): node indices in pick order."""
    return _resolve(lib(), "glia_hmt_", parent, child0, child1, potential, _np)


def label_transform(node_label, child0, child1, picks, key_to_assign=1):
    """genLabelTransform (hmt/tree_segment.hxx:10-21) -> (src, dst) sorted by src."""
    return _label_transform(lib(), "glia_hmt_", node_label, child0, child1, picks, key_to_assign, _np)


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
    """resolveTreeGreedy over several trees (hmt/tree_greedy.hxx:104-152) -> (tree index, node index) per pick."""
    return _resolve_trees(lib(), "glia_hmt_", trees, _np)


class RandomForest:
    """alg::RandomForest / alg::EnsembleRandomForest (alg/rf.hxx) loaded from GLIA model files onto the device."""

    def __init__(self, ctx, model_files, predict_label=-1, distributor_args=None):
        if isinstance(model_files, str):
            model_files = [model_files]
        arr = (C.c_char_p * len(model_files))(*[m.encode() for m in model_files])
        dist = (C.c_double * 3)(*distributor_args) if distributor_args is not None else None
        self.h = C.c_void_p()
        _check(lib().glia_hmt_forest_load(ctx.h, C.c_int(len(model_files)), arr, C.c_int(predict_label), dist,
                                          C.byref(self.h)))

    def close(self):
        if self.h:
            lib().glia_hmt_forest_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FeatureStubClassifier(RandomForest):
    """Diagnostic scorer P(merge) = 1 - x[index] (tests; SURVEY.md Appendix D recipe P4)."""

    def __init__(self, ctx, index):
        self.h = C.c_void_p()
        _check(lib().glia_hmt_forest_stub(ctx.h, C.c_int(index), C.byref(self.h)))


class Slab(C.Structure):
    """glia_hmt_slab (include/glia_hmt.h)"""
    _fields_ = [("dims_local", C.c_int64 * 3), ("z_global_of_plane0", C.c_int64), ("z_begin", C.c_int64), ("z_end", C.c_int64),
                ("d_labels", C.c_void_p), ("d_pb", C.c_void_p), ("cfg", C.POINTER(FeatConfig))]


class DistStats(C.Structure):
    _fields_ = [("records", C.c_uint64), ("cut_records", C.c_uint64), ("bytes_cut_exchange", C.c_uint64), ("bytes_to_loop_owner", C.c_uint64)]


class Comm:
    """glia_hmt_comm: the ranks of this process in the slab route (glia_amd/csrc/slab_dist.cpp)."""

    def __init__(self, ctx, world, rank=None, unique_id=None):
        """rank None: all `world` ranks in this process on ctx's GPU (device copies); else one RCCL rank (unique_id: 128 bytes
        from Comm.unique_id() on one rank)."""
        self.ctx, self.world = ctx, world
        self.h = C.c_void_p()
        if rank is None:
            _check(lib().glia_hmt_comm_create_local(ctx.h, C.c_int(world), C.byref(self.h)))
            self.local = list(range(world))
        else:
            buf = (C.c_char * 128).from_buffer_copy(unique_id)
            _check(lib().glia_hmt_comm_create_rccl(ctx.h, C.c_int(world), C.c_int(rank), buf, C.byref(self.h)))
            self.local = [rank]

    @staticmethod
    def unique_id():
        buf = (C.c_char * 128)()
        _check(lib().glia_hmt_comm_unique_id(buf))
        return bytes(buf)

    def close(self):
        if self.h:
            lib().glia_hmt_comm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def slab_range(nz, world, rank):
    """glia_hmt_slab_range -> (first plane handed in, planes handed in, z_begin, z_end)"""
    a, b, c, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    _check(lib().glia_hmt_slab_range(C.c_int64(nz), C.c_int(world), C.c_int(rank), C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
    return a.value, b.value, c.value, d.value


def build_distributed(ctx, comm, slabs, nz_global, only_contour=False, loop_owner=0, with_values=False):
    """glia_hmt_rag_build_distributed.  slabs: one (labels, pb, first_plane, z_begin, z_end, cfg) per LOCAL rank of comm, in rank
    order (labels / pb: CUDA tensors of the planes handed in).  Returns (RegionMap of the whole volume | None, DistStats)."""
    arr = (Slab * len(slabs))()
    keep = []
    for i, (lab, pb, first, zb, ze, cfg) in enumerate(slabs):
        assert lab.is_cuda and lab.is_contiguous() and lab.dim() == 3
        _fence(lab)
        arr[i].dims_local[0], arr[i].dims_local[1], arr[i].dims_local[2] = lab.shape[2], lab.shape[1], lab.shape[0]
        arr[i].z_global_of_plane0, arr[i].z_begin, arr[i].z_end = first, zb, ze
        arr[i].d_labels = lab.data_ptr()
        arr[i].d_pb = pb.data_ptr() if pb is not None else None
        arr[i].cfg = C.pointer(cfg) if cfg is not None else None
        keep.append((lab, pb, cfg))
    h = C.c_void_p()
    st = DistStats()
    _check(lib().glia_hmt_rag_build_distributed(ctx.h, comm.h, arr, C.c_int64(nz_global), C.c_int(int(only_contour)), C.c_int(int(with_values)), C.c_int(loop_owner),
                                                C.byref(h), C.byref(st)))
    rm = RegionMap(ctx, None, cfg=slabs[0][5], _handle=h) if h else None
    if rm is not None:
        rm._keep = keep
    return rm, st


class RegionMap:
    """Device-resident region adjacency structure with sufficient statistics.
    Mirrors TRegionMap(image, mask, onlyContour) (type/region_map.hxx:38-40)."""

    def __init__(self, ctx, labels, pb=None, mask=None, only_contour=False, cfg=None, slab=None, _handle=None):
        """slab = (z_global_of_plane0, nz_global, z_begin, z_end): build the PARTIAL map of a z-slab whose planes
        (plus halo planes) are `labels` / `pb` (see include/glia_hmt.h, glia_hmt_rag_build_slab)."""
        self.ctx = ctx
        self.cfg = cfg
        self._keep = (labels, pb, mask, cfg)
        self.h = C.c_void_p()
        if _handle is not None:
            self.h = _handle
        else:
            assert labels.is_cuda and labels.is_contiguous() and labels.element_size() == 4
            _fence(labels)
            self.shape = tuple(labels.shape)
            self.dim, d = _dims(self.shape)
            if slab is None:
                _check(lib().glia_hmt_rag_build(
                    ctx.h, C.c_int(self.dim), d, C.c_void_p(labels.data_ptr()),
                    C.c_void_p(mask.data_ptr()) if mask is not None else None, C.c_int(int(only_contour)),
                    C.c_void_p(pb.data_ptr()) if pb is not None else None,
                    C.byref(cfg) if cfg is not None else None, C.byref(self.h)))
            else:
                gz0, gnz, zb, ze = slab
                _check(lib().glia_hmt_rag_build_slab(
                    ctx.h, d, C.c_int64(gz0), C.c_int64(gnz), C.c_int64(zb), C.c_int64(ze), C.c_void_p(labels.data_ptr()),
                    C.c_int(int(only_contour)), C.c_void_p(pb.data_ptr()) if pb is not None else None,
                    C.byref(cfg) if cfg is not None else None, C.byref(self.h)))
        self.bins = cfg.region[0].bins if cfg is not None and cfg.n_region else \
            (cfg.boundary[0].bins if cfg is not None and cfg.n_boundary else 8)
        self.nthr = cfg.n_thresholds if cfg is not None else 0

    def close(self):
        if self.h:
            lib().glia_hmt_rag_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_regions(self):
        return lib().glia_hmt_rag_num_regions(self.h)

    @property
    def num_pairs(self):
        return lib().glia_hmt_rag_num_pairs(self.h)

    def num_channels(self):
        return lib().glia_hmt_rag_num_channels(self.h)

    @staticmethod
    def merge(ctx, parts):
        """glia_hmt_rag_merge: combine the partial maps of several slabs (same configuration)."""
        arr = (C.c_void_p * len(parts))(*[p.h for p in parts])
        h = C.c_void_p()
        _check(lib().glia_hmt_rag_merge(ctx.h, arr, C.c_int(len(parts)), C.byref(h)))
        return RegionMap(ctx, None, cfg=parts[0].cfg, _handle=h)

    def to_tensors(self):
        """The compact record arrays as torch tensors on the context's device (for torch.distributed)."""
        import torch
        dev = torch.device("cuda", self.ctx.device)
        R, P = self.num_regions, self.num_pairs
        rw, pw = C.c_int(), C.c_int()
        _check(lib().glia_hmt_rag_device_arrays(self.h, None, None, None, None, None, C.byref(rw), C.byref(pw)))
        t = dict(rlabel=torch.empty(R, dtype=torch.int32, device=dev), rrec=torch.empty((R, rw.value), dtype=torch.int32, device=dev),
                 pa=torch.empty(P, dtype=torch.int32, device=dev), pb=torch.empty(P, dtype=torch.int32, device=dev),
                 prec=torch.empty((P, pw.value), dtype=torch.int32, device=dev))
        _check(lib().glia_hmt_rag_copy_arrays(self.h, *[C.c_void_p(t[k].data_ptr()) for k in ("rlabel", "rrec", "pa", "pb", "prec")]))
        # further image channels (feature lists with several volumes): same keys, one more record set each
        for k in range(1, lib().glia_hmt_rag_num_channels(self.h)):
            t["rrec%d" % k] = torch.empty((R, rw.value), dtype=torch.int32, device=dev)
            t["prec%d" % k] = torch.empty((P, pw.value), dtype=torch.int32, device=dev)
            _check(lib().glia_hmt_rag_copy_channel(self.h, C.c_int(k), C.c_void_p(t["rrec%d" % k].data_ptr()), C.c_void_p(t["prec%d" % k].data_ptr())))
        return t

    def cut_flags(self, labels_slab, z_begin, z_end):
        """glia_hmt_rag_cut_flags: boolean CUDA tensors (regions, pairs) -- records whose label occurs next to a cut of the slab."""
        import torch
        dev = torch.device("cuda", self.ctx.device)
        _, d = _dims(tuple(labels_slab.shape))
        rcut = torch.zeros(max(self.num_regions, 1), dtype=torch.uint8, device=dev)
        pcut = torch.zeros(max(self.num_pairs, 1), dtype=torch.uint8, device=dev)
        _fence(labels_slab)
        _check(lib().glia_hmt_rag_cut_flags(self.ctx.h, self.h, d, C.c_int64(z_begin), C.c_int64(z_end), C.c_void_p(labels_slab.data_ptr()),
                                            C.c_void_p(rcut.data_ptr()), C.c_void_p(pcut.data_ptr())))
        return rcut[:self.num_regions].bool(), pcut[:self.num_pairs].bool()

    @staticmethod
    def from_tensors(ctx, like, t):
        h = C.c_void_p()
        for k in t:
            assert t[k].is_cuda and t[k].is_contiguous()
        _fence(t["rlabel"])
        _check(lib().glia_hmt_rag_from_arrays(ctx.h, like.h, C.c_int64(t["rlabel"].numel()), C.c_void_p(t["rlabel"].data_ptr()),
                                              C.c_void_p(t["rrec"].data_ptr()), C.c_int64(t["pa"].numel()),
                                              C.c_void_p(t["pa"].data_ptr()), C.c_void_p(t["pb"].data_ptr()),
                                              C.c_void_p(t["prec"].data_ptr()), C.byref(h)))
        k = 1
        while "rrec%d" % k in t:
            _check(lib().glia_hmt_rag_add_channel(ctx.h, h, C.c_void_p(t["rrec%d" % k].data_ptr()), C.c_void_p(t["prec%d" % k].data_ptr())))
            k += 1
        return RegionMap(ctx, None, cfg=like.cfg, _handle=h)

    def last_pass(self):
        ms, by = C.c_double(), C.c_double()
        _check(lib().glia_hmt_rag_last_pass(self.h, C.byref(ms), C.byref(by)))
        return ms.value, by.value

    def regions(self):
        n = self.num_regions
        out = dict(label=np.empty(n, np.uint32), count=np.empty(n, np.int64), border=np.empty(n, np.int64),
                   lo=np.empty((n, 3), np.int64), hi=np.empty((n, 3), np.int64), sum=np.empty(n), sumsq=np.empty(n),
                   min=np.empty(n), max=np.empty(n), hist=np.empty((n, self.bins), np.int64),
                   first=np.empty(n, np.int64))
        _check(lib().glia_hmt_rag_export_regions(self.h, *[_np(out[k]) for k in (
            "label", "count", "border", "lo", "hi", "sum", "sumsq", "min", "max", "hist", "first")]))
        return out

    def pairs(self):
        n = self.num_pairs
        out = dict(a=np.empty(n, np.uint32), b=np.empty(n, np.uint32), count=np.empty(n, np.int64), sum=np.empty(n),
                   sumsq=np.empty(n), min=np.empty(n), max=np.empty(n), hist=np.empty((n, self.bins), np.int64),
                   thr=np.empty((n, max(self.nthr, 1)), np.int64))
        _check(lib().glia_hmt_rag_export_pairs(self.h, *[_np(out[k]) for k in (
            "a", "b", "count", "sum", "sumsq", "min", "max", "hist")], _np(out["thr"]) if self.nthr else None))
        return out

    def merge_order_pb(self, type=1):
        """hmt/main_merge_order_pb.cxx: type 1 = median, 2 = mean.  Returns (order[n,3] uint32, saliency[n])."""
        cap = max(self.num_regions, 1)
        order = np.empty((cap, 3), np.uint32)
        sal = np.empty(cap, np.float64)
        n = C.c_int64(0)
        _check(lib().glia_hmt_merge_order_pb(self.ctx.h, self.h, C.c_int(type), _np(order), _np(sal), C.c_int64(cap),
                                             C.byref(n)))
        return order[:n.value].copy(), sal[:n.value].copy()

    def feat_dim(self):
        return lib().glia_hmt_feat_dim(self.h)

    def merge_order_bc(self, classifier, want_feats=False):
        """hmt/main_merge_order_bc.cxx (--bct 1).  Returns (order, saliency[, feats])."""
        cap = max(self.num_regions, 1)
        order = np.empty((cap, 3), np.uint32)
        sal = np.empty(cap, np.float64)
        d = self.feat_dim()
        feats = np.empty((cap, d), np.float64) if want_feats else None
        n = C.c_int64(0)
        _check(lib().glia_hmt_merge_order_bc(self.ctx.h, self.h, classifier.h, _np(order), _np(sal), _np(feats),
                                             C.c_int64(cap), C.byref(n)))
        if want_feats:
            return order[:n.value].copy(), sal[:n.value].copy(), feats[:n.value].copy()
        return order[:n.value].copy(), sal[:n.value].copy()

    def pre_merge(self, size_thresholds, rpb_threshold):
        """gadget/main_pre_merge.cxx: pb-mean merges restricted by the region-size condition."""
        cap = max(self.num_regions, 1)
        order = np.empty((cap, 3), np.uint32)
        sal = np.empty(cap, np.float64)
        st = (C.c_int * len(size_thresholds))(*size_thresholds)
        n = C.c_int64(0)
        _check(lib().glia_hmt_pre_merge(self.ctx.h, self.h, st, C.c_int(len(size_thresholds)), C.c_double(rpb_threshold),
                                        _np(order), _np(sal), C.c_int64(cap), C.byref(n)))
        return order[:n.value].copy(), sal[:n.value].copy()

    def bc_feat(self, order, saliencies=None, init_sal=1.0, sal_bias=1.0):
        """hmt/main_bc_feat.cxx: feature rows of a given merge order; with `saliencies` (-y) also the saliency features."""
        order = np.ascontiguousarray(order, dtype=np.uint32)
        d = lib().glia_hmt_bc_feat_dim(self.h, C.c_int(1 if saliencies is not None else 0))
        feats = np.empty((len(order), d), np.float64)
        sal = None if saliencies is None else np.ascontiguousarray(saliencies, dtype=np.float64)
        _check(lib().glia_hmt_bc_feat_saliency(self.ctx.h, self.h, _np(order), C.c_int64(len(order)), _np(sal),
                                               C.c_double(init_sal), C.c_double(sal_bias), _np(feats)))
        return feats

    def score_initial_edges(self, classifier):
        n, ms = C.c_int64(0), C.c_double(0)
        _check(lib().glia_hmt_score_initial_edges(self.ctx.h, self.h, classifier.h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def score_initial_edges_shard(self, classifier, shard, n_shards):
        """scores of the initial records e with e % n_shards == shard (-inf elsewhere); max over shards = all scores"""
        cap = max(self.num_pairs, 1)
        out = np.empty(cap, np.float64)
        n = C.c_int64(0)
        _check(lib().glia_hmt_score_initial_edges_shard(self.ctx.h, self.h, classifier.h, C.c_int(shard), C.c_int(n_shards), _np(out),
                                                        C.c_int64(cap), C.byref(n)))
        return out[:n.value].copy()

    def boundary_confidence(self, trees):
        """segment_greedy -b (genBoundaryConfidenceImage, all nodes): float32 CUDA tensor of the volume's shape"""
        import torch
        nt = len(trees)
        keep = [[np.ascontiguousarray(t[0], np.uint32), np.ascontiguousarray(t[1], np.int32), np.ascontiguousarray(t[2], np.int32),
                 np.ascontiguousarray(t[4], np.float64)] for t in trees]
        nn = (C.c_int64 * nt)(*[len(k[0]) for k in keep])
        arr = lambda j: (C.c_void_p * nt)(*[k[j].ctypes.data for k in keep])
        out = torch.empty(self.shape, dtype=torch.float32, device=torch.device("cuda", self.ctx.device))
        _check(lib().glia_hmt_boundary_confidence(self.ctx.h, self.h, C.c_int(nt), nn, arr(0), arr(1), arr(2), arr(3), C.c_void_p(out.data_ptr())))
        return out

    def last_merge_timing(self):
        a, b, c, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        _check(lib().glia_hmt_last_merge_timing(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
        return dict(ms_table=a.value, ms_init=b.value, ms_loop=c.value, n_edges_scored=n.value)
