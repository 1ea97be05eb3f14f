"""Synthetic random forest for benchmarks / smoke tests, written in GLIA's binary model format
(ml/rf/ml_rf_model.cxx:378-455).  There is no network for trained models, so bench.py scores edges with a
random-init forest of the reference's default size (ntree = 255, ml/rf/main_train_rf.cxx:11) whose trees split on the
boundary-mean / |delta mean| / area / boundary-length features."""
import struct

import numpy as np

_HDR = 520
_OFF = dict(nrnodes=128, ntree=132, n_xbestsplit=144, n_classwt=160, n_cutoff=176, n_treemap=192, n_nodestatus=208,
            n_nodeclass=224, n_bestvar=240, n_ndbigtree=256, mtry=264, n_orig_labels=280, n_new_labels=296, nclass=304)


def synthetic_forest(ntree=255, max_depth=12, dim=3, n_thr=3, seed=1234):
    """Each tree votes 'merge' when the shared-boundary mean pb is below a per-tree cut, refined by noise splits on
    other features.  Feature indices follow bc_feat.hxx:232-238 for one --rbi image: BoundaryFeats first."""
    rng = np.random.default_rng(seed)
    bf = 11 + 4 * n_thr + 7 + 5
    rf = 4 + dim + 2 * n_thr + 5 + 5
    i_bmean = 11 + 4 * n_thr + 7 + 1          # mean of pb over the shared boundary
    i_dmean = 11 + 4 * n_thr + 3              # |delta mean| of the two regions
    i_blen = 6                                # boundary length
    i_area0 = bf                              # area of the smaller region
    i_bstd = i_bmean + 1
    noise = [(i_dmean, 0.0, 0.25), (i_blen, 4.0, 400.0), (i_area0, 50.0, 20000.0), (i_bstd, 0.02, 0.3), (i_bmean, 0.15, 0.85)]
    trees = []
    for _ in range(ntree):
        nodes = []

        def grow(depth, vote):
            k = len(nodes)
            nodes.append(None)
            if depth >= max_depth or (depth >= 2 and rng.random() < 0.25):
                nodes[k] = dict(status=-1, var=0, split=0.0, left=0, right=0, cls=vote)
                return k
            if vote == 0:       # undecided: cut on the boundary mean
                cut = float(rng.uniform(0.15, 0.85))
                l = grow(depth + 1, 1)      # low pb -> merge (class 1 = label -1)
                r = grow(depth + 1, 2)
                nodes[k] = dict(status=1, var=i_bmean + 1, split=cut, left=l + 1, right=r + 1, cls=0)
            else:
                v, lo, hi = noise[int(rng.integers(0, len(noise)))]
                flip = rng.random() < 0.1
                l = grow(depth + 1, vote)
                r = grow(depth + 1, (3 - vote) if flip else vote)
                nodes[k] = dict(status=1, var=v + 1, split=float(rng.uniform(lo, hi)), left=l + 1, right=r + 1, cls=0)
            return k

        grow(0, 0)
        for nd in nodes:
            if nd["status"] == -1 and nd["cls"] == 0:
                nd["cls"] = int(rng.integers(1, 3))
        trees.append(nodes)
    nrnodes = max(len(t) for t in trees)
    out = dict(xbestsplit=np.zeros((ntree, nrnodes)), treemap=np.zeros((ntree, nrnodes, 2), np.int32),
               nodestatus=np.zeros((ntree, nrnodes), np.int32), nodeclass=np.zeros((ntree, nrnodes), np.int32),
               bestvar=np.zeros((ntree, nrnodes), np.int32), ndbigtree=np.array([len(t) for t in trees], np.int32),
               orig_labels=np.array([-1, 1], np.int32))
    for j, t in enumerate(trees):
        for k, nd in enumerate(t):
            out["xbestsplit"][j, k] = nd["split"]
            out["treemap"][j, k] = (nd["left"], nd["right"])
            out["nodestatus"][j, k] = nd["status"]
            out["nodeclass"][j, k] = nd["cls"]
            out["bestvar"][j, k] = nd["var"]
    return out


def _arr(f, a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype).reshape(-1)
    if a.size == 0:
        return
    if a.size > 128:
        f.write(struct.pack("<B", 0))      # dense
    f.write(a.tobytes())


def write_model(path, forest):
    ntree, nrnodes = forest["xbestsplit"].shape
    nclass = len(forest["orig_labels"])
    hdr = bytearray(_HDR)

    def n2(name, a, b):
        struct.pack_into("<ii", hdr, _OFF[name], a, b)

    n2("n_xbestsplit", nrnodes, ntree); n2("n_classwt", nclass, 1); n2("n_cutoff", nclass, 1)
    n2("n_treemap", nrnodes, 2 * ntree); n2("n_nodestatus", nrnodes, ntree); n2("n_nodeclass", nrnodes, ntree)
    n2("n_bestvar", nrnodes, ntree); n2("n_ndbigtree", ntree, 1); n2("n_orig_labels", nclass, 1)
    n2("n_new_labels", nclass, 1)
    struct.pack_into("<i", hdr, _OFF["nrnodes"], nrnodes); struct.pack_into("<i", hdr, _OFF["ntree"], ntree)
    struct.pack_into("<i", hdr, _OFF["mtry"], 3); struct.pack_into("<i", hdr, _OFF["nclass"], nclass)

    def filemat(mem, n0, n1):   # memory (node fastest) -> file (row-major n0 x n1); undone by the reader's transpose
        return np.ascontiguousarray(np.asarray(mem).reshape(n1, n0).T)

    with open(path, "wb") as f:
        f.write(bytes(hdr))
        f.write(struct.pack("<ii", nrnodes, ntree))
        _arr(f, filemat(forest["xbestsplit"], nrnodes, ntree), np.float64)
        _arr(f, np.ones(nclass), np.float64)
        _arr(f, np.full(nclass, 1.0 / nclass), np.float64)
        _arr(f, filemat(forest["treemap"], nrnodes, 2 * ntree), np.int32)
        _arr(f, filemat(forest["nodestatus"], nrnodes, ntree), np.int32)
        _arr(f, filemat(forest["nodeclass"], nrnodes, ntree), np.int32)
        _arr(f, filemat(forest["bestvar"], nrnodes, ntree), np.int32)
        _arr(f, forest["ndbigtree"], np.int32)
        f.write(struct.pack("<i", 3))
        _arr(f, forest["orig_labels"], np.int32)
        _arr(f, np.arange(1, nclass + 1), np.int32)
        f.write(struct.pack("<i", nclass))
