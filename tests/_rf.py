"""Test helpers: synthetic random forests in GLIA's binary model format (ml/rf/ml_rf_model.cxx:378-455)."""
import struct

import numpy as np

HDR = 520
OFF = dict(n_ncat=104, n_catf=120, nrnodes=128, ntree=132, n_xbestsplit=144, n_classwt=160, n_cutoff=176, n_treemap=192,
           n_nodestatus=208, n_nodeclass=224, n_bestvar=240, n_ndbigtree=256, mtry=264, n_orig_labels=280,
           n_new_labels=296, nclass=304)


def random_forest(rng, ntree, max_depth, feat_values, nclass=2, p_leaf=0.15):
    """feat_values: [n_samples, D] array; split thresholds are midpoints of neighbouring observed values, so
    splits are informative.  Returns arrays in the in-memory layout classForest walks."""
    D = feat_values.shape[1]
    trees = []
    for _ in range(ntree):
        nodes = [dict(depth=0)]
        k = 0
        while k < len(nodes):
            nd = nodes[k]
            if nd["depth"] >= max_depth or (nd["depth"] > 0 and rng.random() < p_leaf):
                nd.update(status=-1, var=0, split=0.0, left=0, right=0, cls=int(rng.integers(1, nclass + 1)))
            else:
                var = int(rng.integers(0, D))
                # like randomForest's split search, a split sits midway between two observed values, never on one:
                # entropy (log2) and compactness (pow) agree with glibc only to the last ulp on the device
                col = np.unique(feat_values[:, var])
                if len(col) > 1:
                    i = int(rng.integers(0, len(col) - 1))
                    thr = float((col[i] + col[i + 1]) / 2.0)
                else:
                    thr = float(col[0]) + 0.5
                nd.update(status=1, var=var + 1, split=thr, left=len(nodes) + 1, right=len(nodes) + 2, cls=0)
                nodes.append(dict(depth=nd["depth"] + 1))
                nodes.append(dict(depth=nd["depth"] + 1))
            k += 1
        trees.append(nodes)
    nrnodes = max(len(t) for t in trees)
    out = dict(xbestsplit=np.zeros((ntree, nrnodes)), treemap=np.zeros((ntree, nrnodes, 2), np.int32),
               nodestatus=np.zeros((ntree, nrnodes), np.int32), nodeclass=np.zeros((ntree, nrnodes), np.int32),
               bestvar=np.zeros((ntree, nrnodes), np.int32), ndbigtree=np.array([len(t) for t in trees], np.int32),
               orig_labels=np.array([-1, 1][:nclass] + list(range(2, nclass)), np.int32))
    for j, t in enumerate(trees):
        for k, nd in enumerate(t):
            out["xbestsplit"][j, k] = nd["split"]
            out["treemap"][j, k] = (nd["left"], nd["right"])
            out["nodestatus"][j, k] = nd["status"]
            out["nodeclass"][j, k] = nd["cls"]
            out["bestvar"][j, k] = nd["var"]
    return out


def _arr(f, a, dtype, sparse=False):
    a = np.ascontiguousarray(a, dtype=dtype).reshape(-1)
    if a.size == 0:
        return
    if a.size > 128:
        nz = np.flatnonzero(np.abs(a.astype(np.float64)) > 1e-8)
        use_sparse = sparse and len(nz) < a.size // 2          # the reference writer's rule (:11-18)
        f.write(struct.pack("<B", 1 if use_sparse else 0))
        if use_sparse:
            f.write(struct.pack("<i", len(nz)))
            for i in nz:
                f.write(struct.pack("<i", int(i)))
                f.write(a[i:i + 1].tobytes())
            return
    f.write(a.tobytes())


def write_model(path, forest, sparse=False):
    """forest arrays are in memory layout; the file stores each [n0][n1] array row-major such that the reader's
    transpose (ml_rf_model.cxx:542-557) restores that memory."""
    ntree, nrnodes = forest["xbestsplit"].shape
    nclass = len(forest["orig_labels"])
    hdr = bytearray(HDR)

    def n2(name, a, b):
        struct.pack_into("<ii", hdr, OFF[name], a, b)

    n2("n_xbestsplit", nrnodes, ntree); n2("n_classwt", nclass, 1); n2("n_cutoff", nclass, 1)
    n2("n_treemap", nrnodes, 2 * ntree); n2("n_nodestatus", nrnodes, ntree); n2("n_nodeclass", nrnodes, ntree)
    n2("n_bestvar", nrnodes, ntree); n2("n_ndbigtree", ntree, 1); n2("n_orig_labels", nclass, 1)
    n2("n_new_labels", nclass, 1)
    struct.pack_into("<i", hdr, OFF["nrnodes"], nrnodes); struct.pack_into("<i", hdr, OFF["ntree"], ntree)
    struct.pack_into("<i", hdr, OFF["mtry"], 3); struct.pack_into("<i", hdr, OFF["nclass"], nclass)

    def filemat(mem, n0, n1):   # memory (column-major n0 x n1) -> file (row-major n0 x n1)
        return np.ascontiguousarray(np.asarray(mem).reshape(n1, n0).T)

    with open(path, "wb") as f:
        f.write(bytes(hdr))
        f.write(struct.pack("<ii", nrnodes, ntree))
        _arr(f, filemat(forest["xbestsplit"], nrnodes, ntree), np.float64, sparse)
        _arr(f, np.ones(nclass), np.float64)
        _arr(f, np.full(nclass, 1.0 / nclass), np.float64)
        _arr(f, filemat(forest["treemap"], nrnodes, 2 * ntree), np.int32, sparse)
        _arr(f, filemat(forest["nodestatus"], nrnodes, ntree), np.int32, sparse)
        _arr(f, filemat(forest["nodeclass"], nrnodes, ntree), np.int32, sparse)
        _arr(f, filemat(forest["bestvar"], nrnodes, ntree), np.int32, sparse)
        _arr(f, forest["ndbigtree"], np.int32)     # [ntree][1]: its transpose is the same memory
        f.write(struct.pack("<i", 3))
        _arr(f, forest["orig_labels"], np.int32)
        _arr(f, np.arange(1, nclass + 1), np.int32)
        f.write(struct.pack("<i", nclass))
