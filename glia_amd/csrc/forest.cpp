// glia_amd/csrc/forest.cpp -- boundary classifiers: GLIA random-forest model files -> device forests.
//
// File format = rf_old::writeModelToBinaryFile (ml/rf/ml_rf_model.cxx:378-455): a raw dump of the in-memory
// rf_old::Model struct (ml/rf/ml_rf.h:97-156; 520 bytes on x86-64/libstdc++: four std::vectors, then
// {pointer, int n[2]} pairs interleaved with the scalars nrnodes, ntree, mtry, nclass) followed by the arrays,
// each stored dense or -- when longer than 128 elements and mostly zero -- as (index, value) pairs (:48-70).
// readModelFromBinaryFile then transposes the tree arrays (:542-557) so that memory is "node fastest within a
// tree", the layout classForest walks (ml/rf/rf.hxx:392-404).  The walk itself lives in the un-vendored
// randomforest-matlab sources; it is restated from SURVEY.md Appendix B.4 (PARITY UNPINNED).
#include <cstdio>
#include <cstring>
#include <fstream>

#include "forest.hpp"

namespace glia {

namespace {

constexpr size_t kModelStructBytes = 520;
// offsets of the int n[2] pairs / scalars inside the struct dump (x86-64, libstdc++)
constexpr size_t O_N_NCAT = 104, O_N_CATF = 120, O_N_XBEST = 144, O_N_CLASSWT = 160, O_N_CUTOFF = 176,
                 O_N_TREEMAP = 192, O_N_NODESTATUS = 208, O_N_NODECLASS = 224, O_N_BESTVAR = 240, O_N_NDBIGTREE = 256,
                 O_N_ORIG = 280, O_N_NEW = 296, O_N_OUTCL = 320;

struct Reader {
  std::ifstream fs;
  bool ok = true;
  template <typename T> void raw(T* p, size_t n) {
    fs.read(reinterpret_cast<char*>(p), sizeof(T) * n);
    if (!fs) ok = false;
  }
  // readArray (ml_rf_model.cxx:48-70)
  template <typename T> std::vector<T> array(long long size) {
    std::vector<T> d;
    if (size <= 0) return d;
    d.assign((size_t)size, T());
    if (size > 128) {
      unsigned char sparse = 0;
      raw(&sparse, 1);
      if (sparse) {
        int num = 0;
        raw(&num, 1);
        for (int i = 0; i < num && ok; ++i) {
          int index = 0;
          raw(&index, 1);
          T v;
          raw(&v, 1);
          if (index < 0 || index >= size) { ok = false; break; }
          d[index] = v;
        }
        return d;
      }
    }
    raw(d.data(), (size_t)size);
    return d;
  }
};

// transpose (ml_rf.h:350-362): row-major n0 x n1 -> row-major n1 x n0
template <typename T> std::vector<T> transposed(const std::vector<T>& x, int n0, int n1) {
  std::vector<T> r(x.size());
  for (int rr = 0; rr < n1; ++rr)
    for (int cc = 0; cc < n0; ++cc) r[(size_t)rr * n0 + cc] = x[(size_t)cc * n1 + rr];
  return r;
}

}  // namespace

int load_forest_file(const char* path, int predict_label, HostForest* out) {
  Reader rd;
  rd.fs.open(path, std::ios::binary);
  if (!rd.fs.is_open()) { set_error(std::string("Error reading model file... (") + path + ")"); return GLIA_HMT_ERR_IO; }
  unsigned char hdr[kModelStructBytes];
  rd.raw(hdr, kModelStructBytes);
  if (!rd.ok) { set_error("forest: truncated model header"); return GLIA_HMT_ERR_IO; }
  // the four leading std::vectors must be empty (no categorical features): the reader derives their sizes
  // from the dumped begin/end pointers (ml_rf_model.cxx:464,478)
  for (int v = 0; v < 4; ++v) {
    unsigned long long b, e;
    memcpy(&b, hdr + 24 * v, 8);
    memcpy(&e, hdr + 24 * v + 8, 8);
    if (b != e) { set_error("forest: models with categorical feature tables are not supported"); return GLIA_HMT_ERR_UNSUPPORTED; }
  }
  auto n2 = [&](size_t off, int n[2]) { memcpy(n, hdr + off, 8); };
  int n_ncat[2], n_catf[2], n_xb[2], n_cw[2], n_co[2], n_tm[2], n_ns[2], n_nc[2], n_bv[2], n_nd[2], n_ol[2], n_nl[2];
  n2(O_N_NCAT, n_ncat); n2(O_N_CATF, n_catf); n2(O_N_XBEST, n_xb); n2(O_N_CLASSWT, n_cw); n2(O_N_CUTOFF, n_co);
  n2(O_N_TREEMAP, n_tm); n2(O_N_NODESTATUS, n_ns); n2(O_N_NODECLASS, n_nc); n2(O_N_BESTVAR, n_bv); n2(O_N_NDBIGTREE, n_nd);
  n2(O_N_ORIG, n_ol); n2(O_N_NEW, n_nl);
  auto sz = [](const int n[2]) { return (long long)n[0] * (long long)n[1]; };
  for (const int* n : {n_ncat, n_catf, n_xb, n_cw, n_co, n_tm, n_ns, n_nc, n_bv, n_nd, n_ol, n_nl})
    if (n[0] < 0 || n[1] < 0 || sz(n) > (1ll << 31)) { set_error("forest: corrupt model header"); return GLIA_HMT_ERR_IO; }
  std::vector<int> ncat = rd.array<int>(sz(n_ncat));
  std::vector<int> catf = rd.array<int>(sz(n_catf));
  int nrnodes = 0, ntree = 0, mtry = 0, nclass = 0;
  rd.raw(&nrnodes, 1); rd.raw(&ntree, 1);
  std::vector<double> xbest = rd.array<double>(sz(n_xb));
  std::vector<double> classwt = rd.array<double>(sz(n_cw));
  std::vector<double> cutoff = rd.array<double>(sz(n_co));
  std::vector<int> treemap = rd.array<int>(sz(n_tm));
  std::vector<int> nodestatus = rd.array<int>(sz(n_ns));
  std::vector<int> nodeclass = rd.array<int>(sz(n_nc));
  std::vector<int> bestvar = rd.array<int>(sz(n_bv));
  std::vector<int> ndbigtree = rd.array<int>(sz(n_nd));
  rd.raw(&mtry, 1);
  std::vector<int> orig = rd.array<int>(sz(n_ol));
  std::vector<int> newl = rd.array<int>(sz(n_nl));
  rd.raw(&nclass, 1);
  if (!rd.ok) { set_error("forest: truncated model file"); return GLIA_HMT_ERR_IO; }
  for (int c : catf) if (c) { set_error("forest: categorical features are not supported"); return GLIA_HMT_ERR_UNSUPPORTED; }
  if (nrnodes <= 0 || ntree <= 0 || nclass <= 0 || sz(n_xb) != (long long)nrnodes * ntree ||
      sz(n_tm) != 2ll * nrnodes * ntree || sz(n_ns) != (long long)nrnodes * ntree ||
      sz(n_nc) != (long long)nrnodes * ntree || sz(n_bv) != (long long)nrnodes * ntree || (long long)orig.size() < nclass) {
    set_error("forest: inconsistent model dimensions");
    return GLIA_HMT_ERR_IO;
  }
  // post-load transposes (ml_rf_model.cxx:542-557): memory becomes column-major n0 x n1 = node fastest per tree
  xbest = transposed(xbest, n_xb[0], n_xb[1]);
  treemap = transposed(treemap, n_tm[0], n_tm[1]);
  nodestatus = transposed(nodestatus, n_ns[0], n_ns[1]);
  nodeclass = transposed(nodeclass, n_nc[0], n_nc[1]);
  bestvar = transposed(bestvar, n_bv[0], n_bv[1]);

  out->ntree = ntree; out->nrnodes = nrnodes; out->nclass = nclass;
  out->split.assign((size_t)ntree * nrnodes, 0.0);
  out->meta.assign((size_t)ntree * nrnodes * 4, 0);
  int target = -1;
  for (int i = 0; i < nclass; ++i) if (orig[i] == predict_label) { target = i; break; }   // ml/rf/rf.hxx:366-369
  if (target < 0) { set_error("Error: invalid label for random forest predictor"); return GLIA_HMT_ERR_ARG; }
  int maxvar = 0;
  for (int j = 0; j < ntree; ++j) {
    for (int k = 0; k < nrnodes; ++k) {
      const size_t i = (size_t)j * nrnodes + k;
      const bool terminal = nodestatus[i] == -1;
      int* m = &out->meta[i * 4];
      out->split[i] = xbest[i];
      m[0] = bestvar[i] - 1;
      m[1] = treemap[(size_t)2 * j * nrnodes + 2 * k] - 1;
      m[2] = treemap[(size_t)2 * j * nrnodes + 2 * k + 1] - 1;
      m[3] = terminal ? ((nodeclass[i] - 1 == target) ? 1 : 0) : -1;
      if (!terminal) {
        if (m[0] < 0 || m[1] < 0 || m[2] < 0 || m[1] >= nrnodes || m[2] >= nrnodes) {
          // nodes beyond ndbigtree[j] are unused and all-zero: make them harmless terminals
          m[0] = 0; m[1] = m[2] = 0; m[3] = 0;
        } else if (m[0] > maxvar) maxvar = m[0];
      }
    }
  }
  out->max_var = maxvar;
  return GLIA_HMT_OK;
}


int pack_forest(const HostForest& hf, std::vector<PackedNode>* nodes, std::vector<int>* roots) {
  nodes->clear();
  roots->assign(hf.ntree, 0);
  std::vector<int> queue;
  for (int j = 0; j < hf.ntree; ++j) {
    const size_t base = (size_t)j * hf.nrnodes;
    const int first = (int)nodes->size();
    (*roots)[j] = first;
    queue.assign(1, 0);
    nodes->push_back(PackedNode());
    for (size_t q = 0; q < queue.size(); ++q) {
      if ((int)queue.size() > hf.nrnodes) { set_error("forest: tree has a cycle"); return GLIA_HMT_ERR_IO; }
      const int k = queue[q];
      const int* m = &hf.meta[(base + k) * 4];
      PackedNode n;
      n.split = hf.split[base + k];
      if (m[3] >= 0) { n.var = -1 - m[3]; n.left = 0; }
      else {
        n.var = m[0];
        n.left = (int)nodes->size();
        queue.push_back(m[1]); queue.push_back(m[2]);
        nodes->push_back(PackedNode()); nodes->push_back(PackedNode());
      }
      (*nodes)[first + q] = n;
    }
  }
  return GLIA_HMT_OK;
}

// three levels per 128-byte line (forest.hpp): breadth-first over the nodes of depth 0 mod 3
int pack_forest_triples(const HostForest& hf, std::vector<PackedTriple>* lines, std::vector<int>* roots) {
  lines->clear();
  roots->assign(hf.ntree, 0);
  if (hf.max_var > 32767) { set_error("forest: feature index beyond 32767"); return GLIA_HMT_ERR_UNSUPPORTED; }
  std::vector<int> queue;          // original node index of the line with the same position (relative to the tree's first line)
  for (int j = 0; j < hf.ntree; ++j) {
    const size_t base = (size_t)j * hf.nrnodes;
    const int first = (int)lines->size();
    (*roots)[j] = first;
    queue.assign(1, 0);
    lines->push_back(PackedTriple());
    for (size_t q = 0; q < queue.size(); ++q) {
      if ((int)queue.size() > hf.nrnodes) { set_error("forest: tree has a cycle"); return GLIA_HMT_ERR_IO; }
      PackedTriple L;
      memset(&L, 0, sizeof(L));
      int orig[7] = {queue[q], -1, -1, -1, -1, -1, -1};      // original node in each slot (-1: below a terminal node)
      for (int slot = 0; slot < 7; ++slot) {
        const int k = orig[slot];
        if (k < 0) continue;
        const int* m = &hf.meta[(base + k) * 4];
        L.split[slot] = hf.split[base + k];
        if (m[3] >= 0) { L.var[slot] = (short)(-1 - m[3]); continue; }
        L.var[slot] = (short)m[0];
        if (slot < 3) { orig[2 * slot + 1] = m[1]; orig[2 * slot + 2] = m[2]; }
        else {                                                // internal node of the line's last level: its daughters get lines of their own
          L.next[slot - 3] = (int)lines->size();
          queue.push_back(m[1]); queue.push_back(m[2]);
          lines->push_back(PackedTriple()); lines->push_back(PackedTriple());
        }
      }
      (*lines)[first + q] = L;
    }
  }
  return GLIA_HMT_OK;
}

}  // namespace glia
