// glia_amd/csrc/tree.cpp -- the step after the merge path: merge tree with node potentials, greedy
// tree resolution, label transform of the picked nodes.  Host-only (the trees have 2R-1 nodes; the voxel work is
// glia_hmt_transform_image).  Reference: hmt/tree_build.hxx:41-63 (genTreeWithNodePotentials), hmt/tree_greedy.hxx:36-70,
// 76-152 (pickTreeNode / resolveTreeGreedy with comp = potential <), hmt/tree_segment.hxx:10-21 (genLabelTransform),
// hmt/main_segment_greedy.cxx:33-86.  The reference re-scans every node for each pick (O(n^2)); here the same pick
// sequence comes out of a heap ordered by (potential descending, node index ascending) with lazy invalidation.
#include <algorithm>
#include <limits>
#include <queue>
#include <unordered_map>
#include <vector>

#include "hmt_internal.hpp"

using namespace glia;

namespace glia {

// genBoundaryConfidenceMap with all nodes (hmt/tree_segment.hxx:66-143): for every unordered leaf pair the largest
// (float) potential over the tree nodes whose region still holds a directed entry of the pair.  A node N holds a -> b
// iff a is below N and the entry has not been cancelled there: a non-mutual entry never is (type/region.hxx:66-75), a
// mutual one is as soon as b is below N too.  So a non-mutual entry sees a and ALL its ancestors, a mutual one the path
// from a up to just below the lowest common ancestor.  Path maxima come from binary lifting; out[i] belongs to directed
// pair i and is shared by both directions.
int boundary_confidence_values(int n_trees, const int64_t* n_nodes, const uint32_t* const* node_label, const int32_t* const* parent,
                               const int32_t* const* child0, const double* const* potential, const uint32_t* pa, const uint32_t* pb,
                               int64_t P, std::vector<float>* out) {
  out->assign((size_t)P, 0.0f);
  std::vector<char> seen((size_t)P, 0);
  // partner (b -> a) of every directed pair (pairs are sorted by (a, b))
  std::vector<int64_t> partner((size_t)P, -1);
  for (int64_t i = 0; i < P; ++i) {
    int64_t lo = 0, hi = P;
    const uint32_t a = pb[i], b = pa[i];
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (pa[mid] < a || (pa[mid] == a && pb[mid] < b)) lo = mid + 1; else hi = mid; }
    if (lo < P && pa[lo] == a && pb[lo] == b) partner[i] = lo;
  }
  for (int t = 0; t < n_trees; ++t) {
    const int64_t n = n_nodes[t];
    if (n <= 0) continue;
    int LOG = 1;
    while ((1ll << LOG) < n) ++LOG;
    std::vector<int32_t> depth((size_t)n, 0), root((size_t)n, 0);
    std::vector<std::vector<int32_t>> up((size_t)LOG, std::vector<int32_t>((size_t)n));
    std::vector<std::vector<float>> mx((size_t)LOG, std::vector<float>((size_t)n));      // max over 2^k nodes starting at x going up
    std::vector<float> maxUp((size_t)n);
    for (int64_t x = n - 1; x >= 0; --x) {                 // parents have larger indices than children
      const int32_t p = parent[t][x];
      const float v = (float)potential[t][x];
      depth[x] = p >= 0 ? depth[p] + 1 : 0;
      root[x] = p >= 0 ? root[p] : (int32_t)x;
      maxUp[x] = p >= 0 ? std::max(v, maxUp[p]) : v;
      up[0][x] = p >= 0 ? p : (int32_t)x;
      mx[0][x] = v;
    }
    for (int k = 1; k < LOG; ++k)
      for (int64_t x = 0; x < n; ++x) {
        const int32_t h = up[k - 1][x];
        up[k][x] = up[k - 1][h];
        mx[k][x] = std::max(mx[k - 1][x], mx[k - 1][h]);
      }
    std::unordered_map<uint32_t, int32_t> leaf;
    for (int64_t x = 0; x < n; ++x) if (child0[t][x] < 0) leaf[node_label[t][x]] = (int32_t)x;
    // max potential over x and its `steps - 1` nearest ancestors
    auto path_max = [&](int32_t x, int32_t steps) {
      float m = -std::numeric_limits<float>::infinity();
      for (int k = 0; steps > 0; ++k, steps >>= 1)
        if (steps & 1) { m = std::max(m, mx[k][x]); x = up[k][x]; }
      return m;
    };
    auto lca_depth = [&](int32_t x, int32_t y) {
      if (depth[x] < depth[y]) std::swap(x, y);
      int32_t d = depth[x] - depth[y];
      for (int k = 0; d > 0; ++k, d >>= 1) if (d & 1) x = up[k][x];
      if (x == y) return depth[x];
      for (int k = LOG - 1; k >= 0; --k) if (up[k][x] != up[k][y]) { x = up[k][x]; y = up[k][y]; }
      return depth[x] - 1;
    };
    for (int64_t i = 0; i < P; ++i) {
      auto la = leaf.find(pa[i]);
      if (la == leaf.end()) continue;
      const int32_t xa = la->second;
      float v;
      auto lb = partner[i] >= 0 ? leaf.find(pb[i]) : leaf.end();
      if (lb != leaf.end() && root[lb->second] == root[xa]) v = path_max(xa, depth[xa] - lca_depth(xa, lb->second));
      else v = maxUp[xa];
      // the pair's value is shared by both directions (key normalised to (min, max), tree_segment.hxx:76-83)
      const int64_t j = partner[i];
      const float cur = seen[i] ? (*out)[i] : -std::numeric_limits<float>::infinity();
      const float nv = std::max(cur, v);
      (*out)[i] = nv; seen[i] = 1;
      if (j >= 0) { (*out)[j] = seen[j] ? std::max((*out)[j], nv) : nv; seen[j] = 1; if ((*out)[j] > (*out)[i]) (*out)[i] = (*out)[j]; }
    }
  }
  // both directions of a pair hold the same maximum
  for (int64_t i = 0; i < P; ++i) if (partner[i] >= 0) { const float m = std::max((*out)[i], (*out)[partner[i]]); (*out)[i] = m; (*out)[partner[i]] = m; }
  for (int64_t i = 0; i < P; ++i) if (!seen[i]) (*out)[i] = 0.0f;
  return GLIA_HMT_OK;
}

}  // namespace glia

extern "C" {

int64_t glia_hmt_tree_potentials(const uint32_t* h_order, int64_t n_merges, const double* h_merge_probs,
                                 const double* h_region_probs, uint32_t* node_label, int32_t* parent, int32_t* child0,
                                 int32_t* child1, double* potential, int64_t capacity) {
  if (!h_order || !node_label || !parent || !child0 || !child1 || !potential || n_merges < 0) {
    set_error("tree_potentials: invalid argument");
    return GLIA_HMT_ERR_ARG;
  }
  const int64_t n = glia_hmt_gen_tree(h_order, n_merges, node_label, parent, child0, child1, capacity);
  if (n < 0) return n;
  // genTree visits the nodes in index order; an inner node's callback sees its (already created) children
  int64_t mi = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (child0[i] < 0) { potential[i] = 1.0; continue; }           // leaf: set by its parent below (or 1.0 without probabilities)
    if (!h_merge_probs) { potential[i] = 1.0; continue; }          // main_segment_greedy.cxx:46-50
    const double p = h_merge_probs[mi++];
    potential[i] = p;
    const double pSplit = 1.0 - p;
    for (int32_t c : {child0[i], child1[i]}) {
      if (child0[c] < 0) potential[c] = pSplit * pSplit;
      else potential[c] *= pSplit;
    }
  }
  if (h_merge_probs && n > 0) potential[n - 1] *= potential[n - 1];   // root() = last node (type/tree.hxx:100)
  if (h_region_probs) for (int64_t i = 0; i < n; ++i) potential[i] *= std::max(h_region_probs[i], 2.22e-16 /* FEPS */);
  return n;
}

int64_t glia_hmt_resolve_tree_greedy(const int32_t* parent, const int32_t* child0, const int32_t* child1,
                                     const double* potential, int64_t n_nodes, int32_t* h_picks, int64_t capacity) {
  if (!parent || !child0 || !child1 || !potential || !h_picks || n_nodes < 0) {
    set_error("resolve_tree_greedy: invalid argument");
    return GLIA_HMT_ERR_ARG;
  }
  // pickTreeNode keeps the FIRST node (index order) among those of maximal potential: `comp(tree[ret], node)` is a strict <
  struct Item { double p; int32_t i; };
  auto worse = [](const Item& a, const Item& b) { return a.p < b.p || (a.p == b.p && a.i > b.i); };
  std::priority_queue<Item, std::vector<Item>, decltype(worse)> heap(worse);
  for (int64_t i = 0; i < n_nodes; ++i) heap.push(Item{potential[i], (int32_t)i});
  std::vector<char> valid((size_t)n_nodes, 1);
  std::vector<int32_t> stack;
  int64_t np = 0;
  while (!heap.empty()) {
    const Item it = heap.top();
    heap.pop();
    if (!valid[it.i]) continue;
    if (np >= capacity) { set_error("resolve_tree_greedy: output capacity too small"); return GLIA_HMT_ERR_CAPACITY; }
    h_picks[np++] = it.i;
    valid[it.i] = 0;
    for (int32_t a = parent[it.i]; a >= 0; a = parent[a]) valid[a] = 0;          // traverseAncestors
    stack.assign(1, it.i);                                                        // traverseDescendants
    while (!stack.empty()) {
      const int32_t x = stack.back();
      stack.pop_back();
      for (int32_t c : {child0[x], child1[x]}) if (c >= 0 && valid[c]) { valid[c] = 0; stack.push_back(c); }
    }
  }
  return np;
}

// resolveTreeGreedy over several trees (hmt/tree_greedy.hxx:76-92, 104-152): one pick sequence over all trees (first
// maximum in (tree, node) order); a pick invalidates its ancestors and descendants and, for every LEAF below it, the
// leaf of the same label in each other tree (as far as that leaf hangs below the tree's last node, :114-119) together
// with that leaf's ancestors.
int64_t glia_hmt_resolve_trees_greedy(int n_trees, const int64_t* n_nodes, const uint32_t* const* node_label,
                                      const int32_t* const* parent, const int32_t* const* child0, const int32_t* const* child1,
                                      const double* const* potential, int32_t* h_pick_tree, int32_t* h_pick_node, int64_t capacity) {
  if (n_trees < 1 || !n_nodes || !node_label || !parent || !child0 || !child1 || !potential || !h_pick_tree || !h_pick_node) {
    set_error("resolve_trees_greedy: invalid argument");
    return GLIA_HMT_ERR_ARG;
  }
  struct Item { double p; int32_t t, i; };
  auto worse = [](const Item& a, const Item& b) { return a.p < b.p || (a.p == b.p && (a.t > b.t || (a.t == b.t && a.i > b.i))); };
  std::priority_queue<Item, std::vector<Item>, decltype(worse)> heap(worse);
  std::vector<std::vector<char>> valid((size_t)n_trees);
  std::vector<std::unordered_map<uint32_t, int32_t>> lnmap((size_t)n_trees);
  std::vector<int32_t> stack;
  for (int t = 0; t < n_trees; ++t) {
    valid[t].assign((size_t)n_nodes[t], 1);
    for (int64_t i = 0; i < n_nodes[t]; ++i) heap.push(Item{potential[t][i], (int32_t)t, (int32_t)i});
    if (n_nodes[t] > 0) {                       // leaves below root() = the last node
      stack.assign(1, (int32_t)n_nodes[t] - 1);
      while (!stack.empty()) {
        const int32_t x = stack.back(); stack.pop_back();
        if (child0[t][x] < 0) lnmap[t][node_label[t][x]] = x;
        else { stack.push_back(child0[t][x]); stack.push_back(child1[t][x]); }
      }
    }
  }
  int64_t np = 0;
  std::vector<uint32_t> llabels;
  while (!heap.empty()) {
    const Item it = heap.top();
    heap.pop();
    if (!valid[it.t][it.i]) continue;
    if (np >= capacity) { set_error("resolve_trees_greedy: output capacity too small"); return GLIA_HMT_ERR_CAPACITY; }
    h_pick_tree[np] = it.t; h_pick_node[np] = it.i; ++np;
    valid[it.t][it.i] = 0;
    for (int32_t a = parent[it.t][it.i]; a >= 0; a = parent[it.t][a]) valid[it.t][a] = 0;
    llabels.clear();
    stack.clear();
    for (int32_t c : {child0[it.t][it.i], child1[it.t][it.i]}) if (c >= 0) stack.push_back(c);
    while (!stack.empty()) {
      const int32_t x = stack.back(); stack.pop_back();
      valid[it.t][x] = 0;
      if (child0[it.t][x] < 0) llabels.push_back(node_label[it.t][x]);
      else { stack.push_back(child0[it.t][x]); stack.push_back(child1[it.t][x]); }
    }
    for (uint32_t l : llabels)
      for (int t = 0; t < n_trees; ++t) {
        if (t == it.t) continue;
        auto nit = lnmap[t].find(l);
        if (nit == lnmap[t].end()) continue;
        valid[t][nit->second] = 0;
        for (int32_t a = parent[t][nit->second]; a >= 0; a = parent[t][a]) valid[t][a] = 0;
      }
  }
  return np;
}

int64_t glia_hmt_label_transform(const uint32_t* node_label, const int32_t* child0, const int32_t* child1, int64_t n_nodes,
                                 const int32_t* h_picks, int64_t n_picks, uint32_t key_to_assign, uint32_t* h_src,
                                 uint32_t* h_dst, int64_t capacity) {
  if (!node_label || !child0 || !child1 || !h_picks || !h_src || !h_dst || n_nodes < 0 || n_picks < 0) {
    set_error("label_transform: invalid argument");
    return GLIA_HMT_ERR_ARG;
  }
  int64_t m = 0;
  std::vector<int32_t> stack;
  for (int64_t k = 0; k < n_picks; ++k) {
    const int32_t pk = h_picks[k];
    if (pk < 0 || pk >= n_nodes) { set_error("label_transform: pick out of range"); return GLIA_HMT_ERR_ARG; }
    stack.assign(1, pk);
    while (!stack.empty()) {
      const int32_t x = stack.back();
      stack.pop_back();
      if (child0[x] < 0) {
        if (m >= capacity) { set_error("label_transform: output capacity too small"); return GLIA_HMT_ERR_CAPACITY; }
        h_src[m] = node_label[x]; h_dst[m] = key_to_assign; ++m;
      } else { stack.push_back(child0[x]); stack.push_back(child1[x]); }
    }
    ++key_to_assign;
  }
  return m;
}

}  // extern "C"
