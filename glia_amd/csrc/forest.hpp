// glia_amd/csrc/forest.hpp -- boundary classifier objects (alg/rf.hxx:10-122).
#pragma once
#include "hmt_internal.hpp"

namespace glia {

struct HostForest {
  int ntree = 0, nrnodes = 0, nclass = 0, max_var = 0;
  std::vector<double> split;   // [ntree][nrnodes]
  std::vector<int> meta;       // [ntree][nrnodes][4] = {var (0-based), left, right (0-based), vote (-1 = internal node)}
};

// Device layout: only the nodes a walk can reach, renumbered breadth-first per tree so that the two daughters of a
// node are adjacent (right = left + 1).  16 bytes per node = one load per level, and a forest of the reference's
// default size (255 trees) fits the 4 MiB L2 of an XCD.
struct PackedNode {
  double split;
  int var;       // >= 0: internal node, feature index;  < 0: terminal, vote = -1 - var
  int left;      // index of the left daughter in the packed array (absolute)
};
// The same trees for LATENCY-bound walks (one vector, one tree per lane: the classifier loop's helpers; the forest of a large volume
// does not fit the L2 and every trip goes to the memory-side cache): a node of depth 0 mod 3 together with its daughters and
// grand-daughters in one 128-byte line (the L2's line size), so that one trip decides three levels.
//   node 0 = the line's own node; 1, 2 = its left / right daughter; 3, 4 = the daughters of 1; 5, 6 = the daughters of 2
// var >= 0: internal node (feature index); var < 0: terminal, vote = -1 - var; slots below a terminal node hold var = 0, split = 0
// (never looked at by a walk).  next[i]: for an internal node 3 + i, the line of ITS left daughter (the right one follows it).
struct alignas(128) PackedTriple {
  double split[7];
  short var[8];
  int next[4];
  int pad[10];
};
static_assert(sizeof(PackedTriple) == 128, "one L2 line per three levels");
struct DeviceForest {
  int ntree, nrnodes;          // nrnodes: depth bound for the walk
  int nnodes;                  // packed nodes of all trees
  const PackedNode* nodes;
  const int* root;             // [ntree] index of each tree's root
  const PackedTriple* triples; // three levels per line
  const int* troot;            // [ntree] line of each tree's root
};
int pack_forest(const HostForest& hf, std::vector<PackedNode>* nodes, std::vector<int>* roots);
int pack_forest_triples(const HostForest& hf, std::vector<PackedTriple>* lines, std::vector<int>* roots);

// What fBcPred evaluates (hmt/main_merge_order_bc.cxx:127-137): one forest, or three forests behind a
// ThresholdModelDistributor (type/function.hxx:71-85), or -- diagnostics only -- 1 - x[index].
struct DeviceClassifier {
  int kind;              // 0 = forest(s), 1 = feature stub
  int n_models;          // 1 or 3
  DeviceForest f[3];
  int dim0, dim1;
  double threshold;
  int stub_index;
};

int load_forest_file(const char* path, int predict_label, HostForest* out);

}  // namespace glia

struct glia_hmt_forest {
  int device = 0;
  glia::DeviceClassifier dc;
  std::vector<void*> allocs;
  int max_var = 0;
};
