// oracle/ref_tree_driver.cc -- TEST INFRASTRUCTURE ONLY.
//
// Drives the reference's OWN merge-tree code, compiled in place from /root/reference/code (nothing is copied):
//   hmt/tree_build.hxx:12-63   genTree, genTreeWithNodePotentials
//   hmt/tree_greedy.hxx:36-70  resolveTreeGreedy (one tree, validity vector)
//   hmt/tree_greedy.hxx:76-152 resolveTreeGreedy (several trees, (tree, node) picks)
//   type/tree.hxx              TTree (traverseAncestors / Descendants / Leaves)
// with exactly the node data, potential updates and comparison main_segment_greedy.cxx:23-26,36-59,80-83 uses.
// These headers are ITK-free and compile unmodified.
//
// stdin:  nTrees
//         per tree: nMerges hasMergeProbs hasRegionProbs
//                   nMerges lines "x0 x1 x2", then nMerges merge probabilities, then (if any) one region
//                   probability per tree node
// stdout: per tree: "T nNodes", then one "label parent child0 child1 potential(%.17g)" line per node;
//         "S n" + the single-tree picks of tree 0 (tree_greedy.hxx:51-70); "M n" + "(tree node)" picks over all trees.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include "hmt/tree_greedy.hxx"

using namespace glia;
using namespace glia::hmt;

struct NodeData {          // hmt/main_segment_greedy.cxx:23-26
  Label label;
  double potential;
};
typedef TTree<NodeData> Tree;

int main() {
  int nTree;
  if (scanf("%d", &nTree) != 1) return 2;
  std::vector<std::vector<TTriple<Label>>> orders(nTree);
  std::vector<Tree> trees(nTree);
  for (int i = 0; i < nTree; ++i) {
    int n, hasP, hasR;
    if (scanf("%d %d %d", &n, &hasP, &hasR) != 3) return 2;
    orders[i].resize(n);
    for (auto& t : orders[i]) if (scanf("%u %u %u", &t.x0, &t.x1, &t.x2) != 3) return 2;
    if (hasP) {                                                  // main_segment_greedy.cxx:38-44
      std::vector<double> mergeProbs(n);
      for (auto& p : mergeProbs) if (scanf("%lf", &p) != 1) return 2;
      genTreeWithNodePotentials(trees[i], orders[i], mergeProbs.begin());
    } else {                                                     // :45-50
      genTree(trees[i], orders[i], [](Tree::Node& node, Label r) { node.data.label = r; node.data.potential = 1.0; });
    }
    if (hasR) {                                                  // :52-58
      std::vector<double> regionProbs(trees[i].size());
      for (auto& p : regionProbs) if (scanf("%lf", &p) != 1) return 2;
      auto rpit = regionProbs.begin();
      for (auto& tn : trees[i]) tn.data.potential *= std::max(*rpit++, FEPS);
    }
    printf("T %d\n", (int)trees[i].size());
    for (auto const& node : trees[i])
      printf("%u %d %d %d %.17g\n", node.data.label, node.parent, node.children.empty() ? -1 : node.children.front(),
             node.children.empty() ? -1 : node.children.back(), node.data.potential);
  }
  auto comp = [](Tree::Node const& node0, Tree::Node const& node1) -> bool { return node0.data.potential < node1.data.potential; };   // :80-83
  std::vector<int> single;
  resolveTreeGreedy(single, trees[0], comp);
  printf("S %d\n", (int)single.size());
  for (int p : single) printf("%d\n", p);
  std::vector<std::pair<int, int>> picks;
  resolveTreeGreedy(picks, trees, comp);
  printf("M %d\n", (int)picks.size());
  for (auto const& p : picks) printf("%d %d\n", p.first, p.second);
  return 0;
}
