// cli/segment_greedy.cpp -- drop-in for hmt/main_segment_greedy.cxx: node potentials from the merge
// (and optional region) probabilities, greedy tree resolution, final label image.
//   segment_greedy -s seg.mha -o order.txt [-p mergeProbs.txt] [-n regionProbs.txt] [-m mask.mha] [-i 0|1] [-r b] [-u b] -f out.mha
// Several -o / -p / -n files = several trees resolved jointly; -b writes the boundary-confidence image (float).
#include "common.hpp"

using namespace cli;

static std::vector<double> readDoubles(const std::string& file) {
  std::ifstream is(file);
  if (!is) perr("Error: invalid data file dimension in " + file);
  std::vector<double> v;
  double x;
  while (is >> x) v.push_back(x);
  return v;
}

int main(int argc, char* argv[]) {
  const std::string usage = "Usage: segment_greedy -s <seg> -o <order> [-p <mergeProbs>] [-n <regionProbs>] [-m <mask>] [-i b] [-r b] [-u b] "
                            "-f <finalSeg>   (flags as hmt/main_segment_greedy.cxx:98-127)\n";
  Args a = parse(argc, argv, {{"s", "segImage"}, {"o", "mergeOrders"}, {"p", "mergeProbs"}, {"n", "regionProbs"}, {"m", "maskImage"}, {"i", "ignore"},
                              {"r", "relabel"}, {"u", "write16"}, {"z", "compress"}, {"f", "finalSegImage"}, {"b", "bcImage"}},
                 {"segImage", "mergeOrders", "mergeProbs", "regionProbs", "maskImage", "ignore", "relabel", "write16", "compress", "finalSegImage", "bcImage"}, usage);
  for (const char* req : {"segImage", "mergeOrders"})
    if (!a.has(req)) { std::cerr << "Error: the option '--" << req << "' is required but missing\n" << usage; return EXIT_FAILURE; }
  // one tree per merge order file (main_segment_greedy.cxx:33-60)
  const auto orderFiles = a.all("mergeOrders"), probFiles = a.all("mergeProbs"), rprobFiles = a.all("regionProbs");
  const int nTree = (int)orderFiles.size();
  struct Tree { std::vector<uint32_t> lab; std::vector<int32_t> par, c0, c1; std::vector<double> pot; int64_t n = 0; };
  std::vector<Tree> trees((size_t)nTree);
  for (int t = 0; t < nTree; ++t) {
    std::vector<uint32_t> order = readOrder(orderFiles[t]);
    const int64_t n = (int64_t)order.size() / 3, cap = 3 * n + 1;
    std::vector<double> mprobs, rprobs;
    if ((int)probFiles.size() > t) { mprobs = readDoubles(probFiles[t]); if ((int64_t)mprobs.size() < n) perr("Error: too few merge probabilities..."); }
    if (!rprobFiles.empty()) { if ((int)rprobFiles.size() <= t) perr("Error: too few region probability files..."); rprobs = readDoubles(rprobFiles[t]); }
    Tree& T = trees[t];
    T.lab.resize(cap); T.par.resize(cap); T.c0.resize(cap); T.c1.resize(cap); T.pot.resize(cap);
    // region probabilities come one per node: the node count is known after a first pass
    T.n = glia_hmt_tree_potentials(order.data(), n, mprobs.empty() ? nullptr : mprobs.data(), nullptr, T.lab.data(), T.par.data(), T.c0.data(),
                                   T.c1.data(), T.pot.data(), cap);
    if (T.n < 0) perr(glia_hmt_last_error());
    if (!rprobs.empty()) {
      if ((int64_t)rprobs.size() < T.n) perr("Error: too few region probabilities...");
      T.n = glia_hmt_tree_potentials(order.data(), n, mprobs.empty() ? nullptr : mprobs.data(), rprobs.data(), T.lab.data(), T.par.data(),
                                     T.c0.data(), T.c1.data(), T.pot.data(), cap);
      if (T.n < 0) perr(glia_hmt_last_error());
    }
  }
  int64_t total = 0;
  std::vector<int64_t> nn; std::vector<const uint32_t*> pl; std::vector<const int32_t*> pp, p0, p1; std::vector<const double*> pq;
  for (auto& T : trees) { total += T.n; nn.push_back(T.n); pl.push_back(T.lab.data()); pp.push_back(T.par.data()); p0.push_back(T.c0.data());
                          p1.push_back(T.c1.data()); pq.push_back(T.pot.data()); }
  if (a.has("bcImage")) {                                                                                   // :62-70
    Volume segb = readMetaImage(a.str("segImage"), false);
    uint32_t* dL = upload(segb.u32);
    uint32_t* dM = loadMask(a, "maskImage", segb.size());
    std::vector<float> zeros(segb.size(), 0.0f);
    float* dZ = upload(zeros);                                       // the contour-only map needs an image volume; its values are not used
    float* dOut = upload(zeros);
    glia_hmt_ctx* cx; glia_hmt_rag* rag;
    check(glia_hmt_ctx_create(0, nullptr, &cx));
    check(glia_hmt_rag_build(cx, segb.dim, segb.dims, dL, dM, /*only_contour=*/1, dZ, nullptr, &rag));
    check(glia_hmt_boundary_confidence(cx, rag, nTree, nn.data(), pl.data(), pp.data(), p0.data(), pq.data(), dOut));
    hipCheck(hipMemcpy(zeros.data(), dOut, zeros.size() * 4, hipMemcpyDeviceToHost));
    writeMetaImageFloat(a.str("bcImage"), segb.dim, segb.dims, zeros, flagOf(a, "compress"));
    glia_hmt_rag_free(rag); glia_hmt_ctx_destroy(cx);
    (void)hipFree(dL); (void)hipFree(dZ); (void)hipFree(dOut);
    if (dM) (void)hipFree(dM);
  }
  if (!a.has("finalSegImage")) return EXIT_SUCCESS;
  std::vector<int32_t> pickTree((size_t)(total ? total : 1)), pickNode((size_t)(total ? total : 1));
  const int64_t np = glia_hmt_resolve_trees_greedy(nTree, nn.data(), pl.data(), pp.data(), p0.data(), p1.data(), pq.data(), pickTree.data(),
                                                   pickNode.data(), total);                                            // :72-76
  if (np < 0) perr(glia_hmt_last_error());
  // genLabelTransform (hmt/tree_segment.hxx:24-35): the leaves below pick k get the label 1 + k
  std::vector<uint32_t> src, dst, s1((size_t)(total ? total : 1)), d1((size_t)(total ? total : 1));
  for (int64_t k = 0; k < np; ++k) {
    const Tree& T = trees[pickTree[k]];
    const int64_t m1 = glia_hmt_label_transform(T.lab.data(), T.c0.data(), T.c1.data(), T.n, &pickNode[k], 1, (uint32_t)(1 + k), s1.data(), d1.data(),
                                                (int64_t)s1.size());
    if (m1 < 0) perr(glia_hmt_last_error());
    src.insert(src.end(), s1.begin(), s1.begin() + m1); dst.insert(dst.end(), d1.begin(), d1.begin() + m1);
  }
  const int64_t m = (int64_t)src.size();
  Volume seg = readMetaImage(a.str("segImage"), false);
  uint32_t* dLab = upload(seg.u32);
  uint32_t* dMask = loadMask(a, "maskImage", seg.size());
  glia_hmt_ctx* ctx;
  check(glia_hmt_ctx_create(0, nullptr, &ctx));
  const bool ignore = a.has("ignore") ? flagOf(a, "ignore") : true;                                                       // default true
  check(glia_hmt_transform_image(ctx, dLab, (int64_t)seg.size(), src.data(), dst.data(), m, dMask, ignore ? 1 : 0));      // :78 genFinalSegmentation
  uint32_t nl = 0;
  if (flagOf(a, "relabel")) check(glia_hmt_relabel_image(ctx, dLab, (int64_t)seg.size(), 0, &nl));
  hipCheck(hipMemcpy(seg.u32.data(), dLab, seg.size() * 4, hipMemcpyDeviceToHost));
  writeMetaImage(a.str("finalSegImage"), seg.dim, seg.dims, seg.u32, flagOf(a, "write16"), flagOf(a, "compress"));
  glia_hmt_ctx_destroy(ctx);
  (void)hipFree(dLab);
  if (dMask) (void)hipFree(dMask);
  return EXIT_SUCCESS;
}
