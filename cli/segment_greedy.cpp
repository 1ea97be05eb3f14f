// cli/segment_greedy.cpp -- drop-in for hmt/main_segment_greedy.cxx with ONE merge tree: node potentials from the merge
// (and optional region) probabilities, greedy tree resolution, final label image.
//   segment_greedy -s seg.mha -o order.txt [-p mergeProbs.txt] [-n regionProbs.txt] [-m mask.mha] [-i 0|1] [-r b] [-u b] -f out.mha
// Not supported: several trees (-o given more than once), the boundary-confidence image (-b).
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
  if (a.all("mergeOrders").size() != 1) perr("Error: the MI355X path resolves a single merge tree in this version...");
  if (a.has("bcImage")) perr("Error: the boundary confidence image is not supported by the MI355X path yet...");
  if (flagOf(a, "compress")) perr("Error: compressed output is not supported...");
  std::vector<uint32_t> order = readOrder(a.str("mergeOrders"));
  const int64_t n = (int64_t)order.size() / 3, cap = 3 * n + 1;
  std::vector<double> mprobs, rprobs;
  if (a.has("mergeProbs")) { mprobs = readDoubles(a.str("mergeProbs")); if ((int64_t)mprobs.size() < n) perr("Error: too few merge probabilities..."); }
  std::vector<uint32_t> lab(cap), src(cap), dst(cap);
  std::vector<int32_t> par(cap), c0(cap), c1(cap), picks(cap);
  std::vector<double> pot(cap);
  if (a.has("regionProbs")) rprobs = readDoubles(a.str("regionProbs"));
  // potentials need the node count first when region probabilities are given (one per node)
  int64_t nn = glia_hmt_tree_potentials(order.data(), n, mprobs.empty() ? nullptr : mprobs.data(), nullptr, lab.data(), par.data(), c0.data(),
                                        c1.data(), pot.data(), cap);
  if (nn < 0) perr(glia_hmt_last_error());
  if (!rprobs.empty()) {
    if ((int64_t)rprobs.size() < nn) perr("Error: too few region probabilities...");
    nn = glia_hmt_tree_potentials(order.data(), n, mprobs.empty() ? nullptr : mprobs.data(), rprobs.data(), lab.data(), par.data(), c0.data(),
                                  c1.data(), pot.data(), cap);
    if (nn < 0) perr(glia_hmt_last_error());
  }
  if (!a.has("finalSegImage")) return EXIT_SUCCESS;
  const int64_t np = glia_hmt_resolve_tree_greedy(par.data(), c0.data(), c1.data(), pot.data(), nn, picks.data(), cap);   // :72-76
  if (np < 0) perr(glia_hmt_last_error());
  const int64_t m = glia_hmt_label_transform(lab.data(), c0.data(), c1.data(), nn, picks.data(), np, 1u, src.data(), dst.data(), cap);
  if (m < 0) perr(glia_hmt_last_error());
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
  writeMetaImage(a.str("finalSegImage"), seg.dim, seg.dims, seg.u32, flagOf(a, "write16"));
  glia_hmt_ctx_destroy(ctx);
  (void)hipFree(dLab);
  if (dMask) (void)hipFree(dMask);
  return EXIT_SUCCESS;
}
