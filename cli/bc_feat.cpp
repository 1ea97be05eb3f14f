// cli/bc_feat.cpp -- drop-in for hmt/main_bc_feat.cxx: boundary-classifier feature rows of a GIVEN merge order.
//   bc_feat -s seg.mha -o order.txt --pb pb.mha [--rbi/--rbb/--rbl/--rbu ...] [--bt ...] [-n b] [-l b] [--simpf b] -b feats.txt
#include "common.hpp"

using namespace cli;

int main(int argc, char* argv[]) {
  const std::string usage = "Usage: bc_feat -s <seg> -o <order> --pb <pb> [--rbi/--rbb/--rbl/--rbu ...] [--bt ...] [-n b] [-l b] [--simpf b] "
                            "-b <feats>   (flags as hmt/main_bc_feat.cxx:125-185)\n";
  std::vector<std::string> known = {"segImage", "mergeOrder", "saliency", "rbi", "rbb", "rbl", "rbu", "rli", "rlb", "rll", "rlu", "ri", "rb", "rl",
                                    "ru", "bi", "bb", "bl", "bu", "pb", "maskImage", "s0", "sb", "bt", "ns", "logs", "simpf", "histf", "medf", "bfeat"};
  Args a = parse(argc, argv, {{"s", "segImage"}, {"o", "mergeOrder"}, {"y", "saliency"}, {"m", "maskImage"}, {"n", "ns"}, {"l", "logs"}, {"b", "bfeat"}},
                 known, usage);
  for (const char* req : {"segImage", "mergeOrder", "pb"})
    if (!a.has(req)) { std::cerr << "Error: the option '--" << req << "' is required but missing\n" << usage; perr("Error: unable to parse input arguments"); }
  FeatInputs f;
  loadFeatInputs(a, f);
  std::vector<uint32_t> order = readOrder(a.str("mergeOrder"));
  const int64_t n = (int64_t)order.size() / 3;
  if (!a.has("bfeat")) return EXIT_SUCCESS;                                       // :75 nothing else is written
  glia_hmt_ctx* ctx; glia_hmt_rag* rag;
  check(glia_hmt_ctx_create(0, nullptr, &ctx));
  if (glia_hmt_ctx_libm_status(ctx) == 0) std::cerr << glia_hmt_last_error() << std::endl;     // host libm not reproduced: features within 1 ulp, unpinned
  uint32_t* dMask = loadMask(a, "maskImage", f.seg.size());
  check(glia_hmt_rag_build(ctx, f.seg.dim, f.seg.dims, f.dLab, dMask, /*only_contour=*/0, f.dPb, &f.cfg, &rag));
  std::vector<double> sal;
  if (a.has("saliency")) {                                                        // :50-55
    std::ifstream is(a.str("saliency"));
    if (!is) perr("Error: invalid data file dimension in " + a.str("saliency"));
    double x;
    while (is >> x) sal.push_back(x);
    if ((int64_t)sal.size() < n) perr("Error: too few saliencies...");
  }
  const int d = glia_hmt_bc_feat_dim(rag, sal.empty() ? 0 : 1);
  std::vector<double> feats((size_t)(n ? n : 1) * d);
  check(glia_hmt_bc_feat_saliency(ctx, rag, order.data(), n, sal.empty() ? nullptr : sal.data(), atof(a.str("s0", "1.0").c_str()),
                                  atof(a.str("sb", "1.0").c_str()), feats.data()));
  writeRows(a.str("bfeat"), feats.data(), n, d, /*FLT_PREC*/ 8);                  // :103-110
  glia_hmt_rag_free(rag); glia_hmt_ctx_destroy(ctx);
  (void)hipFree(f.dLab); (void)hipFree(f.dPb);
  return EXIT_SUCCESS;
}
