// cli/bc_feat.cpp -- drop-in for hmt/main_bc_feat.cxx: boundary-classifier feature rows of a GIVEN merge order.
//   bc_feat -s seg.mha -o order.txt --pb pb.mha [--rbi/--rbb/--rbl/--rbu ...] [--bt ...] [-n b] [-l b] [--simpf b] -b feats.txt
// Not supported yet: the saliency features (-y/--s0/--sb).
#include "common.hpp"

using namespace cli;

int main(int argc, char* argv[]) {
  const std::string usage = "Usage: bc_feat -s <seg> -o <order> --pb <pb> [--rbi/--rbb/--rbl/--rbu ...] [--bt ...] [-n b] [-l b] [--simpf b] "
                            "-b <feats>   (flags as hmt/main_bc_feat.cxx:125-185)\n";
  std::vector<std::string> known = {"segImage", "mergeOrder", "saliency", "rbi", "rbb", "rbl", "rbu", "rli", "rlb", "rll", "rlu", "ri", "rb", "rl",
                                    "ru", "bi", "bb", "bl", "bu", "pb", "maskImage", "s0", "sb", "bt", "ns", "logs", "simpf", "bfeat"};
  Args a = parse(argc, argv, {{"s", "segImage"}, {"o", "mergeOrder"}, {"y", "saliency"}, {"m", "maskImage"}, {"n", "ns"}, {"l", "logs"}, {"b", "bfeat"}},
                 known, usage);
  for (const char* req : {"segImage", "mergeOrder", "pb"})
    if (!a.has(req)) { std::cerr << "Error: the option '--" << req << "' is required but missing\n" << usage; perr("Error: unable to parse input arguments"); }
  if (a.has("saliency")) perr("Error: saliency features (-y) are not supported by the MI355X path yet...");
  FeatInputs f;
  loadFeatInputs(a, f);
  std::vector<uint32_t> order = readOrder(a.str("mergeOrder"));
  const int64_t n = (int64_t)order.size() / 3;
  if (!a.has("bfeat")) return EXIT_SUCCESS;                                       // :75 nothing else is written
  glia_hmt_ctx* ctx; glia_hmt_rag* rag;
  check(glia_hmt_ctx_create(0, nullptr, &ctx));
  uint32_t* dMask = loadMask(a, "maskImage", f.seg.size());
  check(glia_hmt_rag_build(ctx, f.seg.dim, f.seg.dims, f.dLab, dMask, /*only_contour=*/0, f.dPb, &f.cfg, &rag));
  const int d = glia_hmt_feat_dim(rag);
  std::vector<double> feats((size_t)(n ? n : 1) * d);
  check(glia_hmt_bc_feat(ctx, rag, order.data(), n, feats.data()));
  writeRows(a.str("bfeat"), feats.data(), n, d, /*FLT_PREC*/ 8);                  // :103-110
  glia_hmt_rag_free(rag); glia_hmt_ctx_destroy(ctx);
  (void)hipFree(f.dLab); (void)hipFree(f.dPb);
  return EXIT_SUCCESS;
}
