// cli/merge_order_pb.cpp -- drop-in for hmt/main_merge_order_pb.cxx: same flags, same output files.
//   merge_order_pb -s seg.mha -p pb.mha [-m mask] [-t 1|2] [-o order.txt] [-y saliency.txt]
// The dimension is taken from the image (the reference fixes it at compile time, CMakeLists.txt:17-19).
#include "common.hpp"

using namespace cli;

int main(int argc, char* argv[]) {
  const std::string usage =
      "Usage:\n  --help                 Print usage info\n  -s [ --segImage ] arg  Input initial segmentation image file name\n"
      "  -p [ --pbImage ] arg   Input boundary probability image file name\n  -m [ --maskImage ] arg Input mask image file name (optional)\n"
      "  -t [ --type ] arg      Boundary intensity stats type (1: median, 2: mean) [default: 1]\n"
      "  -o [ --mergeOrder ] arg Output merging order file name (optional)\n  -y [ --saliency ] arg  Output merging saliency file name (optional)\n";
  Args a = parse(argc, argv, {{"s", "segImage"}, {"p", "pbImage"}, {"m", "maskImage"}, {"t", "type"}, {"o", "mergeOrder"}, {"y", "saliency"}},
                 {"segImage", "pbImage", "maskImage", "type", "mergeOrder", "saliency"}, usage);
  if (!a.has("segImage") || !a.has("pbImage")) { std::cerr << "Error: the option '--segImage'/'--pbImage' is required but missing\n" << usage; return EXIT_FAILURE; }
  const int type = atoi(a.str("type", "1").c_str());
  if (type != 1 && type != 2) perr("Error: unsupported boundary stats type...");          // :36
  Volume seg = readMetaImage(a.str("segImage"), false), pb = readMetaImage(a.str("pbImage"), true);
  if (seg.dim != pb.dim || seg.size() != pb.size()) perr("Error: image sizes do not match...");
  uint32_t* dLab = upload(seg.u32);
  float* dPb = upload(pb.f32);
  glia_hmt_ctx* ctx; glia_hmt_rag* rag;
  check(glia_hmt_ctx_create(0, nullptr, &ctx));
  uint32_t* dMask = loadMask(a, "maskImage", seg.size());
  check(glia_hmt_rag_build(ctx, seg.dim, seg.dims, dLab, dMask, /*only_contour=*/1, dPb, nullptr, &rag));   // :27
  int64_t cap = glia_hmt_rag_num_regions(rag), n = 0;
  std::vector<uint32_t> order(3 * (cap ? cap : 1));
  std::vector<double> sal(cap ? cap : 1);
  check(glia_hmt_merge_order_pb(ctx, rag, type, order.data(), sal.data(), cap, &n));
  if (a.has("mergeOrder")) writeOrder(a.str("mergeOrder"), order, n);                    // :37-38
  if (a.has("saliency")) writeDoubles(a.str("saliency"), sal.data(), n);
  glia_hmt_rag_free(rag); glia_hmt_ctx_destroy(ctx);
  (void)hipFree(dLab); (void)hipFree(dPb);
  return EXIT_SUCCESS;
}
