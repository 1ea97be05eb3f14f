// cli/pre_merge.cpp -- drop-in for gadget/main_pre_merge.cxx: merges small regions (pb-mean order under the size
// condition) and writes the relabelled segmentation.
//   pre_merge -s seg.mha -p pb.mha -t t0 [t1] [-b rpbThreshold] [-r 0|1] [-u 0|1] -o out.mha
#include "common.hpp"

using namespace cli;

int main(int argc, char* argv[]) {
  const std::string usage = "Usage: pre_merge -s <seg> -p <pb> [-m mask] -t <size> [<size>] [-b rpb] [-r b] [-u b] [-z b] -o <out>   "
                            "(flags as gadget/main_pre_merge.cxx:88-112)\n";
  Args a = parse(argc, argv, {{"s", "segImage"}, {"p", "pbImage"}, {"m", "maskImage"}, {"t", "sizeThreshold"}, {"b", "rpbThreshold"}, {"r", "relabel"},
                              {"u", "write16"}, {"z", "compress"}, {"o", "outputImage"}},
                 {"segImage", "pbImage", "maskImage", "sizeThreshold", "rpbThreshold", "relabel", "write16", "compress", "outputImage"}, usage);
  for (const char* req : {"segImage", "pbImage", "sizeThreshold", "outputImage"})
    if (!a.has(req)) { std::cerr << "Error: the option '--" << req << "' is required but missing\n" << usage; return EXIT_FAILURE; }
  auto ts = a.all("sizeThreshold");
  if (ts.empty() || ts.size() > 2) perr("Error: one or two size thresholds expected...");
  int sizes[2] = {atoi(ts[0].c_str()), ts.size() > 1 ? atoi(ts[1].c_str()) : 0};
  const double rpb = atof(a.str("rpbThreshold", "0").c_str());
  Volume seg = readMetaImage(a.str("segImage"), false), pb = readMetaImage(a.str("pbImage"), true);
  if (seg.dim != pb.dim || seg.size() != pb.size()) perr("Error: image sizes do not match...");
  uint32_t* dLab = upload(seg.u32);
  float* dPb = upload(pb.f32);
  glia_hmt_ctx* ctx; glia_hmt_rag* rag;
  check(glia_hmt_ctx_create(0, nullptr, &ctx));
  uint32_t* dMask = loadMask(a, "maskImage", seg.size());
  check(glia_hmt_rag_build(ctx, seg.dim, seg.dims, dLab, dMask, /*only_contour=*/0, dPb, nullptr, &rag));   // :20
  int64_t cap = glia_hmt_rag_num_regions(rag), n = 0;
  std::vector<uint32_t> order(3 * (cap ? cap : 1)), src(2 * (cap ? cap : 1)), dst(2 * (cap ? cap : 1));
  std::vector<double> sal(cap ? cap : 1);
  check(glia_hmt_pre_merge(ctx, rag, sizes, (int)ts.size(), rpb, order.data(), sal.data(), cap, &n));
  int64_t m = glia_hmt_transform_keys(order.data(), n, src.data(), dst.data(), (int64_t)src.size());         // :77-78
  if (m < 0) perr(glia_hmt_last_error());
  check(glia_hmt_transform_image(ctx, dLab, (int64_t)seg.size(), src.data(), dst.data(), m, dMask, 0));      // :79 (region points only: unmasked voxels)
  uint32_t nl = 0;
  if (flagOf(a, "relabel")) check(glia_hmt_relabel_image(ctx, dLab, (int64_t)seg.size(), 0, &nl));           // :80
  hipCheck(hipMemcpy(seg.u32.data(), dLab, seg.size() * 4, hipMemcpyDeviceToHost));
  writeMetaImage(a.str("outputImage"), seg.dim, seg.dims, seg.u32, flagOf(a, "write16"), flagOf(a, "compress"));
  glia_hmt_rag_free(rag); glia_hmt_ctx_destroy(ctx);
  (void)hipFree(dLab); (void)hipFree(dPb);
  return EXIT_SUCCESS;
}
