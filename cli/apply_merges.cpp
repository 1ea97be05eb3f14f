// cli/apply_merges.cpp -- drop-in for gadget/main_apply_merges.cxx: applies merge-order file(s) to a label image.
//   apply_merges -i in.mha [-m mask.mha] -g order.txt [-g more.txt] [-r 0|1] [-u 0|1] -o out.mha
#include <algorithm>

#include "common.hpp"

using namespace cli;

int main(int argc, char* argv[]) {
  const std::string usage = "Usage: apply_merges -i <image> [-m <mask>] -g <order>... [-r b] [-u b] [-z b] -o <out>   (flags as gadget/main_apply_merges.cxx:50-66)\n";
  Args a = parse(argc, argv, {{"i", "inputImage"}, {"m", "mask"}, {"g", "merge"}, {"r", "relabel"}, {"u", "write16"}, {"z", "compress"}, {"o", "outputImage"}},
                 {"inputImage", "mask", "merge", "relabel", "write16", "compress", "outputImage"}, usage);
  for (const char* req : {"inputImage", "merge", "outputImage"})
    if (!a.has(req)) { std::cerr << "Error: the option '--" << req << "' is required but missing\n" << usage; return EXIT_FAILURE; }
  struct T3 { uint32_t x0, x1, x2; };
  std::vector<T3> merges;
  for (auto& file : a.all("merge")) {
    std::vector<uint32_t> o = readOrder(file);
    for (size_t i = 0; i + 2 < o.size(); i += 3) merges.push_back({o[i], o[i + 1], o[i + 2]});
  }
  std::sort(merges.begin(), merges.end(), [](T3 const& m0, T3 const& m1) { return m0.x2 < m1.x2; });       // :10-11,26
  std::vector<uint32_t> order;
  for (auto& m : merges) { order.push_back(m.x0); order.push_back(m.x1); order.push_back(m.x2); }
  const int64_t n = (int64_t)merges.size();
  std::vector<uint32_t> src(2 * (n ? n : 1)), dst(2 * (n ? n : 1));
  int64_t m = glia_hmt_transform_keys(order.data(), n, src.data(), dst.data(), (int64_t)src.size());         // :27-28
  if (m < 0) perr(glia_hmt_last_error());
  Volume img = readMetaImage(a.str("inputImage"), false);
  uint32_t* dLab = upload(img.u32);
  uint32_t* dMask = nullptr;
  if (a.has("mask")) {
    Volume mask = readMetaImage(a.str("mask"), false);
    if (mask.size() != img.size()) perr("Error: image sizes do not match...");
    dMask = upload(mask.u32);
  }
  glia_hmt_ctx* ctx;
  check(glia_hmt_ctx_create(0, nullptr, &ctx));
  check(glia_hmt_transform_image(ctx, dLab, (int64_t)img.size(), src.data(), dst.data(), m, dMask, /*fillMissing=*/0));   // :33
  uint32_t nl = 0;
  if (flagOf(a, "relabel")) check(glia_hmt_relabel_image(ctx, dLab, (int64_t)img.size(), 0, &nl));
  hipCheck(hipMemcpy(img.u32.data(), dLab, img.size() * 4, hipMemcpyDeviceToHost));
  writeMetaImage(a.str("outputImage"), img.dim, img.dims, img.u32, flagOf(a, "write16"), flagOf(a, "compress"));
  glia_hmt_ctx_destroy(ctx);
  (void)hipFree(dLab);
  if (dMask) (void)hipFree(dMask);
  return EXIT_SUCCESS;
}
