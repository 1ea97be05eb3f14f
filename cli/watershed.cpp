// cli/watershed.cpp -- drop-in for gadget/main_watershed.cxx: morphological watershed over-segmentation of a float image.
//   watershed -i pb.mha -l level [-r 0|1] [-u 0|1] -o seg.mha
// (glia::watershed, util/image_alg.hxx:9-21; tie rules: see include/glia_hmt.h -- labels are not pinned against ITK)
#include "common.hpp"

using namespace cli;

int main(int argc, char* argv[]) {
  const std::string usage = "Usage: watershed -i <image> -l <level> [-r b] [-u b] [-z b] -o <out>   (flags as gadget/main_watershed.cxx:28-44)\n";
  Args a = parse(argc, argv, {{"i", "inputImage"}, {"l", "level"}, {"r", "relabel"}, {"u", "write16"}, {"z", "compress"}, {"o", "outputImage"}},
                 {"inputImage", "level", "relabel", "write16", "compress", "outputImage"}, usage);
  for (const char* req : {"inputImage", "level", "outputImage"})
    if (!a.has(req)) { std::cerr << "Error: the option '--" << req << "' is required but missing\n" << usage; return EXIT_FAILURE; }
  Volume img = readMetaImage(a.str("inputImage"), true);
  float* dImg = upload(img.f32);
  uint32_t* dLab;
  hipCheck(hipMalloc(&dLab, img.size() * 4));
  glia_hmt_ctx* ctx;
  check(glia_hmt_ctx_create(0, nullptr, &ctx));
  uint32_t nl = 0;
  check(glia_hmt_watershed(ctx, img.dim, img.dims, dImg, atof(a.str("level").c_str()), dLab, &nl, nullptr));     // :10
  if (flagOf(a, "relabel")) check(glia_hmt_relabel_image(ctx, dLab, (int64_t)img.size(), 0, &nl));                // :11
  std::vector<uint32_t> out(img.size());
  hipCheck(hipMemcpy(out.data(), dLab, img.size() * 4, hipMemcpyDeviceToHost));
  writeMetaImage(a.str("outputImage"), img.dim, img.dims, out, flagOf(a, "write16"), flagOf(a, "compress"));                               // :12-16
  glia_hmt_ctx_destroy(ctx);
  (void)hipFree(dLab); (void)hipFree(dImg);
  return EXIT_SUCCESS;
}
