// cli/merge_order_pb.cpp -- drop-in for hmt/main_merge_order_pb.cxx: same flags, same output files.
//   merge_order_pb -s seg.mha -p pb.mha [-m mask] [-t 1|2] [-o order.txt] [-y saliency.txt]
// The dimension is taken from the image (the reference fixes it at compile time, CMakeLists.txt:17-19).
//
// Volumes too big for one GPU (BASELINE.json: "slab-partitioned across the 8 MI355X with RCCL over xGMI exchanging only the
// cross-slab boundary regions"; no counterpart in GLIA): the z-slab route of glia_hmt_rag_build_distributed.
//   --slabs N                      N slabs, one after the other on this GPU (local communicator): every slab is read from the
//                                  file by its own plane range and only ONE slab's volumes are on the device at a time ...
//   --slabs N --rank r --commId f  ... or one process per slab and GPU, launched N times (r = 0..N-1, GPU = r modulo the visible
//                                  GPUs unless --device is given): rank 0 writes RCCL's unique id to file f, the others wait for it;
//                                  the records travel over RCCL, rank 0 runs the merge loop and writes the outputs.
// Median linkage (-t 1, the default) makes the boundary values travel with their pairs.  No mask in this mode.
#include "common.hpp"

using namespace cli;

int main(int argc, char* argv[]) {
  const std::string usage =
      "Usage:\n  --help                 Print usage info\n  -s [ --segImage ] arg  Input initial segmentation image file name\n"
      "  -p [ --pbImage ] arg   Input boundary probability image file name\n  -m [ --maskImage ] arg Input mask image file name (optional)\n"
      "  -t [ --type ] arg      Boundary intensity stats type (1: median, 2: mean) [default: 1]\n"
      "  -o [ --mergeOrder ] arg Output merging order file name (optional)\n  -y [ --saliency ] arg  Output merging saliency file name (optional)\n"
      "  --slabs arg            z slabs of the slab route (optional); with --rank arg --commId arg: one process per slab over RCCL\n";
  Args a = parse(argc, argv, {{"s", "segImage"}, {"p", "pbImage"}, {"m", "maskImage"}, {"t", "type"}, {"o", "mergeOrder"}, {"y", "saliency"}},
                 {"segImage", "pbImage", "maskImage", "type", "mergeOrder", "saliency", "slabs", "rank", "commId", "commNonce", "device"}, usage);
  if (!a.has("segImage") || !a.has("pbImage")) { std::cerr << "Error: the option '--segImage'/'--pbImage' is required but missing\n" << usage; return EXIT_FAILURE; }
  const int type = atoi(a.str("type", "1").c_str());
  if (type != 1 && type != 2) perr("Error: unsupported boundary stats type...");          // :36
  glia_hmt_ctx* ctx; glia_hmt_rag* rag = nullptr;
  const int slabs = atoi(a.str("slabs", "0").c_str());
  std::vector<void*> dev;
  if (slabs > 0) {
    if (a.has("maskImage")) perr("Error: the slab route takes no mask...");
    glia_hmt_comm* comm = makeSlabComm(a, slabs, &ctx);
    int ranks[256];
    const int nl = glia_hmt_comm_local_ranks(comm, ranks, 256);
    // every local rank's planes straight from the files (owned planes + one halo plane per cut)
    int64_t nz = 0;
    std::vector<glia_hmt_slab> sl((size_t)nl);
    for (int i = 0; i < nl; ++i) {
      Volume head = readMetaImage(a.str("segImage"), false, 0, 0);
      nz = head.full_nz;
      if (head.dim != 3 || slabs > nz) perr("Error: the slab route needs a 3D image with at least one plane per slab...");
      int64_t first, np, zb, ze;
      check(glia_hmt_slab_range(nz, slabs, ranks[i], &first, &np, &zb, &ze));
      Volume seg = readMetaImage(a.str("segImage"), false, first, np), pb = readMetaImage(a.str("pbImage"), true, first, np);
      if (seg.size() != pb.size() || pb.full_nz != nz) perr("Error: image sizes do not match...");
      uint32_t* dLab = upload(seg.u32);
      float* dPb = upload(pb.f32);
      dev.push_back(dLab); dev.push_back(dPb);
      memset(&sl[i], 0, sizeof(glia_hmt_slab));
      sl[i].dims_local[0] = seg.dims[0]; sl[i].dims_local[1] = seg.dims[1]; sl[i].dims_local[2] = np;
      sl[i].z_global_of_plane0 = first; sl[i].z_begin = zb; sl[i].z_end = ze; sl[i].d_labels = dLab; sl[i].d_pb = dPb;
    }
    glia_hmt_dist_stats st;
    check(glia_hmt_rag_build_distributed(ctx, comm, sl.data(), nz, /*only_contour=*/1, /*with_values=*/type == 1, /*loop_owner=*/0, &rag, &st));
    for (void* p : dev) (void)hipFree(p);
    dev.clear();
    glia_hmt_comm_destroy(comm);
    if (!rag) { glia_hmt_ctx_destroy(ctx); return EXIT_SUCCESS; }      // not the loop owner: done
  } else {
    Volume seg = readMetaImage(a.str("segImage"), false), pb = readMetaImage(a.str("pbImage"), true);
    if (seg.dim != pb.dim || seg.size() != pb.size()) perr("Error: image sizes do not match...");
    uint32_t* dLab = upload(seg.u32);
    float* dPb = upload(pb.f32);
    dev.push_back(dLab); dev.push_back(dPb);
    check(glia_hmt_ctx_create(0, nullptr, &ctx));
    uint32_t* dMask = loadMask(a, "maskImage", seg.size());
    check(glia_hmt_rag_build(ctx, seg.dim, seg.dims, dLab, dMask, /*only_contour=*/1, dPb, nullptr, &rag));   // :27
  }
  int64_t cap = glia_hmt_rag_num_regions(rag), n = 0;
  std::vector<uint32_t> order(3 * (cap ? cap : 1));
  std::vector<double> sal(cap ? cap : 1);
  check(glia_hmt_merge_order_pb(ctx, rag, type, order.data(), sal.data(), cap, &n));
  if (a.has("mergeOrder")) writeOrder(a.str("mergeOrder"), order, n);                    // :37-38
  if (a.has("saliency")) writeDoubles(a.str("saliency"), sal.data(), n);
  glia_hmt_rag_free(rag); glia_hmt_ctx_destroy(ctx);
  for (void* p : dev) (void)hipFree(p);
  return EXIT_SUCCESS;
}
