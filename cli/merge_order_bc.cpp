// cli/merge_order_bc.cpp -- drop-in for hmt/main_merge_order_bc.cxx (boundary classifier type 1: random forest).
//   merge_order_bc --bct 1 --bcm model.bin [--bcm m1 --bcm m2 --bcmd d0 --bcmd d1 --bcmd thr] -s seg.mha --pb pb.mha
//                  [--rbi img --rbb bins --rbl lo --rbu hi] [--bt t0 t1 ..] [-n 0|1] [-l 0|1] [--simpf 0|1]
//                  -o order.txt [--sal saliency.txt] [-b feats.txt]
// Limits: at most eight images per list, four distinct (volume, histogram) channels in total, 384 feature columns before --simpf.
//   --slabs N [--rank r --commId file]: the z-slab route (cli/merge_order_pb.cpp, include/glia_hmt.h): every image volume is read by
//   plane range, slab by slab; rank 0 scores and merges.  No mask in this mode.
#include "common.hpp"

using namespace cli;

int main(int argc, char* argv[]) {
  const std::string usage = "Usage: merge_order_bc --bct 1 --bcm <model>... -s <seg> --pb <pb> [--rbi/--rbb/--rbl/--rbu ...] [--bt ...] "
                            "[-n b] [-l b] [--simpf b] -o <order> [--sal <file>] [-b <file>]   (flags as hmt/main_merge_order_bc.cxx:172-242)\n";
  std::vector<std::string> known = {"bct", "nn1", "nn2", "bcm", "bcfmm", "bcmd", "segImage", "rbi", "rbb", "rbl", "rbu", "rli", "rlb", "rll", "rlu",
                                    "ri", "rb", "rl", "ru", "bi", "bb", "bl", "bu", "pb", "maskImage", "bt", "ns", "logs", "simpf", "histf", "medf", "mergeOrder",
                                    "sal", "bfeat", "slabs", "rank", "commId", "commNonce", "device"};
  Args a = parse(argc, argv, {{"s", "segImage"}, {"m", "maskImage"}, {"n", "ns"}, {"l", "logs"}, {"o", "mergeOrder"}, {"b", "bfeat"}}, known, usage);
  for (const char* req : {"bct", "bcm", "segImage", "pb", "mergeOrder"})
    if (!a.has(req)) { std::cerr << "Error: the option '--" << req << "' is required but missing\n" << usage; perr("Error: unable to parse input arguments"); }
  if (atoi(a.str("bct").c_str()) != 1) perr("Error: unsupported classifier type...");     // MLP2 (--bct 2) is out of scope
  FeatInputs f;
  glia_hmt_ctx* ctx; glia_hmt_rag* rag = nullptr; glia_hmt_forest* bc;
  const int slabs = atoi(a.str("slabs", "0").c_str());
  if (slabs > 0) {
    if (a.has("maskImage")) perr("Error: the slab route takes no mask...");
    glia_hmt_comm* comm = makeSlabComm(a, slabs, &ctx);
    int ranks[256];
    const int nl = glia_hmt_comm_local_ranks(comm, ranks, 256);
    std::vector<FeatInputs> fs((size_t)nl);
    std::vector<glia_hmt_slab> sl((size_t)nl);
    int64_t nz = 0;
    for (int i = 0; i < nl; ++i) {
      nz = readMetaImage(a.str("segImage"), false, 0, 0).full_nz;
      if (slabs > nz) perr("Error: the slab route needs a 3D image with at least one plane per slab...");
      int64_t first, np, zb, ze;
      check(glia_hmt_slab_range(nz, slabs, ranks[i], &first, &np, &zb, &ze));
      loadFeatInputs(a, fs[i], first, np);
      memset(&sl[i], 0, sizeof(glia_hmt_slab));
      sl[i].dims_local[0] = fs[i].seg.dims[0]; sl[i].dims_local[1] = fs[i].seg.dims[1]; sl[i].dims_local[2] = np;
      sl[i].z_global_of_plane0 = first; sl[i].z_begin = zb; sl[i].z_end = ze; sl[i].d_labels = fs[i].dLab; sl[i].d_pb = fs[i].dPb; sl[i].cfg = &fs[i].cfg;
    }
    check(glia_hmt_rag_build_distributed(ctx, comm, sl.data(), nz, /*only_contour=*/0, /*with_values=*/0, /*loop_owner=*/0, &rag, nullptr));
    for (auto& x : fs) for (auto& kv : x.volumes) (void)hipFree(kv.second);
    for (auto& x : fs) (void)hipFree(x.dLab);
    glia_hmt_comm_destroy(comm);
    if (!rag) { glia_hmt_ctx_destroy(ctx); return EXIT_SUCCESS; }      // not the loop owner: done
  } else {
    loadFeatInputs(a, f);
    check(glia_hmt_ctx_create(0, nullptr, &ctx));
    uint32_t* dMask = loadMask(a, "maskImage", f.seg.size());
    check(glia_hmt_rag_build(ctx, f.seg.dim, f.seg.dims, f.dLab, dMask, /*only_contour=*/0, f.dPb, &f.cfg, &rag));
  }
  if (glia_hmt_ctx_libm_status(ctx) == 0) std::cerr << glia_hmt_last_error() << std::endl;     // host libm not reproduced: features within 1 ulp, unpinned
  auto models = a.all("bcm");
  std::vector<const char*> paths;
  for (auto& m : models) paths.push_back(m.c_str());
  double dist[3] = {0, 0, 0};
  if (models.size() > 1) {
    auto d = a.all("bcmd");
    if (d.size() != 3) perr("Error: model distributor needs 3 arguments...");   // :103-104
    for (int i = 0; i < 3; ++i) dist[i] = atof(d[i].c_str());
  }
  check(glia_hmt_forest_load(ctx, (int)paths.size(), paths.data(), /*BC_LABEL_MERGE*/ -1, models.size() > 1 ? dist : nullptr, &bc));
  int64_t cap = glia_hmt_rag_num_regions(rag), n = 0;
  const int d = glia_hmt_feat_dim(rag);
  std::vector<uint32_t> order(3 * (cap ? cap : 1));
  std::vector<double> sal(cap ? cap : 1), feats(a.has("bfeat") ? (size_t)(cap ? cap : 1) * d : 0);
  check(glia_hmt_merge_order_bc(ctx, rag, bc, order.data(), sal.data(), a.has("bfeat") ? feats.data() : nullptr, cap, &n));
  writeOrder(a.str("mergeOrder"), order, n);                                     // :148-157
  if (a.has("sal")) writeDoubles(a.str("sal"), sal.data(), n);
  if (a.has("bfeat")) writeRows(a.str("bfeat"), feats.data(), n, d, /*FLT_PREC*/ 8);
  glia_hmt_forest_free(bc); glia_hmt_rag_free(rag); glia_hmt_ctx_destroy(ctx);
  if (f.dLab) (void)hipFree(f.dLab);
  if (f.dPb) (void)hipFree(f.dPb);
  return EXIT_SUCCESS;
}
