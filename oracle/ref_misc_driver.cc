// oracle/ref_misc_driver.cc -- TEST INFRASTRUCTURE ONLY.
//
// Two small ITK-free pieces of the reference, compiled in place from /root/reference/code (nothing is copied):
//   type/function.hxx:71-85     opt::ThresholdModelDistributor (which ensemble member scores a vector)
//   util/text_io.hxx:103-133    writeData (merge order as TTriple lines, saliencies at default precision, feature rows at
//                               FLT_PREC, glia_base.hxx:61) and :193-221 readData (the round trip the tools rely on)
//
//   ref_misc dist              stdin: dim0 dim1 threshold n D, then n rows of D doubles       stdout: one model index per row
//   ref_misc write <dir>       stdin: nOrder, nOrder triples; nSal, nSal doubles; rows cols, rows*cols doubles
//                              writes <dir>/order.txt, <dir>/sal.txt, <dir>/feats.txt with the reference's writers as the
//                              tools call them (hmt/main_merge_order_pb.cxx:37-38, hmt/main_merge_order_bc.cxx:148-157), reads
//                              them back with readData and prints "roundtrip <order ok> <sal ok> <rows> <cols> <values>"
#include <cmath>
#include <cstdio>
#include <cstring>
#include "type/function.hxx"
#include "type/tuple.hxx"
#include "util/text_io.hxx"

using namespace glia;

int main(int argc, char* argv[]) {
  if (argc >= 2 && !strcmp(argv[1], "dist")) {
    int dim0, dim1, n, D; double thr;
    if (scanf("%d %d %lf %d %d", &dim0, &dim1, &thr, &n, &D) != 5) return 2;
    opt::ThresholdModelDistributor<double> f(dim0, dim1, thr);
    std::vector<double> x(D);
    for (int i = 0; i < n; ++i) {
      for (int k = 0; k < D; ++k) if (scanf("%lf", &x[k]) != 1) return 2;
      printf("%d\n", f(x));
    }
    return 0;
  }
  if (argc >= 3 && !strcmp(argv[1], "write")) {
    const std::string dir = argv[2];
    long n;
    if (scanf("%ld", &n) != 1) return 2;
    std::vector<TTriple<Label>> order(n);
    for (long i = 0; i < n; ++i) { unsigned a, b, c; if (scanf("%u %u %u", &a, &b, &c) != 3) return 2; order[i] = TTriple<Label>(a, b, c); }
    if (scanf("%ld", &n) != 1) return 2;
    std::vector<double> sal(n);
    for (long i = 0; i < n; ++i) if (scanf("%lf", &sal[i]) != 1) return 2;
    long rows, cols;
    if (scanf("%ld %ld", &rows, &cols) != 2) return 2;
    std::vector<std::vector<FVal>> feats(rows, std::vector<FVal>(cols));
    for (long i = 0; i < rows; ++i) for (long k = 0; k < cols; ++k) if (scanf("%lf", &feats[i][k]) != 1) return 2;
    writeData(dir + "/order.txt", order, "\n");                       // hmt/main_merge_order_pb.cxx:37
    writeData(dir + "/sal.txt", sal, "\n");                           // :38
    writeData(dir + "/feats.txt", feats, " ", "\n", FLT_PREC);        // hmt/main_merge_order_bc.cxx:156
    std::vector<TTriple<Label>> order2;
    readData(order2, dir + "/order.txt", true);                       // hmt/main_bc_feat.cxx:47
    bool ok = order2.size() == order.size();
    for (size_t i = 0; ok && i < order.size(); ++i) ok = order2[i].x0 == order[i].x0 && order2[i].x1 == order[i].x1 && order2[i].x2 == order[i].x2;
    std::vector<double> sal2;
    readData(sal2, dir + "/sal.txt", true);
    bool sok = sal2.size() == sal.size();
    for (size_t i = 0; sok && i < sal.size(); ++i) sok = std::fabs(sal2[i] - sal[i]) <= 1e-5 * std::fabs(sal[i]) + 1e-300;
    std::vector<double> flat;
    std::pair<int, int> size;
    readData(flat, size, dir + "/feats.txt", false);
    printf("roundtrip %d %d %d %d %zu\n", (int)ok, (int)sok, size.first, size.second, flat.size());
    return 0;
  }
  return 2;
}
