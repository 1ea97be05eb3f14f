// cli/text_io_check.cpp -- host-only test helper: writes an order, saliencies and feature rows read from stdin with the tools'
// writers (cli/text_io.hpp) so that the CPU suite can compare the bytes with the reference's writeData.
//   text_io_check <dir>     stdin as oracle/ref_misc_driver.cc "write"
#include <cstdio>
#include "text_io.hpp"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string dir = argv[1];
  long n;
  if (scanf("%ld", &n) != 1) return 2;
  std::vector<uint32_t> order(3 * n);
  for (long i = 0; i < 3 * n; ++i) if (scanf("%u", &order[i]) != 1) return 2;
  cli::writeOrder(dir + "/order.txt", order, n);
  if (scanf("%ld", &n) != 1) return 2;
  std::vector<double> sal(n);
  for (long i = 0; i < n; ++i) if (scanf("%lf", &sal[i]) != 1) return 2;
  cli::writeDoubles(dir + "/sal.txt", sal.data(), n);
  long rows, cols;
  if (scanf("%ld %ld", &rows, &cols) != 2) return 2;
  std::vector<double> f(rows * cols);
  for (long i = 0; i < rows * cols; ++i) if (scanf("%lf", &f[i]) != 1) return 2;
  cli::writeRows(dir + "/feats.txt", f.data(), rows, (int)cols, 8);      // FLT_PREC, glia_base.hxx:61
  return 0;
}
