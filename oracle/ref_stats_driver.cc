// oracle/ref_stats_driver.cc -- TEST INFRASTRUCTURE ONLY.
//
// Drives the reference's OWN statistics helpers, compiled in place from /root/reference/code/util/stats.hxx
// (ITK-free, nothing is copied): amedian :83-91, entropy :145-152, distL1 :155-163, distX2 :177-185, rescale :264-277.
// stdin:  nCases; per case: n, then n values a[i], then n values b[i]
// stdout: per case one line "entropy(a) entropy(b) distL1(a,b) distX2(a,b) amedian(a) amedian(b)" and one line with
//         rescale(a; min = min(a,b) per element, max = max(a,b) per element, -1, 1), all %.17g
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <list>
#include <map>
#include <set>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include "util/stats.hxx"

using namespace glia;

int main() {
  int nCases;
  if (scanf("%d", &nCases) != 1) return 2;
  for (int c = 0; c < nCases; ++c) {
    int n;
    if (scanf("%d", &n) != 1) return 2;
    std::vector<double> a(n), b(n);
    for (auto& v : a) if (scanf("%lf", &v) != 1) return 2;
    for (auto& v : b) if (scanf("%lf", &v) != 1) return 2;
    std::vector<double> ma = a, mb = b;
    printf("%.17g %.17g %.17g %.17g", stats::entropy(a), stats::entropy(b), stats::distL1(a, b), stats::distX2(a, b));
    printf(" %.17g %.17g\n", stats::amedian(ma), stats::amedian(mb));
    std::vector<std::vector<FVal>> minmax(2, std::vector<FVal>(n));
    for (int i = 0; i < n; ++i) { minmax[0][i] = std::min(a[i], b[i]); minmax[1][i] = std::max(a[i], b[i]); }
    std::vector<double> r = a;
    stats::rescale(r, minmax, -1.0, 1.0);
    for (int i = 0; i < n; ++i) printf(i ? " %.17g" : "%.17g", r[i]);
    printf("\n");
  }
  return 0;
}
