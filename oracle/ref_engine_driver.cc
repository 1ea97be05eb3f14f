// oracle/ref_engine_driver.cc -- TEST INFRASTRUCTURE ONLY.
//
// Drives the reference's OWN greedy-merge engine, compiled in place from
// /root/reference/code (nothing is copied into this repository):
//   type/boundary_table.hxx  (TBoundaryTable::init/top/update, the multimap tie rule)
//   type/region.hxx, type/region_map.hxx (TRegion::merge / boundaryWith, TRegionMap::init/merge)
//   util/struct_merge.hxx:13-33 (genMergeOrderGreedy)
// These headers are ITK-free except that type/region_map.hxx includes util/struct.hxx
// (RAG construction from an ITK image).  The build passes -D_glia_util_struct_hxx_ so that
// this one header is skipped: nothing is written in its place, and the only reference
// symbols that go missing are the ITK-based RAG builders, which are never instantiated here.
// The region map's three public leaf maps are filled from a dump of our own RAG, and the
// linkage lambdas below restate util/struct_merge.hxx:45-76 (mean) and :98-132 (median).
//
// Round 3: the size-rule condition of gadget/main_pre_merge.cxx:27-76 is restated here as the fcond handed to the reference's
// TBoundaryTable::top (type/boundary_table.hxx:46-52), so that the condition path of the reference's own queue walk runs, and the
// reference's transformKeys (util/struct_merge.hxx:188-210, same translation unit) is printed for the order it produced.
//
// stdin:  R  P  B  type(1 median / 2 mean) updateRegion
//         R lines:  label nPoints  p0 p1 ...          (voxel ids, raster order)
//         P lines:  label nBorder  p0 p1 ...
//         B lines:  a b n  p0 p1 ...                  (directed boundary voxel ids)
//         then N pb values (N = number of voxels, first line gives N)
//         optionally:  nThresholds t0 [t1] rpbThreshold      (type 2 only: the pre_merge condition, updateRegion is forced on)
// stdout: one "x0 x1 x2 saliency" line per merge (saliency printed with %.17g), then one "K src dst" line per entry of
//         transformKeys(order), sorted by src
#include <chrono>
#include <cmath>
#include <cstring>
#include <cstdio>
#include "util/struct_merge.hxx"

using namespace glia;

struct VoxelId {              // the TPoint template argument: our own voxel id type
  long id;
  VoxelId() : id(0) {}
  VoxelId(long i) : id(i) {}
};

typedef TRegionMap<Label, VoxelId> RegionMap;

int main() {
  long N; int R, P, B, type, updateRegion;
  if (scanf("%ld %d %d %d %d %d", &N, &R, &P, &B, &type, &updateRegion) != 6) return 2;
  RegionMap rmap;
  for (int i = 0; i < R; ++i) {
    unsigned lab; long n;
    if (scanf("%u %ld", &lab, &n) != 2) return 2;
    auto& v = (*rmap.pPointMap)[lab];
    v.reserve(n);
    for (long k = 0; k < n; ++k) { long p; if (scanf("%ld", &p) != 1) return 2; v.push_back(VoxelId(p)); }
  }
  for (int i = 0; i < P; ++i) {
    unsigned lab; long n;
    if (scanf("%u %ld", &lab, &n) != 2) return 2;
    auto& v = (*rmap.pBorderMap)[lab];
    for (long k = 0; k < n; ++k) { long p; if (scanf("%ld", &p) != 1) return 2; v.push_back(VoxelId(p)); }
  }
  for (int i = 0; i < B; ++i) {
    unsigned a, b; long n;
    if (scanf("%u %u %ld", &a, &b, &n) != 3) return 2;
    auto& v = (*rmap.pBoundaryMap)[std::make_pair((Label)a, (Label)b)];
    for (long k = 0; k < n; ++k) { long p; if (scanf("%ld", &p) != 1) return 2; v.push_back(VoxelId(p)); }
  }
  std::vector<float> pb(N);
  for (long i = 0; i < N; ++i) if (scanf("%f", &pb[i]) != 1) return 2;
  if (R > 0) rmap.init(); else rmap.initContour();
  int nThr = 0;
  std::vector<int> sizeThresholds;
  double rpbThreshold = 0.0;
  if (scanf("%d", &nThr) == 1 && nThr > 0) {
    sizeThresholds.resize(nThr);
    for (int i = 0; i < nThr; ++i) if (scanf("%d", &sizeThresholds[i]) != 1) return 2;
    if (scanf("%lf", &rpbThreshold) != 1) return 2;
  }

  std::vector<TTriple<Label>> order;
  std::vector<double> sal;
  typedef RegionMap::Region::Boundary Boundary;
  if (type == 2) {
    typedef std::pair<double, int> ItemData;
    typedef TBoundaryTable<ItemData, RegionMap> BT;
    auto initFb = [&](ItemData& data, Label r0, Label r1) {      // util/struct_merge.hxx:45-56
      Boundary b;
      rmap.find(r0)->second.boundaryWith(b, rmap.find(r1)->second);   // util/struct.hxx:10-16
      rmap.find(r1)->second.boundaryWith(b, rmap.find(r0)->second);
      data.first = 0.0;
      b.traverse([&](VoxelId const& p) { data.first += pb[p.id]; });
      data.second = b.size();
      data.first = sdivide(data.first, data.second, 0.0);
    };
    auto fsal = [](ItemData& data, Label, Label) -> double { return -data.first; };
    auto updateFb = [](ItemData& d2, Label, Label, Label, Label, ItemData* p0, ItemData* p1) {  // :62-76
      d2.first = 0.0; d2.second = 0;
      if (p0) { d2.first += p0->first * p0->second; d2.second += p0->second; }
      if (p1) { d2.first += p1->first * p1->second; d2.second += p1->second; }
      d2.first = sdivide(d2.first, d2.second, 0.0);
    };
    const auto t0 = std::chrono::steady_clock::now();
    if (nThr > 0) {
      // gadget/main_pre_merge.cxx:27-76 with pbImage->GetPixel(p) = pb[p.id]
      std::unordered_map<Label, double> rpbs;
      auto fcond = [&](BT const&, BT::iterator btit) -> bool {
        Label key0 = btit->first.first, key1 = btit->first.second;
        auto const* pr0 = &rmap.find(key0)->second;
        auto const* pr1 = &rmap.find(key1)->second;
        auto sz0 = pr0->size(), sz1 = pr1->size();
        if (sz0 > sz1) { std::swap(key0, key1); std::swap(pr0, pr1); std::swap(sz0, sz1); }
        if (sz0 < sizeThresholds[0]) return true;
        if (sizeThresholds.size() > 1) {
          auto test = [&](Label key, decltype(pr0) pr, decltype(sz0) sz) -> bool {
            auto it = rpbs.find(key);
            if (it != rpbs.end()) return it->second > rpbThreshold;
            double rpb = 0.0;
            pr->traverse([&](VoxelId const& p) { rpb += pb[p.id]; });
            rpb = sdivide(rpb, sz, 0.0);
            rpbs[key] = rpb;
            return rpb > rpbThreshold;
          };
          if (sz0 < sizeThresholds[1] && test(key0, pr0, sz0)) return true;
          if (sz1 < sizeThresholds[1] && test(key1, pr1, sz1)) return true;
        }
        return false;
      };
      genMergeOrderGreedy<ItemData>(order, sal, rmap, true, initFb, fsal, updateFb, fsal, fcond);
    } else
    genMergeOrderGreedy<ItemData>(order, sal, rmap, updateRegion != 0, initFb, fsal, updateFb, fsal,
                                  f_true<BT&, BT::iterator>);
    fprintf(stderr, "engine_seconds %.6f\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  } else {
    typedef std::vector<double> ItemData;
    typedef TBoundaryTable<ItemData, RegionMap> BT;
    auto initFb = [&](ItemData& data, Label r0, Label r1) {      // :98-111
      Boundary b;
      rmap.find(r0)->second.boundaryWith(b, rmap.find(r1)->second);
      rmap.find(r1)->second.boundaryWith(b, rmap.find(r0)->second);
      data.reserve(b.size());
      b.traverse([&](VoxelId const& p) { data.push_back(pb[p.id]); });
    };
    auto fsal = [](ItemData& data, Label, Label) -> double { return -stats::amedian(data); };
    auto updateFb = [](ItemData& d2, Label, Label, Label, Label, ItemData* p0, ItemData* p1) {  // :118-132
      if (p0 && p1) splice(d2, *p0, *p1);
      else if (p0) splice(d2, *p0);
      else if (p1) splice(d2, *p1);
    };
    const auto t0 = std::chrono::steady_clock::now();
    genMergeOrderGreedy<ItemData>(order, sal, rmap, updateRegion != 0, initFb, fsal, updateFb, fsal,
                                  f_true<BT&, BT::iterator>);
    fprintf(stderr, "engine_seconds %.6f\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  }
  for (size_t i = 0; i < order.size(); ++i)
    printf("%u %u %u %.17g\n", order[i].x0, order[i].x1, order[i].x2, sal[i]);
  std::unordered_map<Label, Label> lmap;
  transformKeys(lmap, order);                                     // util/struct_merge.hxx:188-210
  std::map<Label, Label> sorted(lmap.begin(), lmap.end());
  for (auto const& kv : sorted) printf("K %u %u\n", kv.first, kv.second);
  return 0;
}
