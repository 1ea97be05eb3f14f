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
// stdin:  R  P  B  type(1 median / 2 mean) updateRegion
//         R lines:  label nPoints  p0 p1 ...          (voxel ids, raster order)
//         P lines:  label nBorder  p0 p1 ...
//         B lines:  a b n  p0 p1 ...                  (directed boundary voxel ids)
//         then N pb values (N = number of voxels, first line gives N)
// stdout: one "x0 x1 x2 saliency" line per merge (saliency printed with %.17g)
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
  return 0;
}
