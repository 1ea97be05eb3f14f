// oracle/ref_parfor_driver.cc -- TEST / BASELINE INFRASTRUCTURE ONLY (never linked into the product).
//
// The "GLIA_MT OpenMP baseline" of BASELINE.json's north star: the only OpenMP user on the hot path is bc_feat, whose two loops
// run under the reference's parfor (code/util/mp.hxx:24-106, `#pragma omp parallel for` over shuffled indices when GLIA_MT is
// defined).  util/mp.hxx needs no ITK: it is compiled IN PLACE from /root/reference here, twice --
//     ref_parfor_st:  g++ -O2                         (parfor = the serial loop, mp.hxx:41-42 / 72-73)
//     ref_parfor_mt:  g++ -O2 -DGLIA_MT -fopenmp      (the reference's OpenMP loops; threads from OMP_NUM_THREADS, maxThreads = 0 as
//                                                      every call site passes, main_bc_feat.cxx:71,95)
// -- and drives the oracle's restatement of RegionFeats::generate / getBoundary / BoundaryFeats::generate (the ITK-bound feature
// classes cannot be compiled here) over a given merge order exactly as code/hmt/main_bc_feat.cxx:57-101 does: region features of
// every region of the order's region map through parfor(rmap, shuffle = true, ...), then one boundary-feature job per merge through
// parfor(0, bn, shuffle = true, ...), then the serial serialisation.  The reference builds with no optimisation flag at all
// (CMakeLists.txt:24-31); -O2 is in its favour.
//
// usage: ref_parfor_{st,mt} <size> <S> [rows.bin]
//   synthetic size^3 volume (orc_synth: SURVEY 8d, S = supervoxel spacing, G = 8 S), feature configuration of the bench
//   (--rbi pb --rbb 8 --rbl 0 --rbu 1 --bt 0.2 0.5 0.8), merge order = the oracle's pb-mean order of the same volume.
//   stdout: one line  "threads T regions R merges M dim D seconds_regions A seconds_boundaries B seconds_total C rows_fnv H"
//   rows.bin (optional): the M x D feature rows as raw doubles (tests compare them with liboracle's orc_bc_feat).
#include <cmath>
#include <cstring>
#include "hmt_oracle.cc"          // the restatement, in this translation unit (its feature classes live in an anonymous namespace)
#include "util/mp.hxx"            // /root/reference/code/util/mp.hxx, unmodified (-I/root/reference/code)

#include <chrono>

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s size S [rows.bin]\n", argv[0]); return 2; }
  const int size = atoi(argv[1]), S = atoi(argv[2]);
  const int64_t dims[3] = {size, size, size};
  const int64_t N = (int64_t)size * size * size;
  std::vector<orc_label> labels(N);
  std::vector<float> pb(N);
  if (orc_synth(3, dims, S, 8 * S, 0x9E3779B97F4A7C15ull, 0, labels.data(), pb.data())) return 1;
  orc_rag* h = orc_rag_build(3, dims, labels.data(), nullptr, 0);
  if (!h) return 1;
  const int64_t R = orc_rag_num_regions(h);
  std::vector<orc_label> order(3 * (size_t)R);
  std::vector<double> sal(R);
  const int64_t n_merges = orc_merge_order_pb(h, pb.data(), 2, 0, order.data(), sal.data(), R);
  if (n_merges < 0) return 1;
  orc_feat_cfg c;
  memset(&c, 0, sizeof(c));
  c.n_rimg = 1; c.rimg[0] = pb.data(); c.rbins[0] = 8; c.rlo[0] = 0.0; c.rhi[0] = 1.0;
  c.n_bimg = 1; c.bimg[0] = pb.data(); c.bbins[0] = 8; c.blo[0] = 0.0; c.bhi[0] = 1.0;
  c.pb = pb.data(); c.n_thr = 3; c.thr[0] = 0.2; c.thr[1] = 0.5; c.thr[2] = 0.8;
  c.norm_area = 1.0; c.norm_len = 1.0;
  const int d = orc_feat_dim(3, &c);

  RegionMap& rmap = h->rmap;
  Cfg cfg = makeCfg(&h->vol, &c);
  const auto t0 = std::chrono::steady_clock::now();
  // main_bc_feat.cxx:57: RegionMap(seg, mask, order, false) -> set(order) (region_map.hxx:67-68)
  for (int64_t i = 0; i < n_merges; ++i) rmap.merge(order[3 * i], order[3 * i + 1], order[3 * i + 2]);
  const auto t1 = std::chrono::steady_clock::now();
  // main_bc_feat.cxx:58-72: region features, parfor over the region map, shuffled
  const int rn = (int)rmap.size();
  std::vector<std::pair<Label, std::shared_ptr<RegionFeats>>> rfeats(rn);
  glia::parfor(rmap, true, [&rfeats, &cfg](RegionMap::const_iterator rit, int i) {
    rfeats[i].first = rit->first;
    rfeats[i].second = std::make_shared<RegionFeats>();
    rfeats[i].second->generate(rit->second, cfg);
  }, 0);
  std::unordered_map<Label, std::shared_ptr<RegionFeats>> rfmap;
  for (auto const& rfp : rfeats) rfmap[rfp.first] = rfp.second;
  const auto t2 = std::chrono::steady_clock::now();
  // main_bc_feat.cxx:74-95: one boundary-feature job per merge, parfor over the order, shuffled
  const int bn = (int)n_merges;
  std::vector<BoundaryFeats> bfeats(bn);
  std::vector<std::array<RegionFeats*, 3>> xs(bn);
  glia::parfor(0, bn, true, [&rmap, &order, &bfeats, &xs, &rfmap, &cfg](int i) {
    Label r0 = order[3 * i], r1 = order[3 * i + 1], r2 = order[3 * i + 2];
    RegionFeats* x1 = rfmap.find(r0)->second.get();
    RegionFeats* x2 = rfmap.find(r1)->second.get();
    RegionFeats* x3 = rfmap.find(r2)->second.get();
    if (x1->area > x2->area) { std::swap(r0, r1); std::swap(x1, x2); }      // keep region 0 area <= region 1 area (:84-88)
    PtrPairMap b;
    getBoundary(b, rmap.find(r0)->second, rmap.find(r1)->second);
    bfeats[i].generate(b, *x1, *x2, cfg);
    xs[i] = {x1, x2, x3};
  }, 0);
  const auto t3 = std::chrono::steady_clock::now();
  std::vector<double> rows((size_t)bn * d);
  for (int i = 0; i < bn; ++i) {                                             // the writer's serialisation (:105-111), serial
    std::vector<double> f;
    bfeats[i].serialize(f); xs[i][0]->serialize(f); xs[i][1]->serialize(f); xs[i][2]->serialize(f);
    for (int k = 0; k < d; ++k) rows[(size_t)i * d + k] = f[k];
  }
  const auto t4 = std::chrono::steady_clock::now();
  unsigned long long fnv = 1469598103934665603ull;
  for (double v : rows) { unsigned long long u; memcpy(&u, &v, 8); for (int b = 0; b < 8; ++b) { fnv ^= (u >> (8 * b)) & 0xFF; fnv *= 1099511628211ull; } }
  if (argc > 3) { FILE* f = fopen(argv[3], "wb"); if (!f) return 1; fwrite(rows.data(), 8, rows.size(), f); fclose(f); }
  auto sec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  printf("threads %d regions %lld merges %lld dim %d seconds_regionmap %.4f seconds_regions %.4f seconds_boundaries %.4f seconds_serialize %.4f seconds_total %.4f rows_fnv %016llx\n",
         glia::nthreads(), (long long)R, (long long)n_merges, d, sec(t0, t1), sec(t1, t2), sec(t2, t3), sec(t3, t4), sec(t0, t4), fnv);
  orc_rag_free(h);
  return 0;
}
