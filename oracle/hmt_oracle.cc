// oracle/hmt_oracle.cc -- see hmt_oracle.hpp for status and usage rules.
// TEST INFRASTRUCTURE ONLY.  All reference citations are relative to
// /root/reference/code/.
#include "hmt_oracle.hpp"

#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <deque>
#include <limits>
#include <queue>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

namespace {

typedef orc_label Label;
typedef std::vector<int64_t> Points;           // voxel linear indices, raster order (x fastest)
typedef std::pair<Label, Label> LPair;

const Label MASK_OUT_VAL = 0;                  // glia_image.hxx:28
const Label BG_VAL = 0;                        // glia_image.hxx:27
const double DUMMY = -1.0;                     // glia_base.hxx:56
const double FEPS = 2.22e-16;                  // glia_base.hxx:57
const double FMAX = DBL_MAX;                   // glia_base.hxx:59

inline double sdivide(double l, double r, double d) { return std::fabs(r) >= FEPS ? l / r : d; }  // glia_base.hxx:77-78
inline double slog(double x, double d) { return x > 0.0 ? std::log(x) : d; }                      // glia_base.hxx:80-81
inline double ssqrt(double x, double d) { return x >= 0.0 ? std::sqrt(x) : d; }                   // glia_base.hxx:83-84
inline bool isfeq(double a, double b) { return std::fabs(a - b) < FEPS; }                         // glia_base.hxx:71-72

// type/hash.hxx:6-23
struct PairHash {
  static void combine(std::size_t& seed, Label x) {
    std::hash<Label> hasher;
    seed ^= hasher(x) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
  }
  std::size_t operator()(LPair const& p) const {
    std::size_t seed = 0;
    combine(seed, p.first);
    combine(seed, p.second);
    return seed;
  }
};

struct Vol {
  int D;
  int64_t n[3];
  const Label* lab;
  const Label* mask;
  int64_t size() const { return n[0] * n[1] * n[2]; }
  void coords(int64_t idx, int64_t c[3]) const {
    c[0] = idx % n[0];
    c[1] = (idx / n[0]) % n[1];
    c[2] = idx / (n[0] * n[1]);
  }
  int64_t stride(int d) const { return d == 0 ? 1 : (d == 1 ? n[0] : n[0] * n[1]); }
};

// type/neighbor.hxx:72-126  (traverseNeighbors + getNeighborValues + getContourTraits)
// Neighbour order -x,+x,-y,+y,(-z,+z); a neighbour is valid when inside the image
// and not masked out.  Returns (first differing neighbour value or own value, onBorder).
inline std::pair<Label, bool> contourTraits(Vol const& v, int64_t idx) {
  int64_t c[3];
  v.coords(idx, c);
  Label thisVal = v.lab[idx];
  Label nvs[6];
  int nn = 0;
  for (int i = 0; i < v.D; ++i) {
    int64_t s = v.stride(i);
    if (c[i] - 1 >= 0) {
      int64_t j = idx - s;
      if (!v.mask || v.mask[j] != MASK_OUT_VAL) nvs[nn++] = v.lab[j];
    }
    if (c[i] + 1 < v.n[i]) {
      int64_t j = idx + s;
      if (!v.mask || v.mask[j] != MASK_OUT_VAL) nvs[nn++] = v.lab[j];
    }
  }
  std::pair<Label, bool> ret(thisVal, nn < (v.D << 1));
  for (int k = 0; k < nn; ++k) {
    if (nvs[k] != thisVal) { ret.first = nvs[k]; break; }
  }
  return ret;
}

typedef std::unordered_map<Label, Points> PointMap;                 // type/point_map.hxx:11-24

struct PointPairMap : std::unordered_map<LPair, Points, PairHash> {  // type/point_map.hxx:27-79
  std::unordered_map<Label, std::vector<std::pair<Label, Points*>>> umap;
  void prepare() {                                                   // :53-60
    if (!umap.empty()) return;
    for (auto& pp : *this) umap[pp.first.first].push_back(std::make_pair(pp.first.second, &pp.second));
  }
};

typedef std::unordered_map<Label, Points const*> PtrMap;             // type/point_map.hxx:82-126
typedef std::unordered_map<LPair, Points const*, PairHash> PtrPairMap;  // :129-181

template <typename M> int64_t mapSize(M const& m) {
  int64_t r = 0;
  for (auto const& pp : m) r += (int64_t)pp.second->size();
  return r;
}
template <typename M, typename F> void traverse(M const& m, F f) {
  for (auto const& pp : m) for (auto p : *pp.second) f(p);
}

// type/region.hxx:8-87
struct Region {
  PtrMap pts;
  PtrMap border;
  PtrPairMap boundary;

  int64_t size() const { return mapSize(pts); }

  // :42-51  boundary on this region's side
  void boundaryWith(PtrPairMap& b, Region const& region) const {
    for (auto const& bp0 : boundary) {
      for (auto const& bp1 : region.boundary) {
        if (bp0.first.second == bp1.first.first) {
          if (bp0.second) b.insert(bp0);
          break;
        }
      }
    }
  }

  // :66-75
  void merge(Region const& region) {
    for (auto const& pp : region.pts) if (pp.second) pts.insert(pp);
    for (auto const& pp : region.border) if (pp.second) border.insert(pp);
    for (auto const& pp : region.boundary) {
      auto it = boundary.find(std::make_pair(pp.first.second, pp.first.first));
      if (it == boundary.end()) { if (pp.second) boundary.insert(pp); }
      else boundary.erase(it);
    }
  }
};

// util/struct.hxx:10-16
inline void getBoundary(PtrPairMap& b01, Region const& r0, Region const& r1) {
  r0.boundaryWith(b01, r1);
  r1.boundaryWith(b01, r0);
}

// type/region_map.hxx:11-133
struct RegionMap : std::unordered_map<Label, Region> {
  typedef std::unordered_map<Label, Region> Super;
  std::shared_ptr<PointMap> pPointMap = std::make_shared<PointMap>();
  std::shared_ptr<PointMap> pBorderMap = std::make_shared<PointMap>();
  std::shared_ptr<PointPairMap> pBoundaryMap = std::make_shared<PointPairMap>();

  Label maxKey() const {                                              // :70-75
    Label ret = Super::begin()->first;
    for (auto const& rp : *this) if (ret < rp.first) ret = rp.first;
    return ret;
  }

  void init() {                                                       // :79-95
    Super::clear();
    pBoundaryMap->prepare();
    for (auto& pp : *pPointMap) {
      Region reg;
      reg.pts.insert(std::make_pair(pp.first, (Points const*)&pp.second));
      auto it = Super::emplace(pp.first, reg).first;
      auto bit = pBorderMap->find(pp.first);
      if (bit != pBorderMap->end()) it->second.border[pp.first] = &bit->second;
      auto bnit = pBoundaryMap->umap.find(pp.first);
      if (bnit != pBoundaryMap->umap.end()) {
        for (auto const& bp : bnit->second)
          it->second.boundary.emplace(std::make_pair(pp.first, bp.first), (Points const*)bp.second);
      }
    }
  }

  void initContour() {                                                // :99-111
    Super::clear();
    pBoundaryMap->prepare();
    for (auto& pp : pBoundaryMap->umap) {
      auto it = Super::insert(std::make_pair(pp.first, Region())).first;
      auto bit = pBorderMap->find(pp.first);
      if (bit != pBorderMap->end()) it->second.border[pp.first] = &bit->second;
      for (auto const& bp : pp.second)
        it->second.boundary.emplace(std::make_pair(pp.first, bp.first), (Points const*)bp.second);
    }
  }

  Super::iterator merge(Label r0, Label r1, Label r2) {               // :113-118
    auto it = Super::find(r2);
    if (it == Super::end()) it = Super::insert(Super::end(), std::make_pair(r2, Region()));
    if (r2 != r0) it->second.merge(Super::find(r0)->second);
    if (r2 != r1) it->second.merge(Super::find(r1)->second);
    return it;
  }
};

// util/struct.hxx:61-92
void genPointMap(PointMap& pmap, Vol const& v) {
  std::unordered_map<Label, std::size_t> cmap;
  int64_t N = v.size();
  for (int64_t i = 0; i < N; ++i) {
    if (!v.mask || v.mask[i] != MASK_OUT_VAL) {
      Label key = v.lab[i];
      auto cit = cmap.find(key);
      if (cit == cmap.end()) cmap[key] = 1; else ++cit->second;
    }
  }
  for (auto const& cp : cmap) { auto& p = pmap[cp.first]; p.reserve(cp.second); }
  for (int64_t i = 0; i < N; ++i)
    if (!v.mask || v.mask[i] != MASK_OUT_VAL) pmap[v.lab[i]].push_back(i);
}

// util/struct.hxx:95-125  (point-map version)
void genContourMapFromPoints(PointMap& borderMap, PointPairMap& boundaryMap, PointMap const& pointMap,
                             Vol const& v) {
  for (auto const& pp : pointMap) {
    auto it = borderMap.insert(std::make_pair(pp.first, Points())).first;
    for (auto p : pp.second) {
      Label thisVal = v.lab[p];
      auto ct = contourTraits(v, p);
      if (ct.first != thisVal) boundaryMap[std::make_pair(thisVal, ct.first)].push_back(p);
      else if (ct.second) it->second.push_back(p);
    }
    if (it->second.empty()) borderMap.erase(it);
  }
}

// util/struct.hxx:128-144  (image version; note: no mask test on the centre voxel)
void genContourMapFromImage(PointMap& borderMap, PointPairMap& boundaryMap, Vol const& v) {
  int64_t N = v.size();
  for (int64_t p = 0; p < N; ++p) {
    Label thisVal = v.lab[p];
    auto ct = contourTraits(v, p);
    if (ct.first != thisVal) boundaryMap[std::make_pair(thisVal, ct.first)].push_back(p);
    else if (ct.second) borderMap[thisVal].push_back(p);
  }
}

// type/boundary_table.hxx:8-168 -- same containers, same operations
template <typename T>
struct BoundaryTable {
  struct Item;
  typedef std::map<LPair, std::shared_ptr<Item>> Table;
  typedef typename Table::iterator iterator;
  struct Item {
    typename std::multimap<double, iterator>::iterator mqit;
    T data;
  };
  Table table;
  std::multimap<double, iterator> mqueue;

  bool empty() const { return table.empty(); }

  template <typename CFunc> iterator top(CFunc fcond) {               // :46-52
    for (auto it = mqueue.rbegin(); it != mqueue.rend(); ++it)
      if (fcond(*this, it->second)) return it->second;
    return table.end();
  }

  // :91-114.  `order` is the region-map iteration sequence (snapshot taken by the
  // caller: the reference range-iterates rmap while initFb inserts/erases key 0 in
  // it, struct_merge_bc.hxx:18-22, which is UB; iterating a snapshot gives the same
  // sequence whenever the reference's behaviour is defined).
  template <typename BFunc, typename SFunc>
  void init(RegionMap const& rmap, std::vector<Label> const& order, BFunc fb, SFunc fsal) {
    for (Label rk : order) {
      // boundary map of a region is not touched by scratch merges -> safe to iterate
      auto const& reg = rmap.find(rk)->second;
      for (auto const& bp : reg.boundary) {
        Label r0 = bp.first.first, r1 = bp.first.second;
        LPair key(r0, r1);
        if (key.first > key.second) std::swap(key.first, key.second);
        auto rit1 = rmap.find(r1);
        if (table.count(key) == 0 && rit1 != rmap.end() &&
            rit1->second.boundary.count(std::make_pair(r1, r0)) > 0) {
          auto btit = table.emplace(key, std::shared_ptr<Item>(new Item)).first;
          fb(btit->second->data, r0, r1);
        }
      }
    }
    for (auto btit = table.begin(); btit != table.end(); ++btit)
      btit->second->mqit =
          mqueue.insert(std::make_pair(fsal(btit->second->data, btit->first.first, btit->first.second), btit));
  }

  // :121-167
  template <typename BFunc, typename SFunc>
  void update(iterator btit01, Label r2, BFunc fb, SFunc fsal) {
    Label r0 = btit01->first.first, r1 = btit01->first.second;
    mqueue.erase(btit01->second->mqit);
    table.erase(btit01);
    auto btit = table.begin();
    while (btit != table.end() && btit->first.first <= r1) {
      Label rs;
      if (btit->first.first == r0 || btit->first.first == r1) rs = btit->first.second;
      else if (btit->first.second == r0 || btit->first.second == r1) rs = btit->first.first;
      else { ++btit; continue; }
      iterator btit0s, btit1s;
      if (btit->first.first == r0 || btit->first.second == r0) {
        btit0s = btit;
        btit1s = table.find(r1 < rs ? std::make_pair(r1, rs) : std::make_pair(rs, r1));
      } else {
        btit0s = table.find(r0 < rs ? std::make_pair(r0, rs) : std::make_pair(rs, r0));
        btit1s = btit;
      }
      auto btit2s = table.emplace(std::make_pair(rs, r2), std::shared_ptr<Item>(new Item)).first;
      fb(btit2s->second->data, r0, r1, rs, r2,
         btit0s == table.end() ? nullptr : &btit0s->second->data,
         btit1s == table.end() ? nullptr : &btit1s->second->data);
      double sal = fsal(btit2s->second->data, rs, r2);
      btit2s->second->mqit = mqueue.insert(std::make_pair(sal, btit2s));
      if (btit0s != table.end()) {
        mqueue.erase(btit0s->second->mqit);
        if (btit0s != btit) table.erase(btit0s);
      }
      if (btit1s != table.end()) {
        mqueue.erase(btit1s->second->mqit);
        if (btit1s != btit) table.erase(btit1s);
      }
      btit = table.erase(btit);
    }
  }
};

struct Triple { Label x0, x1, x2; };

// util/struct_merge.hxx:13-33
template <typename T, typename IFb, typename IFsal, typename UFb, typename UFsal, typename CFunc>
void genMergeOrderGreedy(std::vector<Triple>& order, std::vector<double>& saliencies, RegionMap& rmap,
                         bool updateRegion, IFb initFb, IFsal initFsal, UFb updateFb, UFsal updateFsal,
                         CFunc fcond) {
  std::vector<Label> iterOrder;
  iterOrder.reserve(rmap.size());
  for (auto const& rp : rmap) iterOrder.push_back(rp.first);
  Label keyToAssign = rmap.maxKey() + 1;   // before init, like the reference (:19 follows :18 -- see note)
  BoundaryTable<T> bt;
  bt.init(rmap, iterOrder, initFb, initFsal);
  // note: the reference computes maxKey() after the table is built, when the scratch key 0
  // may be present; 0 never exceeds a real label so the value is the same.
  while (!bt.empty()) {
    auto btit = bt.top(fcond);
    if (btit == bt.table.end()) break;
    Label r0 = btit->first.first, r1 = btit->first.second;
    order.push_back(Triple{r0, r1, keyToAssign});
    saliencies.push_back(btit->second->mqit->first);
    if (updateRegion) rmap.merge(r0, r1, keyToAssign);
    bt.update(btit, keyToAssign++, updateFb, updateFsal);
  }
}

// util/stats.hxx:83-91 -- order statistic at n/2 (the shuffle only perturbs rand())
template <typename C> double amedian(C& data) {
  if (data.empty()) return DUMMY;
  std::nth_element(data.begin(), data.begin() + data.size() / 2, data.end());
  return *(data.begin() + data.size() / 2);
}

// util/container.hxx:221-246
void splice1(std::vector<double>& dst, std::vector<double>& src) {
  if (src.empty()) return;
  if (dst.empty()) dst = std::move(src);
  else { dst.insert(dst.end(), src.begin(), src.end()); }
  src.clear();
}
void splice2(std::vector<double>& dst, std::vector<double>& s0, std::vector<double>& s1) {
  if (s0.empty()) splice1(dst, s1);
  else if (s1.empty()) splice1(dst, s0);
  else {
    dst.insert(dst.end(), s0.begin(), s0.end());
    dst.insert(dst.end(), s1.begin(), s1.end());
    s0.clear(); s1.clear();
  }
}

// ---------------------------------------------------------------- features ---------
// util/image_stats.hxx:12-52 + util/stats.hxx:145-152
struct HistSpec { int bins; double lo, hi; std::vector<double> bounds; };
HistSpec makeHist(int bins, double lo, double hi) {
  HistSpec h; h.bins = bins; h.lo = lo; h.hi = hi;
  double interval = (hi - lo) / bins;
  h.bounds.resize(bins);
  if (bins > 0) h.bounds[0] = interval;                              // :19 (ignores lo)
  for (int i = 1; i < bins; ++i) h.bounds[i] = h.bounds[i - 1] + interval;
  return h;
}
template <typename M>
void histOver(std::vector<double>& h, M const& points, const float* img, HistSpec const& hs) {
  std::vector<std::size_t> hc(hs.bins, 0);
  traverse(points, [&](int64_t p) {
    float val = img[p];
    if (val > hs.lo && val < hs.hi) {
      for (int i = 0; i < hs.bins; ++i) if (val < hs.bounds[i]) { ++hc[i]; break; }
    } else if (val <= hs.lo) ++hc[0];
    else ++hc[hs.bins - 1];
  });
  std::size_t n = (std::size_t)mapSize(points);
  h.assign(hs.bins, 0.0);
  if (n == 0) return;
  for (int i = 0; i < hs.bins; ++i) h[i] = hc[i] / (double)n;
}
double entropy(std::vector<double> const& d) {
  double ret = 0.0;
  for (double p : d) if (!isfeq(p, 0.0)) ret -= p * log2(p);
  return ret;
}
double distL1(std::vector<double> const& a, std::vector<double> const& b) {   // stats.hxx:155-163
  double r = 0.0;
  for (std::size_t i = 0; i < a.size(); ++i) r += std::fabs(a[i] - b[i]);
  return r;
}
double distX2(std::vector<double> const& a, std::vector<double> const& b) {   // stats.hxx:177-185
  double r = 0.0;
  for (std::size_t i = 0; i < a.size(); ++i) r += std::pow(a[i] - b[i], 2) / (a[i] + b[i] + FEPS);
  return r;
}

// type/feat.hxx:594-639 + 674-738 + 815-852
struct ImageFeats {
  std::vector<double> histogram;
  double entropy = 0.0, mean = 0.0, stddev = 0.0, min = 0.0, max = 0.0, median = 0.0;
  bool hasReal = true;
  bool histAsFeats = false;     // GLIA_USE_HISTOGRAM_AS_FEATS (CMakeLists.txt:54-58): the histogram itself precedes its entropy (feat.hxx:608-621)
  // GLIA_USE_MEDIAN_AS_FEATS (CMakeLists.txt:59-63): a fifth real feature, the median, ahead of the mean -- and mean / standard
  // deviation come from stats::mean / stats::var over the VECTOR of values (feat.hxx:710-722; util/stats.hxx:36-69): the mean is
  // sum / n with a double accumulator, the variance the mean of (x - mean)^2 -- not E[x^2] - mean^2 as in the default build.
  // The reference sums the vector in the order stats::amedian left it in, and amedian shuffles it with rand() first
  // (stats.hxx:87): its own last bits of mean and stddev depend on the history of rand().  The restatement leaves the shuffle
  // out (as everywhere: the order statistic does not depend on it) and sums in the order nth_element leaves; these two columns
  // are therefore comparable to 1e-12 relative, every other column bit for bit.  PARITY UNPINNED (feat.hxx needs ITK).
  bool medianAsFeats = false;
  template <typename M> void generate(M const& points, const float* img, HistSpec const& hs, bool real, bool histFeats = false, bool medFeats = false) {
    hasReal = real; histAsFeats = histFeats; medianAsFeats = medFeats;
    histOver(histogram, points, img, hs);
    entropy = ::entropy(histogram);
    if (!real) return;
    int n = (int)mapSize(points);
    if (n == 0) return;
    if (medFeats) {
      std::vector<float> vals;
      vals.reserve(n);
      traverse(points, [&](int64_t p) { vals.push_back(img[p]); });
      median = amedian(vals);
      double sum = 0.0;                                                   // stats::sum<TContainer, double> (stats.hxx:36-42)
      for (float x : vals) sum += x;
      mean = sdivide(sum, (double)vals.size(), 0.0);                      // stats::mean (:55-57)
      double ret = 0.0;                                                   // stats::var (:60-69)
      for (float x : vals) { double dx = x - mean; ret += dx * dx; }
      stddev = ssqrt(sdivide(ret, (double)vals.size(), 0.0), 0.0);
      float mn = vals.front(), mx = vals.front();                         // stats::min / stats::max (:18-33)
      for (float x : vals) { if (x < mn) mn = x; if (x > mx) mx = x; }
      min = mn; max = mx;
      return;
    }
    mean = 0.0; min = FMAX; max = -FMAX; stddev = 0.0;
    traverse(points, [&](int64_t p) {
      float val = img[p];
      mean += val;
      stddev += (double)val * val;
      if (val < min) min = val;
      if (val > max) max = val;
    });
    mean /= n;
    stddev = ssqrt(stddev / n - mean * mean, 0.0);
  }
  void serialize(std::vector<double>& f) const {
    if (histAsFeats) for (double x : histogram) f.push_back(x);
    f.push_back(entropy);
    if (hasReal) { if (medianAsFeats) f.push_back(median); f.push_back(mean); f.push_back(stddev); f.push_back(min); f.push_back(max); }
  }
};

struct Cfg {
  int D;
  Vol const* vol;
  orc_feat_cfg c;
  std::vector<HistSpec> rh, rlh, bh;
};

// hmt/bc_feat.hxx:46-127 with type/feat.hxx:27-91, 441-503
struct RegionFeats {
  double area = 0, perim = 0, compactness = 0, bboxArea = 0;
  std::vector<double> bboxSize, validPerims, rValidPerims;
  std::vector<ImageFeats> region, labelRegion, boundary;

  void generate(Region const& reg, Cfg const& cfg) {
    const int D = cfg.D;
    const double nArea = cfg.c.norm_area, nLen = cfg.c.norm_len;
    bboxSize.assign(D, 0.0);
    int T = cfg.c.n_thr;
    validPerims.assign(T, 0.0); rValidPerims.assign(T, 0.0);
    // feat.hxx:71-90
    area = (double)reg.size();
    int64_t bsize = mapSize(reg.boundary);
    perim = (double)(bsize + mapSize(reg.border));
    compactness = sdivide(std::pow(perim, (double)D / (D - 1)), area, 0.0);
    area = sdivide(area, nArea, 0.0);
    perim = sdivide(perim, nLen, 0.0);
    // alg/geometry.hxx:21-39
    int64_t lo[3], hi[3];
    {
      int64_t first = reg.pts.begin()->second->front();
      cfg.vol->coords(first, lo);
      cfg.vol->coords(first, hi);
      traverse(reg.pts, [&](int64_t p) {
        int64_t c[3];
        cfg.vol->coords(p, c);
        for (int i = 0; i < D; ++i) {
          if (c[i] < lo[i]) lo[i] = c[i];
          else if (c[i] > hi[i]) hi[i] = c[i];
        }
      });
    }
    bboxArea = 1.0;
    for (int i = 0; i < D; ++i) {
      double bb = (double)(uint64_t)(hi[i] - lo[i]);
      bboxSize[i] = sdivide(bb, nLen, 0.0);
      bboxArea *= bb;
    }
    bboxArea = sdivide(bboxArea, nArea, 0.0);
    // feat.hxx:485-502
    for (int i = 0; i < T; ++i) {
      std::size_t vp = 0;
      double thr = cfg.c.thr[i];
      traverse(reg.boundary, [&](int64_t p) { if (cfg.c.pb[p] >= thr) ++vp; });
      validPerims[i] = sdivide((double)vp, nLen, 0.0);
      rValidPerims[i] = sdivide((double)vp, (double)bsize, 0.0);
    }
    // bc_feat.hxx:100-124
    region.resize(cfg.c.n_rimg);
    for (int i = 0; i < cfg.c.n_rimg; ++i) region[i].generate(reg.pts, cfg.c.rimg[i], cfg.rh[i], true, cfg.c.hist_as_feats != 0, cfg.c.median_as_feats != 0);
    labelRegion.resize(cfg.c.n_rlimg);
    for (int i = 0; i < cfg.c.n_rlimg; ++i) labelRegion[i].generate(reg.pts, cfg.c.rlimg[i], cfg.rlh[i], false, cfg.c.hist_as_feats != 0);
    boundary.resize(cfg.c.n_bimg);
    for (int i = 0; i < cfg.c.n_bimg; ++i) boundary[i].generate(reg.boundary, cfg.c.bimg[i], cfg.bh[i], true, cfg.c.hist_as_feats != 0, cfg.c.median_as_feats != 0);
  }
  void log() {  // feat.hxx:46-52, 463-467
    area = slog(area, 0.0); perim = slog(perim, 0.0); bboxArea = slog(bboxArea, 0.0);
    for (auto& x : bboxSize) x = slog(x, 0.0);
    for (auto& x : validPerims) x = slog(x, 0.0);
  }
  void serialize(std::vector<double>& f) const {  // feat.hxx:54-62, 469-476; bc_feat.hxx:69-77
    f.push_back(area); f.push_back(perim); f.push_back(compactness); f.push_back(bboxArea);
    for (double x : bboxSize) f.push_back(x);
    for (double x : validPerims) f.push_back(x);
    for (double x : rValidPerims) f.push_back(x);
    for (auto const& p : region) p.serialize(f);
    for (auto const& p : labelRegion) p.serialize(f);
    for (auto const& p : boundary) p.serialize(f);
    if (hasSaliency) f.push_back(saliency);          // bc_feat.hxx:76
  }
  bool hasSaliency = false;                          // bc_feat.hxx:53,125-126 (pSaliency)
  double saliency = 0.0;
};

// hmt/bc_feat.hxx:131-214 with type/feat.hxx:94-187, 507-590, 643-670, 769-811, 856-883
struct BoundaryFeats {
  double areaDiff = 0, rAreaDiff0 = 0, rAreaDiff1 = 0, perimDiff = 0, rPerimDiff0 = 0, rPerimDiff1 = 0;
  double boundaryLength = 0, rBLA0 = 0, rBLA1 = 0, rBLP0 = 0, rBLP1 = 0;
  std::vector<double> vbl, rvbl, rvblp0, rvblp1;
  struct Diff { double l1, x2, ed, meanD, stdD, minD, maxD; bool real; double medD = 0.0; bool med = false; };
  std::vector<Diff> region, labelRegion;
  std::vector<ImageFeats> boundary;

  void generate(PtrPairMap const& b, RegionFeats const& rf0, RegionFeats const& rf1, Cfg const& cfg) {
    const double nLen = cfg.c.norm_len;
    // feat.hxx:124-132
    areaDiff = std::fabs(rf0.area - rf1.area);
    rAreaDiff0 = sdivide(areaDiff, rf0.area, 0.0);
    rAreaDiff1 = sdivide(areaDiff, rf1.area, 0.0);
    perimDiff = std::fabs(rf0.perim - rf1.perim);
    rPerimDiff0 = sdivide(perimDiff, rf0.perim, 0.0);
    rPerimDiff1 = sdivide(perimDiff, rf1.perim, 0.0);
    // feat.hxx:176-186
    int64_t bsize = mapSize(b);
    boundaryLength = sdivide(std::ceil(bsize / 2.0), nLen, 0.0);
    rBLA0 = sdivide(boundaryLength, rf0.area, 0.0);
    rBLA1 = sdivide(boundaryLength, rf1.area, 0.0);
    rBLP0 = sdivide(boundaryLength, rf0.perim, 0.0);
    rBLP1 = sdivide(boundaryLength, rf1.perim, 0.0);
    // feat.hxx:567-589
    int T = cfg.c.n_thr;
    vbl.assign(T, 0.0); rvbl.assign(T, 0.0); rvblp0.assign(T, 0.0); rvblp1.assign(T, 0.0);
    for (int i = 0; i < T; ++i) {
      std::size_t vp = 0;
      double thr = cfg.c.thr[i];
      traverse(b, [&](int64_t p) { if (cfg.c.pb[p] >= thr) ++vp; });
      vbl[i] = sdivide(std::ceil(vp / 2.0), nLen, 0.0);
      rvbl[i] = sdivide(vbl[i], boundaryLength, 0.0);
      rvblp0[i] = sdivide(vbl[i], rf0.perim, 0.0);
      rvblp1[i] = sdivide(vbl[i], rf1.perim, 0.0);
    }
    // bc_feat.hxx:188-204
    region.clear();
    for (std::size_t i = 0; i < rf0.region.size(); ++i) {
      auto const& a = rf0.region[i]; auto const& c = rf1.region[i];
      Diff d;
      d.l1 = distL1(a.histogram, c.histogram);
      d.x2 = distX2(a.histogram, c.histogram);
      d.ed = fabs(a.entropy - c.entropy);
      d.meanD = std::fabs(a.mean - c.mean); d.stdD = std::fabs(a.stddev - c.stddev);
      d.minD = std::fabs(a.min - c.min); d.maxD = std::fabs(a.max - c.max);
      d.medD = std::fabs(a.median - c.median); d.med = a.medianAsFeats;       // feat.hxx:803-805
      d.real = true;
      region.push_back(d);
    }
    labelRegion.clear();
    for (std::size_t i = 0; i < rf0.labelRegion.size(); ++i) {
      auto const& a = rf0.labelRegion[i]; auto const& c = rf1.labelRegion[i];
      Diff d = Diff();
      d.l1 = distL1(a.histogram, c.histogram);
      d.x2 = distX2(a.histogram, c.histogram);
      d.ed = fabs(a.entropy - c.entropy);
      d.real = false;
      labelRegion.push_back(d);
    }
    boundary.resize(cfg.c.n_bimg);
    for (int i = 0; i < cfg.c.n_bimg; ++i) boundary[i].generate(b, cfg.c.bimg[i], cfg.bh[i], true, cfg.c.hist_as_feats != 0, cfg.c.median_as_feats != 0);
  }
  void log() {  // feat.hxx:103-106, 148-155, 531-539
    areaDiff = slog(areaDiff, 0.0); perimDiff = slog(perimDiff, 0.0);
    boundaryLength = slog(boundaryLength, 0.0);
    for (auto& x : vbl) x = slog(x, 0.0);
  }
  void serialize(std::vector<double>& f) const {
    f.push_back(areaDiff); f.push_back(rAreaDiff0); f.push_back(rAreaDiff1);
    f.push_back(perimDiff); f.push_back(rPerimDiff0); f.push_back(rPerimDiff1);
    f.push_back(boundaryLength); f.push_back(rBLA0); f.push_back(rBLA1); f.push_back(rBLP0); f.push_back(rBLP1);
    for (double x : vbl) f.push_back(x);
    for (double x : rvbl) f.push_back(x);
    for (double x : rvblp0) f.push_back(x);
    for (double x : rvblp1) f.push_back(x);
    for (auto const& d : region) {
      f.push_back(d.l1); f.push_back(d.x2); f.push_back(d.ed);
      if (d.med) f.push_back(d.medD);                                          // feat.hxx:784-786
      f.push_back(d.meanD); f.push_back(d.stdD); f.push_back(d.minD); f.push_back(d.maxD);
    }
    for (auto const& d : labelRegion) { f.push_back(d.l1); f.push_back(d.x2); f.push_back(d.ed); }
    for (auto const& p : boundary) p.serialize(f);
    if (hasSaliency) { f.push_back(salFirst); f.push_back(salSecond); }     // bc_feat.hxx:163-166
  }
  // bc_feat.hxx:208-213
  void setSaliency(RegionFeats const& rf0, RegionFeats const& rf1, RegionFeats const& rf2) {
    if (rf0.hasSaliency && rf1.hasSaliency && rf2.hasSaliency) {
      double dsal02 = std::fabs(rf0.saliency - rf2.saliency);
      double dsal12 = std::fabs(rf1.saliency - rf2.saliency);
      hasSaliency = true; salFirst = std::min(dsal02, dsal12); salSecond = std::max(dsal02, dsal12);
    }
  }
  bool hasSaliency = false;
  double salFirst = 0.0, salSecond = 0.0;
};

// hmt/bc_feat.hxx:247-279
void selectFeatures(std::vector<double>& f, BoundaryFeats const& x0, RegionFeats const& x1,
                    RegionFeats const& x2) {
  f.push_back(x1.area); f.push_back(x2.area); f.push_back(x1.perim); f.push_back(x2.perim);
  f.push_back(x0.boundaryLength);
  for (auto const& bf : x0.boundary) { f.push_back(bf.mean); if (bf.medianAsFeats) f.push_back(bf.median); }      // bc_feat.hxx:263-268
  for (auto const& rf : x0.region) { f.push_back(rf.meanD); f.push_back(rf.l1); f.push_back(rf.x2); f.push_back(rf.ed); }
  for (auto const& rlf : x0.labelRegion) { f.push_back(rlf.l1); f.push_back(rlf.x2); }
}

// hmt/main_merge_order_bc.cxx:54-95 (fBcFeat), also main_bc_feat.cxx:76-95
void bcFeat(std::vector<double>& data, RegionFeats& rf0, RegionFeats& rf1, RegionFeats& rf2,
            Region const& reg0, Region const& reg1, Cfg const& cfg) {
  RegionFeats* x1 = &rf0; RegionFeats* x2 = &rf1; RegionFeats* x3 = &rf2;
  if (x1->area > x2->area) std::swap(x1, x2);
  PtrPairMap b;
  getBoundary(b, reg0, reg1);
  BoundaryFeats x0;
  x0.generate(b, *x1, *x2, cfg);
  if (cfg.c.use_log) { x0.log(); rf0.log(); rf1.log(); rf2.log(); }
  data.clear();
  if (cfg.c.use_simple) selectFeatures(data, x0, *x1, *x2);
  else { x0.serialize(data); x1->serialize(data); x2->serialize(data); x3->serialize(data); }
}

Cfg makeCfg(Vol const* vol, const orc_feat_cfg* c) {
  Cfg cfg; cfg.D = vol->D; cfg.vol = vol; cfg.c = *c;
  for (int i = 0; i < c->n_rimg; ++i) cfg.rh.push_back(makeHist(c->rbins[i], c->rlo[i], c->rhi[i]));
  for (int i = 0; i < c->n_rlimg; ++i) cfg.rlh.push_back(makeHist(c->rlbins[i], c->rllo[i], c->rlhi[i]));
  for (int i = 0; i < c->n_bimg; ++i) cfg.bh.push_back(makeHist(c->bbins[i], c->blo[i], c->bhi[i]));
  return cfg;
}

// ml/rf/rf.hxx:362-372 + upstream predictClassTree as recalled in SURVEY.md B.4 (PARITY UNPINNED)
double forestPredict(const orc_forest* f, const double* x, int /*d*/) {
  std::vector<int> countts(f->nclass, 0);
  for (int j = 0; j < f->ntree; ++j) {
    const int* treemap = f->treemap + (int64_t)2 * j * f->nrnodes;
    const int* nodestatus = f->nodestatus + (int64_t)j * f->nrnodes;
    const double* xbestsplit = f->xbestsplit + (int64_t)j * f->nrnodes;
    const int* bestvar = f->bestvar + (int64_t)j * f->nrnodes;
    const int* nodeclass = f->nodeclass + (int64_t)j * f->nrnodes;
    int k = 0;
    while (nodestatus[k] != -1) {
      int m = bestvar[k] - 1;
      k = (x[m] <= xbestsplit[k]) ? treemap[k * 2] - 1 : treemap[1 + k * 2] - 1;
    }
    countts[nodeclass[k] - 1] += 1;
  }
  for (int i = 0; i < f->nclass; ++i)
    if (f->predict_label == f->orig_labels[i]) return countts[i] / (double)f->ntree;
  return -1.0;
}

// ---------------------------------------------------------------- synthetic inputs ----
inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

struct CellGrid {
  int D; int64_t n[3]; int S; uint64_t seed; uint64_t salt; int64_t nc[3];
  void init(int D_, const int64_t* dims, int S_, uint64_t seed_, uint64_t salt_) {
    D = D_; S = S_; seed = seed_; salt = salt_;
    for (int i = 0; i < 3; ++i) { n[i] = dims[i]; nc[i] = (dims[i] + S - 1) / S; }
    if (D == 2) { n[2] = 1; nc[2] = 1; }
  }
  void seedPos(int64_t cx, int64_t cy, int64_t cz, int64_t p[3]) const {
    uint64_t lin = (uint64_t)(cx + nc[0] * (cy + nc[1] * cz));
    uint64_t h = splitmix64(seed ^ salt ^ splitmix64(lin));
    p[0] = cx * S + (int64_t)((h & 0xFFFF) % (uint64_t)S);
    p[1] = cy * S + (int64_t)(((h >> 16) & 0xFFFF) % (uint64_t)S);
    p[2] = (D == 3) ? cz * S + (int64_t)(((h >> 32) & 0xFFFF) % (uint64_t)S) : 0;
  }
  // nearest seed among the 3^D surrounding cells; squared L2 in int64; ties -> lower cell index
  uint32_t cellOf(int64_t x, int64_t y, int64_t z) const {
    int64_t cx = x / S, cy = y / S, cz = (D == 3) ? z / S : 0;
    int64_t best = INT64_MAX; uint32_t bestId = 0;
    for (int64_t dz = (D == 3 ? -1 : 0); dz <= (D == 3 ? 1 : 0); ++dz)
      for (int64_t dy = -1; dy <= 1; ++dy)
        for (int64_t dx = -1; dx <= 1; ++dx) {
          int64_t ex = cx + dx, ey = cy + dy, ez = cz + dz;
          if (ex < 0 || ey < 0 || ez < 0 || ex >= nc[0] || ey >= nc[1] || ez >= nc[2]) continue;
          int64_t p[3];
          seedPos(ex, ey, ez, p);
          int64_t d = (p[0] - x) * (p[0] - x) + (p[1] - y) * (p[1] - y) + (p[2] - z) * (p[2] - z);
          uint32_t id = (uint32_t)(ex + nc[0] * (ey + nc[1] * ez));
          if (d < best || (d == best && id < bestId)) { best = d; bestId = id; }
        }
    return bestId;
  }
};

}  // namespace

struct orc_rag {
  Vol vol;
  std::vector<Label> labCopy, maskCopy;
  bool onlyContour;
  RegionMap rmap;
};

extern "C" {

int orc_synth(int dim, const int64_t* dims, int S, int G, uint64_t seed, int variant, orc_label* labels,
              float* pb) {
  if (dim != 2 && dim != 3) return -1;
  CellGrid sv, tr;
  sv.init(dim, dims, S, seed, 0x5350ull);
  tr.init(dim, dims, G, seed, 0x54525554ull);
  int64_t nx = dims[0], ny = dims[1], nz = dim == 3 ? dims[2] : 1;
  int64_t N = nx * ny * nz;
  std::vector<uint32_t> truth(N);
  for (int64_t z = 0; z < nz; ++z)
    for (int64_t y = 0; y < ny; ++y)
      for (int64_t x = 0; x < nx; ++x) {
        int64_t i = x + nx * (y + ny * z);
        labels[i] = 1 + sv.cellOf(x, y, z);
        truth[i] = tr.cellOf(x, y, z);
      }
  const int64_t st[3] = {1, nx, nx * ny};
  const int64_t nn[3] = {nx, ny, nz};
  for (int64_t z = 0; z < nz; ++z)
    for (int64_t y = 0; y < ny; ++y)
      for (int64_t x = 0; x < nx; ++x) {
        int64_t i = x + nx * (y + ny * z);
        int64_t c[3] = {x, y, z};
        int otherTruth = 0, otherSv = 0;
        for (int d = 0; d < dim; ++d) {
          if (c[d] > 0) { int64_t j = i - st[d]; otherTruth |= truth[j] != truth[i]; otherSv |= labels[j] != labels[i]; }
          if (c[d] + 1 < nn[d]) { int64_t j = i + st[d]; otherTruth |= truth[j] != truth[i]; otherSv |= labels[j] != labels[i]; }
        }
        uint64_t r = splitmix64((uint64_t)i ^ seed);
        int q = (int)(r % 77) + 154 * otherTruth + 38 * otherSv;
        if (q > 255) q = 255;
        double v = q / 256.0;
        if (variant == 1) {
          uint64_t r2 = splitmix64(r ^ 0xF32ull);
          v += (double)(r2 >> 40) * (1.0 / 16777216.0) * (1.0 / 256.0);
        }
        pb[i] = (float)v;
      }
  return 0;
}

orc_rag* orc_rag_build(int dim, const int64_t* dims, const orc_label* labels, const orc_label* mask,
                       int only_contour) {
  orc_rag* h = new orc_rag;
  h->vol.D = dim;
  h->vol.n[0] = dims[0]; h->vol.n[1] = dims[1]; h->vol.n[2] = dim == 3 ? dims[2] : 1;
  int64_t N = h->vol.size();
  h->labCopy.assign(labels, labels + N);
  h->vol.lab = h->labCopy.data();
  if (mask) { h->maskCopy.assign(mask, mask + N); h->vol.mask = h->maskCopy.data(); }
  else h->vol.mask = nullptr;
  h->onlyContour = only_contour != 0;
  // type/region_map.hxx:52-65
  if (h->onlyContour) {
    genContourMapFromImage(*h->rmap.pBorderMap, *h->rmap.pBoundaryMap, h->vol);
    h->rmap.initContour();
  } else {
    genPointMap(*h->rmap.pPointMap, h->vol);
    genContourMapFromPoints(*h->rmap.pBorderMap, *h->rmap.pBoundaryMap, *h->rmap.pPointMap, h->vol);
    h->rmap.init();
  }
  return h;
}

void orc_rag_free(orc_rag* h) { delete h; }

int64_t orc_rag_num_regions(const orc_rag* h) { return (int64_t)h->rmap.size(); }
int64_t orc_rag_num_pairs(const orc_rag* h) { return (int64_t)h->rmap.pBoundaryMap->size(); }

static std::vector<Label> sortedRegionKeys(const orc_rag* h) {
  std::vector<Label> keys;
  for (auto const& rp : h->rmap) keys.push_back(rp.first);
  std::sort(keys.begin(), keys.end());
  return keys;
}
static std::vector<LPair> sortedPairKeys(const orc_rag* h) {
  std::vector<LPair> keys;
  for (auto const& bp : *h->rmap.pBoundaryMap) keys.push_back(bp.first);
  std::sort(keys.begin(), keys.end());
  return keys;
}

void orc_rag_regions(const orc_rag* h, orc_label* label, int64_t* npoints, int64_t* nborder) {
  auto keys = sortedRegionKeys(h);
  for (std::size_t i = 0; i < keys.size(); ++i) {
    auto const& reg = h->rmap.find(keys[i])->second;
    label[i] = keys[i];
    npoints[i] = reg.size();
    nborder[i] = mapSize(reg.border);
  }
}

void orc_rag_pairs(const orc_rag* h, orc_label* a, orc_label* b, int64_t* n) {
  auto keys = sortedPairKeys(h);
  for (std::size_t i = 0; i < keys.size(); ++i) {
    a[i] = keys[i].first; b[i] = keys[i].second;
    n[i] = (int64_t)h->rmap.pBoundaryMap->find(keys[i])->second.size();
  }
}

void orc_rag_region_iter_order(const orc_rag* h, orc_label* label) {
  std::size_t i = 0;
  for (auto const& rp : h->rmap) label[i++] = rp.first;
}

static void statsOver(Points const& pts, const float* img, double& sum, double& sumsq, double& mn, double& mx) {
  sum = 0; sumsq = 0; mn = FMAX; mx = -FMAX;
  for (auto p : pts) {
    float v = img[p];
    sum += v; sumsq += (double)v * v;
    if (v < mn) mn = v;
    if (v > mx) mx = v;
  }
}

void orc_rag_pair_stats(const orc_rag* h, const float* img, double* sum, double* sumsq, double* vmin,
                        double* vmax) {
  auto keys = sortedPairKeys(h);
  for (std::size_t i = 0; i < keys.size(); ++i)
    statsOver(h->rmap.pBoundaryMap->find(keys[i])->second, img, sum[i], sumsq[i], vmin[i], vmax[i]);
}

void orc_rag_region_stats(const orc_rag* h, const float* img, double* sum, double* sumsq, double* vmin,
                          double* vmax, int64_t* bbox_lo, int64_t* bbox_hi) {
  auto keys = sortedRegionKeys(h);
  for (std::size_t i = 0; i < keys.size(); ++i) {
    auto pit = h->rmap.pPointMap->find(keys[i]);
    if (pit == h->rmap.pPointMap->end()) {
      sum[i] = sumsq[i] = 0; vmin[i] = FMAX; vmax[i] = -FMAX;
      for (int d = 0; d < 3; ++d) { bbox_lo[3 * i + d] = 0; bbox_hi[3 * i + d] = 0; }
      continue;
    }
    statsOver(pit->second, img, sum[i], sumsq[i], vmin[i], vmax[i]);
    int64_t lo[3] = {INT64_MAX, INT64_MAX, INT64_MAX}, hi[3] = {-1, -1, -1};
    for (auto p : pit->second) {
      int64_t c[3];
      h->vol.coords(p, c);
      for (int d = 0; d < 3; ++d) { lo[d] = std::min(lo[d], c[d]); hi[d] = std::max(hi[d], c[d]); }
    }
    for (int d = 0; d < 3; ++d) { bbox_lo[3 * i + d] = lo[d]; bbox_hi[3 * i + d] = hi[d]; }
  }
}

// Dump the leaf maps + pb in the text format oracle/ref_engine_driver.cc reads, so that the
// reference's own engine headers can be run on exactly this RAG (tests/test_oracle_vs_ref.py).
int orc_rag_dump(const orc_rag* h, const float* pb, int type, int update_region, const char* path) {
  FILE* f = fopen(path, "w");
  if (!f) return -1;
  auto const& rm = h->rmap;
  fprintf(f, "%lld %zu %zu %zu %d %d\n", (long long)h->vol.size(), rm.pPointMap->size(), rm.pBorderMap->size(),
          rm.pBoundaryMap->size(), type, update_region);
  for (auto const& pp : *rm.pPointMap) {
    fprintf(f, "%u %zu", pp.first, pp.second.size());
    for (auto p : pp.second) fprintf(f, " %lld", (long long)p);
    fprintf(f, "\n");
  }
  for (auto const& pp : *rm.pBorderMap) {
    fprintf(f, "%u %zu", pp.first, pp.second.size());
    for (auto p : pp.second) fprintf(f, " %lld", (long long)p);
    fprintf(f, "\n");
  }
  for (auto const& pp : *rm.pBoundaryMap) {
    fprintf(f, "%u %u %zu", pp.first.first, pp.first.second, pp.second.size());
    for (auto p : pp.second) fprintf(f, " %lld", (long long)p);
    fprintf(f, "\n");
  }
  for (int64_t i = 0; i < h->vol.size(); ++i) fprintf(f, "%.9g\n", (double)pb[i]);
  fclose(f);
  return 0;
}

static int64_t emit(std::vector<Triple> const& order, std::vector<double> const& sal, orc_label* o, double* s,
                    int64_t cap) {
  int64_t n = (int64_t)order.size();
  if (n > cap) return -1;
  for (int64_t i = 0; i < n; ++i) {
    o[3 * i] = order[i].x0; o[3 * i + 1] = order[i].x1; o[3 * i + 2] = order[i].x2;
    if (s) s[i] = sal[i];
  }
  return n;
}

int64_t orc_merge_order_pb(orc_rag* h, const float* pb, int type, int update_region, orc_label* order_out,
                           double* sal_out, int64_t cap) {
  RegionMap& rmap = h->rmap;
  if (rmap.empty()) return 0;
  std::vector<Triple> order;
  std::vector<double> sal;
  bool bad = false;
  auto ftrue = [](auto&, auto) { return true; };
  if (type == 2) {
    // util/struct_merge.hxx:38-85
    typedef std::pair<double, int> ItemData;
    auto initFb = [&](ItemData& data, Label r0, Label r1) {
      PtrPairMap b;
      getBoundary(b, rmap.find(r0)->second, rmap.find(r1)->second);
      data.first = 0.0;
      traverse(b, [&](int64_t p) { data.first += pb[p]; });
      data.second = (int)mapSize(b);
      data.first = sdivide(data.first, data.second, 0.0);
    };
    auto fsal = [&](ItemData& data, Label, Label) -> double {
      if (data.first == DUMMY) bad = true;
      return -data.first;
    };
    auto updateFb = [](ItemData& d2, Label, Label, Label, Label, ItemData* p0, ItemData* p1) {
      d2.first = 0.0; d2.second = 0;
      if (p0) { d2.first += p0->first * p0->second; d2.second += p0->second; }
      if (p1) { d2.first += p1->first * p1->second; d2.second += p1->second; }
      d2.first = sdivide(d2.first, d2.second, 0.0);
    };
    genMergeOrderGreedy<ItemData>(order, sal, rmap, update_region != 0, initFb, fsal, updateFb, fsal, ftrue);
  } else if (type == 1) {
    // util/struct_merge.hxx:90-136
    typedef std::vector<double> ItemData;
    auto initFb = [&](ItemData& data, Label r0, Label r1) {
      PtrPairMap b;
      getBoundary(b, rmap.find(r0)->second, rmap.find(r1)->second);
      data.reserve(mapSize(b));
      traverse(b, [&](int64_t p) { data.push_back(pb[p]); });
    };
    auto fsal = [&](ItemData& data, Label, Label) -> double {
      double p = amedian(data);
      if (p == DUMMY) bad = true;
      return -p;
    };
    auto updateFb = [](ItemData& d2, Label, Label, Label, Label, ItemData* p0, ItemData* p1) {
      if (p0 && p1) splice2(d2, *p0, *p1);
      else if (p0) splice1(d2, *p0);
      else if (p1) splice1(d2, *p1);
    };
    genMergeOrderGreedy<ItemData>(order, sal, rmap, update_region != 0, initFb, fsal, updateFb, fsal, ftrue);
  } else if (type == 3) {
    // util/struct_merge.hxx:141-185: median x min(region sizes), updateRegion = true
    typedef std::vector<double> ItemData;
    auto initFb = [&](ItemData& data, Label r0, Label r1) {
      PtrPairMap b;
      getBoundary(b, rmap.find(r0)->second, rmap.find(r1)->second);
      data.reserve(mapSize(b));
      traverse(b, [&](int64_t p) { data.push_back(pb[p]); });
    };
    auto fsal = [&](ItemData& data, Label r0, Label r1) -> double {
      double p = amedian(data);
      if (p == DUMMY) bad = true;
      return -p * std::min(rmap.find(r0)->second.size(), rmap.find(r1)->second.size());
    };
    auto updateFb = [](ItemData& d2, Label, Label, Label, Label, ItemData* p0, ItemData* p1) {
      if (p0 && p1) splice2(d2, *p0, *p1);
      else if (p0) splice1(d2, *p0);
      else if (p1) splice1(d2, *p1);
    };
    genMergeOrderGreedy<ItemData>(order, sal, rmap, true, initFb, fsal, updateFb, fsal, ftrue);
  } else return -1;   // hmt/main_merge_order_pb.cxx:36 "unsupported boundary stats type"
  if (bad) return -2;  // "invalid boundary saliency" (struct_merge.hxx:58-59)
  return emit(order, sal, order_out, sal_out, cap);
}

int orc_feat_dim(int dim, const orc_feat_cfg* c) {
  const int med = c->median_as_feats ? 1 : 0;      // GLIA_USE_MEDIAN_AS_FEATS: one more column per real-feature block (feat.hxx:677-680, 772-775)
  if (c->use_simple) return 5 + (1 + med) * c->n_bimg + 4 * c->n_rimg + 2 * c->n_rlimg;      // bc_feat.hxx:250-256
  int T = c->n_thr;
  int rf = 4 + dim + 2 * T + (5 + med) * c->n_rimg + c->n_rlimg + (5 + med) * c->n_bimg;
  int bf = 11 + 4 * T + (7 + med) * c->n_rimg + 3 * c->n_rlimg + (5 + med) * c->n_bimg;
  if (c->hist_as_feats) {        // every ImageLabelFeats block carries its histogram (feat.hxx:608-621); the diff blocks do not
    int hb = 0;
    for (int i = 0; i < c->n_bimg; ++i) hb += c->bbins[i];
    bf += hb;
    for (int i = 0; i < c->n_rimg; ++i) rf += c->rbins[i];
    for (int i = 0; i < c->n_rlimg; ++i) rf += c->rlbins[i];
    rf += hb;
  }
  return bf + 3 * rf;
}

// alg::EnsembleRandomForest + opt::ThresholdModelDistributor (alg/rf.hxx:63-98, type/function.hxx:71-85): set by
// orc_merge_order_bc_ensemble for the duration of one call.
struct EnsembleSel { const orc_forest* const* models; int dim0, dim1; double threshold; };
static const EnsembleSel* g_ensemble = nullptr;
// type/function.hxx:80-84: model 0 if x[dim1] < thr, else 1 if x[dim0] < thr, else 2 (pinned by oracle/_ref/ref_misc)
int orc_pick_model(int dim0, int dim1, double threshold, const double* x) {
  return x[dim1] < threshold ? 0 : x[dim0] < threshold ? 1 : 2;
}

int64_t orc_merge_order_bc(orc_rag* h, const orc_feat_cfg* c, const orc_forest* forest, int stub_index,
                           orc_label* order_out, double* sal_out, double* feats_out, int64_t cap,
                           int64_t* n_feat_evals) {
  if (h->onlyContour) return -1;
  RegionMap& rmap = h->rmap;
  Cfg cfg = makeCfg(&h->vol, c);
  typedef std::vector<double> ItemData;
  std::unordered_map<LPair, ItemData, PairHash> bcfmap;
  int64_t nEval = 0;
  // util/struct_merge_bc.hxx:18-35 around hmt/main_merge_order_bc.cxx:54-95
  auto feat = [&](ItemData& data, Label r0, Label r1) {
    rmap.erase(BG_VAL);
    auto rit2 = rmap.merge(r0, r1, BG_VAL);
    Region const& reg0 = rmap.find(r0)->second;
    Region const& reg1 = rmap.find(r1)->second;
    RegionFeats rf0, rf1, rf2;
    rf0.generate(reg0, cfg); rf1.generate(reg1, cfg); rf2.generate(rit2->second, cfg);
    // main_merge_order_bc.cxx:77-80: the cache key follows the area-ordered swap but is symmetric
    bcFeat(data, rf0, rf1, rf2, reg0, reg1, cfg);
    bcfmap[std::make_pair(std::min(r0, r1), std::max(r0, r1))] = data;
    ++nEval;
  };
  auto pred = [&](ItemData const& data) -> double {
    if (g_ensemble) {   // type/function.hxx:80-84: model 0 if x[dim1] < thr, else 1 if x[dim0] < thr, else 2
      const EnsembleSel& e = *g_ensemble;
      const int m = orc_pick_model(e.dim0, e.dim1, e.threshold, data.data());
      return forestPredict(e.models[m], data.data(), (int)data.size());   // alg/rf.hxx:97-98
    }
    if (forest) return forestPredict(forest, data.data(), (int)data.size());
    return 1.0 - data[stub_index];
  };
  auto initFb = [&](ItemData& data, Label r0, Label r1) { feat(data, r0, r1); };
  auto fsal = [&](ItemData const& data, Label, Label) -> double { return pred(data); };
  auto updateFb = [&](ItemData& d2, Label, Label, Label rs, Label r2, ItemData*, ItemData*) { feat(d2, rs, r2); };
  auto ftrue = [](auto&, auto) { return true; };
  std::vector<Triple> order;
  std::vector<double> sal;
  genMergeOrderGreedy<ItemData>(order, sal, rmap, true, initFb, fsal, updateFb, fsal, ftrue);
  rmap.erase(BG_VAL);
  if (n_feat_evals) *n_feat_evals = nEval;
  int64_t n = emit(order, sal, order_out, sal_out, cap);
  if (n >= 0 && feats_out) {
    int d = orc_feat_dim(h->vol.D, c);
    for (int64_t i = 0; i < n; ++i) {
      auto const& f = bcfmap.find(std::make_pair(order[i].x0, order[i].x1))->second;   // :152-154
      for (int k = 0; k < d; ++k) feats_out[i * d + k] = f[k];
    }
  }
  return n;
}

// hmt/main_merge_order_bc.cxx:103-109 (--bcm x3 --bcmd dim0 dim1 threshold)
int64_t orc_merge_order_bc_ensemble(orc_rag* h, const orc_feat_cfg* c, const orc_forest* const* models, int dim0, int dim1,
                                    double threshold, orc_label* order_out, double* sal_out, double* feats_out, int64_t cap) {
  EnsembleSel e = {models, dim0, dim1, threshold};
  g_ensemble = &e;
  const int64_t n = orc_merge_order_bc(h, c, nullptr, 0, order_out, sal_out, feats_out, cap, nullptr);
  g_ensemble = nullptr;
  return n;
}

int64_t orc_bc_feat_sal(orc_rag* h, const orc_feat_cfg* c, const orc_label* order, int64_t n_merges, const double* saliencies,
                        double init_sal, double sal_bias, double* feats_out);
int64_t orc_bc_feat(orc_rag* h, const orc_feat_cfg* c, const orc_label* order, int64_t n_merges,
                    double* feats_out) {
  return orc_bc_feat_sal(h, c, order, n_merges, nullptr, 1.0, 1.0, feats_out);
}
// with the saliency features of main_bc_feat.cxx:50-55 when `saliencies` is given; rows have dim + 5 columns then
// (none for the simple selection)
int64_t orc_bc_feat_sal(orc_rag* h, const orc_feat_cfg* c, const orc_label* order, int64_t n_merges, const double* saliencies,
                        double init_sal, double sal_bias, double* feats_out) {
  if (h->onlyContour) return -1;
  RegionMap& rmap = h->rmap;
  Cfg cfg = makeCfg(&h->vol, c);
  std::unordered_map<Label, double> saliencyMap;       // genSaliencyMap, bc_feat.hxx:12-26
  if (saliencies) {
    for (int64_t i = 0; i < n_merges; ++i) {
      if (saliencyMap.count(order[3 * i]) == 0) saliencyMap[order[3 * i]] = init_sal;
      if (saliencyMap.count(order[3 * i + 1]) == 0) saliencyMap[order[3 * i + 1]] = init_sal;
      saliencyMap[order[3 * i + 2]] = saliencies[i] + sal_bias;
    }
  }
  // main_bc_feat.cxx:57: RegionMap(seg, mask, order, false) -> set(order) (region_map.hxx:67-68)
  for (int64_t i = 0; i < n_merges; ++i) rmap.merge(order[3 * i], order[3 * i + 1], order[3 * i + 2]);
  std::unordered_map<Label, RegionFeats> rfmap;
  for (auto const& rp : rmap) {                                                    // :59-71
    RegionFeats& rf = rfmap[rp.first];
    rf.generate(rp.second, cfg);
    auto sit = saliencyMap.find(rp.first);
    if (sit != saliencyMap.end()) { rf.hasSaliency = true; rf.saliency = sit->second; }
  }
  int d = orc_feat_dim(h->vol.D, c) + ((saliencies && !c->use_simple) ? 5 : 0);
  std::vector<BoundaryFeats> bfeats(n_merges);
  std::vector<std::array<RegionFeats*, 3>> xs(n_merges);
  for (int64_t i = 0; i < n_merges; ++i) {                                           // :76-95
    Label r0 = order[3 * i], r1 = order[3 * i + 1], r2 = order[3 * i + 2];
    RegionFeats* x1 = &rfmap.find(r0)->second;
    RegionFeats* x2 = &rfmap.find(r1)->second;
    RegionFeats* x3 = &rfmap.find(r2)->second;
    if (x1->area > x2->area) { std::swap(r0, r1); std::swap(x1, x2); }
    PtrPairMap b;
    getBoundary(b, rmap.find(r0)->second, rmap.find(r1)->second);
    bfeats[i].generate(b, *x1, *x2, cfg);
    bfeats[i].setSaliency(*x1, *x2, *x3);
    xs[i] = {x1, x2, x3};
  }
  if (c->use_log) {                                                                  // :97-102
    for (auto& rp : rfmap) rp.second.log();
    for (auto& bf : bfeats) bf.log();
  }
  for (int64_t i = 0; i < n_merges; ++i) {
    std::vector<double> f;
    if (c->use_simple) selectFeatures(f, bfeats[i], *xs[i][0], *xs[i][1]);
    else { bfeats[i].serialize(f); xs[i][0]->serialize(f); xs[i][1]->serialize(f); xs[i][2]->serialize(f); }
    for (int k = 0; k < d; ++k) feats_out[i * d + k] = f[k];
  }
  return n_merges;
}

// gadget/main_pre_merge.cxx:27-76: mean linkage, updateRegion=true, condition on region sizes
int64_t orc_pre_merge(orc_rag* h, const float* pb, const int* sizeThresholds, int nThresholds,
                      double rpbThreshold, orc_label* order_out, double* sal_out, int64_t cap) {
  if (h->onlyContour) return -1;
  RegionMap& rmap = h->rmap;
  typedef std::pair<double, int> ItemData;
  std::unordered_map<Label, double> rpbs;
  auto regionPb = [&](Label key, Region const* pr, int64_t sz) -> double {
    auto it = rpbs.find(key);
    if (it != rpbs.end()) return it->second;
    double rpb = 0.0;
    traverse(pr->pts, [&](int64_t p) { rpb += pb[p]; });
    rpb = sdivide(rpb, (double)sz, 0.0);
    rpbs[key] = rpb;
    return rpb;
  };
  auto fcond = [&](BoundaryTable<ItemData>& /*bt*/, BoundaryTable<ItemData>::iterator btit) -> bool {
    Label key0 = btit->first.first, key1 = btit->first.second;
    auto const* pr0 = &rmap.find(key0)->second;
    auto const* pr1 = &rmap.find(key1)->second;
    int64_t sz0 = pr0->size(), sz1 = pr1->size();
    if (sz0 > sz1) { std::swap(key0, key1); std::swap(pr0, pr1); std::swap(sz0, sz1); }
    if (sz0 < sizeThresholds[0]) return true;
    if (nThresholds > 1) {
      if (sz0 < sizeThresholds[1] && regionPb(key0, pr0, sz0) > rpbThreshold) return true;
      if (sz1 < sizeThresholds[1] && regionPb(key1, pr1, sz1) > rpbThreshold) return true;
    }
    return false;
  };
  bool bad = false;
  auto initFb = [&](ItemData& data, Label r0, Label r1) {
    PtrPairMap b;
    getBoundary(b, rmap.find(r0)->second, rmap.find(r1)->second);
    data.first = 0.0;
    traverse(b, [&](int64_t p) { data.first += pb[p]; });
    data.second = (int)mapSize(b);
    data.first = sdivide(data.first, data.second, 0.0);
  };
  auto fsal = [&](ItemData& data, Label, Label) -> double { if (data.first == DUMMY) bad = true; return -data.first; };
  auto updateFb = [](ItemData& d2, Label, Label, Label, Label, ItemData* p0, ItemData* p1) {
    d2.first = 0.0; d2.second = 0;
    if (p0) { d2.first += p0->first * p0->second; d2.second += p0->second; }
    if (p1) { d2.first += p1->first * p1->second; d2.second += p1->second; }
    d2.first = sdivide(d2.first, d2.second, 0.0);
  };
  std::vector<Triple> order;
  std::vector<double> sal;
  genMergeOrderGreedy<ItemData>(order, sal, rmap, true, initFb, fsal, updateFb, fsal, fcond);
  if (bad) return -2;
  return emit(order, sal, order_out, sal_out, cap);
}

double orc_forest_predict(const orc_forest* f, const double* x, int d) { return forestPredict(f, x, d); }

// hmt/tree_build.hxx:12-38
int64_t orc_gen_tree(const orc_label* order, int64_t n_merges, orc_label* node_label, int32_t* parent,
                     int32_t* child0, int32_t* child1, int64_t cap) {
  std::unordered_map<Label, int> nmap;
  int ni = 0;
  auto leaf = [&](Label l) {
    if (ni >= cap) return -1;
    node_label[ni] = l; parent[ni] = -1; child0[ni] = -1; child1[ni] = -1;
    nmap.emplace(l, ni);
    return ni++;
  };
  for (int64_t i = 0; i < n_merges; ++i) {
    Label x0 = order[3 * i], x1 = order[3 * i + 1], x2 = order[3 * i + 2];
    auto n0 = nmap.find(x0);
    int i0 = n0 == nmap.end() ? leaf(x0) : n0->second;
    auto n1 = nmap.find(x1);
    int i1 = n1 == nmap.end() ? leaf(x1) : n1->second;
    if (i0 < 0 || i1 < 0 || ni >= cap) return -1;
    parent[i0] = ni; parent[i1] = ni;
    node_label[ni] = x2; parent[ni] = -1; child0[ni] = i0; child1[ni] = i1;
    nmap.emplace(x2, ni++);
  }
  return ni;
}


// util/struct_merge.hxx:188-210 -- same containers, same chain walk; pairs are emitted sorted by source key
int64_t orc_transform_keys(const orc_label* order, int64_t n_merges, orc_label* src, orc_label* dst, int64_t cap) {
  std::unordered_set<Label> newKeys;
  std::unordered_map<Label, Label> omap, lmap;
  for (int64_t i = 0; i < n_merges; ++i) {
    omap[order[3 * i]] = order[3 * i + 2];
    omap[order[3 * i + 1]] = order[3 * i + 2];
    newKeys.insert(order[3 * i + 2]);
  }
  for (auto const& op : omap) {
    if (newKeys.count(op.first) == 0) {
      Label d = op.second;
      auto oit = omap.find(d);
      while (oit != omap.end()) { d = oit->second; oit = omap.find(d); }
      lmap[op.first] = d;
    }
  }
  std::map<Label, Label> sorted(lmap.begin(), lmap.end());
  if ((int64_t)sorted.size() > cap) return -1;
  int64_t k = 0;
  for (auto const& lp : sorted) { src[k] = lp.first; dst[k] = lp.second; ++k; }
  return k;
}

// util/image.hxx:227-242 (mask + optional fill) -- in place
void orc_transform_image(orc_label* lab, int64_t n, const orc_label* src, const orc_label* dst, int64_t m,
                         const orc_label* mask, int fill_missing) {
  std::unordered_map<Label, Label> lmap;
  for (int64_t i = 0; i < m; ++i) lmap[src[i]] = dst[i];
  for (int64_t i = 0; i < n; ++i) {
    if (!mask || mask[i] != MASK_OUT_VAL) {
      auto lit = lmap.find(lab[i]);
      if (lit != lmap.end()) lab[i] = lit->second;
      else if (fill_missing) lab[i] = BG_VAL;
    }
  }
}

// util/image.hxx:992-1001 -> itk::RelabelComponentImageFilter (ITK absent: PARITY UNPINNED).  Restated from its
// documented behaviour: objects sorted by size (largest first, ties by smaller label), background 0 kept.
int64_t orc_relabel_image(orc_label* lab, int64_t n, int64_t min_size) {
  std::map<Label, int64_t> cnt;
  for (int64_t i = 0; i < n; ++i) if (lab[i] != BG_VAL) ++cnt[lab[i]];
  std::vector<std::pair<Label, int64_t>> objs(cnt.begin(), cnt.end());
  std::sort(objs.begin(), objs.end(), [](std::pair<Label, int64_t> const& a, std::pair<Label, int64_t> const& b) {
    return a.second != b.second ? a.second > b.second : a.first < b.first; });
  std::unordered_map<Label, Label> lmap;
  Label next = 1;
  for (auto const& o : objs) lmap[o.first] = (min_size > 0 && o.second < min_size) ? BG_VAL : next++;
  for (int64_t i = 0; i < n; ++i) if (lab[i] != BG_VAL) lab[i] = lmap[lab[i]];
  return (int64_t)next - 1;
}


// hmt/tree_build.hxx:41-63 (+ main_segment_greedy.cxx:46-59): potentials are attached while genTree builds the nodes
int64_t orc_tree_potentials(const orc_label* order, int64_t n_merges, const double* merge_probs, const double* region_probs,
                            orc_label* node_label, int32_t* parent, int32_t* child0, int32_t* child1, double* potential,
                            int64_t cap) {
  std::unordered_map<Label, int> nmap;
  int ni = 0;
  const double* mpit = merge_probs;
  auto visit = [&](int node, Label r) {           // the callback genTree applies to every new node
    node_label[node] = r;
    if (!merge_probs) { potential[node] = 1.0; return; }
    if (child0[node] >= 0) {
      potential[node] = *mpit;
      double pSplit = 1.0 - *mpit;
      for (int c : {child0[node], child1[node]}) {
        if (child0[c] < 0) potential[c] = pSplit * pSplit;
        else potential[c] *= pSplit;
      }
      ++mpit;
    }
  };
  auto add = [&](Label l, int c0, int c1) -> int {
    if (ni >= cap) return -1;
    parent[ni] = -1; child0[ni] = c0; child1[ni] = c1; potential[ni] = 0.0;
    visit(ni, l);
    nmap.emplace(l, ni);
    return ni++;
  };
  for (int64_t i = 0; i < n_merges; ++i) {
    Label x0 = order[3 * i], x1 = order[3 * i + 1], x2 = order[3 * i + 2];
    auto n0 = nmap.find(x0);
    int i0 = n0 == nmap.end() ? add(x0, -1, -1) : n0->second;
    auto n1 = nmap.find(x1);
    int i1 = n1 == nmap.end() ? add(x1, -1, -1) : n1->second;
    if (i0 < 0 || i1 < 0 || ni >= cap) return -1;
    parent[i0] = ni; parent[i1] = ni;
    add(x2, i0, i1);
  }
  if (merge_probs && ni > 0) potential[ni - 1] *= potential[ni - 1];
  if (region_probs) for (int i = 0; i < ni; ++i) potential[i] *= std::max(region_probs[i], FEPS);
  return ni;
}

// hmt/tree_greedy.hxx:76-92 + 104-152 for one tree: full scan per pick, validity flags, ancestors then BFS descendants
int64_t orc_resolve_tree_greedy(const int32_t* parent, const int32_t* child0, const int32_t* child1, const double* potential,
                                int64_t n, int32_t* picks, int64_t cap) {
  std::vector<bool> validity((size_t)n, true);
  int64_t np = 0;
  auto pickNode = [&]() {
    int ret = -1;
    for (int i = 0; i < n; ++i)
      if (validity[i] && (ret < 0 || potential[ret] < potential[i])) ret = i;
    return ret;
  };
  int pi = pickNode();
  while (pi >= 0) {
    if (np >= cap) return -1;
    picks[np++] = pi;
    validity[pi] = false;
    for (int a = parent[pi]; a >= 0; a = parent[a]) validity[a] = false;
    std::queue<int> q;
    for (int c : {child0[pi], child1[pi]}) if (c >= 0) q.push(c);
    while (!q.empty()) {
      int x = q.front(); q.pop();
      validity[x] = false;
      for (int c : {child0[x], child1[x]}) if (c >= 0) q.push(c);
    }
    pi = pickNode();
  }
  return np;
}

// hmt/tree_greedy.hxx:76-92 + 104-152, several trees: full scan per pick over all trees in order, BFS traversals
int64_t orc_resolve_trees_greedy(int n_trees, const int64_t* n_nodes, const orc_label* const* node_label, const int32_t* const* parent,
                                 const int32_t* const* child0, const int32_t* const* child1, const double* const* potential,
                                 int32_t* pick_tree, int32_t* pick_node, int64_t cap) {
  std::vector<std::vector<bool>> validity(n_trees);
  std::vector<std::unordered_map<Label, int>> lnmap(n_trees);
  auto bfs = [&](int t, int root, std::function<void(int)> f) {
    std::queue<int> q; q.push(root);
    while (!q.empty()) { int x = q.front(); q.pop(); f(x); for (int c : {child0[t][x], child1[t][x]}) if (c >= 0) q.push(c); }
  };
  for (int i = 0; i < n_trees; ++i) {
    validity[i].assign((size_t)n_nodes[i], true);
    if (n_nodes[i] > 0) {
      int root = (int)n_nodes[i] - 1;
      if (child0[i][root] < 0) lnmap[i][node_label[i][root]] = root;
      else for (int c : {child0[i][root], child1[i][root]}) bfs(i, c, [&](int x) { if (child0[i][x] < 0) lnmap[i][node_label[i][x]] = x; });
    }
  }
  auto pickNode = [&]() {
    std::pair<int, int> ret(-1, -1);
    for (int i = 0; i < n_trees; ++i)
      for (int x = 0; x < n_nodes[i]; ++x)
        if (validity[i][x] && (ret.first < 0 || potential[ret.first][ret.second] < potential[i][x])) ret = {i, x};
    return ret;
  };
  int64_t np = 0;
  auto pick = pickNode();
  std::vector<Label> llabels;
  while (pick.first >= 0) {
    if (np >= cap) return -1;
    pick_tree[np] = pick.first; pick_node[np] = pick.second; ++np;
    const int t = pick.first;
    validity[t][pick.second] = false;
    for (int a = parent[t][pick.second]; a >= 0; a = parent[t][a]) validity[t][a] = false;
    llabels.clear();
    for (int c : {child0[t][pick.second], child1[t][pick.second]})
      if (c >= 0) bfs(t, c, [&](int x) { validity[t][x] = false; if (child0[t][x] < 0) llabels.push_back(node_label[t][x]); });
    for (Label l : llabels)
      for (int i = 0; i < n_trees; ++i)
        if (i != t) {
          auto nit = lnmap[i].find(l);
          if (nit != lnmap[i].end()) {
            validity[i][nit->second] = false;
            for (int a = parent[i][nit->second]; a >= 0; a = parent[i][a]) validity[i][a] = false;
          }
        }
    pick = pickNode();
  }
  return np;
}

// genBoundaryConfidenceMap / genBoundaryConfidenceImage with all nodes (hmt/tree_segment.hxx:66-203) as
// main_segment_greedy.cxx:62-70 calls them: one copy of the contour-only region map per tree with the tree's merges
// applied (type/region_map.hxx:67-68), every node's surviving boundary keys vote max(float potential), the first map
// paints.  h must be a contour-only rag.
int orc_boundary_confidence(orc_rag* h, int n_trees, const orc_label* const* orders, const int64_t* n_merges,
                            const orc_label* const* node_label, const int64_t* n_nodes, const double* const* potential, float* out) {
  if (!h->onlyContour) return -1;
  typedef std::pair<Label, Label> KeyPair;
  std::unordered_map<KeyPair, float, PairHash> pbmap;
  auto fpb = [&](double val, KeyPair const& key01) {
    KeyPair key10 = key01.first < key01.second ? key01 : std::make_pair(key01.second, key01.first);
    auto pbit = pbmap.find(key10);
    if (pbit == pbmap.end()) pbmap[key10] = (float)val;
    else if (pbit->second < val) pbit->second = (float)val;
  };
  std::vector<RegionMap> rmaps(n_trees, h->rmap);
  for (int i = 0; i < n_trees; ++i)
    for (int64_t m = 0; m < n_merges[i]; ++m) rmaps[i].merge(orders[i][3 * m], orders[i][3 * m + 1], orders[i][3 * m + 2]);
  for (int i = 0; i < n_trees; ++i)
    for (int64_t x = 0; x < n_nodes[i]; ++x) {
      float val = (float)potential[i][x];                     // the node functor returns Real (= float)
      auto rit = rmaps[i].find(node_label[i][x]);
      if (rit == rmaps[i].end()) continue;                    // (the reference would dereference end() here)
      for (auto const& bp : rit->second.boundary) fpb(val, bp.first);
    }
  int64_t N = h->vol.size();
  for (int64_t p = 0; p < N; ++p) out[p] = 0.0f;
  RegionMap const& rmap = rmaps.front();
  for (auto const& pbp : pbmap) {
    KeyPair key = pbp.first;
    auto rit = rmap.find(pbp.first.first);
    if (rit != rmap.end()) {
      auto bit = rit->second.boundary.find(key);
      if (bit != rit->second.boundary.end()) for (auto p : *bit->second) if (out[p] < pbp.second) out[p] = pbp.second;
    }
    std::swap(key.first, key.second);
    rit = rmap.find(pbp.first.second);
    if (rit != rmap.end()) {
      auto bit = rit->second.boundary.find(key);
      if (bit != rit->second.boundary.end()) for (auto p : *bit->second) if (out[p] < pbp.second) out[p] = pbp.second;
    }
  }
  return 0;
}

// hmt/tree_segment.hxx:10-21: pairs sorted by source label (the reference fills an unordered_map)
int64_t orc_label_transform(const orc_label* node_label, const int32_t* child0, const int32_t* child1, int64_t n,
                            const int32_t* picks, int64_t n_picks, orc_label key, orc_label* src, orc_label* dst, int64_t cap) {
  std::map<Label, Label> lmap;
  for (int64_t k = 0; k < n_picks; ++k) {
    std::queue<int> q;
    q.push(picks[k]);
    while (!q.empty()) {
      int x = q.front(); q.pop();
      if (child0[x] < 0) lmap[node_label[x]] = key;
      else { q.push(child0[x]); q.push(child1[x]); }
    }
    ++key;
  }
  if ((int64_t)lmap.size() > cap) return -1;
  int64_t m = 0;
  for (auto const& lp : lmap) { src[m] = lp.first; dst[m] = lp.second; ++m; }
  return m;
}

// ---- morphological watershed (util/image_alg.hxx:9-21 = itk::MorphologicalWatershedImageFilter, level, no watershed line,
// face connectivity).  PARITY WITH ITK IS UNPINNED (ITK is not in this image, the reference holds no fixture): this restates the
// documented pipeline -- h-minima transform, regional minima of it as markers numbered in raster order, flooding of the ORIGINAL
// image from those markers (MorphologicalWatershedFromMarkers keeps the filter's input) -- with the same
// order-free tie rules as the device code (lowest flood level, then fewest steps since the level last rose, then the smaller
// label), by sequential algorithms of its own: a worklist reconstruction, breadth-first plateaus, Dijkstra flooding.
int64_t orc_watershed(int dim, const int64_t* dims, const float* img, double level, orc_label* out) {
  const int64_t nx = dims[0], ny = dims[1], nz = dim == 3 ? dims[2] : 1, n = nx * ny * nz;
  auto nbrs = [&](int64_t p, int64_t* q) {
    int k = 0;
    const int64_t x = p % nx, y = (p / nx) % ny, z = p / (nx * ny);
    if (x > 0) q[k++] = p - 1;
    if (x + 1 < nx) q[k++] = p + 1;
    if (y > 0) q[k++] = p - nx;
    if (y + 1 < ny) q[k++] = p + nx;
    if (dim == 3) { if (z > 0) q[k++] = p - nx * ny; if (z + 1 < nz) q[k++] = p + nx * ny; }
    return k;
  };
  // 1. reconstruction by erosion of (float)(f + level) above f: lower a voxel to max(f, a neighbour's value) until stable
  std::vector<float> g(n);
  for (int64_t p = 0; p < n; ++p) g[p] = (float)((double)img[p] + level);
  {
    std::deque<int64_t> work;
    std::vector<char> queued(n, 1);
    for (int64_t p = 0; p < n; ++p) work.push_back(p);
    int64_t q[6];
    while (!work.empty()) {
      const int64_t p = work.front(); work.pop_front(); queued[p] = 0;
      const int k = nbrs(p, q);
      for (int i = 0; i < k; ++i) {
        const float cand = std::max(img[q[i]], g[p]);
        if (cand < g[q[i]]) { g[q[i]] = cand; if (!queued[q[i]]) { queued[q[i]] = 1; work.push_back(q[i]); } }
      }
    }
  }
  // 2. plateaus in raster order; a plateau without a lower neighbour is a marker
  std::vector<orc_label> lab(n, 0);
  std::vector<char> seen(n, 0);
  orc_label nlab = 0;
  {
    std::vector<int64_t> plateau;
    int64_t q[6];
    for (int64_t s0 = 0; s0 < n; ++s0) {
      if (seen[s0]) continue;
      plateau.clear(); plateau.push_back(s0); seen[s0] = 1;
      bool lower = false;
      for (size_t h = 0; h < plateau.size(); ++h) {
        const int64_t p = plateau[h];
        const int k = nbrs(p, q);
        for (int i = 0; i < k; ++i) {
          if (g[q[i]] < g[p]) lower = true;
          else if (g[q[i]] == g[p] && !seen[q[i]]) { seen[q[i]] = 1; plateau.push_back(q[i]); }
        }
      }
      if (!lower) { ++nlab; for (int64_t p : plateau) lab[p] = nlab; }
    }
  }
  // 3. flooding: Dijkstra on the cost (flood level, steps since it last rose, label)
  struct St { float L; uint32_t d; orc_label l; int64_t p; };
  auto worse = [](St const& a, St const& b) { return a.L > b.L || (a.L == b.L && (a.d > b.d || (a.d == b.d && a.l > b.l))); };
  std::priority_queue<St, std::vector<St>, decltype(worse)> pq(worse);
  std::vector<float> L(n, std::numeric_limits<float>::infinity());
  std::vector<uint32_t> D(n, 0xFFFFFFFFu);
  std::vector<char> marker(n, 0), done(n, 0);
  for (int64_t p = 0; p < n; ++p) if (lab[p]) { marker[p] = 1; L[p] = img[p]; D[p] = 0; pq.push(St{img[p], 0u, lab[p], p}); }
  int64_t q[6];
  while (!pq.empty()) {
    const St s1 = pq.top(); pq.pop();
    if (done[s1.p] || s1.L != L[s1.p] || s1.d != D[s1.p] || s1.l != lab[s1.p]) continue;
    done[s1.p] = 1;
    const int k = nbrs(s1.p, q);
    for (int i = 0; i < k; ++i) {
      const int64_t t = q[i];
      if (marker[t] || done[t]) continue;
      const float Lc = std::max(s1.L, img[t]);
      const uint32_t dc = Lc == s1.L ? s1.d + 1u : 0u;
      if (lab[t] == 0 || Lc < L[t] || (Lc == L[t] && (dc < D[t] || (dc == D[t] && s1.l < lab[t])))) {
        L[t] = Lc; D[t] = dc; lab[t] = s1.l; pq.push(St{Lc, dc, s1.l, t});
      }
    }
  }
  for (int64_t p = 0; p < n; ++p) out[p] = lab[p];
  return (int64_t)nlab;
}

// The restatements of util/stats.hxx used by the features, exported so that tests can pin them against the reference's own
// header (oracle/_ref/ref_stats): out = entropy(a), entropy(b), distL1(a,b), distX2(a,b), amedian(a), amedian(b).
void orc_stats_case(int n, const double* a, const double* b, double* out) {
  std::vector<double> va(a, a + n), vb(b, b + n);
  out[0] = entropy(va); out[1] = entropy(vb); out[2] = distL1(va, vb); out[3] = distX2(va, vb);
  out[4] = amedian(va); out[5] = amedian(vb);
}
// util/stats.hxx:264-277 (MLP input scaling, main_merge_order_bc.cxx:130-137)
void orc_rescale(int n, double* feat, const double* mn, const double* mx, double out_min, double out_max) {
  const double outputDiff = out_max - out_min;
  for (int i = 0; i < n; ++i) feat[i] = outputDiff * (feat[i] - mn[i]) / (mx[i] - mn[i] + FEPS) + out_min;
}

// The host libm functions the reference's features call: std::log2 (util/stats.hxx:150), std::log (glia_base.hxx:80-81),
// std::pow(perim, 1.5) (type/feat.hxx:78-79).  function: 0 / 1 / 2.  volatile: no compile-time folding.
void orc_libm_eval(int function, const double* in, double* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    volatile double x = in[i];
    out[i] = function == 0 ? std::log2(x) : function == 1 ? std::log(x) : std::pow(x, 1.5);
  }
}

}  // extern "C"
