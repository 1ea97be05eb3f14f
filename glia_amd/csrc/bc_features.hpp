// glia_amd/csrc/bc_features.hpp -- K6: boundary-classifier feature vector from sufficient statistics (device).
//
// The reference re-walks the voxels of r0, r1 and r0 u r1 for every candidate edge
// (hmt/main_merge_order_bc.cxx:54-95, 98 % of its run time).  Every feature is a function of
// mergeable statistics -- counts, f64 sums, min/max, integer histograms, bounding boxes, thresholded counts --
// so here a feature vector costs O(D_f) flops.  Formulas follow, operation for operation (device code is built
// with -ffp-contract=off):
//   RegionShapeFeats::generate            type/feat.hxx:71-90     alg::getBoundingBox  alg/geometry.hxx:21-39
//   ImageRegionShapeFeats::generate       type/feat.hxx:485-502
//   ImageLabelFeats / ImageRealFeats      type/feat.hxx:632-638, 706-737   hist: util/image_stats.hxx:41-52
//   stats::entropy / distL1 / distX2      util/stats.hxx:145-152, 155-163, 177-185
//   RegionShapeDiffFeats / IntraDiffFeats type/feat.hxx:124-132, 176-186, 567-589
//   ImageDiffFeats                        type/feat.hxx:663-669, 801-810
//   RegionFeats / BoundaryFeats layout    hmt/bc_feat.hxx:69-77, 156-167, 232-238;  log(): feat.hxx:46-52,...
//   selectFeatures                        hmt/bc_feat.hxx:247-279
#pragma once
#include "hmt_internal.hpp"
#include "glibc_math.hpp"

namespace glia {

constexpr int kMaxFeat = 384;   // longest vector a kernel can hold (per-thread array of the initial-edge kernel); e.g. two 16-bin + two 8-bin image blocks with histogram columns = 314

// statistics of a set of boundary voxels (a commutative monoid under combine)
struct EStats {
  uint32_t n;
  uint32_t thr[GLIA_HMT_MAX_THRESH];
  float mn, mx;          // +inf / -inf when n == 0
  double sum, sq;
  uint32_t hist[GLIA_HMT_MAX_BINS];
};
// statistics of the voxels of a region
struct PStats {
  uint32_t n, border;
  int lo[3], hi[3];
  float mn, mx;
  double sum, sq;
  uint32_t hist[GLIA_HMT_MAX_BINS];
};

__host__ __device__ inline void estats_clear(EStats& s) {
  s.n = 0;
  for (int i = 0; i < GLIA_HMT_MAX_THRESH; ++i) s.thr[i] = 0;
  s.mn = __builtin_inff(); s.mx = -__builtin_inff(); s.sum = 0.0; s.sq = 0.0;
  for (int i = 0; i < GLIA_HMT_MAX_BINS; ++i) s.hist[i] = 0;
}
__host__ __device__ inline void estats_add(EStats& a, const EStats& b) {
  a.n += b.n;
  for (int i = 0; i < GLIA_HMT_MAX_THRESH; ++i) a.thr[i] += b.thr[i];
  a.mn = b.mn < a.mn ? b.mn : a.mn; a.mx = b.mx > a.mx ? b.mx : a.mx;
  a.sum += b.sum; a.sq += b.sq;
  for (int i = 0; i < GLIA_HMT_MAX_BINS; ++i) a.hist[i] += b.hist[i];
}
// additive part only (counts and sums); min/max untouched
__host__ __device__ inline void estats_sub_additive(EStats& a, const EStats& b) {
  a.n -= b.n;
  for (int i = 0; i < GLIA_HMT_MAX_THRESH; ++i) a.thr[i] -= b.thr[i];
  a.sum -= b.sum; a.sq -= b.sq;
  for (int i = 0; i < GLIA_HMT_MAX_BINS; ++i) a.hist[i] -= b.hist[i];
}
__host__ __device__ inline void pstats_add(PStats& a, const PStats& b) {
  a.n += b.n; a.border += b.border;
  for (int i = 0; i < 3; ++i) { a.lo[i] = b.lo[i] < a.lo[i] ? b.lo[i] : a.lo[i]; a.hi[i] = b.hi[i] > a.hi[i] ? b.hi[i] : a.hi[i]; }
  a.mn = b.mn < a.mn ? b.mn : a.mn; a.mx = b.mx > a.mx ? b.mx : a.mx;
  a.sum += b.sum; a.sq += b.sq;
  for (int i = 0; i < GLIA_HMT_MAX_BINS; ++i) a.hist[i] += b.hist[i];
}

struct BcCfg {
  int D, T, bins;                       // bins = bins of channel 0
  int K;                                // image channels (distinct volume + histogram), channel 0 = boundary probability
  int cbins[kMaxChannels];
  int n_region, n_rlabel, n_boundary;   // lengths of the three image lists (prepareImages, hmt/hmt_util.hxx:17-56)
  int rc[kMaxListed], lc[kMaxListed], bc[kMaxListed];   // list entry -> channel
  int use_log, use_simple;
  int use_hist;                         // GLIA_USE_HISTOGRAM_AS_FEATS: every image block carries its histogram ahead of the entropy
  double norm_area, norm_len;
  int rfdim, bfdim, fdim;
  int libm_log2, libm_log, libm_pow;    // which restatement of the host libm log2 / log / pow use (glibc_math.hpp)
};

// GLIA_BC_COMMON (one more pair of greedy_bc instances, hmt_internal.hpp): the configuration nearly every run has -- ONE image
// (the boundary probability map, on the region and the boundary list: "--rbi pb"), full feature vector, no --logs, no
// histogram columns -- with those switches and list lengths fixed at compile time (-7 % loop time)
#ifdef GLIA_BC_COMMON
#define BC_HIST(c) 0
#define BC_LOG(c) 0
#define BC_SIMPLE(c) 0
#define BC_K(c) 1
#define BC_NR(c) 1
#define BC_NL(c) 0
#define BC_NB(c) 1
#else
#define BC_HIST(c) (c).use_hist
#define BC_LOG(c) (c).use_log
#define BC_SIMPLE(c) (c).use_simple
#define BC_K(c) (c).K
#define BC_NR(c) (c).n_region
#define BC_NL(c) (c).n_rlabel
#define BC_NB(c) (c).n_boundary
#endif
// histogram columns of the image lists (0 unless use_hist): ImageLabelFeats::dim = histBin + 1 (type/feat.hxx:608-612)
__host__ __device__ inline int bc_hist_cols(const BcCfg& c, int kind) {
  if (!BC_HIST(c)) return 0;
  int n = 0;
  const int cnt = kind == 0 ? BC_NR(c) : kind == 1 ? BC_NL(c) : BC_NB(c);
  for (int i = 0; i < cnt; ++i) n += c.cbins[kind == 0 ? c.rc[i] : kind == 1 ? c.lc[i] : c.bc[i]];
  return n;
}
__host__ __device__ inline int bc_rf_dim(const BcCfg& c) {
  return 4 + c.D + 2 * c.T + 5 * BC_NR(c) + BC_NL(c) + 5 * BC_NB(c) + bc_hist_cols(c, 0) + bc_hist_cols(c, 1) + bc_hist_cols(c, 2);
}
__host__ __device__ inline int bc_bf_dim(const BcCfg& c) { return 11 + 4 * c.T + 7 * BC_NR(c) + 3 * BC_NL(c) + 5 * BC_NB(c) + bc_hist_cols(c, 2); }
__host__ __device__ inline int bc_full_dim(const BcCfg& c) { return bc_bf_dim(c) + 3 * bc_rf_dim(c); }
__host__ __device__ inline int bc_feat_dim(const BcCfg& c) {
  return BC_SIMPLE(c) ? 5 + BC_NB(c) + 4 * BC_NR(c) + 2 * BC_NL(c) : bc_bf_dim(c) + 3 * bc_rf_dim(c);
}

namespace feat {

__device__ inline double sdiv(double l, double r, double d) { return fabs(r) >= 2.22e-16 ? l / r : d; }
// std::log2 / std::log as the HOST libm computes them (glibc_math.hpp); variant 0 = device libm (<= 1 ulp off, unpinned).
// GLIA_LIBM_FIXED = log2 | log << 4 | pow << 8 fixes the variants at compile time (hmt_internal.hpp: greedy_bc instances).
#ifdef GLIA_LIBM_FIXED
#define GLIA_LIBM_SEL(run_time, shift) (((GLIA_LIBM_FIXED) >> (shift)) & 15)
#else
#define GLIA_LIBM_SEL(run_time, shift) (run_time)
#endif
__device__ __forceinline__ double host_log2(double x, int variant) { return GLIA_LIBM_SEL(variant, 0) == kLibmSse2 ? glibc::log2_sse2(x) : log2(x); }
// the same with the glibc tables taken from `tab` (a kernel's LDS copy: kLog2Head | kLog2Tab | kLog2Tab2)
__device__ __forceinline__ double host_log2(double x, int variant, const uint64_t* tab) {
  return GLIA_LIBM_SEL(variant, 0) == kLibmSse2 ? glibc::log2_sse2_tab(x, tab, tab + 18, tab + 18 + 128) : log2(x);
}
__device__ __forceinline__ double host_log(double x, int variant) {
  const int v = GLIA_LIBM_SEL(variant, 4);
  return v == kLibmFma ? glibc::log_fma(x) : v == kLibmSse2 ? glibc::log_sse2(x) : log(x);
}
__device__ inline double slog(double x, double d, int variant) { return x > 0.0 ? host_log(x, variant) : d; }
__device__ inline double ssqrt(double x, double d) { return x >= 0.0 ? sqrt(x) : d; }

// std::pow(perim, D/(D-1)) of type/feat.hxx:78-79 for an integer-valued perim: exponent 2 (2D) is an exact square (glibc's
// pow returns it: the exact value is representable and pow's error is far below half an ulp of it); exponent 1.5 (3D) is the
// restatement of the host's pow (glibc_math.hpp) the context probed.  Variant 0 (no restatement matched the host: UNPINNED)
// is perim*sqrt(perim) evaluated in double-double and rounded once, i.e. the correctly rounded value, which glibc's pow
// misses by one ulp on ~0.1 % of integer arguments.
__device__ inline double pow_perim(double x, int D, int variant) {
  if (D == 2) return x * x;
  if (!(x > 0.0)) return 0.0;
  const int v = GLIA_LIBM_SEL(variant, 8);
  if (v == kLibmFma) return glibc::pow_fma(x, 1.5);
  if (v == kLibmSse2) return glibc::pow_sse2(x, 1.5);
  const double s = sqrt(x);                       // correctly rounded
  const double r = __builtin_fma(-s, s, x);       // x - s*s, exact
  const double sl = r / (2.0 * s);                // sqrt(x) = s + sl (+ O(ulp^2))
  const double ph = x * s;
  const double pl = __builtin_fma(x, s, -ph);     // x*s = ph + pl exactly
  return ph + (pl + x * sl);
}

// the five ImageFeats numbers (entropy, mean, std, min, max) of a voxel set
struct ImgFeats { double entropy, mean, stddev, mn, mx; };

__device__ inline void region_log(const BcCfg& c, double* rf) {    // feat.hxx:46-52, 463-467
  const int v = c.libm_log;
  rf[0] = slog(rf[0], 0.0, v); rf[1] = slog(rf[1], 0.0, v); rf[3] = slog(rf[3], 0.0, v);
  for (int i = 0; i < c.D; ++i) rf[4 + i] = slog(rf[4 + i], 0.0, v);
  for (int i = 0; i < c.T; ++i) rf[4 + c.D + i] = slog(rf[4 + c.D + i], 0.0, v);
}

__device__ inline void boundary_log(const BcCfg& c, double* bf) {   // feat.hxx:103-106, 148-155, 531-539
  const int v = c.libm_log;
  bf[0] = slog(bf[0], 0.0, v); bf[3] = slog(bf[3], 0.0, v); bf[6] = slog(bf[6], 0.0, v);
  for (int i = 0; i < c.T; ++i) bf[11 + i] = slog(bf[11 + i], 0.0, v);
}

// ---- the vector of hmt/main_merge_order_bc.cxx:54-95, without private arrays --------------------------------------
// A first version kept four feature arrays and several statistics structs per thread; with run-time bin counts the
// compiler placed all of them in scratch memory and the greedy loop waited on scratch round trips (measured: 92 k
// cycles per vector).  The functions below read the statistics where they live, form the merged sets on the fly, loop
// over bins with compile-time bounds and write every feature straight to its final slot of `out` (LDS in the loop).
struct ImgSrc {               // an image-statistics set: hist = a + b - c (null pointers contribute nothing)
  const uint32_t* ha; const uint32_t* hb; const uint32_t* hc;
  uint32_t n; double sum, sq; float mn, mx;
  const double* ent;          // entropy computed beforehand (lane-parallel pass of the greedy loop), or null
  __device__ __forceinline__ uint32_t h(int i) const { return (ha ? ha[i] : 0u) + (hb ? hb[i] : 0u) - (hc ? hc[i] : 0u); }
};
__device__ __forceinline__ ImgFeats image_feats_src(const ImgSrc& s, int bins, int libm_log2) {
  ImgFeats f;
  double ent = 0.0;
  if (s.ent) ent = *s.ent;
  else {
#pragma unroll
    for (int i = 0; i < GLIA_HMT_MAX_BINS; ++i) {
      if (i < bins) {
        const double p = s.n ? s.h(i) / (double)s.n : 0.0;
        if (!(fabs(p - 0.0) < 2.22e-16)) ent -= p * host_log2(p, libm_log2);
      }
    }
  }
  f.entropy = ent; f.mean = 0.0; f.stddev = 0.0; f.mn = 0.0; f.mx = 0.0;
  const int ni = (int)s.n;
  if (ni != 0) {
    f.mean = s.sum / ni;
    f.stddev = ssqrt(s.sq / ni - f.mean * f.mean, 0.0);
    f.mn = (double)s.mn; f.mx = (double)s.mx;
  }
  return f;
}
// one bin's term of an entropy sum / of the two histogram distances (the lane-parallel pass adds them in bin order)
__device__ __forceinline__ double entropy_term(uint32_t cnt, uint32_t n, int libm_log2) {
  const double p = n ? cnt / (double)n : 0.0;
  return (fabs(p - 0.0) < 2.22e-16) ? 0.0 : p * host_log2(p, libm_log2);
}
// ... for the greedy loop: tables from LDS, and a full bin (p == 1: log2 is +0, the term 0.0) answered without the call -- one
// lane with p near 1 sends its whole wave through the near-one polynomial AND the table path of the restatement
__device__ __forceinline__ double entropy_term(uint32_t cnt, uint32_t n, int libm_log2, const uint64_t* tab) {
  const double p = n ? cnt / (double)n : 0.0;
  return ((fabs(p - 0.0) < 2.22e-16) | (cnt == n)) ? 0.0 : p * host_log2(p, libm_log2, tab);
}
__device__ __forceinline__ void dist_terms(uint32_t c0, uint32_t n0, uint32_t c1, uint32_t n1, double& tl, double& tx) {
  const double p0 = n0 ? c0 / (double)n0 : 0.0, p1 = n1 ? c1 / (double)n1 : 0.0;
  const double d = p0 - p1;
  tl = fabs(d);
  tx = (d * d) / (p0 + p1 + 2.22e-16);
}

// the normalised histogram of a voxel set (util/image_stats.hxx:45-52) as feature columns; returns their number
__device__ __forceinline__ int put_hist(const ImgSrc& s, int bins, double* out) {
#pragma unroll
  for (int i = 0; i < GLIA_HMT_MAX_BINS; ++i) if (i < bins) out[i] = s.n ? s.h(i) / (double)s.n : 0.0;
  return bins;
}

// Image lists: a source functor hands out the statistics of one list entry -- src(kind, i) with kind 0 = the voxel set on
// region-list image i, 1 = the voxel set on label-list image i, 2 = the boundary set on boundary-list image i.
struct ShapeIn { uint32_t n, border; int lo[3], hi[3]; uint32_t bn; uint32_t thr[GLIA_HMT_MAX_THRESH]; };

// layout of the values the lane-parallel pass precomputes per record: region / label entry i -> 5 doubles (entropy of
// the first, second, merged voxel set, L1, chi-square), boundary entry i -> 4 (entropy of the first, second, merged
// boundary set and of the shared boundary)
__host__ __device__ inline int pre_region(const BcCfg& c, int kind, int i) { return 5 * ((kind ? BC_NR(c) : 0) + i); }
__host__ __device__ inline int pre_boundary(const BcCfg& c, int i) { return 5 * (BC_NR(c) + BC_NL(c)) + 4 * i; }
__host__ __device__ inline int pre_count(const BcCfg& c) { return 5 * (BC_NR(c) + BC_NL(c)) + 4 * BC_NB(c); }

// half: 0 = the whole block; 1 = its first 4 + D columns (area, perimeter, compactness, bounding box); 2 = the rest (thresholded
// counts and the image lists) -- the greedy loop's helpers give the halves of a block to two waves (two chains of divisions of about
// equal length)
template <class Src>
__device__ __forceinline__ void region_feats_multi(const BcCfg& c, const ShapeIn& r, Src src, double* out, double& area_o, double& perim_o,
                                                   int half = 0) {
  const int D = c.D, T = c.T;
  int k = 4 + D;
  if (half != 2) {
    double area = (double)r.n;
    double perim = (double)((unsigned long long)r.bn + (unsigned long long)r.border);
    const double compactness = sdiv(pow_perim(perim, D, c.libm_pow), area, 0.0);
    area = sdiv(area, c.norm_area, 0.0);
    perim = sdiv(perim, c.norm_len, 0.0);
    double bboxArea = 1.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i < D) {
        const double bb = (double)(unsigned long long)(r.hi[i] - r.lo[i]);
        out[4 + i] = sdiv(bb, c.norm_len, 0.0);
        bboxArea *= bb;
      }
    }
    out[0] = area; out[1] = perim; out[2] = compactness; out[3] = sdiv(bboxArea, c.norm_area, 0.0);
    area_o = area; perim_o = perim;
  }
  if (half == 1) return;
#pragma unroll
  for (int i = 0; i < GLIA_HMT_MAX_THRESH; ++i) if (i < T) out[k + i] = sdiv((double)r.thr[i], c.norm_len, 0.0);
#pragma unroll
  for (int i = 0; i < GLIA_HMT_MAX_THRESH; ++i) if (i < T) out[k + T + i] = sdiv((double)r.thr[i], (double)r.bn, 0.0);
  k += 2 * T;
  for (int i = 0; i < BC_NR(c); ++i) {
    const ImgSrc s = src(0, i);
    const ImgFeats f = image_feats_src(s, c.cbins[c.rc[i]], c.libm_log2);
    if (BC_HIST(c)) k += put_hist(s, c.cbins[c.rc[i]], out + k);
    out[k++] = f.entropy; out[k++] = f.mean; out[k++] = f.stddev; out[k++] = f.mn; out[k++] = f.mx;
  }
  for (int i = 0; i < BC_NL(c); ++i) {
    const ImgSrc s = src(1, i);
    if (BC_HIST(c)) k += put_hist(s, c.cbins[c.lc[i]], out + k);
    out[k++] = image_feats_src(s, c.cbins[c.lc[i]], c.libm_log2).entropy;
  }
  for (int i = 0; i < BC_NB(c); ++i) {
    const ImgSrc s = src(2, i);
    const ImgFeats f = image_feats_src(s, c.cbins[c.bc[i]], c.libm_log2);
    if (BC_HIST(c)) k += put_hist(s, c.cbins[c.bc[i]], out + k);
    out[k++] = f.entropy; out[k++] = f.mean; out[k++] = f.stddev; out[k++] = f.mn; out[k++] = f.mx;
  }
}

// shn / shthr: voxel and thresholded counts of the shared boundary; src0 / src1: the area-ordered regions (kinds 0, 1);
// srcSh(i): the shared boundary on boundary-list image i; pre: precomputed distances (see pre_region) or null
// half: 0 = the whole block; 1 = the area / perimeter / length columns and the image lists; 2 = the 4 T thresholded-length columns
template <class Src0, class Src1, class SrcSh>
__device__ __forceinline__ void boundary_feats_multi(const BcCfg& c, uint32_t shn, const uint32_t* shthr, double a0area, double a0perim,
                                                     double a1area, double a1perim, Src0 src0, Src1 src1, SrcSh srcSh, const double* pre,
                                                     double* out, int half = 0) {
  const int T = c.T;
  int k = 0;
  const double bl = sdiv(ceil(shn / 2.0), c.norm_len, 0.0);
  if (half != 2) {
    const double areaDiff = fabs(a0area - a1area);
    out[k++] = areaDiff; out[k++] = sdiv(areaDiff, a0area, 0.0); out[k++] = sdiv(areaDiff, a1area, 0.0);
    const double perimDiff = fabs(a0perim - a1perim);
    out[k++] = perimDiff; out[k++] = sdiv(perimDiff, a0perim, 0.0); out[k++] = sdiv(perimDiff, a1perim, 0.0);
    out[k++] = bl; out[k++] = sdiv(bl, a0area, 0.0); out[k++] = sdiv(bl, a1area, 0.0);
    out[k++] = sdiv(bl, a0perim, 0.0); out[k++] = sdiv(bl, a1perim, 0.0);
  }
  k = 11;
  if (half != 1) {
#pragma unroll
    for (int i = 0; i < GLIA_HMT_MAX_THRESH; ++i) {
      if (i < T) {
        const double vbl = sdiv(ceil(shthr[i] / 2.0), c.norm_len, 0.0);
        out[k + i] = vbl; out[k + T + i] = sdiv(vbl, bl, 0.0);
        out[k + 2 * T + i] = sdiv(vbl, a0perim, 0.0); out[k + 3 * T + i] = sdiv(vbl, a1perim, 0.0);
      }
    }
  }
  if (half == 2) return;
  k += 4 * T;
  for (int kind = 0; kind < 2; ++kind) {
    const int cnt = kind ? BC_NL(c) : BC_NR(c);
    for (int i = 0; i < cnt; ++i) {
      const ImgSrc s0 = src0(kind, i), s1 = src1(kind, i);
      const int bins = c.cbins[kind ? c.lc[i] : c.rc[i]];
      double l1 = 0.0, x2 = 0.0;
      if (pre) { l1 = pre[pre_region(c, kind, i) + 3]; x2 = pre[pre_region(c, kind, i) + 4]; }
      else {
#pragma unroll
        for (int b = 0; b < GLIA_HMT_MAX_BINS; ++b) {
          if (b < bins) {
            double tl, tx;
            dist_terms(s0.h(b), s0.n, s1.h(b), s1.n, tl, tx);
            l1 += tl; x2 += tx;
          }
        }
      }
      const ImgFeats f0 = image_feats_src(s0, bins, c.libm_log2), f1 = image_feats_src(s1, bins, c.libm_log2);
      out[k++] = l1; out[k++] = x2; out[k++] = fabs(f0.entropy - f1.entropy);
      if (kind == 0) {
        out[k++] = fabs(f0.mean - f1.mean); out[k++] = fabs(f0.stddev - f1.stddev);
        out[k++] = fabs(f0.mn - f1.mn); out[k++] = fabs(f0.mx - f1.mx);
      }
    }
  }
  for (int i = 0; i < BC_NB(c); ++i) {
    const ImgSrc s = srcSh(i);
    const ImgFeats f = image_feats_src(s, c.cbins[c.bc[i]], c.libm_log2);
    if (BC_HIST(c)) k += put_hist(s, c.cbins[c.bc[i]], out + k);
    out[k++] = f.entropy; out[k++] = f.mean; out[k++] = f.stddev; out[k++] = f.mn; out[k++] = f.mx;
  }
}

// log() and selectFeatures applied in place to a vector laid out as [boundary | region 0 | region 1 | merged]
// the slots boundary_log / region_log touch, in order; returns their number (<= kMaxLogSlots)
constexpr int kMaxLogSlots = 3 + GLIA_HMT_MAX_THRESH + 3 * (3 + 3 + GLIA_HMT_MAX_THRESH);
template <class Out>
__device__ __forceinline__ int log_slots(const BcCfg& c, Out* pos) {
  int n = 0;
  pos[n++] = 0; pos[n++] = 3; pos[n++] = 6;
  for (int i = 0; i < c.T; ++i) pos[n++] = (Out)(11 + i);
  for (int b = 0; b < 3; ++b) {
    const int o = c.bfdim + b * c.rfdim;
    pos[n++] = (Out)(o + 0); pos[n++] = (Out)(o + 1); pos[n++] = (Out)(o + 3);
    for (int i = 0; i < c.D + c.T; ++i) pos[n++] = (Out)(o + 4 + i);
  }
  return n;
}
__device__ __forceinline__ void simple_selection(const BcCfg& c, double* out);
__device__ __forceinline__ void finish_features(const BcCfg& c, double* out) {
  if (BC_LOG(c)) {
    boundary_log(c, out);
    region_log(c, out + c.bfdim); region_log(c, out + c.bfdim + c.rfdim); region_log(c, out + c.bfdim + 2 * c.rfdim);
  }
  simple_selection(c, out);
}
__device__ __forceinline__ void simple_selection(const BcCfg& c, double* out) {
  if (BC_SIMPLE(c)) {   // hmt/bc_feat.hxx:247-279; every source index lies beyond the slot it is copied to
    const double* bf = out; const double* x1 = out + c.bfdim; const double* x2 = out + c.bfdim + c.rfdim;
    const double v0 = x1[0], v1 = x2[0], v2 = x1[1], v3 = x2[1], v4 = bf[6];
    const int r = 11 + 4 * c.T, rl = r + 7 * BC_NR(c), bimg = rl + 3 * BC_NL(c);
    int k = 0;
    out[k++] = v0; out[k++] = v1; out[k++] = v2; out[k++] = v3; out[k++] = v4;     // slots 0..4 < every source below (>= 11)
    for (int i = 0, o = bimg; i < BC_NB(c); ++i) {            // the mean of the shared boundary on boundary image i
      const int hb = BC_HIST(c) ? c.cbins[c.bc[i]] : 0;
      out[k++] = bf[o + hb + 1];
      o += hb + 5;
    }
    for (int i = 0; i < BC_NR(c); ++i) {
      const double m = bf[r + 7 * i + 3], l1 = bf[r + 7 * i + 0], xx = bf[r + 7 * i + 1], en = bf[r + 7 * i + 2];
      out[k++] = m; out[k++] = l1; out[k++] = xx; out[k++] = en;
    }
    for (int i = 0; i < BC_NL(c); ++i) { const double l1 = bf[rl + 3 * i + 0], xx = bf[rl + 3 * i + 1]; out[k++] = l1; out[k++] = xx; }
  }
}

}  // namespace feat
}  // namespace glia
