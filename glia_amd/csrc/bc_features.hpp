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

namespace glia {

constexpr int kMaxFeat = 160;   // 11+4T+7+3+5 + 3*(4+D+2T+5+1+5) with D=3, T=4

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
  int D, T, bins;
  int n_region, n_rlabel, n_boundary;   // 0 or 1 each: all lists share one image volume (see rag_build)
  int use_log, use_simple;
  double norm_area, norm_len;
  int rfdim, bfdim, fdim;
};

__host__ __device__ inline int bc_rf_dim(const BcCfg& c) { return 4 + c.D + 2 * c.T + 5 * c.n_region + c.n_rlabel + 5 * c.n_boundary; }
__host__ __device__ inline int bc_bf_dim(const BcCfg& c) { return 11 + 4 * c.T + 7 * c.n_region + 3 * c.n_rlabel + 5 * c.n_boundary; }
__host__ __device__ inline int bc_full_dim(const BcCfg& c) { return bc_bf_dim(c) + 3 * bc_rf_dim(c); }
__host__ __device__ inline int bc_feat_dim(const BcCfg& c) {
  return c.use_simple ? 5 + c.n_boundary + 4 * c.n_region + 2 * c.n_rlabel : bc_bf_dim(c) + 3 * bc_rf_dim(c);
}

namespace feat {

__device__ inline double sdiv(double l, double r, double d) { return fabs(r) >= 2.22e-16 ? l / r : d; }
__device__ inline double slog(double x, double d) { return x > 0.0 ? log(x) : d; }
__device__ inline double ssqrt(double x, double d) { return x >= 0.0 ? sqrt(x) : d; }

// std::pow(perim, D/(D-1)) of type/feat.hxx:78-79 for an integer-valued perim: exponent 2 (2D) is an exact
// square; exponent 1.5 (3D) is perim*sqrt(perim) evaluated in double-double and rounded once, i.e. the correctly
// rounded value, which is what glibc's pow returns except in vanishingly rare near-tie cases.
__device__ inline double pow_perim(double x, int D) {
  if (D == 2) return x * x;
  if (!(x > 0.0)) return 0.0;
  const double s = sqrt(x);                       // correctly rounded
  const double r = __builtin_fma(-s, s, x);       // x - s*s, exact
  const double sl = r / (2.0 * s);                // sqrt(x) = s + sl (+ O(ulp^2))
  const double ph = x * s;
  const double pl = __builtin_fma(x, s, -ph);     // x*s = ph + pl exactly
  return ph + (pl + x * sl);
}

// histogram -> entropy (normalised by the set's voxel count), also returns the normalised histogram
__device__ inline double hist_entropy(const uint32_t* hc, uint32_t n, int bins, double* h) {
  double ent = 0.0;
  for (int i = 0; i < bins; ++i) {
    double p = n ? hc[i] / (double)n : 0.0;
    h[i] = p;
    if (!(fabs(p - 0.0) < 2.22e-16)) ent -= p * log2(p);
  }
  return ent;
}

// the five ImageFeats numbers (entropy, mean, std, min, max) of a voxel set
struct ImgFeats { double entropy, mean, stddev, mn, mx; };
__device__ inline ImgFeats image_feats(const uint32_t* hc, uint32_t n, double sum, double sq, float mn, float mx, int bins,
                                       double* h) {
  ImgFeats f;
  f.entropy = hist_entropy(hc, n, bins, h);
  f.mean = 0.0; f.stddev = 0.0; f.mn = 0.0; f.mx = 0.0;
  const int ni = (int)n;
  if (ni != 0) {
    f.mean = sum / ni;
    f.stddev = ssqrt(sq / ni - f.mean * f.mean, 0.0);
    f.mn = (double)mn; f.mx = (double)mx;
  }
  return f;
}

// RegionFeats of one region: p = its voxels, b = its un-cancelled boundary set.  Writes rfdim doubles to out and
// the pieces BoundaryFeats needs to `aux` (area, perim after normalisation + region-image feats + histograms).
struct RegionAux {
  double area, perim;
  ImgFeats rimg, rlimg;
  double rh[GLIA_HMT_MAX_BINS];
};

__device__ inline void region_feats(const BcCfg& c, const PStats& p, const EStats& b, double* out, RegionAux& aux) {
  const int D = c.D, T = c.T;
  double area = (double)p.n;
  double perim = (double)((unsigned long long)b.n + (unsigned long long)p.border);
  double compactness = sdiv(pow_perim(perim, D), area, 0.0);
  area = sdiv(area, c.norm_area, 0.0);
  perim = sdiv(perim, c.norm_len, 0.0);
  double bboxArea = 1.0;
  double bboxSize[3];
  for (int i = 0; i < D; ++i) {
    double bb = (double)(unsigned long long)(p.hi[i] - p.lo[i]);
    bboxSize[i] = sdiv(bb, c.norm_len, 0.0);
    bboxArea *= bb;
  }
  bboxArea = sdiv(bboxArea, c.norm_area, 0.0);
  int k = 0;
  out[k++] = area; out[k++] = perim; out[k++] = compactness; out[k++] = bboxArea;
  for (int i = 0; i < D; ++i) out[k++] = bboxSize[i];
  for (int i = 0; i < T; ++i) out[k++] = sdiv((double)b.thr[i], c.norm_len, 0.0);
  for (int i = 0; i < T; ++i) out[k++] = sdiv((double)b.thr[i], (double)b.n, 0.0);
  aux.area = area; aux.perim = perim;
  double htmp[GLIA_HMT_MAX_BINS];
  if (c.n_region) {
    aux.rimg = image_feats(p.hist, p.n, p.sum, p.sq, p.mn, p.mx, c.bins, aux.rh);
    out[k++] = aux.rimg.entropy; out[k++] = aux.rimg.mean; out[k++] = aux.rimg.stddev; out[k++] = aux.rimg.mn; out[k++] = aux.rimg.mx;
  }
  if (c.n_rlabel) {
    aux.rlimg.entropy = hist_entropy(p.hist, p.n, c.bins, aux.rh);
    out[k++] = aux.rlimg.entropy;
  }
  if (c.n_boundary) {
    ImgFeats f = image_feats(b.hist, b.n, b.sum, b.sq, b.mn, b.mx, c.bins, htmp);
    out[k++] = f.entropy; out[k++] = f.mean; out[k++] = f.stddev; out[k++] = f.mn; out[k++] = f.mx;
  }
}

__device__ inline void region_log(const BcCfg& c, double* rf) {    // feat.hxx:46-52, 463-467
  rf[0] = slog(rf[0], 0.0); rf[1] = slog(rf[1], 0.0); rf[3] = slog(rf[3], 0.0);
  for (int i = 0; i < c.D; ++i) rf[4 + i] = slog(rf[4 + i], 0.0);
  for (int i = 0; i < c.T; ++i) rf[4 + c.D + i] = slog(rf[4 + c.D + i], 0.0);
}

// BoundaryFeats from the shared boundary set `sh` and the two (area-ordered) regions
__device__ inline void boundary_feats(const BcCfg& c, const EStats& sh, const RegionAux& a0, const RegionAux& a1, double* out) {
  const int T = c.T;
  int k = 0;
  const double areaDiff = fabs(a0.area - a1.area);
  out[k++] = areaDiff; out[k++] = sdiv(areaDiff, a0.area, 0.0); out[k++] = sdiv(areaDiff, a1.area, 0.0);
  const double perimDiff = fabs(a0.perim - a1.perim);
  out[k++] = perimDiff; out[k++] = sdiv(perimDiff, a0.perim, 0.0); out[k++] = sdiv(perimDiff, a1.perim, 0.0);
  const double bl = sdiv(ceil(sh.n / 2.0), c.norm_len, 0.0);
  out[k++] = bl; out[k++] = sdiv(bl, a0.area, 0.0); out[k++] = sdiv(bl, a1.area, 0.0);
  out[k++] = sdiv(bl, a0.perim, 0.0); out[k++] = sdiv(bl, a1.perim, 0.0);
  double vbl[GLIA_HMT_MAX_THRESH];
  for (int i = 0; i < T; ++i) { vbl[i] = sdiv(ceil(sh.thr[i] / 2.0), c.norm_len, 0.0); out[k++] = vbl[i]; }
  for (int i = 0; i < T; ++i) out[k++] = sdiv(vbl[i], bl, 0.0);
  for (int i = 0; i < T; ++i) out[k++] = sdiv(vbl[i], a0.perim, 0.0);
  for (int i = 0; i < T; ++i) out[k++] = sdiv(vbl[i], a1.perim, 0.0);
  if (c.n_region || c.n_rlabel) {
    double l1 = 0.0, x2 = 0.0;
    for (int i = 0; i < c.bins; ++i) {
      const double d = a0.rh[i] - a1.rh[i];
      l1 += fabs(d);
      x2 += (d * d) / (a0.rh[i] + a1.rh[i] + 2.22e-16);
    }
    if (c.n_region) {
      out[k++] = l1; out[k++] = x2; out[k++] = fabs(a0.rimg.entropy - a1.rimg.entropy);
      out[k++] = fabs(a0.rimg.mean - a1.rimg.mean); out[k++] = fabs(a0.rimg.stddev - a1.rimg.stddev);
      out[k++] = fabs(a0.rimg.mn - a1.rimg.mn); out[k++] = fabs(a0.rimg.mx - a1.rimg.mx);
    }
    if (c.n_rlabel) {
      const double e0 = c.n_region ? a0.rimg.entropy : a0.rlimg.entropy, e1 = c.n_region ? a1.rimg.entropy : a1.rlimg.entropy;
      out[k++] = l1; out[k++] = x2; out[k++] = fabs(e0 - e1);
    }
  }
  if (c.n_boundary) {
    double htmp[GLIA_HMT_MAX_BINS];
    ImgFeats f = image_feats(sh.hist, sh.n, sh.sum, sh.sq, sh.mn, sh.mx, c.bins, htmp);
    out[k++] = f.entropy; out[k++] = f.mean; out[k++] = f.stddev; out[k++] = f.mn; out[k++] = f.mx;
  }
}

__device__ inline void boundary_log(const BcCfg& c, double* bf) {   // feat.hxx:103-106, 148-155, 531-539
  bf[0] = slog(bf[0], 0.0); bf[3] = slog(bf[3], 0.0); bf[6] = slog(bf[6], 0.0);
  for (int i = 0; i < c.T; ++i) bf[11 + i] = slog(bf[11 + i], 0.0);
}

// The whole vector of hmt/main_merge_order_bc.cxx:54-95.  (p0,b0) / (p1,b1) are the regions in the orientation
// the reference passes them (reg0, reg1), (p2,b2) the scratch-merged region, sh their shared boundary.
__device__ inline void bc_features(const BcCfg& c, const PStats& p0, const EStats& b0, const PStats& p1, const EStats& b1,
                                   const PStats& p2, const EStats& b2, const EStats& sh, double* out) {
  double rf0[48], rf1[48], rf2[48], bf[48];
  RegionAux a0, a1, a2;
  region_feats(c, p0, b0, rf0, a0);
  region_feats(c, p1, b1, rf1, a1);
  region_feats(c, p2, b2, rf2, a2);
  // keep region 0 area <= region 1 area (main_merge_order_bc.cxx:77-80)
  const bool swap = a0.area > a1.area;
  const double* x1 = swap ? rf1 : rf0;
  const double* x2 = swap ? rf0 : rf1;
  boundary_feats(c, sh, swap ? a1 : a0, swap ? a0 : a1, bf);
  if (c.use_log) { boundary_log(c, bf); region_log(c, rf0); region_log(c, rf1); region_log(c, rf2); }
  int k = 0;
  if (c.use_simple) {   // hmt/bc_feat.hxx:247-279
    out[k++] = x1[0]; out[k++] = x2[0]; out[k++] = x1[1]; out[k++] = x2[1]; out[k++] = bf[6];
    const int bimg = 11 + 4 * c.T + 7 * c.n_region + 3 * c.n_rlabel;
    if (c.n_boundary) out[k++] = bf[bimg + 1];
    if (c.n_region) { const int r = 11 + 4 * c.T; out[k++] = bf[r + 3]; out[k++] = bf[r + 0]; out[k++] = bf[r + 1]; out[k++] = bf[r + 2]; }
    if (c.n_rlabel) { const int r = 11 + 4 * c.T + 7 * c.n_region; out[k++] = bf[r + 0]; out[k++] = bf[r + 1]; }
    return;
  }
  for (int i = 0; i < c.bfdim; ++i) out[k++] = bf[i];
  for (int i = 0; i < c.rfdim; ++i) out[k++] = x1[i];
  for (int i = 0; i < c.rfdim; ++i) out[k++] = x2[i];
  for (int i = 0; i < c.rfdim; ++i) out[k++] = rf2[i];
}


// ---- the same vector without private arrays ------------------------------------------------------------------------
// bc_features() above keeps four feature arrays and several statistics structs per thread; with run-time bin counts the
// compiler places all of them in scratch memory, and the greedy loop then waits on scratch round trips (measured: 92 k
// cycles per vector).  The variant below reads the statistics where they live, forms the merged sets on the fly, loops
// over bins with compile-time bounds and writes every feature straight to its final slot of `out` (LDS in the loop).
struct ImgSrc {               // an image-statistics set: hist = a + b - c (null pointers contribute nothing)
  const uint32_t* ha; const uint32_t* hb; const uint32_t* hc;
  uint32_t n; double sum, sq; float mn, mx;
  const double* ent;          // entropy computed beforehand (lane-parallel pass of the greedy loop), or null
  __device__ __forceinline__ uint32_t h(int i) const { return (ha ? ha[i] : 0u) + (hb ? hb[i] : 0u) - (hc ? hc[i] : 0u); }
};
__device__ __forceinline__ ImgFeats image_feats_src(const ImgSrc& s, int bins) {
  ImgFeats f;
  double ent = 0.0;
  if (s.ent) ent = *s.ent;
  else {
#pragma unroll
    for (int i = 0; i < GLIA_HMT_MAX_BINS; ++i) {
      if (i < bins) {
        const double p = s.n ? s.h(i) / (double)s.n : 0.0;
        if (!(fabs(p - 0.0) < 2.22e-16)) ent -= p * log2(p);
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
struct RegionIn {             // what RegionFeats::generate reads of one region
  uint32_t n, border; int lo[3], hi[3];
  ImgSrc pimg;                // statistics over the region's voxels
  uint32_t bn; uint32_t thr[GLIA_HMT_MAX_THRESH];
  ImgSrc bimg;                // statistics over its boundary set
};
struct RegionOut { double area, perim; ImgFeats rimg; };
__device__ __forceinline__ void region_feats_direct(const BcCfg& c, const RegionIn& r, double* out, RegionOut& aux) {
  const int D = c.D, T = c.T;
  double area = (double)r.n;
  double perim = (double)((unsigned long long)r.bn + (unsigned long long)r.border);
  const double compactness = sdiv(pow_perim(perim, D), area, 0.0);
  area = sdiv(area, c.norm_area, 0.0);
  perim = sdiv(perim, c.norm_len, 0.0);
  double bboxArea = 1.0;
  int k = 4;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (i < D) {
      const double bb = (double)(unsigned long long)(r.hi[i] - r.lo[i]);
      out[k++] = sdiv(bb, c.norm_len, 0.0);
      bboxArea *= bb;
    }
  }
  out[0] = area; out[1] = perim; out[2] = compactness; out[3] = sdiv(bboxArea, c.norm_area, 0.0);
#pragma unroll
  for (int i = 0; i < GLIA_HMT_MAX_THRESH; ++i) if (i < T) out[k + i] = sdiv((double)r.thr[i], c.norm_len, 0.0);
#pragma unroll
  for (int i = 0; i < GLIA_HMT_MAX_THRESH; ++i) if (i < T) out[k + T + i] = sdiv((double)r.thr[i], (double)r.bn, 0.0);
  k += 2 * T;
  aux.area = area; aux.perim = perim;
  aux.rimg = image_feats_src(r.pimg, c.bins);           // entropy is shared by the region- and label-image features
  if (c.n_region) { out[k++] = aux.rimg.entropy; out[k++] = aux.rimg.mean; out[k++] = aux.rimg.stddev; out[k++] = aux.rimg.mn; out[k++] = aux.rimg.mx; }
  if (c.n_rlabel) out[k++] = aux.rimg.entropy;
  if (c.n_boundary) {
    const ImgFeats f = image_feats_src(r.bimg, c.bins);
    out[k++] = f.entropy; out[k++] = f.mean; out[k++] = f.stddev; out[k++] = f.mn; out[k++] = f.mx;
  }
}

// sh: shared boundary set (thresholds + image statistics); h0/n0, h1/n1: voxel histograms of the area-ordered regions
__device__ __forceinline__ void boundary_feats_direct(const BcCfg& c, uint32_t shn, const uint32_t* shthr, const ImgSrc& shimg,
                                                      const RegionOut& a0, const RegionOut& a1, const uint32_t* h0, uint32_t n0,
                                                      const uint32_t* h1, uint32_t n1, const double* l1x2, double* out) {
  const int T = c.T;
  int k = 0;
  const double areaDiff = fabs(a0.area - a1.area);
  out[k++] = areaDiff; out[k++] = sdiv(areaDiff, a0.area, 0.0); out[k++] = sdiv(areaDiff, a1.area, 0.0);
  const double perimDiff = fabs(a0.perim - a1.perim);
  out[k++] = perimDiff; out[k++] = sdiv(perimDiff, a0.perim, 0.0); out[k++] = sdiv(perimDiff, a1.perim, 0.0);
  const double bl = sdiv(ceil(shn / 2.0), c.norm_len, 0.0);
  out[k++] = bl; out[k++] = sdiv(bl, a0.area, 0.0); out[k++] = sdiv(bl, a1.area, 0.0);
  out[k++] = sdiv(bl, a0.perim, 0.0); out[k++] = sdiv(bl, a1.perim, 0.0);
#pragma unroll
  for (int i = 0; i < GLIA_HMT_MAX_THRESH; ++i) {
    if (i < T) {
      const double vbl = sdiv(ceil(shthr[i] / 2.0), c.norm_len, 0.0);
      out[k + i] = vbl; out[k + T + i] = sdiv(vbl, bl, 0.0);
      out[k + 2 * T + i] = sdiv(vbl, a0.perim, 0.0); out[k + 3 * T + i] = sdiv(vbl, a1.perim, 0.0);
    }
  }
  k += 4 * T;
  if (c.n_region || c.n_rlabel) {
    double l1 = 0.0, x2 = 0.0;
    if (l1x2) { l1 = l1x2[0]; x2 = l1x2[1]; }
    else {
#pragma unroll
      for (int i = 0; i < GLIA_HMT_MAX_BINS; ++i) {
        if (i < c.bins) {
          const double p0 = n0 ? h0[i] / (double)n0 : 0.0, p1 = n1 ? h1[i] / (double)n1 : 0.0;
          const double d = p0 - p1;
          l1 += fabs(d);
          x2 += (d * d) / (p0 + p1 + 2.22e-16);
        }
      }
    }
    if (c.n_region) {
      out[k++] = l1; out[k++] = x2; out[k++] = fabs(a0.rimg.entropy - a1.rimg.entropy);
      out[k++] = fabs(a0.rimg.mean - a1.rimg.mean); out[k++] = fabs(a0.rimg.stddev - a1.rimg.stddev);
      out[k++] = fabs(a0.rimg.mn - a1.rimg.mn); out[k++] = fabs(a0.rimg.mx - a1.rimg.mx);
    }
    if (c.n_rlabel) { out[k++] = l1; out[k++] = x2; out[k++] = fabs(a0.rimg.entropy - a1.rimg.entropy); }
  }
  if (c.n_boundary) {
    const ImgFeats f = image_feats_src(shimg, c.bins);
    out[k++] = f.entropy; out[k++] = f.mean; out[k++] = f.stddev; out[k++] = f.mn; out[k++] = f.mx;
  }
}

// one bin's term of an entropy sum / of the two histogram distances (the lane-parallel pass adds them in bin order)
__device__ __forceinline__ double entropy_term(uint32_t cnt, uint32_t n) {
  const double p = n ? cnt / (double)n : 0.0;
  return (fabs(p - 0.0) < 2.22e-16) ? 0.0 : p * log2(p);
}
__device__ __forceinline__ void dist_terms(uint32_t c0, uint32_t n0, uint32_t c1, uint32_t n1, double& tl, double& tx) {
  const double p0 = n0 ? c0 / (double)n0 : 0.0, p1 = n1 ? c1 / (double)n1 : 0.0;
  const double d = p0 - p1;
  tl = fabs(d);
  tx = (d * d) / (p0 + p1 + 2.22e-16);
}

// log() and selectFeatures applied in place to a vector laid out as [boundary | region 0 | region 1 | merged]
__device__ __forceinline__ void finish_features(const BcCfg& c, double* out) {
  if (c.use_log) {
    boundary_log(c, out);
    region_log(c, out + c.bfdim); region_log(c, out + c.bfdim + c.rfdim); region_log(c, out + c.bfdim + 2 * c.rfdim);
  }
  if (c.use_simple) {   // hmt/bc_feat.hxx:247-279; every source index lies beyond the slot it is copied to
    const double* bf = out; const double* x1 = out + c.bfdim; const double* x2 = out + c.bfdim + c.rfdim;
    const double v0 = x1[0], v1 = x2[0], v2 = x1[1], v3 = x2[1], v4 = bf[6];
    const int bimg = 11 + 4 * c.T + 7 * c.n_region + 3 * c.n_rlabel, r = 11 + 4 * c.T, rl = 11 + 4 * c.T + 7 * c.n_region;
    const double b1 = c.n_boundary ? bf[bimg + 1] : 0.0;
    const double g3 = c.n_region ? bf[r + 3] : 0.0, g0 = c.n_region ? bf[r + 0] : 0.0, g1 = c.n_region ? bf[r + 1] : 0.0, g2 = c.n_region ? bf[r + 2] : 0.0;
    const double l0 = c.n_rlabel ? bf[rl + 0] : 0.0, l1 = c.n_rlabel ? bf[rl + 1] : 0.0;
    int k = 0;
    out[k++] = v0; out[k++] = v1; out[k++] = v2; out[k++] = v3; out[k++] = v4;
    if (c.n_boundary) out[k++] = b1;
    if (c.n_region) { out[k++] = g3; out[k++] = g0; out[k++] = g1; out[k++] = g2; }
    if (c.n_rlabel) { out[k++] = l0; out[k++] = l1; }
  }
}

}  // namespace feat
}  // namespace glia
