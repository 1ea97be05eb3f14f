// glia_amd/csrc/median_feats.hip -- GLIA_USE_MEDIAN_AS_FEATS for a GIVEN merge order (bc_feat).
//
// Reference: with GLIA_USE_MEDIAN_AS_FEATS (code/CMakeLists.txt:55,62-64) ImageRealFeats gains `median = stats::amedian(values)`
// ahead of the mean and takes mean / standard deviation from the VECTOR of values (type/feat.hxx:677-722: stats::mean, stats::var --
// the mean of (x - mean)^2, not E[x^2] - mean^2), and ImageDiffFeats gains |median0 - median1| (:772-808).  stats::amedian is the
// order statistic at n / 2 (util/stats.hxx:83-91).  That holds for every voxel set a feature row looks at (hmt/bc_feat.hxx:88-214):
//   P(R)      the voxels of a region R                    -- R = the two regions of a merge and the region it creates
//   B(R)      R's boundary voxels (TRegion::merge, type/region.hxx:66-75: the directed leaf entries (a -> b), a in R, minus every
//             MUTUAL pair whose two leaves are both inside R; a non-mutual entry is never cancelled)
//   Sh(R0,R1) the shared boundary (getBoundary / boundaryWith, util/struct.hxx:10-16, type/region.hxx:42-51): entries (a -> b) of
//             B(R0) whose target leaf b lies in R1 and still owns an un-cancelled entry there, and the symmetric set
// A statistic cannot give an order statistic; the value multisets themselves are needed.  For a GIVEN order (hmt/main_bc_feat.cxx) the
// whole merge tree is known beforehand, so every set is a list of RUNS of two value arrays built once per image:
//   * voxel values grouped by leaf, the leaves laid out in depth-first order of the tree: P(R) is ONE contiguous run;
//   * boundary-voxel values grouped by directed leaf pair (the neighbour rule of type/neighbor.hxx:109-126): B(R) and Sh are lists of
//     pair runs -- the host walks every entry up the tree from its source leaf to the merge that joins it with its target (there a
//     mutual entry is cancelled and becomes part of that merge's shared boundary; a non-mutual one goes on to the root).
// The device gathers the runs of a batch of sets into one buffer, sorts every set (rocPRIM segmented radix sort) and reduces it: median
// = sorted[n / 2] (bit-exact, an order statistic of f32 values), mean = sum / n, stddev = sqrt(sum (x - mean)^2 / n) in f64 -- the
// reference adds in the order nth_element leaves, which depends on rand() (stats.hxx:87): these two columns are comparable to 1e-12
// relative, every other column bit for bit.  Memory bound: the sets of one merge order hold sum over tree nodes of |P| + |B|
// values (the volume times the mean depth of a leaf); they are processed in batches of at most 2^27 values, one set may not exceed 2^30.
// The greedy LOOP with this layout (sets that change with every contraction) is not implemented: GLIA_HMT_ERR_UNSUPPORTED.
#include <algorithm>
#include <cmath>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include "greedy_common.hpp"

namespace glia {

namespace {

constexpr unsigned long long kBatchValues = 1ull << 27;      // values gathered, sorted and reduced at a time
constexpr unsigned long long kSetLimit = 1ull << 30;         // one set

__global__ void mf_gather_u32(const uint32_t* rec, long long n, int words, int word, uint32_t* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = rec[(size_t)i * words + word];
}

// voxel values grouped by leaf: dst = leaf_off[leaf] + running count (the order inside a leaf does not matter: the sets are sorted)
__global__ void mf_scatter_regions(VolumeRef vol, const float* img, const uint32_t* rlabel, uint32_t R, const unsigned long long* leaf_off, uint32_t* cursor, float* out) {
  const long long N = vol.nx * vol.ny * vol.nz;
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  const uint32_t t = vol.lab[p];
  if (t == kMaskedLabel) return;                          // masked-out centre (point-map mode: util/struct.hxx:86-91)
  const uint32_t leaf = find_label(rlabel, R, t);
  if (leaf >= R || rlabel[leaf] != t) return;
  out[leaf_off[leaf] + atomicAdd(&cursor[leaf], 1u)] = img[p];
}

// boundary-voxel values grouped by directed pair, the neighbour rule of type/neighbor.hxx:109-126 (masked-out neighbours are invalid)
__global__ void mf_scatter_pairs(VolumeRef vol, const float* img, const uint32_t* pa, const uint32_t* pb, long long P, const unsigned long long* off, uint32_t* cursor,
                                 float* out) {
  const long long N = vol.nx * vol.ny * vol.nz;
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  const long long x = p % vol.nx, y = (p / vol.nx) % vol.ny, z = p / (vol.nx * vol.ny);
  const uint32_t t = vol.lab[p];
  if (t == kMaskedLabel) return;
  uint32_t nb = t;
  const long long sy = vol.nx, sz = vol.nx * vol.ny;
  const uint32_t* L = vol.lab_nb;
  do {
    uint32_t q;
    if (x > 0 && (q = L[p - 1]) != t && q != kMaskedLabel) { nb = q; break; }
    if (x + 1 < vol.nx && (q = L[p + 1]) != t && q != kMaskedLabel) { nb = q; break; }
    if (y > 0 && (q = L[p - sy]) != t && q != kMaskedLabel) { nb = q; break; }
    if (y + 1 < vol.ny && (q = L[p + sy]) != t && q != kMaskedLabel) { nb = q; break; }
    if (vol.dim == 3) {
      if (z > 0 && (q = L[p - sz]) != t && q != kMaskedLabel) { nb = q; break; }
      if (z + 1 < vol.nz && (q = L[p + sz]) != t && q != kMaskedLabel) { nb = q; break; }
    }
  } while (false);
  if (nb == t) return;
  const long long i = find_pair(pa, pb, P, t, nb);
  if (i < 0) return;
  out[off[i] + atomicAdd(&cursor[i], 1u)] = img[p];
}

// the runs of a batch of sets into one buffer: element k of the buffer belongs to the run r with dst[r] <= k < dst[r + 1]
__global__ void mf_expand(const float* src, const unsigned long long* run_src, const unsigned long long* run_dst, uint32_t n_runs, unsigned long long total, float* out) {
  const unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= total) return;
  uint32_t lo = 0, hi = n_runs;                // last run with dst <= k
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (run_dst[mid] <= k) lo = mid; else hi = mid; }
  out[k] = src[run_src[lo] + (k - run_dst[lo])];
}

// one workgroup per sorted set: median, mean, standard deviation (two passes, f64)
__global__ __launch_bounds__(256) void mf_set_stats(const float* sorted, const uint32_t* seg_off, uint32_t n_sets, double* out) {
  __shared__ double part[256];
  const uint32_t s = blockIdx.x;
  if (s >= n_sets) return;
  const uint32_t b = seg_off[s], e = seg_off[s + 1], n = e - b;
  double acc = 0.0;
  for (uint32_t i = b + threadIdx.x; i < e; i += 256) acc += (double)sorted[i];
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int d = 128; d >= 1; d >>= 1) { if ((int)threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d]; __syncthreads(); }
  const double mean = n ? part[0] / (double)n : 0.0;             // stats::mean (util/stats.hxx:55-57); an empty set leaves 0 (feat.hxx:708-709)
  __syncthreads();
  acc = 0.0;
  for (uint32_t i = b + threadIdx.x; i < e; i += 256) { const double dx = (double)sorted[i] - mean; acc += dx * dx; }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int d = 128; d >= 1; d >>= 1) { if ((int)threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d]; __syncthreads(); }
  if (threadIdx.x == 0) {
    const double var = n ? part[0] / (double)n : 0.0;            // stats::var (:60-69)
    out[3 * (size_t)s + 0] = n ? (double)sorted[b + n / 2] : 0.0;     // stats::amedian (:83-91)
    out[3 * (size_t)s + 1] = mean;
    out[3 * (size_t)s + 2] = var >= 0.0 ? sqrt(var) : 0.0;
  }
}

struct Run { unsigned long long src; unsigned long long len; };

// median / mean / stddev of every set; set j = runs[first[j] .. first[j + 1]) of the device array `src`
int set_stats(const float* src, const std::vector<Run>& runs, const std::vector<uint32_t>& first, hipStream_t stream, std::vector<double>* out) {
  const size_t J = first.size() - 1;
  out->assign(3 * J, 0.0);
  size_t j0 = 0;
  while (j0 < J) {
    // a batch: consecutive sets up to kBatchValues values (one oversized set goes alone)
    size_t j1 = j0;
    unsigned long long total = 0;
    while (j1 < J) {
      unsigned long long n = 0;
      for (uint32_t r = first[j1]; r < first[j1 + 1]; ++r) n += runs[r].len;
      if (n > kSetLimit) { set_error("bc_feat (median features): a voxel set of this merge order holds " + std::to_string(n) + " values, more than the bound of 2^30"); return GLIA_HMT_ERR_UNSUPPORTED; }
      if (j1 > j0 && total + n > kBatchValues) break;
      total += n; ++j1;
    }
    const uint32_t nr = first[j1] - first[j0], ns = (uint32_t)(j1 - j0);
    if (total) {
      std::vector<unsigned long long> h_src(nr), h_dst((size_t)nr + 1);
      std::vector<uint32_t> h_seg((size_t)ns + 1);
      unsigned long long d = 0;
      for (size_t j = j0; j < j1; ++j) {
        h_seg[j - j0] = (uint32_t)d;
        for (uint32_t r = first[j]; r < first[j + 1]; ++r) { h_src[r - first[j0]] = runs[r].src; h_dst[r - first[j0]] = d; d += runs[r].len; }
      }
      h_seg[ns] = (uint32_t)d; h_dst[nr] = d;
      DeviceBuffers buf;
      int rc;
      unsigned long long *d_src, *d_dst; uint32_t* d_seg; float *d_a, *d_b; double* d_out;
      if ((rc = buf.get(&d_src, nr ? nr : 1, false, stream)) || (rc = buf.get(&d_dst, (size_t)nr + 1, false, stream)) || (rc = buf.get(&d_seg, (size_t)ns + 1, false, stream)) ||
          (rc = buf.get(&d_a, (size_t)total, false, stream)) || (rc = buf.get(&d_b, (size_t)total, false, stream)) || (rc = buf.get(&d_out, 3 * (size_t)ns, false, stream))) return rc;
      GLIA_HIP_TRY(hipMemcpyAsync(d_src, h_src.data(), 8 * (size_t)nr, hipMemcpyHostToDevice, stream));
      GLIA_HIP_TRY(hipMemcpyAsync(d_dst, h_dst.data(), 8 * ((size_t)nr + 1), hipMemcpyHostToDevice, stream));
      GLIA_HIP_TRY(hipMemcpyAsync(d_seg, h_seg.data(), 4 * ((size_t)ns + 1), hipMemcpyHostToDevice, stream));
      hipLaunchKernelGGL(mf_expand, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, d_src, d_dst, nr, total, d_a);
      size_t tmp = 0;
      GLIA_HIP_TRY(rocprim::segmented_radix_sort_keys(nullptr, tmp, d_a, d_b, (unsigned)total, ns, d_seg, d_seg + 1, 0, 32, stream));
      char* d_tmp;
      if ((rc = buf.get(&d_tmp, tmp ? tmp : 16, false, stream))) return rc;
      GLIA_HIP_TRY(rocprim::segmented_radix_sort_keys((void*)d_tmp, tmp, d_a, d_b, (unsigned)total, ns, d_seg, d_seg + 1, 0, 32, stream));
      hipLaunchKernelGGL(mf_set_stats, dim3(ns), dim3(256), 0, stream, d_b, d_seg, ns, d_out);
      GLIA_HIP_TRY(hipGetLastError());
      GLIA_HIP_TRY(hipMemcpyAsync(out->data() + 3 * j0, d_out, sizeof(double) * 3 * (size_t)ns, hipMemcpyDeviceToHost, stream));
      GLIA_HIP_TRY(hipStreamSynchronize(stream));      // (the host vectors and the buffers of this batch die here)
    }
    j0 = j1;
  }
  return GLIA_HMT_OK;
}

}  // namespace

int median_feature_stats(const MedianFeatIn& in, hipStream_t stream, std::vector<double>* reg, std::vector<double>* bnd, std::vector<unsigned long long>* area_out) {
  const RagArrays& rag = *in.rag;
  const uint32_t R = (uint32_t)rag.R;
  const long long P = rag.P;
  const int64_t M = in.n_merges;
  const long long N = in.vol.nx * in.vol.ny * in.vol.nz;
  if (!in.vol.lab) { set_error("bc_feat (median features): needs the volumes the region map was built from (whole-volume build)"); return GLIA_HMT_ERR_UNSUPPORTED; }
  DeviceBuffers buf;
  int rc;
  // ---- host copies of the map's structure ----
  std::vector<uint32_t> lab(R), pa((size_t)P), pb((size_t)P), leaf_n(R), pair_n((size_t)P);
  {
    uint32_t *d_ln, *d_pn;
    if ((rc = buf.get(&d_ln, R, false, stream)) || (rc = buf.get(&d_pn, (size_t)(P ? P : 1), false, stream))) return rc;
    hipLaunchKernelGGL(mf_gather_u32, dim3((R + 255) / 256), dim3(256), 0, stream, rag.d_rrec, (long long)R, kRegionWords, R_CNT, d_ln);
    if (P) hipLaunchKernelGGL(mf_gather_u32, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, rag.d_prec, P, kPairWords, P_CNT, d_pn);
    GLIA_HIP_TRY(hipMemcpyAsync(lab.data(), rag.d_rlabel, 4 * (size_t)R, hipMemcpyDeviceToHost, stream));
    GLIA_HIP_TRY(hipMemcpyAsync(leaf_n.data(), d_ln, 4 * (size_t)R, hipMemcpyDeviceToHost, stream));
    if (P) {
      GLIA_HIP_TRY(hipMemcpyAsync(pa.data(), rag.d_pa, 4 * (size_t)P, hipMemcpyDeviceToHost, stream));
      GLIA_HIP_TRY(hipMemcpyAsync(pb.data(), rag.d_pb, 4 * (size_t)P, hipMemcpyDeviceToHost, stream));
      GLIA_HIP_TRY(hipMemcpyAsync(pair_n.data(), d_pn, 4 * (size_t)P, hipMemcpyDeviceToHost, stream));
    }
    GLIA_HIP_TRY(hipStreamSynchronize(stream));
  }
  auto leaf_of = [&](uint32_t label) { return (uint32_t)(std::lower_bound(lab.begin(), lab.end(), label) - lab.begin()); };
  // ---- the merge tree: leaves 0 .. R-1, the region of merge i = R + i; depth-first leaf positions, node = interval of positions ----
  const size_t nn = (size_t)R + (size_t)M;
  std::vector<int64_t> parent(nn, -1);
  for (int64_t i = 0; i < M; ++i) { parent[in.forced[2 * i]] = (int64_t)R + i; parent[in.forced[2 * i + 1]] = (int64_t)R + i; }
  std::vector<uint32_t> pos(R), at(R), lo(nn), hi(nn);            // position of a leaf; leaf at a position; interval [lo, hi) of a node
  {
    uint32_t next = 0;
    std::vector<std::pair<size_t, int>> stack;
    for (size_t root = nn; root-- > 0;) {                        // (any order of the roots will do)
      if (parent[root] >= 0) continue;
      stack.push_back({root, 0});
      while (!stack.empty()) {
        auto& top = stack.back();
        const size_t x = top.first;
        if (x < R) { pos[x] = next; at[next] = (uint32_t)x; lo[x] = next; hi[x] = ++next; stack.pop_back(); continue; }
        if (top.second == 0) { lo[x] = next; top.second = 1; stack.push_back({in.forced[2 * (x - R)], 0}); continue; }
        if (top.second == 1) { top.second = 2; stack.push_back({in.forced[2 * (x - R) + 1], 0}); continue; }
        hi[x] = next; stack.pop_back();
      }
    }
  }
  std::vector<unsigned long long> voff((size_t)R + 1, 0), leaf_off(R);      // voxel-value offsets by position / by leaf
  for (uint32_t q = 0; q < R; ++q) { voff[q + 1] = voff[q] + leaf_n[at[q]]; leaf_off[at[q]] = voff[q]; }
  area_out->assign(nn, 0);
  for (size_t x = 0; x < nn; ++x) (*area_out)[x] = voff[hi[x]] - voff[lo[x]];
  // the nodes a row looks at: both regions of every merge and the region it creates
  std::vector<uint8_t> needed(nn, 0);
  for (int64_t i = 0; i < M; ++i) { needed[in.forced[2 * i]] = needed[in.forced[2 * i + 1]] = 1; needed[(size_t)R + i] = 1; }
  std::vector<uint32_t> set_of(nn, 0xFFFFFFFFu);                           // node -> its set index
  uint32_t n_sets = 0;
  for (size_t x = 0; x < nn; ++x) if (needed[x]) set_of[x] = n_sets++;

  // ---- region sets: one run each ----
  reg->assign((size_t)M * 3 * (in.n_r > 0 ? in.n_r : 0) * 3, 0.0);
  if (in.n_r > 0 && M > 0) {
    std::vector<Run> runs(n_sets);
    std::vector<uint32_t> first((size_t)n_sets + 1);
    for (size_t x = 0; x < nn; ++x) if (needed[x]) { runs[set_of[x]] = {voff[lo[x]], voff[hi[x]] - voff[lo[x]]}; }
    for (uint32_t j = 0; j <= n_sets; ++j) first[j] = j;
    unsigned long long* d_leaf_off; uint32_t* d_cursor; float* d_v;
    if ((rc = buf.get(&d_leaf_off, R, false, stream)) || (rc = buf.get(&d_cursor, R, false, stream)) || (rc = buf.get(&d_v, (size_t)(voff[R] ? voff[R] : 1), false, stream))) return rc;
    GLIA_HIP_TRY(hipMemcpyAsync(d_leaf_off, leaf_off.data(), 8 * (size_t)R, hipMemcpyHostToDevice, stream));
    for (int c = 0; c < in.n_r; ++c) {
      GLIA_HIP_TRY(hipMemsetAsync(d_cursor, 0, 4 * (size_t)R, stream));
      hipLaunchKernelGGL(mf_scatter_regions, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, in.vol, in.r_img[c], rag.d_rlabel, R, d_leaf_off, d_cursor, d_v);
      GLIA_HIP_TRY(hipGetLastError());
      std::vector<double> st;
      if ((rc = set_stats(d_v, runs, first, stream, &st))) return rc;
      for (int64_t i = 0; i < M; ++i) {
        const size_t node[3] = {in.forced[2 * i], in.forced[2 * i + 1], (size_t)R + i};
        for (int k = 0; k < 3; ++k)
          for (int q = 0; q < 3; ++q) (*reg)[(((size_t)i * 3 + k) * in.n_r + c) * 3 + q] = st[3 * (size_t)set_of[node[k]] + q];
      }
    }
  }

  // ---- boundary sets: lists of directed-pair runs ----
  bnd->assign((size_t)M * 4 * (in.n_b > 0 ? in.n_b : 0) * 3, 0.0);
  if (in.n_b > 0 && M > 0) {
    // per directed entry: leaves, mutual partner, value run
    std::vector<uint32_t> ea((size_t)P), eb((size_t)P);
    std::vector<uint8_t> mutual((size_t)P, 0);
    std::vector<unsigned long long> pv_off((size_t)P + 1, 0);
    for (long long i = 0; i < P; ++i) { ea[i] = leaf_of(pa[i]); eb[i] = leaf_of(pb[i]); pv_off[i + 1] = pv_off[i] + pair_n[i]; }
    auto find_entry = [&](uint32_t a, uint32_t b) -> long long {      // pairs ascend by (a, b)
      long long l = 0, h = P;
      while (l < h) { const long long mid = (l + h) >> 1; if (pa[mid] < a || (pa[mid] == a && pb[mid] < b)) l = mid + 1; else h = mid; }
      return (l < P && pa[l] == a && pb[l] == b) ? l : -1;
    };
    // "alive" test of a target leaf inside a node (boundaryWith): the leaf owns a non-mutual entry, or a mutual partner outside the node
    std::vector<uint32_t> nm_out(R, 0), pmin(R, 0xFFFFFFFFu), pmax(R, 0);
    for (long long i = 0; i < P; ++i) {
      mutual[i] = find_entry(pb[i], pa[i]) >= 0;
      if (!mutual[i]) nm_out[ea[i]] += 1;
      else { pmin[ea[i]] = std::min(pmin[ea[i]], pos[eb[i]]); pmax[ea[i]] = std::max(pmax[ea[i]], pos[eb[i]]); }
    }
    auto alive_in = [&](uint32_t leaf, size_t node) { return nm_out[leaf] > 0 || (pmin[leaf] != 0xFFFFFFFFu && (pmin[leaf] < lo[node] || pmax[leaf] >= hi[node])); };
    // incidences (set, entry): B(x) for the needed nodes (sets 0 .. n_sets-1), Sh of merge i (set n_sets + i); counted, then filled
    const size_t n_all = (size_t)n_sets + (size_t)M;
    std::vector<uint32_t> first(n_all + 1, 0);
    std::vector<Run> runs;
    for (int pass = 0; pass < 2; ++pass) {
      std::vector<uint32_t> fill;
      std::vector<Run>* runs_p = nullptr;
      if (pass == 1) {
        uint32_t s = 0;
        for (size_t j = 0; j < n_all; ++j) { const uint32_t c = first[j]; first[j] = s; s += c; }
        first[n_all] = s;
        runs.assign(s, Run{0, 0});
        fill.assign(first.begin(), first.end() - 1);
        runs_p = &runs;
      }
      auto add = [&](size_t set, long long e) {
        if (pass == 0) first[set] += 1;
        else (*runs_p)[fill[set]++] = {pv_off[e], pv_off[e + 1] - pv_off[e]};
      };
      for (long long e = 0; e < P; ++e) {
        if (pair_n[e] == 0) continue;
        const uint32_t a = ea[e], b = eb[e], pb_pos = pos[b];
        for (int64_t x = a; x >= 0; x = parent[x]) {
          const bool inside = pb_pos >= lo[x] && pb_pos < hi[x];        // the target leaf lies in x: x is where the pair is joined (or above)
          if (inside && x >= (int64_t)R) {
            const size_t m = (size_t)x - R;
            // the entry belongs to the shared boundary of the merge that joins the two leaves (only there: above, both are inside)
            const size_t c0 = in.forced[2 * m], c1 = in.forced[2 * m + 1];
            const bool first_join = !(pb_pos >= lo[c0] && pb_pos < hi[c0] && pos[a] >= lo[c0] && pos[a] < hi[c0]) &&
                                    !(pb_pos >= lo[c1] && pb_pos < hi[c1] && pos[a] >= lo[c1] && pos[a] < hi[c1]);
            if (first_join) {
              const size_t side_b = (pb_pos >= lo[c0] && pb_pos < hi[c0]) ? c0 : c1;
              if (mutual[e] || alive_in(b, side_b)) add((size_t)n_sets + m, e);
            }
            if (mutual[e]) break;                                     // cancelled from here on (type/region.hxx:66-75)
          }
          if (needed[x]) add(set_of[x], e);                            // un-cancelled: part of B(x)
        }
      }
      if (pass == 1) {
        unsigned long long* d_off; uint32_t* d_cursor; float* d_pv;
        if ((rc = buf.get(&d_off, (size_t)P + 1, false, stream)) || (rc = buf.get(&d_cursor, (size_t)(P ? P : 1), false, stream)) ||
            (rc = buf.get(&d_pv, (size_t)(pv_off[P] ? pv_off[P] : 1), false, stream))) return rc;
        GLIA_HIP_TRY(hipMemcpyAsync(d_off, pv_off.data(), 8 * ((size_t)P + 1), hipMemcpyHostToDevice, stream));
        for (int c = 0; c < in.n_b; ++c) {
          GLIA_HIP_TRY(hipMemsetAsync(d_cursor, 0, 4 * (size_t)(P ? P : 1), stream));
          hipLaunchKernelGGL(mf_scatter_pairs, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, in.vol, in.b_img[c], rag.d_pa, rag.d_pb, P, d_off, d_cursor, d_pv);
          GLIA_HIP_TRY(hipGetLastError());
          std::vector<double> st;
          if ((rc = set_stats(d_pv, runs, first, stream, &st))) return rc;
          for (int64_t i = 0; i < M; ++i) {
            const size_t sets[4] = {set_of[in.forced[2 * i]], set_of[in.forced[2 * i + 1]], set_of[(size_t)R + i], (size_t)n_sets + (size_t)i};
            for (int k = 0; k < 4; ++k)
              for (int q = 0; q < 3; ++q) (*bnd)[(((size_t)i * 4 + k) * in.n_b + c) * 3 + q] = st[3 * sets[k] + q];
          }
        }
      }
    }
  }
  return GLIA_HMT_OK;
}

}  // namespace glia
