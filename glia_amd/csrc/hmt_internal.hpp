// glia_amd/csrc/hmt_internal.hpp -- shared definitions of the HIP implementation (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/glia_hmt.h"

namespace glia {

void set_error(const std::string& msg);
// process-wide tuning / test switches (api.cpp; glia_hmt_set_option): true when `key` is set, its text in *value
bool option(const char* key, std::string* value = nullptr);
// merge loops: calls that ended with GLIA_HMT_ERR_INTERNAL (a kernel's own consistency stop, or an order that fails the replay)
unsigned long long internal_errors();
void count_internal_error();
// dense ids: every merge k joins two regions that still exist and creates region R + k (util/struct_merge.hxx:19-31)
bool merge_order_is_consistent(const uint32_t* dense_order, int64_t n, uint32_t R, int64_t* first_bad);
#define GLIA_HIP_TRY(expr)                                                                       \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      ::glia::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                      \
      return GLIA_HMT_ERR_HIP;                                                                   \
    }                                                                                            \
  } while (0)

// ---- accumulation tile geometry -------------------------------------------------------------
constexpr int kTileX = 64;         // one wave = one row of 64 voxels, one voxel per lane
#ifndef GLIA_ACC_TILEY
#define GLIA_ACC_TILEY 24
#endif
constexpr int kTileY = GLIA_ACC_TILEY;   // rows of a tile = marching waves x rows per wave (rag_accumulate.hip, Geo)
constexpr int kTZ = 32;            // planes a workgroup marches through (a run of a lane has at most kTZ voxels: 8-bit counters)

// ---- record layouts (32-bit words).  All zero == "empty": minima / lower bounds are stored
// complemented so that every reduction is an add or an unsigned max and tables initialise by memset.
// region record
constexpr int R_CNT = 0, R_BORDER = 1, R_LO = 2 /*3: 0x7fffffff-lo*/, R_HI = 5 /*3: hi+1*/, R_SUM = 8 /*f64*/,
              R_SQ = 10 /*f64*/, R_MIN = 12 /*~ord*/, R_MAX = 13 /*ord*/, R_FIRST = 14 /*u64 ~idx*/, R_HIST = 16;
constexpr int kRegionWords = 32;
// directed pair record
constexpr int P_CNT = 0, P_MIN = 1, P_MAX = 2, P_THR = 4 /*4*/, P_SUM = 8, P_SQ = 10, P_HIST = 12;
constexpr int kPairWords = 32;      // global stride (128-byte lines)


struct HistSpec {
  int bins;
  float fb[GLIA_HMT_MAX_BINS];  // smallest float >= bounds[i] (util/image_stats.hxx:17-22), +inf padded
  float lo_f;                   // largest float <= range.first   (val >  lo  <=> val >  lo_f ; val <= lo <=> val <= lo_f)
  float hi_f;                   // smallest float >= range.second (val <  hi  <=> val <  hi_f)
};

struct AccParams {
  const uint32_t* lab;               // labels; with a mask: the folded copy (masked-out voxels hold kMaskedLabel)
  const uint32_t* lab_c;             // labels of the centre voxels (differs from lab only for contour-only mode + mask)
  int masked;                        // a mask was given
  const float* img;
  int64_t nx, ny, nz;                // extent of the arrays handed in (nz includes halo planes of a slab)
  int64_t gz0, gnz;                  // slab mode: global z of local plane 0 and global depth (gz0 = 0, gnz = nz otherwise)
  int64_t zb, ze;                    // local planes [zb, ze) are accumulated (the halo planes only feed the neighbour rule)
  int dim;
  int nbx, nby, nbz;
  int tz;                            // planes a workgroup marches through: kTZ, or fewer when supervoxels are small
  HistSpec hist;
  int nthr;
  float thr_f[GLIA_HMT_MAX_THRESH];  // smallest float >= threshold (val >= thr <=> val >= thr_f)
  uint32_t* rkeys;                   // label + 1, 0 = empty
  uint32_t* rrec;
  uint32_t rmask;
  unsigned long long* pkeys;         // ((a+1) << 32) | (b+1), 0 = empty
  uint32_t* prec;
  uint32_t pmask;
  uint32_t* flags;                   // [0] region table full, [1] pair table full
  uint32_t debug;                    // ablation switches for profiling builds (GLIA_HMT_DEBUG), 0 in production
};

constexpr int kMaxChannels = 4;     // distinct (image volume, histogram) pairs over all feature lists
constexpr int kMaxListed = GLIA_HMT_MAX_IMAGES;   // images per feature list (region / label / boundary); the vector they give must still fit kMaxFeat columns

// compact, sorted RAG as produced by the edge-table step
struct RagArrays {
  int64_t R = 0, P = 0;       // regions, directed pairs
  uint32_t* d_rlabel = nullptr;   // [R] ascending
  uint32_t* d_rrec = nullptr;     // [R][kRegionWords]
  uint32_t* d_pa = nullptr;       // [P] label a (ascending (a,b))
  uint32_t* d_pb = nullptr;       // [P]
  uint32_t* d_prec = nullptr;     // [P][kPairWords]
  // Feature lists with several image volumes: one accumulation pass per distinct (volume, histogram) CHANNEL.  Channel
  // 0 is the boundary-probability volume (d_rrec / d_prec above = c_rrec[0] / c_prec[0]); the keys are common.
  int K = 1;
  uint32_t* c_rrec[kMaxChannels] = {nullptr, nullptr, nullptr, nullptr};
  uint32_t* c_prec[kMaxChannels] = {nullptr, nullptr, nullptr, nullptr};
  int c_bins[kMaxChannels] = {0, 0, 0, 0};
  // Optional (slab route, median linkage): the image value of every boundary voxel, grouped by directed pair -- values of pair i
  // are d_pv[d_pv_off[i] .. d_pv_off[i + 1]) in no particular order (the median loop sorts its runs itself).  The reference keeps
  // these lists per edge (util/struct_merge.hxx:97-111); a whole-volume map re-reads them from the volume instead.
  unsigned long long* d_pv_off = nullptr;   // [P + 1]
  float* d_pv = nullptr;                    // [nV]
  unsigned long long nV = 0;
};

int launch_accumulate(const AccParams& p, hipStream_t stream);
int launch_synth(int dim, const int64_t dims[3], int S, int G, uint64_t seed, int variant, uint32_t* d_labels,
                 uint32_t* d_truth_tmp, float* d_pb, hipStream_t stream);
// Masks (type/neighbor.hxx:80-86, util/struct.hxx:86-91,133-143): a masked-out NEIGHBOUR is invalid like an out-of-bounds one;
// a masked-out CENTRE voxel is skipped in point-map mode and processed in contour-only mode.  The mask is folded into a
// copy of the label volume once (sentinel label), so the streaming pass keeps its label loads.
constexpr uint32_t kMaskedLabel = 0xFFFFFFFFu;
int launch_mask_fold(const uint32_t* lab, const uint32_t* mask, uint32_t* out, int64_t n, hipStream_t stream);

// the volume a RAG was built from: median linkage re-reads the boundary voxels' values (util/struct_merge.hxx:97-103)
struct VolumeRef {
  const uint32_t* lab = nullptr;      // centre labels
  const uint32_t* lab_nb = nullptr;   // neighbour labels (the folded copy when a mask was given, else = lab)
  const float* pb = nullptr;
  int dim = 3;
  long long nx = 1, ny = 1, nz = 1;
  long long zb = 0, ze = -1;           // planes whose voxels count (a slab's owned planes; ze < 0: all)
};
// GLIA_USE_MEDIAN_AS_FEATS for a given merge order (median_feats.hip): median, mean and standard deviation of the value multiset of
// every voxel set a bc_feat row looks at.  forced = dense region pairs of the merges; list entry c of the region / boundary lists
// reads r_img[c] / b_img[c] (device).  reg[((i * 3 + k) * n_r + c) * 3 + q]: merge i, k = first region | second region | region
// created, q = median | mean | stddev; bnd[((i * 4 + k) * n_b + c) * 3 + q]: k = B(first) | B(second) | B(created) | shared boundary.
struct MedianFeatIn {
  const RagArrays* rag; VolumeRef vol;
  int n_r; const float* r_img[GLIA_HMT_MAX_IMAGES]; int n_b; const float* b_img[GLIA_HMT_MAX_IMAGES];
  const uint32_t* forced; int64_t n_merges;
};
int median_feature_stats(const MedianFeatIn& in, hipStream_t stream, std::vector<double>* reg, std::vector<double>* bnd, std::vector<unsigned long long>* area);
int greedy_mean(const RagArrays& rag, hipStream_t stream, uint32_t* h_order, double* h_sal, int64_t capacity,
                int64_t* n_merges, double* ms_table, double* ms_loop, int64_t* n_scored, int cond_n = 0,
                const long long* cond_sizes = nullptr, double cond_rpb = 0.0, const VolumeRef* median_of = nullptr,
                bool size_weight = false);
struct BcCfg;
struct DeviceClassifier;
// greedy_bc.hip is compiled five times: with the libm restatements of the feature code (glibc_math.hpp) selected at run time
// (any combination, incl. "unpinned"), and with the two combinations real hosts have -- glibc's FMA build and its non-FMA
// build -- fixed at compile time (a run-time choice between three logarithms at every call site costs the classifier loop
// 5 %).  greedy_bc() picks the instance from cfg.libm_* (api.cpp).
#define GLIA_DECLARE_GREEDY_BC(name)                                                                                          \
  int name(const RagArrays& rag, const BcCfg& cfg, const DeviceClassifier& clf, hipStream_t stream, uint32_t* h_order,       \
           double* h_sal, double* h_feats, int64_t capacity, int64_t* n_merges, double* ms_table, double* ms_init,           \
           double* ms_loop, int64_t* n_scored, bool init_only, const uint32_t* h_forced, int64_t n_forced, int shard,        \
           int n_shards, double* h_scores)
GLIA_DECLARE_GREEDY_BC(greedy_bc_generic);
GLIA_DECLARE_GREEDY_BC(greedy_bc_fma);
GLIA_DECLARE_GREEDY_BC(greedy_bc_sse2);
GLIA_DECLARE_GREEDY_BC(greedy_bc_fma_common);      // + GLIA_BC_COMMON (bc_features.hpp): one image on the region and boundary lists, no --logs / --simpf / histogram columns
GLIA_DECLARE_GREEDY_BC(greedy_bc_sse2_common);
int greedy_bc(const RagArrays& rag, const BcCfg& cfg, const DeviceClassifier& clf, hipStream_t stream, uint32_t* h_order,
              double* h_sal, double* h_feats, int64_t capacity, int64_t* n_merges, double* ms_table, double* ms_init,
              double* ms_loop, int64_t* n_scored, bool init_only, const uint32_t* h_forced = nullptr,
              int64_t n_forced = 0, int shard = 0, int n_shards = 1, double* h_scores = nullptr);
int compact_tables(const AccParams& p, uint32_t rcap, uint32_t pcap, RagArrays* out, hipStream_t stream);
int transform_keys(const uint32_t* order, int64_t n, std::vector<uint32_t>* src, std::vector<uint32_t>* dst);
int transform_image(uint32_t* d_lab, int64_t n, const uint32_t* h_src, const uint32_t* h_dst, int64_t m, const uint32_t* d_mask,
                    int fill_missing, hipStream_t stream, double* ms);
int relabel_image(uint32_t* d_lab, int64_t n, int64_t min_size, uint32_t* n_labels, hipStream_t stream);
int paint_pair_values(const VolumeRef& vol, const uint32_t* d_pa, const uint32_t* d_pb, int64_t P, const float* h_val, float* d_out,
                      hipStream_t stream);
int boundary_confidence_values(int n_trees, const int64_t* n_nodes, const uint32_t* const* node_label, const int32_t* const* parent,
                               const int32_t* const* child0, const double* const* potential, const uint32_t* pa, const uint32_t* pb,
                               int64_t P, std::vector<float>* out);
int watershed_labels(int dim, const int64_t dims[3], const float* d_img, double level, uint32_t* d_out, uint32_t* n_labels, int* sweeps,
                     hipStream_t stream);
int launch_libm_eval(int function, int variant, const double* d_in, double* d_out, int64_t n, hipStream_t stream);
int merge_rag_arrays(const RagArrays* parts, int n_parts, RagArrays* out, hipStream_t stream);
// boundary-voxel values per directed pair of a map built from `vol` (its owned planes): fills rag->d_pv_off / d_pv / nV (owned)
int collect_pair_values(RagArrays* rag, const VolumeRef& vol, hipStream_t stream);
// value runs in a new pair order: out run j = source run order[j] (counts from the source offsets); allocates *out_off [n + 1], *out_vals
int gather_value_runs(const unsigned long long* src_off, const float* src_vals, const uint32_t* order, uint32_t n, unsigned long long** out_off,
                      float** out_vals, unsigned long long* nV, hipStream_t stream);
int rag_cut_flags(const RagArrays& rag, const uint32_t* d_lab, int64_t nx, int64_t ny, int64_t nzl, int64_t zb, int64_t ze,
                  uint8_t* d_rflag, uint8_t* d_pflag, hipStream_t stream);

__host__ __device__ inline uint32_t float_ord(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__host__ __device__ inline float ord_float(uint32_t o) {
  uint32_t u = o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu);
  return __builtin_bit_cast(float, u);
}

}  // namespace glia
