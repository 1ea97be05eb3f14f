/* include/glia_hmt.h -- C ABI of the MI355X-native HMT hot path (libglia_hmt.so).
 *
 * Drop-in boundary for GLIA's hierarchical-merge-tree path (code/hmt + the L1/L2 types it
 * drives).  GLIA has no FFI/plugin registry: its operator API is header templates whose
 * behaviour is injected through lambdas, plus CLIs and file formats.  Each
 * entry point below therefore replaces one reference operator together with the fixed set
 * of lambdas the reference's own callers pass to it; the citation names file:line under
 * /root/reference/code/.
 *
 * Conventions: plain C, opaque handles, int status (0 = ok, <0 = error, message via
 * glia_hmt_last_error()); never exits the process (the reference's perr() does,
 * glia_base.hxx:66-69).  Pointers prefixed d_ are DEVICE (HBM) pointers on the context's
 * GPU, h_ are host pointers.  Volumes are dense, x fastest: index = x + nx*(y + ny*z),
 * labels uint32 (glia_base.hxx:43), images float (:44), features double (:45).
 * A handle is not thread-safe; all work of a context is ordered on its HIP stream.
 */
#ifndef GLIA_HMT_H
#define GLIA_HMT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLIA_HMT_OK 0
#define GLIA_HMT_ERR_ARG (-1)          /* invalid argument */
#define GLIA_HMT_ERR_HIP (-2)          /* HIP runtime error */
#define GLIA_HMT_ERR_UNSUPPORTED (-3)  /* valid in the reference, not implemented here yet */
#define GLIA_HMT_ERR_CAPACITY (-4)     /* caller-provided output buffer too small */
#define GLIA_HMT_ERR_SALIENCY (-5)     /* "Error: invalid boundary saliency..." (util/struct_merge.hxx:58-59) */
#define GLIA_HMT_ERR_IO (-6)           /* file could not be read / malformed */
#define GLIA_HMT_ERR_INTERNAL (-7)     /* a merge loop stopped on its own consistency check (see glia_hmt_internal_errors) */

#define GLIA_HMT_MAX_IMAGES 8
#define GLIA_HMT_MAX_BINS 16
#define GLIA_HMT_MAX_THRESH 4

typedef struct glia_hmt_ctx glia_hmt_ctx;
typedef struct glia_hmt_rag glia_hmt_rag;
typedef struct glia_hmt_forest glia_hmt_forest;

const char* glia_hmt_last_error(void);
const char* glia_hmt_version(void);

/* Context: device + stream + reusable workspaces.  hip_stream may be NULL (own stream) or a
 * hipStream_t the caller owns (e.g. torch.cuda.current_stream().cuda_stream). */
int glia_hmt_ctx_create(int device, void* hip_stream, glia_hmt_ctx** out);
void glia_hmt_ctx_destroy(glia_hmt_ctx* ctx);
/* The library parks scratch blocks of finished calls for reuse (at most 12 GiB per process; a hipMalloc / hipFree pair is a
 * device-wide synchronisation).  This returns them to the driver -- for callers that share the device with another allocator
 * (torch, a second library).  Returns the number of bytes released.  Destroying the last context of a process does the same. */
unsigned long long glia_hmt_release_cached_memory(void);
/* The invariant of genMergeOrderGreedy's output (util/struct_merge.hxx:19-31: the loop appends (r0, r1, key++) and erases the two
 * regions' items): merge k joins two regions that still exist and creates region n_regions + k.  h_order holds DENSE ids (leaf i =
 * i-th label ascending, merged region n_regions + k).  Returns GLIA_HMT_OK, or GLIA_HMT_ERR_ARG with *first_bad = the first merge
 * that breaks the rule.  The pb / pre_merge loops run this O(R) replay on every order before they return it; a violation ends the
 * call with GLIA_HMT_ERR_INTERNAL -- it is never repaired by running the loop again. */
int glia_hmt_check_merge_order(const uint32_t* h_order, int64_t n_merges, int64_t n_regions, int64_t* first_bad);
/* Calls of this process that ended with GLIA_HMT_ERR_INTERNAL.  0 is the only healthy answer: bench.py prints it, the GPU test
 * session and __graft_entry__.smoke() assert it. */
unsigned long long glia_hmt_internal_errors(void);
/* Tuning and test switches of the loops, process-wide: which queue the pb-mean loop runs on (GLIA_HMT_PB_WINDOW, GLIA_HMT_PB_BATCH,
 * GLIA_HMT_FORCE_TREE), its window size / baseline interval / horizon (GLIA_HMT_WINCAP, GLIA_HMT_REBASE, GLIA_HMT_HORIZON), the
 * classifier loop's helper workgroups and instance tier (GLIA_HMT_HELPERS, GLIA_HMT_BC_NOCOMMON, GLIA_HMT_BC_GENERIC), the libm
 * variant (GLIA_HMT_LIBM, read when a context is created), GLIA_HMT_TRACE, GLIA_HMT_DEBUG (profiling build).  No setting changes a
 * result.  value = NULL unsets.  The table starts from the environment variables of the same names, read once; nothing in the
 * library calls getenv() per call.  Unknown key: GLIA_HMT_ERR_ARG. */
int glia_hmt_set_option(const char* key, const char* value);
int glia_hmt_ctx_sync(glia_hmt_ctx* ctx);
/* Logarithms of the feature vector.  The reference computes histogram entropies with std::log2 (util/stats.hxx:145-152)
 * and the --logs features with std::log (glia_base.hxx:80-81), the compactness with std::pow (type/feat.hxx:78-79), i.e.
 * with the host's glibc, which is not correctly
 * rounded; the kernels carry restatements of glibc's algorithms (glia_amd/csrc/glibc_math.hpp) and the context selects,
 * by probing the host's libm when it is created, the one that reproduces it bit for bit: 1 = non-FMA build ("sse2"),
 * 2 = FMA build, 0 = no restatement matches this host (device libm, within 1 ulp: entropy features then UNPINNED).
 * glia_hmt_host_libm_probe runs the same probe without a GPU; glia_hmt_libm_eval evaluates function (0 = log2, 1 = log,
 * 2 = the pow(perim, 1.5) of type/feat.hxx:78-79) in the given variant over a device array (parity tests). */
int glia_hmt_ctx_libm(const glia_hmt_ctx* ctx, int* log2_variant, int* log_variant);
int glia_hmt_ctx_libm_pow(const glia_hmt_ctx* ctx, int* pow_variant);        /* same, for std::pow(perim, 1.5) (type/feat.hxx:78-79) */
/* 1 = all three host functions are reproduced bit for bit, 0 = at least one is not (glia_hmt_ctx_create then also leaves a
 * "warning: ..." text in glia_hmt_last_error(); the tools print it to stderr), < 0 = error */
int glia_hmt_ctx_libm_status(const glia_hmt_ctx* ctx);
int glia_hmt_host_libm_probe(int* log2_variant, int* log_variant);
int glia_hmt_host_libm_probe_pow(int* pow_variant);
int glia_hmt_host_libm_eval(int function, int variant, const double* h_in, double* h_out, int64_t n);  /* host code of the same restatement, no GPU */
int glia_hmt_libm_eval(glia_hmt_ctx* ctx, int function, int variant, const double* d_in, double* d_out, int64_t n);
/* Which region is "region 0" of an initial edge, everything else being equal, follows the iteration order of the reference's
 * std::unordered_map region map (type/region_map.hxx:79-95 filled from genPointMap, util/struct.hxx:77-92).  The library
 * reproduces it by replaying the insertions under libstdc++'s hashtable rules (glia_amd/csrc/rmap_order.cpp).  This host-only
 * entry evaluates the replay for n leaves (labels ascending, first_voxel = raster index of a label's first voxel): rank[i] =
 * position of leaf i in that order.  mode 0 = as the library does, 1 = through the real container, 2 = the array emulation
 * (tests compare them). */
int glia_hmt_host_rmap_ranks(const uint32_t* labels, const int64_t* first_voxel, int64_t n, int mode, uint32_t* rank);
/* Optional sizing hint for the accumulation hash tables (0 = derive from the volume size; they grow and
 * the pass is redone if they fill up). */
int glia_hmt_ctx_set_table_hint(glia_hmt_ctx* ctx, int64_t expected_regions, int64_t expected_pairs);

/* Image + histogram spec: one ImageHistPair of hmt/bc_feat.hxx:29-43. */
typedef struct {
  const float* d_image;
  int bins;            /* --rbb/--rlb/--rb/--bb, <= GLIA_HMT_MAX_BINS */
  double lo, hi;       /* histogram range (--rbl/--rbu ...) */
} glia_hmt_image;

/* Feature configuration = the image lists prepareImages builds (hmt/hmt_util.hxx:17-56: an --rbi
 * image is appended to BOTH the region and the boundary list) plus the scalar flags of
 * hmt/main_merge_order_bc.cxx:172-242.  At most 4 entries per list; every distinct (d_image, bins, lo, hi) over
 * all lists is one accumulation pass over the volume (at most 4 distinct ones). */
typedef struct {
  int n_region;  glia_hmt_image region[GLIA_HMT_MAX_IMAGES];         /* rImages  */
  int n_rlabel;  glia_hmt_image rlabel[GLIA_HMT_MAX_IMAGES];         /* rlImages */
  int n_boundary; glia_hmt_image boundary[GLIA_HMT_MAX_IMAGES];      /* bImages  */
  const float* d_pb;                 /* --pb */
  int n_thresholds;                  /* --bt, <= GLIA_HMT_MAX_THRESH */
  double thresholds[GLIA_HMT_MAX_THRESH];
  double normalizing_area;           /* 1.0 unless --ns (main_merge_order_bc.cxx:36-39) */
  double normalizing_length;
  int use_log_shape;                 /* --logs */
  int use_simple_features;           /* --simpf */
  /* Build options of the reference that change the vector layout (CMakeLists.txt:54-64, type/feat.hxx:608-621, 677-722):
   * GLIA_HMT_HIST_FEAT -> GLIA_USE_HISTOGRAM_AS_FEATS: every image-feature block carries its normalised histogram ahead of
   * the entropy (bins more columns per block).  GLIA_HMT_MEDIAN_FEAT -> GLIA_USE_MEDIAN_AS_FEATS: every real-feature block carries
   * the MEDIAN of its voxel set ahead of the mean, mean / standard deviation come from the value vector (stats::mean, stats::var),
   * the diff blocks gain |median0 - median1|, --simpf carries the shared boundary's median beside its mean (hmt/bc_feat.hxx:252-268).
   * Implemented for a GIVEN merge order (glia_hmt_bc_feat / _saliency: glia_amd/csrc/median_feats.hip; the value multisets of one
   * order are processed in batches of 2^27 values, one set may hold at most 2^30); the greedy loop (glia_hmt_merge_order_bc,
   * glia_hmt_score_initial_edges) returns GLIA_HMT_ERR_UNSUPPORTED with this layout.  Median columns are bit-exact, the mean / stddev
   * columns comparable to 1e-12 (the reference sums in an order that depends on rand(), util/stats.hxx:87). */
  int use_histogram_features;
  int use_median_features;
} glia_hmt_feat_config;

/* ---- region adjacency structure -------------------------------------------------------
 * Replaces TRegionMap(image, mask, onlyContour) (type/region_map.hxx:38-40,52-65), i.e.
 * genPointMap/genContourMap (util/struct.hxx:77-144) with getContourTraits
 * (type/neighbor.hxx:109-126).  Instead of voxel lists it accumulates, in ONE pass over the
 * label volume and `cfg->d_pb`/image volumes, the sufficient statistics every downstream
 * operator needs: per label (count, border count, bbox, image moments, histogram) and per
 * DIRECTED label pair (a->b) (boundary voxel count, image moments, histogram, threshold counts).
 * cfg may be NULL when only merge_order_pb will be called (then d_pb must be given).
 * d_mask (optional, u32 volume, MASK_OUT_VAL = 0 masks a voxel out): masked-out neighbours are invalid like
 * out-of-bounds ones (type/neighbor.hxx:80-86); a masked-out centre voxel belongs to no region when only_contour = 0
 * (util/struct.hxx:86-91) and is still processed when only_contour = 1 (:133-143 has no mask test).
 * The label / mask / image volumes must stay alive while the handle lives: median linkage re-reads them. */
int glia_hmt_rag_build(glia_hmt_ctx* ctx, int dim, const int64_t dims[3], const uint32_t* d_labels,
                       const uint32_t* d_mask, int only_contour, const float* d_pb,
                       const glia_hmt_feat_config* cfg, glia_hmt_rag** out);
/* ---- z-slab partition for volumes split across GPUs (no reference counterpart: GLIA is single-node) ----------
 * A rank owns global planes [z_global_of_plane0 + z_begin, z_global_of_plane0 + z_end) and hands in its planes plus
 * one halo plane on each side that is not a face of the volume (dims_local[2] = planes handed in).  The result is a
 * PARTIAL region map: every statistic is a commutative monoid, so partial maps of all slabs are combined with
 * glia_hmt_rag_merge into exactly the map glia_hmt_rag_build gives for the whole volume.  Between ranks the partial
 * records travel as the plain device arrays glia_hmt_rag_device_arrays exposes -- only the records glia_hmt_rag_cut_flags marks
 * take the keyed owner exchange (unpadded point-to-point messages over RCCL), everything goes once to the rank that runs the
 * merge loop (glia_amd/slab.py) -- and are wrapped again with glia_hmt_rag_from_arrays (`like` supplies configuration and
 * dimensions). */
int glia_hmt_rag_build_slab(glia_hmt_ctx* ctx, const int64_t dims_local[3], int64_t z_global_of_plane0,
                            int64_t nz_global, int64_t z_begin, int64_t z_end, const uint32_t* d_labels,
                            int only_contour, const float* d_pb, const glia_hmt_feat_config* cfg, glia_hmt_rag** out);
int glia_hmt_rag_merge(glia_hmt_ctx* ctx, glia_hmt_rag* const* parts, int n_parts, glia_hmt_rag** out);
/* ---- the slab route end to end (glia_amd/csrc/slab_dist.cpp): build, keyed owner exchange of the cut records, hand-over ----
 * A communicator holds the ranks of ONE process.  glia_hmt_comm_create_rccl: one rank per process and GPU, transfers between
 * processes are ncclSend / ncclRecv over RCCL (xGMI inside a node); the 128-byte id comes from glia_hmt_comm_unique_id on one rank
 * and reaches the others through the launcher (a file, an environment variable).  glia_hmt_comm_create_local: all `world` ranks
 * in this process on this context's GPU, transfers are device copies -- the same code path otherwise (single-GPU runs of volumes
 * processed slab by slab, tests).  librccl.so is loaded when the first RCCL communicator is made, not linked. */
typedef struct glia_hmt_comm glia_hmt_comm;
int glia_hmt_comm_unique_id(void* id128);
int glia_hmt_comm_create_rccl(glia_hmt_ctx* ctx, int world, int rank, const void* id128, glia_hmt_comm** out);
int glia_hmt_comm_create_local(glia_hmt_ctx* ctx, int world, glia_hmt_comm** out);
void glia_hmt_comm_destroy(glia_hmt_comm* comm);
int glia_hmt_comm_world(const glia_hmt_comm* comm);
int glia_hmt_comm_local_ranks(const glia_hmt_comm* comm, int* ranks, int capacity);      /* returns their number */
/* the z range of rank `rank` of `world`: planes [first_plane, first_plane + n_planes) are handed in (owned planes + one halo
 * plane per cut), of which the local planes [z_begin, z_end) are owned (the arguments of glia_hmt_rag_build_slab) */
int glia_hmt_slab_range(int64_t nz, int world, int rank, int64_t* first_plane, int64_t* n_planes, int64_t* z_begin, int64_t* z_end);
typedef struct {
  int64_t dims_local[3];             /* x, y, planes handed in */
  int64_t z_global_of_plane0;        /* global z of local plane 0 */
  int64_t z_begin, z_end;            /* owned local planes */
  const uint32_t* d_labels;          /* device, dims_local */
  const float* d_pb;                 /* device, dims_local (or NULL when cfg lists the images) */
  const glia_hmt_feat_config* cfg;   /* image lists of THIS slab's sub-volumes, or NULL */
} glia_hmt_slab;
typedef struct {
  uint64_t records, cut_records;                     /* of the local ranks: all records / records that took the owner exchange */
  uint64_t bytes_cut_exchange, bytes_to_loop_owner;  /* bytes the local ranks sent to OTHER ranks in step 2 / step 3 */
} glia_hmt_dist_stats;
/* slabs[i] belongs to the i-th local rank of the communicator (rank order).  *out = the whole volume's map when loop_owner is a
 * local rank, NULL otherwise.  Collective: every process of the communicator calls it.
 * with_values != 0: the image value (d_pb) of every boundary voxel travels with its directed pair, so that the merged map can run
 * the MEDIAN linkage (glia_hmt_merge_order_pb type 1, the reference tool's default, hmt/main_merge_order_pb.cxx:10; the reference
 * keeps these value lists per edge, util/struct_merge.hxx:97-111) -- 4 bytes per boundary voxel on top of the records. */
int glia_hmt_rag_build_distributed(glia_hmt_ctx* ctx, glia_hmt_comm* comm, const glia_hmt_slab* slabs, int64_t nz_global, int only_contour,
                                   int with_values, int loop_owner, glia_hmt_rag** out, glia_hmt_dist_stats* stats);
/* Which records of a slab's partial map may have a counterpart in another slab ("exchanging only the cross-slab boundary
 * regions"): d_region_cut[i] / d_pair_cut[i] (device, one byte per record, in the order of glia_hmt_rag_device_arrays) = 1 iff
 * the region's label / one of the pair's labels occurs on a plane next to a cut (first / last owned plane, halo plane).  Only
 * flagged records take the keyed owner exchange (glia_amd/slab.py); the others go once to the rank that runs the merge loop.
 * dims_local / z_begin / z_end / d_labels: as handed to glia_hmt_rag_build_slab. */
int glia_hmt_rag_cut_flags(glia_hmt_ctx* ctx, const glia_hmt_rag* rag, const int64_t dims_local[3], int64_t z_begin, int64_t z_end,
                           const uint32_t* d_labels, uint8_t* d_region_cut, uint8_t* d_pair_cut);
int glia_hmt_rag_device_arrays(const glia_hmt_rag* rag, const uint32_t** d_region_label, const uint32_t** d_region_rec,
                               const uint32_t** d_pair_a, const uint32_t** d_pair_b, const uint32_t** d_pair_rec,
                               int* region_words, int* pair_words);
/* copies the compact arrays into caller-owned device buffers ([R], [R][region_words], [P], [P], [P][pair_words]) */
int glia_hmt_rag_copy_arrays(const glia_hmt_rag* rag, uint32_t* d_region_label, uint32_t* d_region_rec,
                             uint32_t* d_pair_a, uint32_t* d_pair_b, uint32_t* d_pair_rec);
int glia_hmt_rag_from_arrays(glia_hmt_ctx* ctx, const glia_hmt_rag* like, int64_t n_regions,
                             const uint32_t* d_region_label, const uint32_t* d_region_rec, int64_t n_pairs,
                             const uint32_t* d_pair_a, const uint32_t* d_pair_b, const uint32_t* d_pair_rec,
                             glia_hmt_rag** out);
/* Feature lists with several image volumes give a map one record set per CHANNEL (distinct volume + histogram); the keys are
 * common.  glia_hmt_rag_device_arrays / _copy_arrays / _from_arrays handle channel 0; the further channels of a partial map
 * are copied out with glia_hmt_rag_copy_channel and attached, in order, to a map made by glia_hmt_rag_from_arrays with
 * glia_hmt_rag_add_channel (`like` supplied the histogram layout of every channel). */
int glia_hmt_rag_num_channels(const glia_hmt_rag* rag);
int glia_hmt_rag_copy_channel(const glia_hmt_rag* rag, int channel, uint32_t* d_region_rec, uint32_t* d_pair_rec);
int glia_hmt_rag_add_channel(glia_hmt_ctx* ctx, glia_hmt_rag* rag, const uint32_t* d_region_rec, const uint32_t* d_pair_rec);
void glia_hmt_rag_free(glia_hmt_rag* rag);
int64_t glia_hmt_rag_num_regions(const glia_hmt_rag* rag);
int64_t glia_hmt_rag_num_pairs(const glia_hmt_rag* rag);     /* directed label pairs */
/* Export for inspection / parity tests (host arrays sized by the counts above; any may be NULL).
 * Regions ascend by label; pairs ascend by (a,b). */
int glia_hmt_rag_export_regions(const glia_hmt_rag* rag, uint32_t* h_label, int64_t* h_count,
                                int64_t* h_border, int64_t* h_bbox_lo /*[n][3]*/, int64_t* h_bbox_hi,
                                double* h_sum, double* h_sumsq, double* h_min, double* h_max,
                                int64_t* h_hist /*[n][bins]*/, int64_t* h_first_voxel);
int glia_hmt_rag_export_pairs(const glia_hmt_rag* rag, uint32_t* h_a, uint32_t* h_b, int64_t* h_count,
                              double* h_sum, double* h_sumsq, double* h_min, double* h_max,
                              int64_t* h_hist /*[n][bins]*/, int64_t* h_thr /*[n][n_thresholds]*/);
/* Kernel time of the accumulation pass of the last build on this handle, measured with HIP events
 * on the context stream (milliseconds), and the algorithmic bytes it covers. */
int glia_hmt_rag_last_pass(const glia_hmt_rag* rag, double* ms, double* algorithmic_bytes);

/* ---- greedy merge orders -----------------------------------------------------------------
 * glia_hmt_merge_order_pb replaces genMergeOrderGreedyUsingPbApproxMedian (type 1,
 * util/struct_merge.hxx:90-136) and genMergeOrderGreedyUsingPbMean (type 2, :38-85) as called by
 * merge_order_pb (hmt/main_merge_order_pb.cxx:27-36) with fcond = f_true.
 * Output: h_order[3*i..] = (x0, x1, x2) of merge i (TTriple, type/tuple.hxx:8-29),
 * h_saliency[i] = popped queue key.  *n_merges <= capacity.  Type 1 (the tool's default) keeps, per table edge, the
 * SORTED run of its boundary voxels' values; a contraction merges runs (merge path) and reads the order statistic
 * n/2 (stats::amedian, util/stats.hxx:83-91).  It needs a handle built by glia_hmt_rag_build (whole volume).
 * Type 3 (no tool of the reference calls it) = genMergeOrderGreedyUsingPbApproxMedianAndMinSize (:141-185): saliency
 * -median * min(size of the two regions), regions merged as the loop goes; needs only_contour = 0. */
int glia_hmt_merge_order_pb(glia_hmt_ctx* ctx, glia_hmt_rag* rag, int type, uint32_t* h_order,
                            double* h_saliency, int64_t capacity, int64_t* n_merges);

/* ---- boundary classifier -------------------------------------------------------------------------
 * glia_hmt_forest_load replaces alg::RandomForest(predictLabel, modelFile) (alg/rf.hxx:22-33) for n_models == 1
 * and alg::EnsembleRandomForest + opt::ThresholdModelDistributor(dim0, dim1, threshold) (alg/rf.hxx:63-98,
 * type/function.hxx:71-85; --bcmd) for n_models == 3.  Files are GLIA's binary RF models
 * (ml/rf/ml_rf_model.cxx:378-455).  predict_label is BC_LABEL_MERGE = -1 in every reference caller.
 * glia_hmt_forest_stub builds the diagnostic scorer P = 1 - x[feature_index] (tests only). */
int glia_hmt_forest_load(glia_hmt_ctx* ctx, int n_models, const char* const* paths, int predict_label,
                         const double* distributor_args /*[3] or NULL*/, glia_hmt_forest** out);
/* Host-only: parse a GLIA model file (rf_old::readModelFromBinaryFile, ml/rf/ml_rf_model.cxx:459-563) into the node
 * arrays the device walks: h_split[tree][node], h_meta[tree][node][4] = {variable (0-based), left, right (0-based node),
 * vote (-1 = internal node, else 1 iff the leaf's class is predict_label)}.  Needs no GPU. */
int glia_hmt_forest_file_parse(const char* path, int predict_label, int* ntree, int* nrnodes, int* nclass,
                               double* h_split, int* h_meta, int64_t capacity_nodes);
int glia_hmt_forest_stub(glia_hmt_ctx* ctx, int feature_index, glia_hmt_forest** out);
void glia_hmt_forest_free(glia_hmt_forest* forest);

/* Length of one feature vector for the configuration the rag was built with (BoundaryClassificationFeats::dim,
 * hmt/bc_feat.hxx:225-230; or selectFeatures' length with --simpf). */
int glia_hmt_feat_dim(const glia_hmt_rag* rag);

/* Replaces genMergeOrderGreedyUsingBoundaryClassifier (util/struct_merge_bc.hxx:45-58) with the fBcFeat / fBcPred
 * pair of hmt/main_merge_order_bc.cxx:54-137 (bcType 1).  The rag must have been built with a feature
 * configuration and only_contour = 0.  h_feats (optional, [capacity][feat_dim]) receives the feature vector each
 * merged edge was scored with (the -b output, :148-157). */
int glia_hmt_merge_order_bc(glia_hmt_ctx* ctx, glia_hmt_rag* rag, const glia_hmt_forest* forest, uint32_t* h_order,
                            double* h_saliency, double* h_feats, int64_t capacity, int64_t* n_merges);

/* Replaces the merge engine of pre_merge (gadget/main_pre_merge.cxx:20-76): pb-mean linkage with updateRegion = true
 * and the size condition -- an edge may merge only if its smaller region has fewer than size_thresholds[0] voxels, or
 * (n_thresholds == 2) a region smaller than size_thresholds[1] has mean pb above rpb_threshold.  The region map must
 * have been built with only_contour = 0 and the pb image.  Relabelling the volume (transformKeys/transformImage,
 * :77-79) stays with the caller. */
int glia_hmt_pre_merge(glia_hmt_ctx* ctx, glia_hmt_rag* rag, const int* size_thresholds, int n_thresholds,
                       double rpb_threshold, uint32_t* h_order, double* h_saliency, int64_t capacity, int64_t* n_merges);

/* Replaces the bc_feat pipeline (hmt/main_bc_feat.cxx:27-112) for a GIVEN merge order: RegionMap(seg, mask, order,
 * false) + RegionFeats of every tree node + BoundaryFeats of every merge (the OpenMP parfor loops of :59-101),
 * without the optional saliency features (see glia_hmt_bc_feat_saliency).  h_order: n_merges triples (x0, x1, x2); h_feats: [n_merges][feat_dim],
 * row i = features of merge i with regions in the file's orientation and the area-ordered swap of :88-91. */
int glia_hmt_bc_feat(glia_hmt_ctx* ctx, glia_hmt_rag* rag, const uint32_t* h_order, int64_t n_merges, double* h_feats);
/* The same with the saliency features of bc_feat -y/--s0/--sb (hmt/main_bc_feat.cxx:50-55, genSaliencyMap
 * hmt/bc_feat.hxx:12-26): h_saliencies[i] belongs to merge i; rows then have glia_hmt_bc_feat_dim(rag, 1) columns
 * (5 more: two in the boundary block, one per region block; none for the simple selection). */
int glia_hmt_bc_feat_saliency(glia_hmt_ctx* ctx, glia_hmt_rag* rag, const uint32_t* h_order, int64_t n_merges,
                              const double* h_saliencies, double init_saliency, double saliency_bias, double* h_feats);
int glia_hmt_bc_feat_dim(const glia_hmt_rag* rag, int with_saliency);

/* ---- the step after the merge path: tree resolution (hmt/main_segment_greedy.cxx:33-86), host-only ----
 * glia_hmt_tree_potentials = genTree / genTreeWithNodePotentials (hmt/tree_build.hxx:12-63): array tree of the merge
 * order plus a potential per node -- merge probability of the node, times (1 - p) of its parent; leaves: (1 - p_parent)^2;
 * the root (= last node) squared; optionally times max(region_prob[node], FEPS).  h_merge_probs == NULL: all 1.0.
 * Returns the number of nodes. */
int64_t glia_hmt_tree_potentials(const uint32_t* h_order, int64_t n_merges, const double* h_merge_probs,
                                 const double* h_region_probs, uint32_t* node_label, int32_t* parent, int32_t* child0,
                                 int32_t* child1, double* potential, int64_t capacity);
/* resolveTreeGreedy (hmt/tree_greedy.hxx:36-70,104-152 for one tree, comp = potential <): repeatedly picks the valid
 * node of highest potential (first in node order among equals) and invalidates its ancestors and descendants.
 * h_picks receives node indices in pick order; returns their number. */
int64_t glia_hmt_resolve_tree_greedy(const int32_t* parent, const int32_t* child0, const int32_t* child1,
                                     const double* potential, int64_t n_nodes, int32_t* h_picks, int64_t capacity);
/* The same over several trees (alternative merge orders of overlapping supervoxel sets; hmt/tree_greedy.hxx:104-152):
 * per-tree arrays are passed as arrays of pointers; picks come out as (tree, node) pairs in pick order. */
int64_t glia_hmt_resolve_trees_greedy(int n_trees, const int64_t* n_nodes, const uint32_t* const* node_label,
                                      const int32_t* const* parent, const int32_t* const* child0, const int32_t* const* child1,
                                      const double* const* potential, int32_t* h_pick_tree, int32_t* h_pick_node, int64_t capacity);
/* genLabelTransform (hmt/tree_segment.hxx:10-21): every leaf label under pick k maps to key_to_assign + k.  The pairs
 * feed glia_hmt_transform_image (fill_missing = 1 reproduces segment_greedy's default --ignore true). */
int64_t glia_hmt_label_transform(const uint32_t* node_label, const int32_t* child0, const int32_t* child1, int64_t n_nodes,
                                 const int32_t* h_picks, int64_t n_picks, uint32_t key_to_assign, uint32_t* h_src,
                                 uint32_t* h_dst, int64_t capacity);

/* genBoundaryConfidenceImage with all tree nodes (hmt/tree_segment.hxx:66-203; segment_greedy -b): every voxel on a
 * directed boundary of two supervoxels receives the largest (float) node potential among the tree nodes whose region
 * still owns an entry of that supervoxel pair; every other voxel 0.  rag: the region map of the segmentation image
 * (glia_hmt_rag_build, whole volume); trees as produced by glia_hmt_tree_potentials; d_out: float volume. */
int glia_hmt_boundary_confidence(glia_hmt_ctx* ctx, glia_hmt_rag* rag, int n_trees, const int64_t* n_nodes,
                                 const uint32_t* const* node_label, const int32_t* const* parent, const int32_t* const* child0,
                                 const double* const* potential, float* d_out);

/* ---- label-volume rewrites either side of the path (gadget/main_pre_merge.cxx:77-79, gadget/main_apply_merges.cxx:28-34) ----
 * transformKeys (util/struct_merge.hxx:188-210): every key that is merged and is not itself created by a merge maps
 * to the key it finally ends up in.  Host-only; pairs come out sorted by source key.  Returns the number of pairs
 * (capacity: 2 * n_merges is always enough) or a negative status. */
int64_t glia_hmt_transform_keys(const uint32_t* h_order, int64_t n_merges, uint32_t* h_src, uint32_t* h_dst, int64_t capacity);

/* transformImage (util/image.hxx:227-242 and :246-257): rewrites d_labels IN PLACE -- every voxel whose mask value is
 * not MASK_OUT_VAL (0; d_mask may be NULL) and whose label has an entry in (h_src -> h_dst) receives the mapped label;
 * labels without an entry are kept, or set to BG_VAL (0) when fill_missing != 0.  Volumes must be 16-byte aligned. */
int glia_hmt_transform_image(glia_hmt_ctx* ctx, uint32_t* d_labels, int64_t n_voxels, const uint32_t* h_src,
                             const uint32_t* h_dst, int64_t n_map, const uint32_t* d_mask, int fill_missing);

/* relabelImage (util/image.hxx:992-1001) = itk::RelabelComponentImageFilter, in place: objects (labels != 0) get the
 * consecutive labels 1..n by decreasing voxel count (ties: smaller original label first), objects smaller than
 * min_size (> 0) become 0.  ITK is not available to pin this against. */
int glia_hmt_relabel_image(glia_hmt_ctx* ctx, uint32_t* d_labels, int64_t n_voxels, int64_t min_size, uint32_t* n_labels);

/* milliseconds of the last glia_hmt_transform_image kernel on this context (HIP events) */
double glia_hmt_last_transform_ms(const glia_hmt_ctx* ctx);

/* Replaces hmt::genTree (hmt/tree_build.hxx:12-38): merge order -> array tree (children before parents, root last).
 * Host-only.  Returns the number of nodes (2 * n_merges + 1 for one connected tree) or a negative status. */
int64_t glia_hmt_gen_tree(const uint32_t* h_order, int64_t n_merges, uint32_t* node_label, int32_t* parent,
                          int32_t* child0, int32_t* child1, int64_t capacity);

/* TBoundaryTable::init with the classifier linkage only (type/boundary_table.hxx:91-114 driven by
 * util/struct_merge_bc.hxx:18-27): feature vector + score of every initial table edge, no merging.
 * *ms = device time of the feature + forest kernel. */
int glia_hmt_score_initial_edges(glia_hmt_ctx* ctx, glia_hmt_rag* rag, const glia_hmt_forest* forest,
                                 int64_t* n_edges, double* ms);
/* The same, sharded for several GPUs (scoring of the initial edges is independent per edge): this call
 * scores the records e with e % n_shards == shard.  h_scores[e] (e < *n_records, one record per unordered adjacent
 * leaf pair in lexicographic order) = P(merge) for the table edges of the shard, -inf for every other record; the
 * element-wise maximum over the shards is the full result.  h_scores may be NULL to query *n_records (<= directed pairs). */
int glia_hmt_score_initial_edges_shard(glia_hmt_ctx* ctx, glia_hmt_rag* rag, const glia_hmt_forest* forest, int shard,
                                       int n_shards, double* h_scores, int64_t capacity, int64_t* n_records);

/* Phase timings of the last merge_order_* call on this rag (ms): edge-table build, init, greedy loop. */
int glia_hmt_last_merge_timing(const glia_hmt_rag* rag, double* ms_table, double* ms_init, double* ms_loop,
                               int64_t* n_edges_scored);

/* ---- the step before the path: watershed over-segmentation (gadget/main_watershed.cxx, util/image_alg.hxx:9-21) ----------
 * glia::watershed = itk::MorphologicalWatershedImageFilter(level, MarkWatershedLineOff, face connectivity): h-minima transform
 * of the image (minima shallower than `level` vanish), regional minima as markers numbered in raster order, flooding without
 * watershed lines.  ITK decides flooding ties by the arrival order of a sequential hierarchical queue; here every tie has an
 * order-free rule (lowest flood level, then shortest way on the plateau, then smaller label), so labels are reproducible but
 * NOT pinned against ITK (absent from this image).  d_labels: uint32 volume, labels 1..*n_labels; *sweeps (optional): whole-volume
 * passes it took. */
int glia_hmt_watershed(glia_hmt_ctx* ctx, int dim, const int64_t dims[3], const float* d_image, double level, uint32_t* d_labels,
                       uint32_t* n_labels, int* sweeps);

/* ---- synthetic inputs for tests / bench, generated on the device -------------- */
int glia_hmt_synth(glia_hmt_ctx* ctx, int dim, const int64_t dims[3], int S, int G, uint64_t seed,
                   int variant, uint32_t* d_labels, float* d_pb);

#ifdef __cplusplus
}
#endif
#endif
