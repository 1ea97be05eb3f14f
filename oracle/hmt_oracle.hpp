// oracle/hmt_oracle.hpp
//
// TEST INFRASTRUCTURE ONLY -- never linked, imported or executed by the
// product path (glia_amd/, include/).  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may use it, and only as the checker.
//
// CPU restatement of GLIA's hierarchical-merge-tree hot path
//   label image -> region adjacency structure -> per-edge features ->
//   greedy merge loop -> merge order
// following the reference's data structures one for one (voxel lists,
// std::unordered_map / std::map / std::multimap with the reference's hash), so
// that every order-dependent behaviour of the reference (multimap tie rule,
// map-scan visit order in TBoundaryTable::update, unordered_map iteration order
// deciding the orientation of an initial edge) is reproduced by the same
// libstdc++ containers fed the same insertion sequences.
//
// PARITY PIN STATUS: the reference ships no tests, fixtures or golden vectors
//  and its hot-path headers need ITK, which this image
// lacks, so the RAG/feature code cannot be built from the reference here.
//  * The ITK-free engine headers (type/boundary_table.hxx, type/region*.hxx,
//    util/struct_merge.hxx:13-33) ARE compiled in place from /root/reference
//    by oracle/Makefile (oracle/_ref/ref_engine) and this restatement is checked
//    against them (tests/test_oracle_vs_ref.py).
//  * RAG construction + linkages + features are pinned against the known
//    answers recorded from the reference's own headers in SURVEY.md Appendix D
//    (tests/test_oracle_known_answers.py).
//  * The random-forest tree walk lives in an un-vendored third party
//    (randomforest-matlab, no version pinned): PARITY UNPINNED for that part.
#ifndef HMT_ORACLE_HPP
#define HMT_ORACLE_HPP

#include <cstdint>
#include <cstddef>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint32_t orc_label;

#define ORC_MAX_IMAGES 8
#define ORC_MAX_THRESH 8

// Feature configuration, mirrors the flags of hmt/main_merge_order_bc.cxx:172-242
// after prepareImages (hmt/hmt_util.hxx:17-56) has expanded --rbi into both lists.
typedef struct {
  int n_rimg;                          // region images  (rb images first, then r images)
  const float* rimg[ORC_MAX_IMAGES];
  int rbins[ORC_MAX_IMAGES];
  double rlo[ORC_MAX_IMAGES], rhi[ORC_MAX_IMAGES];
  int n_rlimg;                         // region "label" images (histogram + entropy only)
  const float* rlimg[ORC_MAX_IMAGES];
  int rlbins[ORC_MAX_IMAGES];
  double rllo[ORC_MAX_IMAGES], rlhi[ORC_MAX_IMAGES];
  int n_bimg;                          // boundary images (rb images first, then b images)
  const float* bimg[ORC_MAX_IMAGES];
  int bbins[ORC_MAX_IMAGES];
  double blo[ORC_MAX_IMAGES], bhi[ORC_MAX_IMAGES];
  const float* pb;                     // --pb image used for thresholded shape features
  int n_thr;
  double thr[ORC_MAX_THRESH];
  double norm_area, norm_len;          // 1.0 unless --ns
  int use_log;                         // --logs
  int use_simple;                      // --simpf
  int hist_as_feats;                   // build option GLIA_HMT_HIST_FEAT -> GLIA_USE_HISTOGRAM_AS_FEATS (CMakeLists.txt:54-58, feat.hxx:608-621)
  int median_as_feats;                 // build option GLIA_HMT_MEDIAN_FEAT -> GLIA_USE_MEDIAN_AS_FEATS (CMakeLists.txt:59-63, feat.hxx:677-722, 772-808; bc_feat.hxx:252-268)
} orc_feat_cfg;

// Random forest in the layout produced by rf_old::readModelFromBinaryFile after
// its transposes (ml/rf/ml_rf_model.cxx:542-557): node index fastest within a tree.
typedef struct {
  int nrnodes, ntree, nclass;
  const double* xbestsplit;   // [ntree][nrnodes]
  const int* treemap;         // [ntree][nrnodes][2]  (left, right daughters, 1-based)
  const int* nodestatus;      // [ntree][nrnodes]
  const int* nodeclass;       // [ntree][nrnodes]
  const int* bestvar;         // [ntree][nrnodes]
  const int* orig_labels;     // [nclass]
  int predict_label;          // BC_LABEL_MERGE = -1 (hmt/bc_label.hxx:11)
} orc_forest;

// ---- synthetic inputs -------------------------------------------
// variant 0: Q8 (pb = q/256 exactly), 1: F32 (adds sub-quantum noise)
int orc_synth(int dim, const int64_t* dims, int S, int G, uint64_t seed, int variant,
              orc_label* labels, float* pb);

// ---- RAG ---------------------------------------------------------------------------
typedef struct orc_rag orc_rag;
orc_rag* orc_rag_build(int dim, const int64_t* dims, const orc_label* labels,
                       const orc_label* mask, int only_contour);
void orc_rag_free(orc_rag*);
int64_t orc_rag_num_regions(const orc_rag*);
int64_t orc_rag_num_pairs(const orc_rag*);
// regions ascending by label: label, #voxels (0 in contour-only mode), #border voxels
void orc_rag_regions(const orc_rag*, orc_label* label, int64_t* npoints, int64_t* nborder);
// directed pairs ascending by (a,b): a, b, #boundary voxels
void orc_rag_pairs(const orc_rag*, orc_label* a, orc_label* b, int64_t* n);
// iteration order of the region map (the order TBoundaryTable::init walks it in)
void orc_rag_region_iter_order(const orc_rag*, orc_label* label);
// per directed pair (same order as orc_rag_pairs): sum, sumsq, min, max of img over its voxels
void orc_rag_pair_stats(const orc_rag*, const float* img, double* sum, double* sumsq,
                        double* vmin, double* vmax);
void orc_rag_region_stats(const orc_rag*, const float* img, double* sum, double* sumsq,
                          double* vmin, double* vmax, int64_t* bbox_lo, int64_t* bbox_hi);

// text dump consumed by oracle/_ref/ref_engine (the reference's own engine headers built in place)
int orc_rag_dump(const orc_rag*, const float* pb, int type, int update_region, const char* path);

// ---- greedy merge orders ------------------------------------------------------------
// type 1 = median, 2 = mean (hmt/main_merge_order_pb.cxx:10).  order: 3 labels per merge.
// Returns number of merges (<= R-1), or -1 on error.  update_region mirrors the
// updateRegion argument of util/struct_merge.hxx:13-16.
int64_t orc_merge_order_pb(orc_rag*, const float* pb, int type, int update_region,
                           orc_label* order, double* sal, int64_t cap);

// Classifier linkage (util/struct_merge_bc.hxx:10-43 + hmt/main_merge_order_bc.cxx:54-95).
// forest == NULL selects the stub scorer P(merge) = 1 - x[stub_index]
// (SURVEY.md Appendix D recipe P4 uses stub_index = 31).
// feats_out (optional): feat_dim doubles per merge, the vector cached for the popped edge.
int64_t orc_merge_order_bc(orc_rag*, const orc_feat_cfg*, const orc_forest* forest,
                           int stub_index, orc_label* order, double* sal,
                           double* feats_out, int64_t cap, int64_t* n_feat_evals);
// alg::EnsembleRandomForest with opt::ThresholdModelDistributor(dim0, dim1, threshold) (alg/rf.hxx:63-98,
// type/function.hxx:71-85; hmt/main_merge_order_bc.cxx:103-109): three models.
int64_t orc_merge_order_bc_ensemble(orc_rag*, const orc_feat_cfg*, const orc_forest* const* models, int dim0, int dim1,
                                    double threshold, orc_label* order, double* sal, double* feats_out, int64_t cap);
int orc_feat_dim(int dim, const orc_feat_cfg*);

// bc_feat pipeline (hmt/main_bc_feat.cxx:27-112): features for a given order.
int64_t orc_bc_feat(orc_rag*, const orc_feat_cfg*, const orc_label* order, int64_t n_merges,
                    double* feats_out);

int64_t orc_bc_feat_sal(orc_rag*, const orc_feat_cfg*, const orc_label* order, int64_t n_merges, const double* saliencies,
                        double init_sal, double sal_bias, double* feats_out);

// pre_merge condition engine (gadget/main_pre_merge.cxx:27-76)
int64_t orc_pre_merge(orc_rag*, const float* pb, const int* size_thresholds, int n_thresholds,
                      double rpb_threshold, orc_label* order, double* sal, int64_t cap);

// Single-sample forest score, votes[label]/ntree (ml/rf/rf.hxx:362-372)
double orc_forest_predict(const orc_forest*, const double* x, int d);

// merge tree (hmt/tree_build.hxx:12-38): node labels, parents, children (-1 for leaves)
int64_t orc_gen_tree(const orc_label* order, int64_t n_merges, orc_label* node_label,
                     int32_t* parent, int32_t* child0, int32_t* child1, int64_t cap);

// tree resolution (hmt/tree_build.hxx:41-63, hmt/tree_greedy.hxx:76-152, hmt/tree_segment.hxx:10-21)
int64_t orc_tree_potentials(const orc_label* order, int64_t n_merges, const double* merge_probs, const double* region_probs,
                            orc_label* node_label, int32_t* parent, int32_t* child0, int32_t* child1, double* potential,
                            int64_t cap);
int64_t orc_resolve_tree_greedy(const int32_t* parent, const int32_t* child0, const int32_t* child1, const double* potential,
                                int64_t n, int32_t* picks, int64_t cap);
int64_t orc_resolve_trees_greedy(int n_trees, const int64_t* n_nodes, const orc_label* const* node_label, const int32_t* const* parent,
                                 const int32_t* const* child0, const int32_t* const* child1, const double* const* potential,
                                 int32_t* pick_tree, int32_t* pick_node, int64_t cap);
int orc_boundary_confidence(orc_rag* h, int n_trees, const orc_label* const* orders, const int64_t* n_merges,
                            const orc_label* const* node_label, const int64_t* n_nodes, const double* const* potential, float* out);
int64_t orc_label_transform(const orc_label* node_label, const int32_t* child0, const int32_t* child1, int64_t n,
                            const int32_t* picks, int64_t n_picks, orc_label key, orc_label* src, orc_label* dst, int64_t cap);

// opt::ThresholdModelDistributor (type/function.hxx:71-85): the ensemble member that scores vector x
int orc_pick_model(int dim0, int dim1, double threshold, const double* x);
// label-volume rewrites (util/struct_merge.hxx:188-210, util/image.hxx:227-242, :992-1001)
int64_t orc_transform_keys(const orc_label* order, int64_t n_merges, orc_label* src, orc_label* dst, int64_t cap);
void orc_transform_image(orc_label* lab, int64_t n, const orc_label* src, const orc_label* dst, int64_t m,
                         const orc_label* mask, int fill_missing);
int64_t orc_relabel_image(orc_label* lab, int64_t n, int64_t min_size);
// host libm as the reference calls it: function 0 = std::log2, 1 = std::log, 2 = std::pow(x, 1.5)
// morphological watershed (util/image_alg.hxx:9-21; ITK absent: parity unpinned, tie rules as in glia_amd/csrc/watershed.hip)
int64_t orc_watershed(int dim, const int64_t* dims, const float* img, double level, orc_label* out);
// util/stats.hxx restatements (entropy :145-152, distL1 :155-163, distX2 :177-185, amedian :83-91, rescale :264-277)
void orc_stats_case(int n, const double* a, const double* b, double* out6);
void orc_rescale(int n, double* feat, const double* mn, const double* mx, double out_min, double out_max);
void orc_libm_eval(int function, const double* in, double* out, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
