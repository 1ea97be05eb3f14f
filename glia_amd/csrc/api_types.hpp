// glia_amd/csrc/api_types.hpp -- the opaque handles of include/glia_hmt.h as the library sees them (api.cpp, slab_dist.cpp).
#pragma once
#include "glibc_math.hpp"
#include "greedy_common.hpp"
#include "hmt_internal.hpp"

using namespace glia;

struct glia_hmt_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // accumulation hash tables (kept clean between builds: compaction clears what it consumes)
  uint32_t rcap = 0, pcap = 0;
  uint32_t* rkeys = nullptr; uint32_t* rrec = nullptr;
  unsigned long long* pkeys = nullptr; uint32_t* prec = nullptr;
  uint32_t* flags = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  uint32_t hint_rcap = 0, hint_pcap = 0;
  double transform_ms = 0;
  int tz = kTZ;                  // tile depth of the accumulation pass; halved when the LDS tables of a pass overflowed a lot
  LibmSel libm = {kLibmDevice, kLibmDevice, kLibmDevice};   // restatement of the host's log2 / log the kernels use (glibc_math.hpp)
};

struct glia_hmt_rag {
  glia_hmt_ctx* ctx = nullptr;
  int dim = 3;
  int64_t dims[3] = {1, 1, 1};
  bool only_contour = false;
  int bins = 0, nthr = 0;
  RagArrays arr;
  double pass_ms = 0, alg_bytes = 0;
  double ms_table = 0, ms_init = 0, ms_loop = 0;
  int64_t n_scored = 0;
  glia_hmt_feat_config cfg;
  bool has_cfg = false;
  VolumeRef vol;                 // whole-volume builds only: the caller keeps the volumes alive while the handle lives
  VolumeRef slab;                // slab builds: the planes handed in and the owned range (valid while the caller keeps them)
  uint32_t* d_folded = nullptr;  // labels with the mask folded in (owned)
  int map_region[GLIA_HMT_MAX_IMAGES] = {0}, map_rlabel[GLIA_HMT_MAX_IMAGES] = {0}, map_boundary[GLIA_HMT_MAX_IMAGES] = {0};   // list entry -> channel
};

