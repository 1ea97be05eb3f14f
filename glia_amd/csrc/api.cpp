// glia_amd/csrc/api.cpp -- C ABI of libglia_hmt.so (see include/glia_hmt.h for the reference
// operators each entry point replaces).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <unordered_map>

#include "bc_features.hpp"
#include "forest.hpp"
#include "rmap_order.hpp"
#include "hmt_internal.hpp"

namespace glia {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

// ---- options ------------------------------------------------------------------------------------------------------------
// Tuning and test switches of the loops (which queue, window size, baseline interval, helper count, instance tier ...).  One
// process-wide table, set through glia_hmt_set_option(); it starts from the environment variables of the same names, read ONCE
// when the table is first used -- no entry point calls getenv() per call.
static const char* const kOptionKeys[] = {"GLIA_HMT_PB_WINDOW", "GLIA_HMT_PB_BATCH", "GLIA_HMT_WINCAP", "GLIA_HMT_REBASE", "GLIA_HMT_HORIZON",
                                          "GLIA_HMT_FORCE_TREE", "GLIA_HMT_HELPERS", "GLIA_HMT_TRACE", "GLIA_HMT_LIBM", "GLIA_HMT_BC_NOCOMMON",
                                          "GLIA_HMT_BC_GENERIC", "GLIA_HMT_DEBUG", "GLIA_HMT_MAXITERS"};
static std::mutex g_opt_mu;
static std::unordered_map<std::string, std::string> g_opt;
static bool g_opt_init = false;
static void opt_init_locked() {
  if (g_opt_init) return;
  for (const char* k : kOptionKeys) if (const char* e = getenv(k)) g_opt[k] = e;
  g_opt_init = true;
}
static bool opt_known(const char* key) { for (const char* k : kOptionKeys) if (!strcmp(k, key)) return true; return false; }
bool option(const char* key, std::string* value) {
  std::lock_guard<std::mutex> lock(g_opt_mu);
  opt_init_locked();
  auto it = g_opt.find(key);
  if (it == g_opt.end()) return false;
  if (value) *value = it->second;
  return true;
}
int set_option(const char* key, const char* value) {
  if (!key || !opt_known(key)) { set_error(std::string("set_option: unknown option ") + (key ? key : "(null)")); return GLIA_HMT_ERR_ARG; }
  std::lock_guard<std::mutex> lock(g_opt_mu);
  opt_init_locked();
  if (value) g_opt[key] = value; else g_opt.erase(key);
  return GLIA_HMT_OK;
}

static float ceil_f32(double d) {   // smallest float >= d
  if (std::isinf(d) || std::isnan(d)) return (float)d;
  float f = (float)d;
  if ((double)f < d) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
  return f;
}
static float floor_f32(double d) {  // largest float <= d
  if (std::isinf(d) || std::isnan(d)) return (float)d;
  float f = (float)d;
  if ((double)f > d) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
  return f;
}
static uint32_t next_pow2(uint64_t v) {
  uint64_t p = 1;
  while (p < v) p <<= 1;
  return (uint32_t)std::min<uint64_t>(p, 1ull << 31);
}

// util/image_stats.hxx:17-22: bounds[0] = interval (range.first ignored), bounds[i] = bounds[i-1] + interval
HistSpec make_hist_spec(int bins, double lo, double hi) {
  HistSpec h;
  h.bins = bins;
  double interval = (hi - lo) / bins;
  double b = 0.0;
  for (int i = 0; i < GLIA_HMT_MAX_BINS; ++i) {
    if (i < bins) { b = (i == 0) ? interval : b + interval; h.fb[i] = ceil_f32(b); }
    else h.fb[i] = std::numeric_limits<float>::infinity();
  }
  h.lo_f = floor_f32(lo);
  h.hi_f = ceil_f32(hi);
  return h;
}


// Which restatement of glibc's log2 / log reproduces THIS host's libm bit for bit (the reference calls std::log2 in
// stats::entropy, util/stats.hxx:150, and std::log in slog, glia_base.hxx:80-81; the device must return the same bits).
// Probe vector: histogram fractions c/n, values around 1 (the near-one branch), and a sweep over exponents.
// GLIA_HMT_LIBM = device | sse2 | fma overrides the choice (tests).
static LibmSel probe_host_libm() {
  std::vector<double> xs;
  uint64_t s = 0x9E3779B97F4A7C15ull;
  auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (int i = 0; i < 6000; ++i) { const uint64_t n = 1 + next() % 100000, c = 1 + next() % n; xs.push_back((double)c / (double)n); }
  for (int i = 0; i < 3000; ++i) xs.push_back(1.0 + ((double)(int64_t)(next() % 2000001) - 1000000.0) * 1e-7);
  for (int i = 0; i < 3000; ++i) xs.push_back(std::ldexp(1.0 + (double)(next() >> 12) * 0x1p-52, (int)(next() % 120) - 60));
  for (int e = -1074; e < 1024; e += 37) xs.push_back(std::ldexp(1.0, e));
  LibmSel sel = {kLibmDevice, kLibmDevice, kLibmDevice};
  bool l2 = true, ls = true, lf = true, ps = true, pf = true;
  for (double x : xs) {
    volatile double vx = x;
    const double h2 = std::log2(vx), h1 = std::log(vx);
    l2 = l2 && glibc::as_u64(h2) == glibc::as_u64(glibc::log2_sse2(x));
    ls = ls && glibc::as_u64(h1) == glibc::as_u64(glibc::log_sse2(x));
    lf = lf && glibc::as_u64(h1) == glibc::as_u64(glibc::log_fma(x));
  }
  // std::pow(perim, 1.5) of the compactness feature (type/feat.hxx:78-79): integer perimeters, small and large
  for (int i = 0; i < 12000; ++i) {
    const double x = i < 4000 ? (double)(i + 1) : (double)(1 + next() % (i < 8000 ? (1ull << 24) : (1ull << 40)));
    volatile double vx = x, vy = 1.5;
    const uint64_t h = glibc::as_u64(std::pow(vx, vy));
    ps = ps && h == glibc::as_u64(glibc::pow_sse2(x, 1.5));
    pf = pf && h == glibc::as_u64(glibc::pow_fma(x, 1.5));
  }
  if (l2) sel.log2_variant = kLibmSse2;
  if (lf) sel.log_variant = kLibmFma; else if (ls) sel.log_variant = kLibmSse2;
  if (pf) sel.pow_variant = kLibmFma; else if (ps) sel.pow_variant = kLibmSse2;
  std::string ov;
  if (option("GLIA_HMT_LIBM", &ov)) {
    const char* e = ov.c_str();
    const int v = !strcmp(e, "sse2") ? kLibmSse2 : !strcmp(e, "fma") ? kLibmFma : kLibmDevice;
    sel.log_variant = v; sel.log2_variant = v == kLibmFma ? kLibmSse2 : v; sel.pow_variant = v;
  }
  return sel;
}

}  // namespace glia

namespace glia {
int greedy_bc(const RagArrays& rag, const BcCfg& cfg, const DeviceClassifier& clf, hipStream_t stream, uint32_t* h_order,
              double* h_sal, double* h_feats, int64_t capacity, int64_t* n_merges, double* ms_table, double* ms_init,
              double* ms_loop, int64_t* n_scored, bool init_only, const uint32_t* h_forced, int64_t n_forced, int shard,
              int n_shards, double* h_scores) {
  auto* fn = &greedy_bc_generic;
  const bool common = cfg.K == 1 && cfg.n_region == 1 && cfg.n_rlabel == 0 && cfg.n_boundary == 1 && !cfg.use_hist && !cfg.use_log &&
                      !cfg.use_simple && !option("GLIA_HMT_BC_NOCOMMON");
  if (cfg.libm_log2 == kLibmSse2 && cfg.libm_log == kLibmFma && cfg.libm_pow == kLibmFma) fn = common ? &greedy_bc_fma_common : &greedy_bc_fma;
  else if (cfg.libm_log2 == kLibmSse2 && cfg.libm_log == kLibmSse2 && cfg.libm_pow == kLibmSse2) fn = common ? &greedy_bc_sse2_common : &greedy_bc_sse2;
  if (option("GLIA_HMT_BC_GENERIC")) fn = &greedy_bc_generic;       // tests: the run-time-dispatch instance
  return fn(rag, cfg, clf, stream, h_order, h_sal, h_feats, capacity, n_merges, ms_table, ms_init, ms_loop, n_scored, init_only,
            h_forced, n_forced, shard, n_shards, h_scores);
}
}  // namespace glia

using namespace glia;

#include "api_types.hpp"

static std::atomic<int> g_live_contexts{0};

static int free_tables(glia_hmt_ctx* c) {
  if (c->rkeys) GLIA_HIP_TRY(hipFree(c->rkeys));
  if (c->rrec) GLIA_HIP_TRY(hipFree(c->rrec));
  if (c->pkeys) GLIA_HIP_TRY(hipFree(c->pkeys));
  if (c->prec) GLIA_HIP_TRY(hipFree(c->prec));
  c->rkeys = nullptr; c->rrec = nullptr; c->pkeys = nullptr; c->prec = nullptr;
  c->rcap = c->pcap = 0;
  return GLIA_HMT_OK;
}

static int ensure_tables(glia_hmt_ctx* c, uint32_t rcap, uint32_t pcap) {
  if (rcap <= c->rcap && pcap <= c->pcap) return GLIA_HMT_OK;
  rcap = std::max(rcap, c->rcap);
  pcap = std::max(pcap, c->pcap);
  int rc = free_tables(c);
  if (rc) return rc;
  GLIA_HIP_TRY(hipMalloc(&c->rkeys, sizeof(uint32_t) * (size_t)rcap));
  GLIA_HIP_TRY(hipMalloc(&c->rrec, sizeof(uint32_t) * kRegionWords * (size_t)rcap));
  GLIA_HIP_TRY(hipMalloc(&c->pkeys, sizeof(unsigned long long) * (size_t)pcap));
  GLIA_HIP_TRY(hipMalloc(&c->prec, sizeof(uint32_t) * kPairWords * (size_t)pcap));
  GLIA_HIP_TRY(hipMemsetAsync(c->rkeys, 0, sizeof(uint32_t) * (size_t)rcap, c->stream));
  GLIA_HIP_TRY(hipMemsetAsync(c->rrec, 0, sizeof(uint32_t) * kRegionWords * (size_t)rcap, c->stream));
  GLIA_HIP_TRY(hipMemsetAsync(c->pkeys, 0, sizeof(unsigned long long) * (size_t)pcap, c->stream));
  GLIA_HIP_TRY(hipMemsetAsync(c->prec, 0, sizeof(uint32_t) * kPairWords * (size_t)pcap, c->stream));
  c->rcap = rcap;
  c->pcap = pcap;
  return GLIA_HMT_OK;
}

static int clear_tables(glia_hmt_ctx* c) {
  GLIA_HIP_TRY(hipMemsetAsync(c->rkeys, 0, sizeof(uint32_t) * (size_t)c->rcap, c->stream));
  GLIA_HIP_TRY(hipMemsetAsync(c->rrec, 0, sizeof(uint32_t) * kRegionWords * (size_t)c->rcap, c->stream));
  GLIA_HIP_TRY(hipMemsetAsync(c->pkeys, 0, sizeof(unsigned long long) * (size_t)c->pcap, c->stream));
  GLIA_HIP_TRY(hipMemsetAsync(c->prec, 0, sizeof(uint32_t) * kPairWords * (size_t)c->pcap, c->stream));
  return GLIA_HMT_OK;
}

extern "C" {

const char* glia_hmt_last_error(void) { return g_err.c_str(); }
const char* glia_hmt_version(void) { return "glia_hmt 0.1 (gfx950)"; }

int glia_hmt_ctx_create(int device, void* hip_stream, glia_hmt_ctx** out) {
  if (!out) { set_error("ctx_create: out is NULL"); return GLIA_HMT_ERR_ARG; }
  int n = 0;
  GLIA_HIP_TRY(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) { set_error("ctx_create: no such HIP device"); return GLIA_HMT_ERR_ARG; }
  GLIA_HIP_TRY(hipSetDevice(device));
  glia_hmt_ctx* c = new glia_hmt_ctx;
  c->device = device;
  if (hip_stream) c->stream = (hipStream_t)hip_stream;
  else { GLIA_HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
  GLIA_HIP_TRY(hipMalloc(&c->flags, 256));      // 8 status words + profiling counters of the accumulation pass (profiling builds)
  GLIA_HIP_TRY(hipMemsetAsync(c->flags, 0, 256, c->stream));
  GLIA_HIP_TRY(hipEventCreate(&c->ev0));
  GLIA_HIP_TRY(hipEventCreate(&c->ev1));
  static const LibmSel sel = probe_host_libm();
  c->libm = sel;
  *out = c;
  ++g_live_contexts;
  // The call succeeds either way; a host libm none of the restatements reproduces is reported through glia_hmt_last_error()
  // (and glia_hmt_ctx_libm_status): entropy, --logs and compactness columns are then within 1 ulp of the host's, not pinned.
  if (sel.log2_variant == 0 || sel.log_variant == 0 || sel.pow_variant == 0)
    set_error(std::string("warning: the host libm's ") + (sel.log2_variant == 0 ? "log2 " : "") + (sel.log_variant == 0 ? "log " : "") +
              (sel.pow_variant == 0 ? "pow " : "") + "is not reproduced bit for bit by a restatement (glibc 2.35 FMA / non-FMA builds): "
              "the entropy / log / compactness feature columns are within 1 ulp of this host's values, unpinned; equal classifier scores may order differently");
  return GLIA_HMT_OK;
}

int glia_hmt_ctx_libm(const glia_hmt_ctx* c, int* log2_variant, int* log_variant) {
  if (!c) return GLIA_HMT_ERR_ARG;
  if (log2_variant) *log2_variant = c->libm.log2_variant;
  if (log_variant) *log_variant = c->libm.log_variant;
  return GLIA_HMT_OK;
}

int glia_hmt_ctx_libm_status(const glia_hmt_ctx* c) {
  if (!c) return GLIA_HMT_ERR_ARG;
  return (c->libm.log2_variant != 0 && c->libm.log_variant != 0 && c->libm.pow_variant != 0) ? 1 : 0;
}

int glia_hmt_ctx_libm_pow(const glia_hmt_ctx* c, int* pow_variant) {
  if (!c || !pow_variant) return GLIA_HMT_ERR_ARG;
  *pow_variant = c->libm.pow_variant;
  return GLIA_HMT_OK;
}

int glia_hmt_host_libm_probe_pow(int* pow_variant) {
  if (!pow_variant) return GLIA_HMT_ERR_ARG;
  *pow_variant = probe_host_libm().pow_variant;
  return GLIA_HMT_OK;
}

int glia_hmt_host_rmap_ranks(const uint32_t* labels, const int64_t* first_voxel, int64_t n, int mode, uint32_t* rank) {
  if (!labels || !first_voxel || !rank || n < 0 || mode < 0 || mode > 2) { set_error("host_rmap_ranks: invalid argument"); return GLIA_HMT_ERR_ARG; }
  std::vector<uint32_t> lab(labels, labels + n), out;
  std::vector<long long> first(first_voxel, first_voxel + n);
  for (int64_t i = 1; i < n; ++i) if (lab[i] <= lab[i - 1]) { set_error("host_rmap_ranks: labels must be ascending"); return GLIA_HMT_ERR_ARG; }
  rmap_ranks(lab, first, &out, mode);
  std::copy(out.begin(), out.end(), rank);
  return GLIA_HMT_OK;
}

int glia_hmt_host_libm_probe(int* log2_variant, int* log_variant) {
  const LibmSel sel = probe_host_libm();
  if (log2_variant) *log2_variant = sel.log2_variant;
  if (log_variant) *log_variant = sel.log_variant;
  return GLIA_HMT_OK;
}

int glia_hmt_host_libm_eval(int function, int variant, const double* h_in, double* h_out, int64_t n) {
  if (!h_in || !h_out || n < 0 || function < 0 || function > 2 || (variant != kLibmSse2 && variant != kLibmFma)) {
    set_error("host_libm_eval: invalid argument");
    return GLIA_HMT_ERR_ARG;
  }
  for (int64_t i = 0; i < n; ++i) {
    const double x = h_in[i];
    if (function == 2) h_out[i] = !glibc::pow_in_domain(x, 1.5) ? std::pow(x, 1.5) : variant == kLibmFma ? glibc::pow_fma(x, 1.5) : glibc::pow_sse2(x, 1.5);
    else h_out[i] = function == 0 ? glibc::log2_sse2(x) : variant == kLibmFma ? glibc::log_fma(x) : glibc::log_sse2(x);
  }
  return GLIA_HMT_OK;
}

int glia_hmt_libm_eval(glia_hmt_ctx* c, int function, int variant, const double* d_in, double* d_out, int64_t n) {
  if (!c || !d_in || !d_out || n < 0 || function < 0 || function > 2) { set_error("libm_eval: invalid argument"); return GLIA_HMT_ERR_ARG; }
  GLIA_HIP_TRY(hipSetDevice(c->device));
  return launch_libm_eval(function, variant, d_in, d_out, n, c->stream);
}

unsigned long long glia_hmt_release_cached_memory(void) { return (unsigned long long)glia::BlockCache::get().trim(); }
unsigned long long glia_hmt_internal_errors(void) { return glia::internal_errors(); }
int glia_hmt_set_option(const char* key, const char* value) { return glia::set_option(key, value); }
int glia_hmt_check_merge_order(const uint32_t* h_order, int64_t n_merges, int64_t n_regions, int64_t* first_bad) {
  if ((!h_order && n_merges) || n_merges < 0 || n_regions < 0 || n_regions > 0x7FFFFFFFll) { set_error("check_merge_order: invalid argument"); return GLIA_HMT_ERR_ARG; }
  int64_t bad = -1;
  if (glia::merge_order_is_consistent(h_order, n_merges, (uint32_t)n_regions, &bad)) { if (first_bad) *first_bad = -1; return GLIA_HMT_OK; }
  if (first_bad) *first_bad = bad;
  set_error("check_merge_order: merge " + std::to_string(bad) + " joins a region that does not exist (any more) or creates the wrong one");
  return GLIA_HMT_ERR_ARG;
}

void glia_hmt_ctx_destroy(glia_hmt_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  (void)free_tables(c);
  if (c->flags) (void)hipFree(c->flags);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
  if (--g_live_contexts <= 0) (void)glia::BlockCache::get().trim();      // nobody left to reuse the parked blocks
}

int glia_hmt_ctx_sync(glia_hmt_ctx* c) {
  if (!c) return GLIA_HMT_ERR_ARG;
  GLIA_HIP_TRY(hipStreamSynchronize(c->stream));
  return GLIA_HMT_OK;
}

int glia_hmt_ctx_set_table_hint(glia_hmt_ctx* c, int64_t expected_regions, int64_t expected_pairs) {
  if (!c) return GLIA_HMT_ERR_ARG;
  c->hint_rcap = expected_regions > 0 ? next_pow2((uint64_t)expected_regions * 2) : 0;
  c->hint_pcap = expected_pairs > 0 ? next_pow2((uint64_t)expected_pairs * 2) : 0;
  return GLIA_HMT_OK;
}

int glia_hmt_synth(glia_hmt_ctx* c, int dim, const int64_t dims[3], int S, int G, uint64_t seed, int variant,
                   uint32_t* d_labels, float* d_pb) {
  if (!c || !dims || !d_labels || !d_pb || (dim != 2 && dim != 3) || S <= 0 || G <= 0) {
    set_error("synth: invalid argument");
    return GLIA_HMT_ERR_ARG;
  }
  GLIA_HIP_TRY(hipSetDevice(c->device));
  int64_t N = dims[0] * dims[1] * (dim == 3 ? dims[2] : 1);
  uint32_t* d_truth = nullptr;
  GLIA_HIP_TRY(hipMalloc(&d_truth, sizeof(uint32_t) * (size_t)N));
  int rc = launch_synth(dim, dims, S, G, seed, variant, d_labels, d_truth, d_pb, c->stream);
  if (rc == GLIA_HMT_OK) GLIA_HIP_TRY(hipStreamSynchronize(c->stream));
  GLIA_HIP_TRY(hipFree(d_truth));
  return rc;
}

static int rag_build_impl(glia_hmt_ctx* c, int dim, const int64_t dims[3], int64_t gz0, int64_t gnz, int64_t zb, int64_t ze,
                          const uint32_t* d_labels, const uint32_t* d_mask, int only_contour, const float* d_pb,
                          const glia_hmt_feat_config* cfg, glia_hmt_rag** out) {
  if (!c || !dims || !d_labels || !out || (dim != 2 && dim != 3)) {
    set_error("rag_build: invalid argument");
    return GLIA_HMT_ERR_ARG;
  }
  const int64_t nx = dims[0], ny = dims[1], nz = dim == 3 ? dims[2] : 1;
  if (nx <= 0 || ny <= 0 || nz <= 0 || nx >= (1ll << 31) || ny >= (1ll << 31) || nz >= (1ll << 31)) {
    set_error("rag_build: bad dimensions");
    return GLIA_HMT_ERR_ARG;
  }
  GLIA_HIP_TRY(hipSetDevice(c->device));
  // ---- channels: one accumulation pass per distinct (volume, bins, lo, hi) over the three feature lists; channel 0
  // is the boundary-probability volume (thresholded counts, pb linkages) ----
  struct Chan { const float* img; int bins; double lo, hi; };
  std::vector<Chan> chans;
  int map_region[GLIA_HMT_MAX_IMAGES] = {0}, map_rlabel[GLIA_HMT_MAX_IMAGES] = {0}, map_boundary[GLIA_HMT_MAX_IMAGES] = {0};
  int nthr = 0;
  double thr[GLIA_HMT_MAX_THRESH] = {0, 0, 0, 0};
  if (cfg) {
    if (cfg->n_region < 0 || cfg->n_rlabel < 0 || cfg->n_boundary < 0 || cfg->n_region > kMaxListed ||
        cfg->n_rlabel > kMaxListed || cfg->n_boundary > kMaxListed ||
        cfg->n_thresholds < 0 || cfg->n_thresholds > GLIA_HMT_MAX_THRESH) {
      set_error("rag_build: feature configuration out of range (at most " + std::to_string(kMaxListed) + " images per list, " + std::to_string(GLIA_HMT_MAX_THRESH) + " thresholds)");
      return GLIA_HMT_ERR_ARG;
    }
    const float* pbv = cfg->d_pb ? cfg->d_pb : d_pb;
    auto same = [](const Chan& c, const glia_hmt_image& g) { return c.img == g.d_image && c.bins == g.bins && c.lo == g.lo && c.hi == g.hi; };
    // channel 0: the first listed image that IS the pb volume (its histogram spec comes along), else pb with 8 bins on [0,1]
    const glia_hmt_image* lists[3] = {cfg->region, cfg->rlabel, cfg->boundary};
    const int counts[3] = {cfg->n_region, cfg->n_rlabel, cfg->n_boundary};
    for (int l = 0; l < 3 && chans.empty(); ++l)
      for (int i = 0; i < counts[l] && chans.empty(); ++i)
        if (pbv && lists[l][i].d_image == pbv) chans.push_back(Chan{pbv, lists[l][i].bins, lists[l][i].lo, lists[l][i].hi});
    if (chans.empty()) {
      if (pbv) chans.push_back(Chan{pbv, 8, 0.0, 1.0});
      else if (counts[0] + counts[1] + counts[2] > 0) {      // no pb at all: thresholds count on the first image
        const glia_hmt_image& g = counts[0] ? cfg->region[0] : (counts[1] ? cfg->rlabel[0] : cfg->boundary[0]);
        chans.push_back(Chan{g.d_image, g.bins, g.lo, g.hi});
      }
    }
    int* maps[3] = {map_region, map_rlabel, map_boundary};
    for (int l = 0; l < 3; ++l)
      for (int i = 0; i < counts[l]; ++i) {
        const glia_hmt_image& g = lists[l][i];
        if (!g.d_image) { set_error("rag_build: null image in a feature list"); return GLIA_HMT_ERR_ARG; }
        int k = -1;
        for (size_t q = 0; q < chans.size(); ++q) if (same(chans[q], g)) { k = (int)q; break; }
        if (k < 0) {
          if ((int)chans.size() >= kMaxChannels) { set_error("rag_build: more than 4 distinct (image volume, histogram) channels"); return GLIA_HMT_ERR_UNSUPPORTED; }
          chans.push_back(Chan{g.d_image, g.bins, g.lo, g.hi});
          k = (int)chans.size() - 1;
        }
        maps[l][i] = k;
      }
    nthr = cfg->n_thresholds;
    for (int i = 0; i < nthr; ++i) thr[i] = cfg->thresholds[i];
  } else if (d_pb) chans.push_back(Chan{d_pb, 8, 0.0, 1.0});
  if (chans.empty()) { set_error("rag_build: no image volume given"); return GLIA_HMT_ERR_ARG; }
  for (const Chan& ch : chans)
    if (ch.bins < 1 || ch.bins > GLIA_HMT_MAX_BINS || !(ch.hi > ch.lo)) {
      set_error("rag_build: histogram bins must be 1..16 and hi > lo");
      return GLIA_HMT_ERR_ARG;
    }
  const float* img = chans[0].img;
  const int bins = chans[0].bins;

  const int64_t N = nx * ny * nz;
  uint32_t rcap = c->hint_rcap ? c->hint_rcap : next_pow2(std::max<int64_t>(1 << 12, N / 128));
  uint32_t pcap = c->hint_pcap ? c->hint_pcap : next_pow2(std::max<int64_t>(1 << 14, N / 32));

  glia_hmt_rag* rag = new glia_hmt_rag;
  rag->ctx = c; rag->dim = dim; rag->dims[0] = nx; rag->dims[1] = ny; rag->dims[2] = gnz;
  rag->only_contour = only_contour != 0;
  rag->bins = bins; rag->nthr = nthr;
  if (cfg) { rag->cfg = *cfg; rag->has_cfg = true; }
  const uint32_t* lab_nb = d_labels;
  const uint32_t* lab_c = d_labels;
  if (d_mask) {
    GLIA_HIP_TRY(hipMalloc(&rag->d_folded, sizeof(uint32_t) * (size_t)N));
    int rcm = launch_mask_fold(d_labels, d_mask, rag->d_folded, N, c->stream);
    if (rcm) { glia_hmt_rag_free(rag); return rcm; }
    lab_nb = rag->d_folded;
    lab_c = only_contour ? d_labels : rag->d_folded;      // util/struct.hxx:133-143 has no mask test on the centre voxel
  }
  if (zb == 0 && ze == nz && gz0 == 0 && gnz == nz) {
    rag->vol.lab = lab_c; rag->vol.lab_nb = lab_nb; rag->vol.pb = d_pb ? d_pb : img; rag->vol.dim = dim;
    rag->vol.nx = nx; rag->vol.ny = ny; rag->vol.nz = nz;
  }
  if (!d_mask && dim == 3) {
    // a slab: the planes handed in, of which [zb, ze) are owned -- what collect_pair_values walks for the median linkage of the
    // slab route (the caller keeps the planes alive until glia_hmt_rag_build_distributed returns)
    rag->slab.lab = lab_c; rag->slab.lab_nb = lab_nb; rag->slab.pb = d_pb ? d_pb : img; rag->slab.dim = 3;
    rag->slab.nx = nx; rag->slab.ny = ny; rag->slab.nz = nz; rag->slab.zb = zb; rag->slab.ze = ze;
  }

  memcpy(rag->map_region, map_region, sizeof(map_region)); memcpy(rag->map_rlabel, map_rlabel, sizeof(map_rlabel));
  memcpy(rag->map_boundary, map_boundary, sizeof(map_boundary));
  double pass_ms_total = 0.0;
  for (size_t ci = 0; ci < chans.size(); ++ci) {
  RagArrays pass_arr;
  for (int attempt = 0;; ++attempt) {
    int rc = ensure_tables(c, rcap, pcap);
    if (rc) { glia_hmt_rag_free(rag); return rc; }
    AccParams p;
    p.lab = lab_nb; p.lab_c = lab_c; p.masked = d_mask ? 1 : 0; p.img = chans[ci].img;
    p.nx = nx; p.ny = ny; p.nz = nz; p.dim = dim;
    p.gz0 = gz0; p.gnz = gnz; p.zb = zb; p.ze = ze;
    p.nbx = (int)((nx + kTileX - 1) / kTileX);
    p.nby = (int)((ny + kTileY - 1) / kTileY);
    p.tz = c->tz;
    p.nbz = (int)((ze - zb + p.tz - 1) / p.tz);
    p.hist = make_hist_spec(chans[ci].bins, chans[ci].lo, chans[ci].hi);
    p.nthr = nthr;
    for (int i = 0; i < GLIA_HMT_MAX_THRESH; ++i) p.thr_f[i] = i < nthr ? ceil_f32(thr[i]) : std::numeric_limits<float>::infinity();
    p.rkeys = c->rkeys; p.rrec = c->rrec; p.rmask = c->rcap - 1;
    p.pkeys = c->pkeys; p.prec = c->prec; p.pmask = c->pcap - 1;
    p.flags = c->flags;
    { std::string dbg; p.debug = option("GLIA_HMT_DEBUG", &dbg) ? (uint32_t)strtoul(dbg.c_str(), nullptr, 0) : 0u; }
    if ((int64_t)p.nbx * p.nby * p.nbz >= (1ll << 31)) { glia_hmt_rag_free(rag); set_error("rag_build: volume too large"); return GLIA_HMT_ERR_ARG; }
    // the pass addresses a column's rows with 32-bit byte offsets from a base that moves every four planes (rag_accumulate.hip)
    if (nx * ny * 4 * 11 >= (1ll << 32)) { glia_hmt_rag_free(rag); set_error("rag_build: planes of more than 97 M voxels are not supported"); return GLIA_HMT_ERR_ARG; }
    hipError_t e = hipEventRecord(c->ev0, c->stream);
    if (e == hipSuccess) { rc = launch_accumulate(p, c->stream); e = hipEventRecord(c->ev1, c->stream); }
    if (e != hipSuccess) { glia_hmt_rag_free(rag); set_error(hipGetErrorString(e)); return GLIA_HMT_ERR_HIP; }
    if (rc) { glia_hmt_rag_free(rag); return rc; }
    uint32_t flags[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    e = hipMemcpyAsync(flags, c->flags, sizeof(flags), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { glia_hmt_rag_free(rag); set_error(std::string("rag_build: ") + hipGetErrorString(e)); return GLIA_HMT_ERR_HIP; }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
    pass_ms_total += ms;
    rag->pass_ms = pass_ms_total;
    if (p.debug & 32) {
      unsigned long long pc[12];
      (void)hipMemcpy(pc, c->flags + 16, sizeof(pc), hipMemcpyDeviceToHost);
      fprintf(stderr, "[glia_hmt debug] marching waves: %llu waits for room in a ring, %llu cycles waiting of %llu cycles marching\n", pc[9], pc[8], pc[10]);
      fprintf(stderr, "[glia_hmt debug] drainers: region batches %llu entries %llu cycles %llu | pair batches %llu entries %llu cycles %llu | idle polls %llu, drainer cycles %llu\n",
              pc[0], pc[1], pc[2], pc[3], pc[4], pc[5], pc[6], pc[7]);
      (void)hipMemsetAsync(c->flags, 0, 256, c->stream);
    }
    rag->alg_bytes = (double)(nx * ny * (ze - zb)) * 8.0 * (double)chans.size();
    // runs that found the tile's LDS tables full went to the global tables one by one (exact, slow): with more than one
    // such run per 256 voxels the supervoxels are too small for this tile depth -- use shallower tiles from now on
    if (c->tz > 4 && (double)flags[7] * 256.0 > (double)(nx * ny * (ze - zb))) c->tz /= 2;
    if (flags[7]) (void)hipMemsetAsync(c->flags + 7, 0, sizeof(uint32_t), c->stream);
    if (flags[0] || flags[1]) {
      // a table filled up: drop the partial result, grow and redo the pass
      if (attempt >= 6) { glia_hmt_rag_free(rag); set_error("rag_build: hash tables keep overflowing"); return GLIA_HMT_ERR_HIP; }
      (void)hipMemsetAsync(c->flags, 0, 64, c->stream);
      rc = clear_tables(c);
      if (rc) { glia_hmt_rag_free(rag); return rc; }
      if (flags[0]) rcap = c->rcap * 4;
      if (flags[1]) pcap = c->pcap * 4;
      continue;
    }
    rc = compact_tables(p, c->rcap, c->pcap, &pass_arr, c->stream);
    if (rc) { glia_hmt_rag_free(rag); return rc; }
    break;
  }
  if (ci == 0) rag->arr = pass_arr;
  else {
    // the keys depend on the labels only: every pass yields the same sorted key arrays
    (void)hipFree(pass_arr.d_rlabel); (void)hipFree(pass_arr.d_pa); (void)hipFree(pass_arr.d_pb);
    if (pass_arr.R != rag->arr.R || pass_arr.P != rag->arr.P) {
      (void)hipFree(pass_arr.d_rrec); (void)hipFree(pass_arr.d_prec);
      glia_hmt_rag_free(rag); set_error("rag_build: channel passes disagree on the region set"); return GLIA_HMT_ERR_HIP;
    }
  }
  rag->arr.c_rrec[ci] = pass_arr.d_rrec; rag->arr.c_prec[ci] = pass_arr.d_prec; rag->arr.c_bins[ci] = chans[ci].bins;
  rag->arr.K = (int)ci + 1;
  }
  *out = rag;
  return GLIA_HMT_OK;
}

int glia_hmt_rag_build(glia_hmt_ctx* c, int dim, const int64_t dims[3], const uint32_t* d_labels,
                       const uint32_t* d_mask, int only_contour, const float* d_pb,
                       const glia_hmt_feat_config* cfg, glia_hmt_rag** out) {
  if (!dims) { set_error("rag_build: invalid argument"); return GLIA_HMT_ERR_ARG; }
  const int64_t nz = dim == 3 ? dims[2] : 1;
  return rag_build_impl(c, dim, dims, 0, nz, 0, nz, d_labels, d_mask, only_contour, d_pb, cfg, out);
}

int glia_hmt_rag_build_slab(glia_hmt_ctx* c, const int64_t dims_local[3], int64_t z_global_of_plane0, int64_t nz_global,
                            int64_t z_begin, int64_t z_end, const uint32_t* d_labels, int only_contour, const float* d_pb,
                            const glia_hmt_feat_config* cfg, glia_hmt_rag** out) {
  if (!dims_local || z_begin < 0 || z_end > dims_local[2] || z_begin >= z_end || z_global_of_plane0 < 0 ||
      z_global_of_plane0 + dims_local[2] > nz_global ||
      (z_begin > 0 ? false : z_global_of_plane0 != 0) || (z_end < dims_local[2] ? false : z_global_of_plane0 + z_end != nz_global)) {
    // the plane below z_begin / above z_end-1 must be present unless the slab touches the volume face
    set_error("rag_build_slab: slab range or halo planes inconsistent");
    return GLIA_HMT_ERR_ARG;
  }
  return rag_build_impl(c, 3, dims_local, z_global_of_plane0, nz_global, z_begin, z_end, d_labels, nullptr, only_contour, d_pb,
                        cfg, out);
}

int glia_hmt_rag_merge(glia_hmt_ctx* c, glia_hmt_rag* const* parts, int n_parts, glia_hmt_rag** out) {
  if (!c || !parts || n_parts < 1 || n_parts > 255 || !out) { set_error("rag_merge: invalid argument"); return GLIA_HMT_ERR_ARG; }
  GLIA_HIP_TRY(hipSetDevice(c->device));
  std::vector<RagArrays> arrs;
  for (int i = 0; i < n_parts; ++i) {
    if (!parts[i] || parts[i]->ctx != c || parts[i]->bins != parts[0]->bins || parts[i]->nthr != parts[0]->nthr || parts[i]->arr.K != parts[0]->arr.K ||
        parts[i]->dims[0] != parts[0]->dims[0] || parts[i]->dims[1] != parts[0]->dims[1] || parts[i]->dims[2] != parts[0]->dims[2]) {
      set_error("rag_merge: parts do not belong together");
      return GLIA_HMT_ERR_ARG;
    }
    arrs.push_back(parts[i]->arr);
  }
  glia_hmt_rag* rag = new glia_hmt_rag(*parts[0]);
  rag->arr = RagArrays();
  rag->pass_ms = 0; rag->alg_bytes = 0;
  for (int i = 0; i < n_parts; ++i) { rag->pass_ms += parts[i]->pass_ms; rag->alg_bytes += parts[i]->alg_bytes; }
  int rc = merge_rag_arrays(arrs.data(), n_parts, &rag->arr, c->stream);
  rag->d_folded = nullptr; rag->vol = VolumeRef();
  if (rc) { delete rag; return rc; }
  *out = rag;
  return GLIA_HMT_OK;
}

int glia_hmt_watershed(glia_hmt_ctx* c, int dim, const int64_t dims[3], const float* d_image, double level, uint32_t* d_labels, uint32_t* n_labels,
                       int* sweeps) {
  if (!c || !dims || !d_image || !d_labels || (dim != 2 && dim != 3) || !(level >= 0.0)) { set_error("watershed: invalid argument"); return GLIA_HMT_ERR_ARG; }
  GLIA_HIP_TRY(hipSetDevice(c->device));
  return watershed_labels(dim, dims, d_image, level, d_labels, n_labels, sweeps, c->stream);
}

int glia_hmt_rag_num_channels(const glia_hmt_rag* r) { return r ? r->arr.K : -1; }

int glia_hmt_rag_copy_channel(const glia_hmt_rag* r, int channel, uint32_t* d_region_rec, uint32_t* d_pair_rec) {
  if (!r || channel < 0 || channel >= r->arr.K || !d_region_rec || !d_pair_rec) { set_error("rag_copy_channel: invalid argument"); return GLIA_HMT_ERR_ARG; }
  GLIA_HIP_TRY(hipSetDevice(r->ctx->device));
  hipStream_t s = r->ctx->stream;
  if (r->arr.R) GLIA_HIP_TRY(hipMemcpyAsync(d_region_rec, r->arr.c_rrec[channel], sizeof(uint32_t) * (size_t)r->arr.R * kRegionWords, hipMemcpyDeviceToDevice, s));
  if (r->arr.P) GLIA_HIP_TRY(hipMemcpyAsync(d_pair_rec, r->arr.c_prec[channel], sizeof(uint32_t) * (size_t)r->arr.P * kPairWords, hipMemcpyDeviceToDevice, s));
  GLIA_HIP_TRY(hipStreamSynchronize(s));
  return GLIA_HMT_OK;
}

int glia_hmt_rag_add_channel(glia_hmt_ctx* c, glia_hmt_rag* r, const uint32_t* d_region_rec, const uint32_t* d_pair_rec) {
  if (!c || !r || r->ctx != c || !d_region_rec || !d_pair_rec || r->arr.K < 1 || r->arr.K >= kMaxChannels) { set_error("rag_add_channel: invalid argument"); return GLIA_HMT_ERR_ARG; }
  GLIA_HIP_TRY(hipSetDevice(c->device));
  const int k = r->arr.K;
  uint32_t *rr, *pr;
  GLIA_HIP_TRY(hipMalloc(&rr, sizeof(uint32_t) * kRegionWords * (size_t)(r->arr.R ? r->arr.R : 1)));
  GLIA_HIP_TRY(hipMalloc(&pr, sizeof(uint32_t) * kPairWords * (size_t)(r->arr.P ? r->arr.P : 1)));
  if (r->arr.R) GLIA_HIP_TRY(hipMemcpyAsync(rr, d_region_rec, sizeof(uint32_t) * (size_t)r->arr.R * kRegionWords, hipMemcpyDeviceToDevice, c->stream));
  if (r->arr.P) GLIA_HIP_TRY(hipMemcpyAsync(pr, d_pair_rec, sizeof(uint32_t) * (size_t)r->arr.P * kPairWords, hipMemcpyDeviceToDevice, c->stream));
  GLIA_HIP_TRY(hipStreamSynchronize(c->stream));
  r->arr.c_rrec[k] = rr; r->arr.c_prec[k] = pr; r->arr.K = k + 1;
  return GLIA_HMT_OK;
}

int glia_hmt_rag_cut_flags(glia_hmt_ctx* c, const glia_hmt_rag* rag, const int64_t dims_local[3], int64_t z_begin, int64_t z_end,
                           const uint32_t* d_labels, uint8_t* d_region_cut, uint8_t* d_pair_cut) {
  if (!c || !rag || !dims_local || !d_labels || !d_region_cut || !d_pair_cut || rag->ctx != c || z_begin < 0 || z_end > dims_local[2] || z_begin >= z_end) {
    set_error("rag_cut_flags: invalid argument");
    return GLIA_HMT_ERR_ARG;
  }
  GLIA_HIP_TRY(hipSetDevice(c->device));
  return rag_cut_flags(rag->arr, d_labels, dims_local[0], dims_local[1], dims_local[2], z_begin, z_end, d_region_cut, d_pair_cut, c->stream);
}

int glia_hmt_rag_device_arrays(const glia_hmt_rag* r, const uint32_t** d_region_label, const uint32_t** d_region_rec,
                               const uint32_t** d_pair_a, const uint32_t** d_pair_b, const uint32_t** d_pair_rec,
                               int* region_words, int* pair_words) {
  if (!r) return GLIA_HMT_ERR_ARG;
  if (d_region_label) *d_region_label = r->arr.d_rlabel;
  if (d_region_rec) *d_region_rec = r->arr.d_rrec;
  if (d_pair_a) *d_pair_a = r->arr.d_pa;
  if (d_pair_b) *d_pair_b = r->arr.d_pb;
  if (d_pair_rec) *d_pair_rec = r->arr.d_prec;
  if (region_words) *region_words = kRegionWords;
  if (pair_words) *pair_words = kPairWords;
  return GLIA_HMT_OK;
}

int glia_hmt_rag_copy_arrays(const glia_hmt_rag* r, uint32_t* d_region_label, uint32_t* d_region_rec, uint32_t* d_pair_a,
                             uint32_t* d_pair_b, uint32_t* d_pair_rec) {
  if (!r) return GLIA_HMT_ERR_ARG;
  GLIA_HIP_TRY(hipSetDevice(r->ctx->device));
  hipStream_t st = r->ctx->stream;
  const size_t R = (size_t)r->arr.R, P = (size_t)r->arr.P;
  if (d_region_label && R) GLIA_HIP_TRY(hipMemcpyAsync(d_region_label, r->arr.d_rlabel, 4 * R, hipMemcpyDeviceToDevice, st));
  if (d_region_rec && R) GLIA_HIP_TRY(hipMemcpyAsync(d_region_rec, r->arr.d_rrec, 4 * R * kRegionWords, hipMemcpyDeviceToDevice, st));
  if (d_pair_a && P) GLIA_HIP_TRY(hipMemcpyAsync(d_pair_a, r->arr.d_pa, 4 * P, hipMemcpyDeviceToDevice, st));
  if (d_pair_b && P) GLIA_HIP_TRY(hipMemcpyAsync(d_pair_b, r->arr.d_pb, 4 * P, hipMemcpyDeviceToDevice, st));
  if (d_pair_rec && P) GLIA_HIP_TRY(hipMemcpyAsync(d_pair_rec, r->arr.d_prec, 4 * P * kPairWords, hipMemcpyDeviceToDevice, st));
  GLIA_HIP_TRY(hipStreamSynchronize(st));
  return GLIA_HMT_OK;
}

int glia_hmt_rag_from_arrays(glia_hmt_ctx* c, const glia_hmt_rag* like, int64_t n_regions, const uint32_t* d_region_label,
                             const uint32_t* d_region_rec, int64_t n_pairs, const uint32_t* d_pair_a, const uint32_t* d_pair_b,
                             const uint32_t* d_pair_rec, glia_hmt_rag** out) {
  if (!c || !like || !out || n_regions < 0 || n_pairs < 0) { set_error("rag_from_arrays: invalid argument"); return GLIA_HMT_ERR_ARG; }
  GLIA_HIP_TRY(hipSetDevice(c->device));
  glia_hmt_rag* rag = new glia_hmt_rag(*like);
  rag->ctx = c;
  rag->d_folded = nullptr; rag->vol = VolumeRef();      // the copy owns neither the folded labels nor a whole volume
  rag->arr = RagArrays();
  rag->arr.R = n_regions; rag->arr.P = n_pairs;
  auto dup = [&](uint32_t** dst, const uint32_t* src, size_t n) -> int {
    GLIA_HIP_TRY(hipMalloc(dst, sizeof(uint32_t) * (n ? n : 1)));
    if (n) GLIA_HIP_TRY(hipMemcpyAsync(*dst, src, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, c->stream));
    return GLIA_HMT_OK;
  };
  int rc;
  if ((rc = dup(&rag->arr.d_rlabel, d_region_label, (size_t)n_regions)) || (rc = dup(&rag->arr.d_rrec, d_region_rec, (size_t)n_regions * kRegionWords)) ||
      (rc = dup(&rag->arr.d_pa, d_pair_a, (size_t)n_pairs)) || (rc = dup(&rag->arr.d_pb, d_pair_b, (size_t)n_pairs)) ||
      (rc = dup(&rag->arr.d_prec, d_pair_rec, (size_t)n_pairs * kPairWords))) { glia_hmt_rag_free(rag); return rc; }
  rag->arr.K = 1; rag->arr.c_rrec[0] = rag->arr.d_rrec; rag->arr.c_prec[0] = rag->arr.d_prec;
  for (int k = 0; k < kMaxChannels; ++k) rag->arr.c_bins[k] = like->arr.c_bins[k];      // (further channels: glia_hmt_rag_add_channel)
  rag->arr.c_bins[0] = rag->bins;
  GLIA_HIP_TRY(hipStreamSynchronize(c->stream));
  *out = rag;
  return GLIA_HMT_OK;
}

void glia_hmt_rag_free(glia_hmt_rag* r) {
  if (!r) return;
  (void)hipSetDevice(r->ctx->device);
  (void)hipFree(r->arr.d_rlabel); (void)hipFree(r->arr.d_rrec);
  (void)hipFree(r->arr.d_pa); (void)hipFree(r->arr.d_pb); (void)hipFree(r->arr.d_prec);
  for (int k = 1; k < r->arr.K; ++k) { (void)hipFree(r->arr.c_rrec[k]); (void)hipFree(r->arr.c_prec[k]); }
  if (r->arr.d_pv_off) (void)hipFree(r->arr.d_pv_off);
  if (r->arr.d_pv) (void)hipFree(r->arr.d_pv);
  if (r->d_folded) (void)hipFree(r->d_folded);
  delete r;
}

int64_t glia_hmt_rag_num_regions(const glia_hmt_rag* r) { return r ? r->arr.R : -1; }
int64_t glia_hmt_rag_num_pairs(const glia_hmt_rag* r) { return r ? r->arr.P : -1; }

int glia_hmt_rag_last_pass(const glia_hmt_rag* r, double* ms, double* bytes) {
  if (!r) return GLIA_HMT_ERR_ARG;
  if (ms) *ms = r->pass_ms;
  if (bytes) *bytes = r->alg_bytes;
  return GLIA_HMT_OK;
}

static double dword(const uint32_t* w) { double d; std::memcpy(&d, w, 8); return d; }

int glia_hmt_rag_export_regions(const glia_hmt_rag* r, uint32_t* h_label, int64_t* h_count, int64_t* h_border,
                                int64_t* h_lo, int64_t* h_hi, double* h_sum, double* h_sumsq, double* h_min,
                                double* h_max, int64_t* h_hist, int64_t* h_first) {
  if (!r) return GLIA_HMT_ERR_ARG;
  GLIA_HIP_TRY(hipSetDevice(r->ctx->device));
  const int64_t R = r->arr.R;
  std::vector<uint32_t> lab(R), rec((size_t)R * kRegionWords);
  if (R) {
    GLIA_HIP_TRY(hipMemcpy(lab.data(), r->arr.d_rlabel, sizeof(uint32_t) * R, hipMemcpyDeviceToHost));
    GLIA_HIP_TRY(hipMemcpy(rec.data(), r->arr.d_rrec, sizeof(uint32_t) * kRegionWords * R, hipMemcpyDeviceToHost));
  }
  for (int64_t i = 0; i < R; ++i) {
    const uint32_t* w = &rec[(size_t)i * kRegionWords];
    if (h_label) h_label[i] = lab[i];
    if (h_count) h_count[i] = w[R_CNT];
    if (h_border) h_border[i] = w[R_BORDER];
    for (int d = 0; d < 3; ++d) {
      if (h_lo) h_lo[3 * i + d] = (int64_t)(0x7fffffffu - w[R_LO + d]);
      if (h_hi) h_hi[3 * i + d] = (int64_t)w[R_HI + d] - 1;
    }
    if (h_sum) h_sum[i] = dword(&w[R_SUM]);
    if (h_sumsq) h_sumsq[i] = dword(&w[R_SQ]);
    if (h_min) h_min[i] = (double)ord_float(~w[R_MIN]);
    if (h_max) h_max[i] = (double)ord_float(w[R_MAX]);
    if (h_hist) for (int b = 0; b < r->bins; ++b) h_hist[i * r->bins + b] = w[R_HIST + b];
    if (h_first) { unsigned long long f; std::memcpy(&f, &w[R_FIRST], 8); h_first[i] = (int64_t)~f; }
  }
  return GLIA_HMT_OK;
}

int glia_hmt_rag_export_pairs(const glia_hmt_rag* r, uint32_t* h_a, uint32_t* h_b, int64_t* h_count, double* h_sum,
                              double* h_sumsq, double* h_min, double* h_max, int64_t* h_hist, int64_t* h_thr) {
  if (!r) return GLIA_HMT_ERR_ARG;
  GLIA_HIP_TRY(hipSetDevice(r->ctx->device));
  const int64_t P = r->arr.P;
  std::vector<uint32_t> a(P), b(P), rec((size_t)P * kPairWords);
  if (P) {
    GLIA_HIP_TRY(hipMemcpy(a.data(), r->arr.d_pa, sizeof(uint32_t) * P, hipMemcpyDeviceToHost));
    GLIA_HIP_TRY(hipMemcpy(b.data(), r->arr.d_pb, sizeof(uint32_t) * P, hipMemcpyDeviceToHost));
    GLIA_HIP_TRY(hipMemcpy(rec.data(), r->arr.d_prec, sizeof(uint32_t) * kPairWords * P, hipMemcpyDeviceToHost));
  }
  for (int64_t i = 0; i < P; ++i) {
    const uint32_t* w = &rec[(size_t)i * kPairWords];
    if (h_a) h_a[i] = a[i];
    if (h_b) h_b[i] = b[i];
    if (h_count) h_count[i] = w[P_CNT];
    if (h_sum) h_sum[i] = dword(&w[P_SUM]);
    if (h_sumsq) h_sumsq[i] = dword(&w[P_SQ]);
    if (h_min) h_min[i] = (double)ord_float(~w[P_MIN]);
    if (h_max) h_max[i] = (double)ord_float(w[P_MAX]);
    if (h_hist) for (int k = 0; k < r->bins; ++k) h_hist[i * r->bins + k] = w[P_HIST + k];
    if (h_thr) for (int k = 0; k < r->nthr; ++k) h_thr[i * r->nthr + k] = w[P_THR + k];
  }
  return GLIA_HMT_OK;
}

int glia_hmt_merge_order_pb(glia_hmt_ctx* c, glia_hmt_rag* rag, int type, uint32_t* h_order, double* h_sal,
                            int64_t capacity, int64_t* n_merges) {
  if (!c || !rag || !h_order || !h_sal || !n_merges || rag->ctx != c) {
    set_error("merge_order_pb: invalid argument");
    return GLIA_HMT_ERR_ARG;
  }
  if (type == 3 && rag->only_contour) {
    set_error("merge_order_pb: the median x min-size linkage needs region sizes (only_contour = 0)");
    return GLIA_HMT_ERR_ARG;
  }
  if ((type == 1 || type == 3) && !rag->vol.lab && !rag->arr.d_pv_off) {
    set_error("merge_order_pb: median linkage needs the volumes the RAG was built from (whole-volume build) or a map that carries its "
              "boundary values (glia_hmt_rag_build_distributed with with_values)");
    return GLIA_HMT_ERR_UNSUPPORTED;
  }
  if (type != 1 && type != 2 && type != 3) {   // hmt/main_merge_order_pb.cxx:36 (3: this library's name for ...AndMinSize)
    set_error("Error: unsupported boundary stats type...");
    return GLIA_HMT_ERR_ARG;
  }
  GLIA_HIP_TRY(hipSetDevice(c->device));
  const int64_t R = rag->arr.R, P = rag->arr.P;
  *n_merges = 0;
  if (R == 0 || P == 0) return GLIA_HMT_OK;
  std::vector<uint32_t> order((size_t)3 * R);
  std::vector<double> sal((size_t)R);
  int64_t n = 0;
  int rc = greedy_mean(rag->arr, c->stream, order.data(), sal.data(), R, &n, &rag->ms_table, &rag->ms_loop,
                       &rag->n_scored, 0, nullptr, 0.0, (type == 1 || type == 3) ? &rag->vol : nullptr, type == 3);
  if (rc) return rc;
  if (n > capacity) { set_error("merge_order_pb: output capacity too small"); return GLIA_HMT_ERR_CAPACITY; }
  // dense id -> key.  Leaves: i-th label ascending.  Merged regions: maxKey + 1 + k (util/struct_merge.hxx:19,27-31),
  // maxKey over the regions the reference's map holds: every label (point-map mode) or every label that owns a
  // directed boundary (contour-only mode, type/region_map.hxx:99-111).
  std::vector<uint32_t> lab((size_t)R);
  GLIA_HIP_TRY(hipMemcpy(lab.data(), rag->arr.d_rlabel, sizeof(uint32_t) * R, hipMemcpyDeviceToHost));
  uint32_t maxKey = lab[R - 1];
  if (rag->only_contour)
    GLIA_HIP_TRY(hipMemcpy(&maxKey, rag->arr.d_pa + (P - 1), sizeof(uint32_t), hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < 3 * n; ++i) {
    const uint32_t id = order[i];
    h_order[i] = id < (uint32_t)R ? lab[id] : maxKey + 1u + (id - (uint32_t)R);
  }
  for (int64_t i = 0; i < n; ++i) h_sal[i] = sal[i];
  *n_merges = n;
  return GLIA_HMT_OK;
}

static int upload_forest(const HostForest& hf, glia_hmt_forest* f, int slot, hipStream_t stream) {
  std::vector<PackedNode> nodes;
  std::vector<int> roots;
  int rc = pack_forest(hf, &nodes, &roots);
  if (rc) return rc;
  PackedNode* d_nodes = nullptr;
  int* d_roots = nullptr;
  GLIA_HIP_TRY(hipMalloc(&d_nodes, sizeof(PackedNode) * nodes.size()));
  f->allocs.push_back(d_nodes);
  GLIA_HIP_TRY(hipMalloc(&d_roots, sizeof(int) * roots.size()));
  f->allocs.push_back(d_roots);
  GLIA_HIP_TRY(hipMemcpyAsync(d_nodes, nodes.data(), sizeof(PackedNode) * nodes.size(), hipMemcpyHostToDevice, stream));
  GLIA_HIP_TRY(hipMemcpyAsync(d_roots, roots.data(), sizeof(int) * roots.size(), hipMemcpyHostToDevice, stream));
  std::vector<PackedTriple> lines;
  std::vector<int> proots;
  rc = pack_forest_triples(hf, &lines, &proots);
  if (rc) return rc;
  PackedTriple* d_lines = nullptr;
  int* d_proots = nullptr;
  GLIA_HIP_TRY(hipMalloc(&d_lines, sizeof(PackedTriple) * lines.size()));
  f->allocs.push_back(d_lines);
  GLIA_HIP_TRY(hipMalloc(&d_proots, sizeof(int) * proots.size()));
  f->allocs.push_back(d_proots);
  GLIA_HIP_TRY(hipMemcpyAsync(d_lines, lines.data(), sizeof(PackedTriple) * lines.size(), hipMemcpyHostToDevice, stream));
  GLIA_HIP_TRY(hipMemcpyAsync(d_proots, proots.data(), sizeof(int) * proots.size(), hipMemcpyHostToDevice, stream));
  GLIA_HIP_TRY(hipStreamSynchronize(stream));
  f->dc.f[slot].ntree = hf.ntree; f->dc.f[slot].nrnodes = hf.nrnodes; f->dc.f[slot].nnodes = (int)nodes.size();
  f->dc.f[slot].nodes = d_nodes; f->dc.f[slot].root = d_roots;
  f->dc.f[slot].triples = d_lines; f->dc.f[slot].troot = d_proots;
  if (hf.max_var > f->max_var) f->max_var = hf.max_var;
  return GLIA_HMT_OK;
}

int glia_hmt_forest_load(glia_hmt_ctx* c, int n_models, const char* const* paths, int predict_label,
                         const double* dist, glia_hmt_forest** out) {
  if (!c || !paths || !out || (n_models != 1 && n_models != 3)) {
    set_error("forest_load: one model, or three models with distributor arguments, expected");
    return GLIA_HMT_ERR_ARG;
  }
  if (n_models == 3 && !dist) { set_error("Error: model distributor needs 3 arguments..."); return GLIA_HMT_ERR_ARG; }
  GLIA_HIP_TRY(hipSetDevice(c->device));
  glia_hmt_forest* f = new glia_hmt_forest;
  f->device = c->device;
  memset(&f->dc, 0, sizeof(f->dc));
  f->dc.kind = 0; f->dc.n_models = n_models;
  if (dist) {
    f->dc.dim0 = (int)dist[0]; f->dc.dim1 = (int)dist[1]; f->dc.threshold = dist[2];
    if (f->dc.dim0 < 0 || f->dc.dim1 < 0) { delete f; set_error("forest_load: negative distributor dimension"); return GLIA_HMT_ERR_ARG; }
    f->max_var = std::max(f->dc.dim0, f->dc.dim1);
  }
  for (int i = 0; i < n_models; ++i) {
    HostForest hf;
    int rc = load_forest_file(paths[i], predict_label, &hf);
    if (rc == GLIA_HMT_OK) rc = upload_forest(hf, f, i, c->stream);
    if (rc) { glia_hmt_forest_free(f); return rc; }
  }
  *out = f;
  return GLIA_HMT_OK;
}

int glia_hmt_forest_file_parse(const char* path, int predict_label, int* ntree, int* nrnodes, int* nclass,
                               double* h_split, int* h_meta, int64_t capacity_nodes) {
  if (!path || !ntree || !nrnodes || !nclass) { set_error("forest_file_parse: invalid argument"); return GLIA_HMT_ERR_ARG; }
  HostForest hf;
  int rc = load_forest_file(path, predict_label, &hf);
  if (rc) return rc;
  *ntree = hf.ntree; *nrnodes = hf.nrnodes; *nclass = hf.nclass;
  const int64_t nodes = (int64_t)hf.ntree * hf.nrnodes;
  if (h_split || h_meta) {
    if (nodes > capacity_nodes) { set_error("forest_file_parse: capacity too small"); return GLIA_HMT_ERR_CAPACITY; }
    if (h_split) memcpy(h_split, hf.split.data(), sizeof(double) * nodes);
    if (h_meta) memcpy(h_meta, hf.meta.data(), sizeof(int) * 4 * nodes);
  }
  return GLIA_HMT_OK;
}

int glia_hmt_forest_stub(glia_hmt_ctx* c, int feature_index, glia_hmt_forest** out) {
  if (!c || !out || feature_index < 0) { set_error("forest_stub: invalid argument"); return GLIA_HMT_ERR_ARG; }
  glia_hmt_forest* f = new glia_hmt_forest;
  f->device = c->device;
  memset(&f->dc, 0, sizeof(f->dc));
  f->dc.kind = 1; f->dc.n_models = 1; f->dc.stub_index = feature_index; f->max_var = feature_index;
  *out = f;
  return GLIA_HMT_OK;
}

void glia_hmt_forest_free(glia_hmt_forest* f) {
  if (!f) return;
  (void)hipSetDevice(f->device);
  for (void* p : f->allocs) (void)hipFree(p);
  delete f;
}

static inline double sdivide_host(double l, double r) { return std::fabs(r) >= 2.22e-16 ? l / r : 0.0; }   // glia_base.hxx:77-78

static bool make_bc_cfg(const glia_hmt_rag* rag, BcCfg* c) {
  if (!rag->has_cfg) return false;
  const glia_hmt_feat_config& g = rag->cfg;
  memset(c, 0, sizeof(*c));
  c->D = rag->dim; c->T = g.n_thresholds; c->bins = rag->bins;
  c->K = rag->arr.K;
  for (int k = 0; k < kMaxChannels; ++k) c->cbins[k] = k < rag->arr.K ? rag->arr.c_bins[k] : 0;
  c->n_region = g.n_region; c->n_rlabel = g.n_rlabel; c->n_boundary = g.n_boundary;
  for (int i = 0; i < kMaxListed; ++i) { c->rc[i] = rag->map_region[i]; c->lc[i] = rag->map_rlabel[i]; c->bc[i] = rag->map_boundary[i]; }
  c->use_log = g.use_log_shape; c->use_simple = g.use_simple_features; c->use_hist = g.use_histogram_features;
  c->norm_area = g.normalizing_area; c->norm_len = g.normalizing_length;
  c->rfdim = bc_rf_dim(*c); c->bfdim = bc_bf_dim(*c); c->fdim = bc_feat_dim(*c);
  c->libm_log2 = rag->ctx->libm.log2_variant; c->libm_log = rag->ctx->libm.log_variant; c->libm_pow = rag->ctx->libm.pow_variant;
  return true;
}

// GLIA_USE_MEDIAN_AS_FEATS (type/feat.hxx:677-722, 772-808; hmt/bc_feat.hxx:252-268): one more column per real-feature block -- the
// diff block of every region-list image, the shared-boundary block of every boundary-list image, and both kinds of block in each of
// the three region blocks; the simple selection carries the shared boundary's median beside its mean.
static int median_extra_cols(const glia_hmt_rag* rag, const BcCfg& c) {
  if (!rag->cfg.use_median_features) return 0;
  return c.use_simple ? c.n_boundary : 4 * c.n_region + 4 * c.n_boundary;
}
// the greedy loop keeps mergeable statistics, not value multisets: with this layout only bc_feat (a given order) is implemented
static int refuse_median_in_loop(const glia_hmt_rag* rag, const char* who) {
  if (!rag->has_cfg || !rag->cfg.use_median_features) return GLIA_HMT_OK;
  set_error(std::string(who) + ": the GLIA_USE_MEDIAN_AS_FEATS layout (type/feat.hxx:677-722) is implemented for a given merge order only (glia_hmt_bc_feat): "
            "inside the greedy loop every contraction changes the value multisets of its regions and boundary sets");
  return GLIA_HMT_ERR_UNSUPPORTED;
}

int glia_hmt_feat_dim(const glia_hmt_rag* rag) {
  BcCfg c;
  if (!rag || !make_bc_cfg(rag, &c)) return -1;
  return c.fdim + median_extra_cols(rag, c);
}

int glia_hmt_merge_order_bc(glia_hmt_ctx* c, glia_hmt_rag* rag, const glia_hmt_forest* forest, uint32_t* h_order,
                            double* h_sal, double* h_feats, int64_t capacity, int64_t* n_merges) {
  if (!c || !rag || !forest || !h_order || !h_sal || !n_merges || rag->ctx != c) {
    set_error("merge_order_bc: invalid argument");
    return GLIA_HMT_ERR_ARG;
  }
  BcCfg cfg;
  if (!make_bc_cfg(rag, &cfg) || rag->only_contour) {
    set_error("merge_order_bc: the region map must be built with a feature configuration and with region points");
    return GLIA_HMT_ERR_ARG;
  }
  if (int rm = refuse_median_in_loop(rag, "merge_order_bc")) return rm;
  if (forest->max_var >= cfg.fdim) { set_error("merge_order_bc: the classifier reads features beyond the vector"); return GLIA_HMT_ERR_ARG; }
  GLIA_HIP_TRY(hipSetDevice(c->device));
  const int64_t R = rag->arr.R;
  *n_merges = 0;
  if (R == 0 || rag->arr.P == 0) return GLIA_HMT_OK;
  std::vector<uint32_t> order((size_t)3 * R);
  std::vector<double> sal((size_t)R);
  std::vector<double> feats;
  if (h_feats) feats.resize((size_t)R * cfg.fdim);
  int64_t n = 0;
  int rc = greedy_bc(rag->arr, cfg, forest->dc, c->stream, order.data(), sal.data(), h_feats ? feats.data() : nullptr, R, &n,
                     &rag->ms_table, &rag->ms_init, &rag->ms_loop, &rag->n_scored, false);
  if (rc) return rc;
  if (n > capacity) { set_error("merge_order_bc: output capacity too small"); return GLIA_HMT_ERR_CAPACITY; }
  std::vector<uint32_t> lab((size_t)R);
  GLIA_HIP_TRY(hipMemcpy(lab.data(), rag->arr.d_rlabel, sizeof(uint32_t) * R, hipMemcpyDeviceToHost));
  const uint32_t maxKey = lab[R - 1];
  for (int64_t i = 0; i < 3 * n; ++i) {
    const uint32_t id = order[i];
    h_order[i] = id < (uint32_t)R ? lab[id] : maxKey + 1u + (id - (uint32_t)R);
  }
  for (int64_t i = 0; i < n; ++i) h_sal[i] = sal[i];
  if (h_feats) memcpy(h_feats, feats.data(), sizeof(double) * (size_t)n * cfg.fdim);
  *n_merges = n;
  return GLIA_HMT_OK;
}

int glia_hmt_pre_merge(glia_hmt_ctx* c, glia_hmt_rag* rag, const int* size_thresholds, int n_thresholds,
                       double rpb_threshold, uint32_t* h_order, double* h_sal, int64_t capacity, int64_t* n_merges) {
  if (!c || !rag || !size_thresholds || n_thresholds < 1 || n_thresholds > 2 || !h_order || !h_sal || !n_merges || rag->ctx != c) {
    set_error("pre_merge: invalid argument");
    return GLIA_HMT_ERR_ARG;
  }
  if (rag->only_contour) { set_error("pre_merge: the region map must hold region points (only_contour = 0)"); return GLIA_HMT_ERR_ARG; }
  GLIA_HIP_TRY(hipSetDevice(c->device));
  const int64_t R = rag->arr.R, P = rag->arr.P;
  *n_merges = 0;
  if (R == 0 || P == 0) return GLIA_HMT_OK;
  std::vector<uint32_t> order((size_t)3 * R);
  std::vector<double> sal((size_t)R);
  long long sizes[2] = {size_thresholds[0], n_thresholds > 1 ? size_thresholds[1] : 0};
  int64_t n = 0;
  int rc = greedy_mean(rag->arr, c->stream, order.data(), sal.data(), R, &n, &rag->ms_table, &rag->ms_loop, &rag->n_scored,
                       n_thresholds, sizes, rpb_threshold);
  if (rc) return rc;
  if (n > capacity) { set_error("pre_merge: output capacity too small"); return GLIA_HMT_ERR_CAPACITY; }
  std::vector<uint32_t> lab((size_t)R);
  GLIA_HIP_TRY(hipMemcpy(lab.data(), rag->arr.d_rlabel, sizeof(uint32_t) * R, hipMemcpyDeviceToHost));
  const uint32_t maxKey = lab[R - 1];
  for (int64_t i = 0; i < 3 * n; ++i) {
    const uint32_t id = order[i];
    h_order[i] = id < (uint32_t)R ? lab[id] : maxKey + 1u + (id - (uint32_t)R);
  }
  for (int64_t i = 0; i < n; ++i) h_sal[i] = sal[i];
  *n_merges = n;
  return GLIA_HMT_OK;
}

int glia_hmt_bc_feat(glia_hmt_ctx* c, glia_hmt_rag* rag, const uint32_t* h_order, int64_t n_merges, double* h_feats) {
  return glia_hmt_bc_feat_saliency(c, rag, h_order, n_merges, nullptr, 1.0, 1.0, h_feats);
}

int glia_hmt_bc_feat_dim(const glia_hmt_rag* rag, int with_saliency) {
  BcCfg cfg;
  if (!rag || !make_bc_cfg(rag, &cfg)) return -1;
  return cfg.fdim + median_extra_cols(rag, cfg) + ((with_saliency && !cfg.use_simple) ? 5 : 0);
}

int glia_hmt_bc_feat_saliency(glia_hmt_ctx* c, glia_hmt_rag* rag, const uint32_t* h_order, int64_t n_merges,
                              const double* h_saliencies, double init_saliency, double saliency_bias, double* h_feats) {
  if (!c || !rag || !h_order || !h_feats || n_merges < 0 || rag->ctx != c) { set_error("bc_feat: invalid argument"); return GLIA_HMT_ERR_ARG; }
  BcCfg cfg;
  if (!make_bc_cfg(rag, &cfg) || rag->only_contour) {
    set_error("bc_feat: the region map must be built with a feature configuration and with region points");
    return GLIA_HMT_ERR_ARG;
  }
  GLIA_HIP_TRY(hipSetDevice(c->device));
  const int64_t R = rag->arr.R;
  if (n_merges == 0) return GLIA_HMT_OK;
  if (n_merges >= R) { set_error("bc_feat: more merges than regions"); return GLIA_HMT_ERR_ARG; }
  // keys -> dense ids: leaves by label, the region created by merge i is R + i (RegionMap(seg, mask, order, false),
  // type/region_map.hxx:42-48,67-68)
  std::vector<uint32_t> lab((size_t)R);
  GLIA_HIP_TRY(hipMemcpy(lab.data(), rag->arr.d_rlabel, sizeof(uint32_t) * R, hipMemcpyDeviceToHost));
  std::unordered_map<uint32_t, uint32_t> id;
  id.reserve((size_t)R * 2);
  for (int64_t i = 0; i < R; ++i) id[lab[i]] = (uint32_t)i;
  std::vector<uint32_t> forced((size_t)2 * n_merges);
  std::vector<uint8_t> used((size_t)R + n_merges, 0);
  for (int64_t i = 0; i < n_merges; ++i) {
    for (int s2 = 0; s2 < 2; ++s2) {
      auto it = id.find(h_order[3 * i + s2]);
      if (it == id.end() || used[it->second]) { set_error("bc_feat: merge order refers to an unknown or already merged region"); return GLIA_HMT_ERR_ARG; }
      used[it->second] = 1;
      forced[2 * i + s2] = it->second;
    }
    if (id.count(h_order[3 * i + 2])) { set_error("bc_feat: merge order reuses a region key"); return GLIA_HMT_ERR_ARG; }
    id[h_order[3 * i + 2]] = (uint32_t)(R + i);
  }
  std::vector<uint32_t> order((size_t)3 * R);
  std::vector<double> sal((size_t)R), feats((size_t)R * cfg.fdim);
  DeviceClassifier none;
  memset(&none, 0, sizeof(none));
  none.kind = 1;
  int64_t n = 0;
  int rc = greedy_bc(rag->arr, cfg, none, c->stream, order.data(), sal.data(), feats.data(), R, &n, &rag->ms_table, &rag->ms_init,
                     &rag->ms_loop, &rag->n_scored, false, forced.data(), n_merges);
  if (rc) return rc;
  if (n != n_merges) { set_error("bc_feat: internal error, merges not completed"); return GLIA_HMT_ERR_HIP; }
  int bfdim = cfg.bfdim, rfdim = cfg.rfdim, fdim = cfg.fdim;
  if (rag->cfg.use_median_features) {
    // GLIA_USE_MEDIAN_AS_FEATS: the kernel's rows (built from the statistics monoids) + the medians, means and standard deviations
    // of the value multisets (median_feats.hip), spliced into the reference's layout
    const glia_hmt_feat_config& g = rag->cfg;
    MedianFeatIn in;
    memset(&in, 0, sizeof(in));
    in.rag = &rag->arr; in.vol = rag->vol; in.forced = forced.data(); in.n_merges = n_merges;
    in.n_r = g.n_region; in.n_b = g.n_boundary;
    for (int i = 0; i < g.n_region; ++i) in.r_img[i] = (const float*)g.region[i].d_image;
    for (int i = 0; i < g.n_boundary; ++i) in.b_img[i] = (const float*)g.boundary[i].d_image;
    std::vector<double> reg, bnd;
    std::vector<unsigned long long> marea;
    if ((rc = median_feature_stats(in, c->stream, &reg, &bnd, &marea))) return rc;
    const int nr = g.n_region, nl = g.n_rlabel, nb = g.n_boundary, T = cfg.T, D = cfg.D;
    const int extra = median_extra_cols(rag, cfg);
    const int nd = cfg.fdim + extra;
    std::vector<double> rows((size_t)n * nd);
    auto hb = [&](int kind, int i) { return cfg.use_hist ? cfg.cbins[kind == 0 ? cfg.rc[i] : kind == 1 ? cfg.lc[i] : cfg.bc[i]] : 0; };
    for (int64_t i = 0; i < n; ++i) {
      const bool swap = sdivide_host((double)marea[forced[2 * i]], cfg.norm_area) > sdivide_host((double)marea[forced[2 * i + 1]], cfg.norm_area);
      // x1 = the smaller region (main_bc_feat.cxx:84-88): k of the statistics arrays for block b of the row
      const int kreg[3] = {swap ? 1 : 0, swap ? 0 : 1, 2};
      auto RS = [&](int blk, int img, int q) { return reg[(((size_t)i * 3 + kreg[blk]) * nr + img) * 3 + q]; };
      auto BS = [&](int blk, int img, int q) { return bnd[(((size_t)i * 4 + (blk < 3 ? kreg[blk] : 3)) * nb + img) * 3 + q]; };
      const double* in_row = &feats[(size_t)i * cfg.fdim];
      double* out = &rows[(size_t)i * nd];
      int p = 0, k = 0;
      if (cfg.use_simple) {                                         // hmt/bc_feat.hxx:247-279
        for (int q = 0; q < 5; ++q) out[k++] = in_row[p++];
        for (int j = 0; j < nb; ++j) { out[k++] = BS(3, j, 1); out[k++] = BS(3, j, 0); ++p; }
        for (int j = 0; j < nr; ++j) { out[k++] = std::fabs(RS(0, j, 1) - RS(1, j, 1)); out[k++] = in_row[p + 1]; out[k++] = in_row[p + 2]; out[k++] = in_row[p + 3]; p += 4; }
        for (int j = 0; j < 2 * nl; ++j) out[k++] = in_row[p++];
      } else {
        for (int q = 0; q < 11 + 4 * T; ++q) out[k++] = in_row[p++];
        for (int j = 0; j < nr; ++j) {                               // feat.hxx:782-808: l1, x2, |d entropy|, |d median|, |d mean|, |d std|, |d min|, |d max|
          out[k++] = in_row[p]; out[k++] = in_row[p + 1]; out[k++] = in_row[p + 2];
          out[k++] = std::fabs(RS(0, j, 0) - RS(1, j, 0)); out[k++] = std::fabs(RS(0, j, 1) - RS(1, j, 1)); out[k++] = std::fabs(RS(0, j, 2) - RS(1, j, 2));
          out[k++] = in_row[p + 5]; out[k++] = in_row[p + 6];
          p += 7;
        }
        for (int q = 0; q < 3 * nl; ++q) out[k++] = in_row[p++];
        auto real_block = [&](int h, double med, double mean, double sd) {     // [histogram] entropy | median mean std | min max
          for (int q = 0; q < h + 1; ++q) out[k++] = in_row[p++];
          out[k++] = med; out[k++] = mean; out[k++] = sd;
          out[k++] = in_row[p + 2]; out[k++] = in_row[p + 3];
          p += 4;
        };
        for (int j = 0; j < nb; ++j) real_block(hb(2, j), BS(3, j, 0), BS(3, j, 1), BS(3, j, 2));
        for (int blk = 0; blk < 3; ++blk) {
          for (int q = 0; q < 4 + D + 2 * T; ++q) out[k++] = in_row[p++];
          for (int j = 0; j < nr; ++j) real_block(hb(0, j), RS(blk, j, 0), RS(blk, j, 1), RS(blk, j, 2));
          for (int j = 0; j < nl; ++j) for (int q = 0; q < hb(1, j) + 1; ++q) out[k++] = in_row[p++];
          for (int j = 0; j < nb; ++j) real_block(hb(2, j), BS(blk, j, 0), BS(blk, j, 1), BS(blk, j, 2));
        }
      }
      if (p != cfg.fdim || k != nd) { set_error("bc_feat: internal error, median feature layout"); return GLIA_HMT_ERR_INTERNAL; }
    }
    feats.swap(rows);
    fdim = nd;
    if (!cfg.use_simple) { bfdim = cfg.bfdim + nr + nb; rfdim = cfg.rfdim + nr + nb; }
  }
  if (!h_saliencies || cfg.use_simple) {       // selectFeatures carries no saliency (hmt/bc_feat.hxx:247-279)
    memcpy(h_feats, feats.data(), sizeof(double) * (size_t)n * fdim);
    return GLIA_HMT_OK;
  }
  // Saliency features (hmt/main_bc_feat.cxx:50-55): every region of the order carries a number -- initSal for the
  // regions that are only merged, saliency + bias for the region a merge creates (genSaliencyMap, hmt/bc_feat.hxx:12-26).
  // They do not depend on the image: each region block gets its number appended (bc_feat.hxx:76), the boundary block
  // (min, max) of |s(x1) - s(x3)|, |s(x2) - s(x3)| (:163-166, :208-213).  Which region is x1 follows the area rule of
  // main_bc_feat.cxx:88-91, for which the voxel counts of the tree nodes are summed up here.
  std::vector<uint32_t> rrec((size_t)R * kRegionWords);
  GLIA_HIP_TRY(hipMemcpy(rrec.data(), rag->arr.d_rrec, sizeof(uint32_t) * kRegionWords * R, hipMemcpyDeviceToHost));
  std::vector<unsigned long long> area((size_t)R + n_merges);
  for (int64_t i = 0; i < R; ++i) area[i] = rrec[(size_t)i * kRegionWords + R_CNT];
  std::unordered_map<uint32_t, double> smap;
  for (int64_t i = 0; i < n_merges; ++i) {
    if (!smap.count(h_order[3 * i])) smap[h_order[3 * i]] = init_saliency;
    if (!smap.count(h_order[3 * i + 1])) smap[h_order[3 * i + 1]] = init_saliency;
    smap[h_order[3 * i + 2]] = h_saliencies[i] + saliency_bias;
  }
  const int od = fdim + 5;
  for (int64_t i = 0; i < n_merges; ++i) {
    const uint32_t a = forced[2 * i], b = forced[2 * i + 1];
    area[R + i] = area[a] + area[b];
    const bool swap = sdivide_host((double)area[a], cfg.norm_area) > sdivide_host((double)area[b], cfg.norm_area);
    const double s0 = smap[h_order[3 * i]], s1 = smap[h_order[3 * i + 1]], s2 = smap[h_order[3 * i + 2]];
    const double sx1 = swap ? s1 : s0, sx2 = swap ? s0 : s1;
    const double d02 = std::fabs(sx1 - s2), d12 = std::fabs(sx2 - s2);
    const double* in = &feats[(size_t)i * fdim];
    double* out = &h_feats[(size_t)i * od];
    int k = 0;
    for (int q = 0; q < bfdim; ++q) out[k++] = in[q];
    out[k++] = std::min(d02, d12); out[k++] = std::max(d02, d12);
    for (int q = 0; q < rfdim; ++q) out[k++] = in[bfdim + q];
    out[k++] = sx1;
    for (int q = 0; q < rfdim; ++q) out[k++] = in[bfdim + rfdim + q];
    out[k++] = sx2;
    for (int q = 0; q < rfdim; ++q) out[k++] = in[bfdim + 2 * rfdim + q];
    out[k++] = s2;
  }
  return GLIA_HMT_OK;
}

// hmt::genTree (hmt/tree_build.hxx:12-38): order -> array tree, children before parents
int64_t glia_hmt_transform_keys(const uint32_t* h_order, int64_t n_merges, uint32_t* h_src, uint32_t* h_dst, int64_t capacity) {
  if (!h_order || !h_src || !h_dst || n_merges < 0) { set_error("transform_keys: invalid argument"); return GLIA_HMT_ERR_ARG; }
  std::vector<uint32_t> s, d;
  int rc = transform_keys(h_order, n_merges, &s, &d);
  if (rc) return rc;
  if ((int64_t)s.size() > capacity) { set_error("transform_keys: output capacity too small"); return GLIA_HMT_ERR_CAPACITY; }
  std::copy(s.begin(), s.end(), h_src);
  std::copy(d.begin(), d.end(), h_dst);
  return (int64_t)s.size();
}

int glia_hmt_transform_image(glia_hmt_ctx* c, uint32_t* d_labels, int64_t n_voxels, const uint32_t* h_src, const uint32_t* h_dst,
                             int64_t n_map, const uint32_t* d_mask, int fill_missing) {
  if (!c || !d_labels || n_voxels < 0 || n_map < 0 || (n_map && (!h_src || !h_dst))) { set_error("transform_image: invalid argument"); return GLIA_HMT_ERR_ARG; }
  GLIA_HIP_TRY(hipSetDevice(c->device));
  return transform_image(d_labels, n_voxels, h_src, h_dst, n_map, d_mask, fill_missing, c->stream, &c->transform_ms);
}

int glia_hmt_relabel_image(glia_hmt_ctx* c, uint32_t* d_labels, int64_t n_voxels, int64_t min_size, uint32_t* n_labels) {
  if (!c || !d_labels || n_voxels < 0 || !n_labels) { set_error("relabel_image: invalid argument"); return GLIA_HMT_ERR_ARG; }
  GLIA_HIP_TRY(hipSetDevice(c->device));
  return relabel_image(d_labels, n_voxels, min_size, n_labels, c->stream);
}

int glia_hmt_boundary_confidence(glia_hmt_ctx* c, glia_hmt_rag* rag, int n_trees, const int64_t* n_nodes, const uint32_t* const* node_label,
                                 const int32_t* const* parent, const int32_t* const* child0, const double* const* potential, float* d_out) {
  if (!c || !rag || rag->ctx != c || n_trees < 1 || !n_nodes || !node_label || !parent || !child0 || !potential || !d_out) {
    set_error("boundary_confidence: invalid argument");
    return GLIA_HMT_ERR_ARG;
  }
  if (!rag->vol.lab) { set_error("boundary_confidence: needs the volumes the region map was built from (whole-volume build)"); return GLIA_HMT_ERR_UNSUPPORTED; }
  GLIA_HIP_TRY(hipSetDevice(c->device));
  const int64_t P = rag->arr.P;
  std::vector<uint32_t> pa((size_t)(P ? P : 1)), pb((size_t)(P ? P : 1));
  if (P) {
    GLIA_HIP_TRY(hipMemcpy(pa.data(), rag->arr.d_pa, sizeof(uint32_t) * P, hipMemcpyDeviceToHost));
    GLIA_HIP_TRY(hipMemcpy(pb.data(), rag->arr.d_pb, sizeof(uint32_t) * P, hipMemcpyDeviceToHost));
  }
  std::vector<float> val;
  int rc = boundary_confidence_values(n_trees, n_nodes, node_label, parent, child0, potential, pa.data(), pb.data(), P, &val);
  if (rc) return rc;
  return paint_pair_values(rag->vol, rag->arr.d_pa, rag->arr.d_pb, P, val.data(), d_out, c->stream);
}

double glia_hmt_last_transform_ms(const glia_hmt_ctx* c) { return c ? c->transform_ms : 0.0; }

int64_t glia_hmt_gen_tree(const uint32_t* h_order, int64_t n_merges, uint32_t* node_label, int32_t* parent, int32_t* child0,
                          int32_t* child1, int64_t capacity) {
  if (!h_order || !node_label || !parent || !child0 || !child1 || n_merges < 0) { set_error("gen_tree: invalid argument"); return GLIA_HMT_ERR_ARG; }
  std::unordered_map<uint32_t, int> nmap;
  int64_t ni = 0;
  auto add = [&](uint32_t label, int c0, int c1) -> int64_t {
    if (ni >= capacity) return -1;
    node_label[ni] = label; parent[ni] = -1; child0[ni] = c0; child1[ni] = c1;
    nmap.emplace(label, (int)ni);
    return ni++;
  };
  for (int64_t i = 0; i < n_merges; ++i) {
    const uint32_t x0 = h_order[3 * i], x1 = h_order[3 * i + 1], x2 = h_order[3 * i + 2];
    auto n0 = nmap.find(x0);
    int64_t i0 = n0 == nmap.end() ? add(x0, -1, -1) : n0->second;
    auto n1 = nmap.find(x1);
    int64_t i1 = n1 == nmap.end() ? add(x1, -1, -1) : n1->second;
    if (i0 < 0 || i1 < 0 || ni >= capacity) { set_error("gen_tree: output capacity too small"); return GLIA_HMT_ERR_CAPACITY; }
    parent[i0] = (int32_t)ni; parent[i1] = (int32_t)ni;
    add(x2, (int)i0, (int)i1);
  }
  return ni;
}

int glia_hmt_score_initial_edges(glia_hmt_ctx* c, glia_hmt_rag* rag, const glia_hmt_forest* forest, int64_t* n_edges,
                                 double* ms) {
  if (!c || !rag || !forest || rag->ctx != c) { set_error("score_initial_edges: invalid argument"); return GLIA_HMT_ERR_ARG; }
  BcCfg cfg;
  if (!make_bc_cfg(rag, &cfg) || rag->only_contour) {
    set_error("score_initial_edges: the region map must be built with a feature configuration and with region points");
    return GLIA_HMT_ERR_ARG;
  }
  if (int rm = refuse_median_in_loop(rag, "score_initial_edges")) return rm;
  if (forest->max_var >= cfg.fdim) { set_error("score_initial_edges: the classifier reads features beyond the vector"); return GLIA_HMT_ERR_ARG; }
  GLIA_HIP_TRY(hipSetDevice(c->device));
  int64_t n = 0;
  int rc = greedy_bc(rag->arr, cfg, forest->dc, c->stream, nullptr, nullptr, nullptr, 0, &n, &rag->ms_table, &rag->ms_init,
                     &rag->ms_loop, &rag->n_scored, true);
  if (rc) return rc;
  if (n_edges) *n_edges = rag->n_scored;
  if (ms) *ms = rag->ms_init;
  return GLIA_HMT_OK;
}

int glia_hmt_score_initial_edges_shard(glia_hmt_ctx* c, glia_hmt_rag* rag, const glia_hmt_forest* forest, int shard, int n_shards,
                                       double* h_scores, int64_t capacity, int64_t* n_records) {
  if (!c || !rag || !forest || rag->ctx != c || n_shards < 1 || shard < 0 || shard >= n_shards || !n_records) {
    set_error("score_initial_edges_shard: invalid argument");
    return GLIA_HMT_ERR_ARG;
  }
  BcCfg cfg;
  if (!make_bc_cfg(rag, &cfg) || rag->only_contour) {
    set_error("score_initial_edges: the region map must be built with a feature configuration and with region points");
    return GLIA_HMT_ERR_ARG;
  }
  if (int rm = refuse_median_in_loop(rag, "score_initial_edges")) return rm;
  if (forest->max_var >= cfg.fdim) { set_error("score_initial_edges: the classifier reads features beyond the vector"); return GLIA_HMT_ERR_ARG; }
  GLIA_HIP_TRY(hipSetDevice(c->device));
  // one record per unordered leaf pair; the count is needed before the scores can be copied out
  std::vector<double> scores;
  int64_t n = 0;
  const int64_t upper = rag->arr.P;            // records <= directed pairs
  scores.assign((size_t)(upper ? upper : 1), 0.0);
  int rc = greedy_bc(rag->arr, cfg, forest->dc, c->stream, nullptr, nullptr, nullptr, 0, &n, &rag->ms_table, &rag->ms_init,
                     &rag->ms_loop, &rag->n_scored, true, nullptr, 0, shard, n_shards, scores.data());
  if (rc) return rc;
  *n_records = n;
  if (h_scores) {
    if (n > capacity) { set_error("score_initial_edges_shard: output capacity too small"); return GLIA_HMT_ERR_CAPACITY; }
    memcpy(h_scores, scores.data(), sizeof(double) * (size_t)n);
  }
  return GLIA_HMT_OK;
}

int glia_hmt_last_merge_timing(const glia_hmt_rag* r, double* ms_table, double* ms_init, double* ms_loop,
                               int64_t* n_scored) {
  if (!r) return GLIA_HMT_ERR_ARG;
  if (ms_table) *ms_table = r->ms_table;
  if (ms_init) *ms_init = r->ms_init;
  if (ms_loop) *ms_loop = r->ms_loop;
  if (n_scored) *n_scored = r->n_scored;
  return GLIA_HMT_OK;
}

}  // extern "C"
