// glia_amd/csrc/greedy_bc.hip -- K5/K6/K7 for the classifier linkage: greedy agglomeration where every new
// edge gets a feature vector and a boundary-classifier score, on the device.
//
// Reference semantics reproduced (all under /root/reference/code/):
//   * genMergeOrderGreedyUsingBoundaryClassifier (util/struct_merge_bc.hxx:10-43): saliency of an edge =
//     classifier(features(r0, r1, r0 u r1, shared boundary)); regions ARE merged (updateRegion = true).
//   * TRegion::merge (type/region.hxx:66-75): a merged region's boundary set is the union of its leaves' DIRECTED
//     boundary entries minus every MUTUAL pair (a->b, b->a) whose two leaves are both inside; a non-mutual
//     entry is never cancelled.  So  B(R) = Bn(R) + Bm(R):  Bn = all non-mutual out-entries of R's leaves (only
//     ever grows), Bm = mutual entries to leaves outside R (lives on the table edges of R).
//   * getBoundary / boundaryWith (util/struct.hxx:10-16, type/region.hxx:42-51): the shared boundary of R0,R1 =
//     entries (a->x) of R0 whose target leaf x is in R1 AND still owns an un-cancelled entry there, plus the
//     symmetric set.  A mutual entry's target always qualifies; a non-mutual entry's target x qualifies iff x
//     has a non-mutual out-entry of its own or a mutual partner outside its region ("alive").
//   * TBoundaryTable (type/boundary_table.hxx): only MUTUAL leaf pairs start as table edges (:99-102); update
//     creates (rs,r2) iff (r0,rs) or (r1,rs) is a table edge (:127-156).  Directed-only adjacencies are kept
//     here as non-table records so that their entries are found when a later contraction makes the pair a
//     table edge.  Tie rule / visit order: see greedy.hip.
//   * feature vector and orientation: bc_features.hpp; initial edges are passed to fBcFeat in the orientation
//     in which TBoundaryTable::init meets them, i.e. the region that comes first in the unordered_map
//     iteration order of the region map (rank[] is computed on the host by replaying the reference's insertion
//     sequences through libstdc++'s hashtable rules, rmap_order.cpp); updated edges are passed as (rs, r2).
//
// MI355X mapping: one persistent workgroup (the contraction chain is sequential); per contraction the new
// records are built data-parallel, features are computed one thread per new edge into a workspace, and the
// forest is evaluated with (edge, tree) pairs spread over the whole workgroup.
#include <algorithm>
#include <chrono>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "bc_features.hpp"
#include "forest.hpp"
#include "greedy_common.hpp"
#include "rmap_order.hpp"

namespace glia {

#ifndef GLIA_BC_THREADS
#define GLIA_BC_THREADS 512
#endif
constexpr int kBcThreads = GLIA_BC_THREADS;   // 512: 256 VGPRs per thread (1024 spilled into scratch, measured 8% slower); >= 512 needed: the feature blocks of a 96-record chunk take 8 waves
#ifndef GLIA_BC_CHUNK
#define GLIA_BC_CHUNK 96
#endif
constexpr int kChunk = GLIA_BC_CHUNK;              // new edges scored per round (their feature vectors live in LDS)

// statistics of one image channel (a distinct volume + histogram among the feature lists); channel 0 = boundary
// probability: its counts and thresholded counts also serve the shape features
struct BcChan {
  // regions [2*R0]
  PStats* pts;
  EStats* Bn;          // non-mutual out-entries of the region's leaves
  EStats* Bt;          // additive totals of the whole boundary set (min/max fields unused)
  float* Bmn; float* Bmx;   // min / max over the whole boundary set
  // histogram entropies of pts[r] and Bt[r] (both fixed for the lifetime of region r): the leaf kernel files them for the
  // leaves, the scoring pass of the contraction that creates r for merged regions; a record's neighbour-side entropies are
  // then two loads instead of two passes over the bins (a division and a logarithm per bin)
  double* entP; double* entB;
  // records [Ecap]
  EStats* e_A;         // mutual entries, both directions
  EStats* e_NA;        // always-alive non-mutual entries, both directions
  float* e_dir;        // [Ecap][4]  mutual u->v min,max ; v->u min,max
  float2* pool_dir;    // [pool_cap] parallel to the incident lists: (min, max) the list's owner sends along that entry's
                       // record, (+inf, -inf) for dead entries -- the "all but one record" scans read it contiguously
  // leaf entries [P]
  EStats* le_stats;
};

struct BcState {
  uint32_t R0;
  BcChan ch[kMaxChannels];
  uint32_t* parent;    // merge forest (find -> current region of a leaf)
  uint32_t* adj_off; uint32_t* adj_len; uint32_t* pool; unsigned long long pool_cap;
  // records [Ecap]
  uint32_t Ecap;
  uint32_t *e_u, *e_v, *e_posu, *e_posv;
  uint8_t *e_alive, *e_table, *e_orient;    // orient: 1 = features take (u, v), 0 = (v, u)
  uint32_t *e_fhead, *e_ftail;   // fragile non-mutual leaf entries (linked through le_next), kNone = empty
  PqTree pq;
  // leaf entries [P] (directed label pairs, ascending (a,b))
  long long P;
  uint32_t *le_src, *le_dst;     // dense leaf ids
  uint32_t* le_next;
  uint8_t* le_mutual;
  uint32_t* le_start;            // [R0+1] first out-entry of each leaf
  uint32_t* nm_out;              // [R0] non-mutual out-entries per leaf
  uint32_t *mark0, *mark1;       // [2*R0]
  uint32_t* order; double* sal_out; double* feats_out;
  // Helper workgroups score WHOLE records (round 3): the loop's workgroup publishes a job -- "records ne0 .. ne0 + cnt of the region
  // r2 just created" -- and helper h takes the records h, h + H, ...: it stages everything a record's vector needs with agent-scope
  // loads, computes neighbour extremes, shared-boundary set, entropies, the vector and the forest's votes, and answers with one
  // 64-bit word per record.  Everything a helper reads that this launch writes is stored write-through (st_agent) by the loop.
  // What the loop's workgroup re-reads itself (the records' own fields and statistics, list headers) it keeps in plain stores --
  // a write-through store drops the line from its XCD's L2 and every later read of it went to memory (measured: +10 k cycles
  // per contraction) -- and hands the helpers a COPY: one packed 256-byte row per new record and channel (hrec), hadj for the
  // list headers.  Region statistics, set extremes, entropies, pool_dir and the merge forest are stored write-through only.
  uint32_t* hctl;                // [kFlagReps * kFlagStride] job sequence number (never 0; 0xFFFFFFFF = quit), replicated: helper h polls copy h % kFlagReps
  unsigned long long* hjob;      // [kJobBufs][kJobWords] job descriptors, slot = sequence % kJobBufs
  // Round 4: a job goes out in TWO parts.  Part 1 (hjob + hctl) as soon as the contraction's records are built -- the helpers stage
  // rows and region statistics, form the shared sets and the entropies (S0-S3) while the loop's workgroup is still finding the top two
  // of r2's extremes; part 2 (hjob2 + hctl2: those extremes and min / max of B(r2)) is what the vector assembly (S4) waits for.
  uint32_t* hctl2;               // [kFlagReps * kFlagStride] sequence number of the newest job whose part 2 is out
  unsigned long long* hjob2;     // [kJobBufs][kJob2Words] sequence | per channel: best_mn, best_mx, second_mn, second_mx, (Bmn | Bmx << 32) of r2
  unsigned long long* hvotes;    // [kJobMax] sequence << 32 | model << 24 | votes, indexed by the record's position in the job
  unsigned long long* hrec;      // [kJobMax][K][kRowWords] rows of the current job: e_A | e_NA | rs, own slot | table flag, fragile head
  unsigned long long* hadj;      // [2*R0] adj_off | adj_len << 32 for the helpers
  uint32_t n_helpers;            // workgroups 1..n_helpers score records; 0 = the loop's own workgroup does
  uint32_t shard, n_shards;      // initial scoring: this call scores the records e with e % n_shards == shard (multi-GPU K7)
  unsigned long long* ctrl;
  unsigned long long max_iters;
  const uint32_t* forced;        // bc_feat mode: [forced_n][2] region pairs to merge, in this order (no queue, no scoring)
  unsigned long long forced_n;
  BcCfg cfg;
  DeviceClassifier clf;
};

namespace {

// Channel c of the state BY VALUE, field by field.  The loop kernel takes its state as a kernel argument (pointers read from the
// kernel-argument segment are known to be GLOBAL pointers: global_load / global_store; through a pointer to a state in memory
// every access was a FLAT instruction, which counts in lgkmcnt too and ties every LDS wait to the stores in flight).  A pointer
// fetched with a run-time index loses that, and selecting a whole struct goes through private memory: hence the field-wise
// selects (c is uniform: scalar selects).  GLIA_BC_COMMON instances have one channel.
__device__ __forceinline__ BcChan chan_of(const BcState& st, int c) {
  BcChan r = st.ch[0];
#ifndef GLIA_BC_COMMON
#define GLIA_SEL(f) r.f = c == 1 ? st.ch[1].f : c == 2 ? st.ch[2].f : c == 3 ? st.ch[3].f : r.f
  GLIA_SEL(pts); GLIA_SEL(Bn); GLIA_SEL(Bt); GLIA_SEL(Bmn); GLIA_SEL(Bmx); GLIA_SEL(entP); GLIA_SEL(entB); GLIA_SEL(e_A); GLIA_SEL(e_NA);
  GLIA_SEL(e_dir); GLIA_SEL(pool_dir); GLIA_SEL(le_stats);
#undef GLIA_SEL
#endif
  return r;
}

// ---- memory access forms -----------------------------------------------------------------------------------------------
// Everything the loop's workgroup and the helpers exchange goes through agent-scope accesses: write-through (sc1) stores by the
// producer, "my stores are acknowledged" (s_waitcnt vmcnt(0) in every storing wave + the workgroup's barrier) before ONE lane
// stores the flag, and sc1 loads -- which bypass the reader's L1 -- for EVERY load of such bytes by the consumer
// (cdna_hip_programming.md Guideline 16, form R1 with sc1 loads in place of the acquire).  A release / acquire fence pair at
// agent scope would write back and invalidate whole caches at every hand-off (measured in round 1: -12 % loop time without).
__device__ __forceinline__ uint32_t ld_relaxed(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_release(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(reinterpret_cast<uint32_t*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint8_t* p, uint8_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(float2* p, float2 v) {
  st_agent(reinterpret_cast<unsigned long long*>(p), ((unsigned long long)__float_as_uint(v.y) << 32) | (unsigned long long)__float_as_uint(v.x));
}
// a statistics struct (a whole number of 8-byte words, 8-byte aligned) with write-through stores
template <class T>
__device__ __forceinline__ void st_agent_struct(T* dst, const T& v) {
  static_assert(sizeof(T) % 8 == 0 && alignof(T) >= 8, "8-byte words");
  unsigned long long w[sizeof(T) / 8];
  __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
  for (int i = 0; i < (int)(sizeof(T) / 8); ++i) st_agent(reinterpret_cast<unsigned long long*>(dst) + i, w[i]);
}
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// loads of bytes another workgroup may have written in this launch: AG = true in the helpers (sc1: past the L1), plain in the
// loop's own workgroup (the only writer of that state)
template <bool AG> __device__ __forceinline__ uint32_t ldm(const uint32_t* p) { if (AG) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return *p; }
template <bool AG> __device__ __forceinline__ uint8_t ldm(const uint8_t* p) { if (AG) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return *p; }
template <bool AG> __device__ __forceinline__ unsigned long long ldm(const unsigned long long* p) { if (AG) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return *p; }
template <bool AG> __device__ __forceinline__ float ldm(const float* p) { return __uint_as_float(ldm<AG>(reinterpret_cast<const uint32_t*>(p))); }
template <bool AG> __device__ __forceinline__ double ldm(const double* p) { return __longlong_as_double((long long)ldm<AG>(reinterpret_cast<const unsigned long long*>(p))); }
template <bool AG> __device__ __forceinline__ float2 ldm(const float2* p) {
  const unsigned long long w = ldm<AG>(reinterpret_cast<const unsigned long long*>(p));
  return make_float2(__uint_as_float((uint32_t)w), __uint_as_float((uint32_t)(w >> 32)));
}
// this wave's earlier stores have been acknowledged (and the compiler keeps later accesses behind this point)
__device__ __forceinline__ void stores_done() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void after_flag() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

template <bool AG>
__device__ __forceinline__ uint32_t find_root(const BcState& st, uint32_t x) {
  uint32_t p = ldm<AG>(&st.parent[x]);
  while (p != x) {
    const uint32_t g = ldm<AG>(&st.parent[p]);
    if (g != p) st_agent(&st.parent[x], g);    // path halving (benign race: only ever points to an ancestor)
    x = p; p = g;
  }
  return x;
}

// does leaf x still own an un-cancelled boundary entry inside its region?  (nm_out, le_* are constant during the loop)
template <bool AG>
__device__ __forceinline__ bool leaf_alive(const BcState& st, uint32_t x) {
  if (st.nm_out[x]) return true;
  const uint32_t rx = find_root<AG>(st, x);
  for (uint32_t i = st.le_start[x]; i < st.le_start[x + 1]; ++i)
    if (st.le_mutual[i] && find_root<AG>(st, st.le_dst[i]) != rx) return true;
  return false;
}

// min / max (channel c) over region r's boundary set without the mutual entries it sends along record `skip`
__device__ __forceinline__ void excl_minmax(const BcState& st, int c, uint32_t r, uint32_t skip, float& mn, float& mx) {
  const BcChan ch = chan_of(st, c);
  mn = ch.Bn[r].mn; mx = ch.Bn[r].mx;
  const uint32_t off = st.adj_off[r], len = st.adj_len[r];
  for (uint32_t i = 0; i < len; ++i) {
    const uint32_t e = st.pool[off + i];
    if (e == skip || !st.e_alive[e]) continue;
    const float* d = &ch.e_dir[(size_t)e * 4 + (st.e_u[e] == r ? 0 : 2)];
    mn = fminf(mn, d[0]); mx = fmaxf(mx, d[1]);
  }
}

// shared boundary of the two regions of record rec on channel c: mutual entries + always-alive non-mutual ones + the
// fragile ones whose target leaf is still alive
__device__ __forceinline__ void shared_boundary(const BcState& st, int c, uint32_t rec, EStats& sh) {
  estats_clear(sh);
  if (rec != kNone) {
    const BcChan ch = chan_of(st, c);
    sh = ch.e_A[rec];
    estats_add(sh, ch.e_NA[rec]);
    for (uint32_t f = st.e_fhead[rec]; f != kNone; f = st.le_next[f])
      if (leaf_alive<false>(st, st.le_dst[f])) estats_add(sh, ch.le_stats[f]);
  }
}

// ---- where a vector's statistics live ------------------------------------------------------------------------------------
// edge_features reads them through a view: the global arrays (initial edges, the popped edge's vector), or the copies a scoring
// round has staged in LDS (RecIn / R2In below).  cc = image channel.
struct GlobalView {
  const BcState* st; uint32_t first, second, rec; const EStats* shv;      // shv[cc]: shared boundary sets, computed beforehand
  __device__ __forceinline__ const PStats* P0(int cc) const { return &chan_of(*st, cc).pts[first]; }
  __device__ __forceinline__ const PStats* P1(int cc) const { return &chan_of(*st, cc).pts[second]; }
  __device__ __forceinline__ const EStats* B0(int cc) const { return &chan_of(*st, cc).Bt[first]; }
  __device__ __forceinline__ const EStats* B1(int cc) const { return &chan_of(*st, cc).Bt[second]; }
  __device__ __forceinline__ const EStats* A(int cc) const { return rec != kNone ? &chan_of(*st, cc).e_A[rec] : nullptr; }   // kNone: no shared record (bc_feat on non-neighbours)
  __device__ __forceinline__ const EStats* sh(int cc) const { return &shv[cc]; }
  __device__ __forceinline__ float bmn0(int cc) const { return chan_of(*st, cc).Bmn[first]; }
  __device__ __forceinline__ float bmx0(int cc) const { return chan_of(*st, cc).Bmx[first]; }
  __device__ __forceinline__ float bmn1(int cc) const { return chan_of(*st, cc).Bmn[second]; }
  __device__ __forceinline__ float bmx1(int cc) const { return chan_of(*st, cc).Bmx[second]; }
};
// one record's staged inputs on one channel (first = the neighbour rs) ...
struct alignas(16) RecIn {
  PStats P; uint32_t padP[2];     // pts[rs]
  EStats B;                       // Bt[rs]
  EStats A;                       // e_A[rec]
  EStats sh;                      // staged: e_NA[rec]; then the shared boundary set (A + NA + alive fragile entries)
  float bnmn, bnmx;               // Bn[rs] extremes
  float bmn, bmx;                 // Bmn / Bmx[rs]
  float exmn, exmx;               // extremes of rs's boundary set without this record's mutual entries
  uint32_t pad[2];
  double entP, entB;              // entropies of pts[rs], Bt[rs]
};
static_assert(sizeof(PStats) == 120 && sizeof(EStats) == 112 && sizeof(RecIn) == 512, "staging layout");
// ... and the region being created (second = r2), once per round
struct alignas(16) R2In {
  PStats P; uint32_t padP[2];
  EStats B;
  float bnmn, bnmx, bmn, bmx;
  double entP, entB;
};
static_assert(sizeof(R2In) == 272, "staging layout");
struct StagedView {
  const RecIn* in; const R2In* r2;        // [K] each
  __device__ __forceinline__ const PStats* P0(int cc) const { return &in[cc].P; }
  __device__ __forceinline__ const PStats* P1(int cc) const { return &r2[cc].P; }
  __device__ __forceinline__ const EStats* B0(int cc) const { return &in[cc].B; }
  __device__ __forceinline__ const EStats* B1(int cc) const { return &r2[cc].B; }
  __device__ __forceinline__ const EStats* A(int cc) const { return &in[cc].A; }
  __device__ __forceinline__ const EStats* sh(int cc) const { return &in[cc].sh; }
  __device__ __forceinline__ float bmn0(int cc) const { return in[cc].bmn; }
  __device__ __forceinline__ float bmx0(int cc) const { return in[cc].bmx; }
  __device__ __forceinline__ float bmn1(int cc) const { return r2[cc].bmn; }
  __device__ __forceinline__ float bmx1(int cc) const { return r2[cc].bmx; }
};

// feature vector of one record, read through a view V (above); `out` must hold bc_full_dim doubles (the simple selection is
// compacted in place).  No private arrays: see bc_features.hpp.
// ex: per channel {min, max of first's boundary set without the record's mutual entries, the same for second};
// pre (optional): values of the lane-parallel pass (feat::pre_region / pre_boundary)
// parts: which blocks to write (1 first region, 2 second region, 4 merged region, 8 boundary block) -- the greedy loop
// gives each block to a different wave, and with few records each HALF of a block (half = 1 / 2, bc_features.hpp; 0 = whole
// blocks); finish: apply the log / simple selection (the loop does both lane-parallel)
template <class V>
__device__ __forceinline__ void edge_features(const BcCfg& c, const V& v, const float* ex, double* out, const double* pre = nullptr, int parts = 15,
                                              bool finish = true, int half = 0) {
  const PStats* P0 = v.P0(0);
  const PStats* P1 = v.P1(0);
  const EStats* B0 = v.B0(0);
  const EStats* B1 = v.B1(0);
  const EStats* A0 = v.A(0);
  const uint32_t n0 = P0->n, n1 = P1->n;
  // keep region 0 area <= region 1 area (main_merge_order_bc.cxx:77-80): decides the slots of the two region blocks
  const bool swap = feat::sdiv((double)n0, c.norm_area, 0.0) > feat::sdiv((double)n1, c.norm_area, 0.0);
  double* o_bf = out;
  double* o_first = out + c.bfdim + (swap ? c.rfdim : 0);
  double* o_second = out + c.bfdim + (swap ? 0 : c.rfdim);
  double* o_merged = out + c.bfdim + 2 * c.rfdim;
  // statistics sources of one region (which = 0 first, 1 second) and of the scratch-merged region (which = 2)
  auto src_of = [&](int which) {
    return [&c, &v, ex, pre, which](int kind, int i) -> feat::ImgSrc {
      const int cc = kind == 0 ? c.rc[i] : (kind == 1 ? c.lc[i] : c.bc[i]);
      const double* pe = nullptr;
      if (pre) pe = kind < 2 ? pre + feat::pre_region(c, kind, i) + which : pre + feat::pre_boundary(c, i) + which;
      if (kind < 2) {
        const PStats* p0 = v.P0(cc); const PStats* p1 = v.P1(cc);
        if (which == 0) return feat::ImgSrc{p0->hist, nullptr, nullptr, p0->n, p0->sum, p0->sq, p0->mn, p0->mx, pe};
        if (which == 1) return feat::ImgSrc{p1->hist, nullptr, nullptr, p1->n, p1->sum, p1->sq, p1->mn, p1->mx, pe};
        return feat::ImgSrc{p0->hist, p1->hist, nullptr, p0->n + p1->n, p0->sum + p1->sum, p0->sq + p1->sq,
                            p1->mn < p0->mn ? p1->mn : p0->mn, p1->mx > p0->mx ? p1->mx : p0->mx, pe};
      }
      const EStats* b0 = v.B0(cc); const EStats* b1 = v.B1(cc);
      if (which == 0) return feat::ImgSrc{b0->hist, nullptr, nullptr, b0->n, b0->sum, b0->sq, v.bmn0(cc), v.bmx0(cc), pe};
      if (which == 1) return feat::ImgSrc{b1->hist, nullptr, nullptr, b1->n, b1->sum, b1->sq, v.bmn1(cc), v.bmx1(cc), pe};
      // the merged boundary set: both sets minus the mutual entries of the shared record
      const EStats* a = v.A(cc);
      uint32_t bn = b0->n + b1->n;
      double bsum = b0->sum + b1->sum, bsq = b0->sq + b1->sq;
      if (a) { bn -= a->n; bsum -= a->sum; bsq -= a->sq; }
      return feat::ImgSrc{b0->hist, b1->hist, a ? a->hist : nullptr, bn, bsum, bsq, fminf(ex[4 * cc + 0], ex[4 * cc + 2]),
                          fmaxf(ex[4 * cc + 1], ex[4 * cc + 3]), pe};
    };
  };
  // area / perimeter as region_feats_multi normalises them (the boundary block needs both regions')
  double ar_first = feat::sdiv((double)n0, c.norm_area, 0.0);
  double pe_first = feat::sdiv((double)((unsigned long long)B0->n + (unsigned long long)P0->border), c.norm_len, 0.0);
  double ar_second = feat::sdiv((double)n1, c.norm_area, 0.0);
  double pe_second = feat::sdiv((double)((unsigned long long)B1->n + (unsigned long long)P1->border), c.norm_len, 0.0);
  double ar_m, pe_m;
  if (parts & 1) {
    feat::ShapeIn r;
    r.n = n0; r.border = P0->border; r.bn = B0->n;
    for (int i = 0; i < 3; ++i) { r.lo[i] = P0->lo[i]; r.hi[i] = P0->hi[i]; }
    for (int i = 0; i < GLIA_HMT_MAX_THRESH; ++i) r.thr[i] = B0->thr[i];
    feat::region_feats_multi(c, r, src_of(0), o_first, ar_first, pe_first, half);
  }
  if (parts & 2) {
    feat::ShapeIn r;
    r.n = n1; r.border = P1->border; r.bn = B1->n;
    for (int i = 0; i < 3; ++i) { r.lo[i] = P1->lo[i]; r.hi[i] = P1->hi[i]; }
    for (int i = 0; i < GLIA_HMT_MAX_THRESH; ++i) r.thr[i] = B1->thr[i];
    feat::region_feats_multi(c, r, src_of(1), o_second, ar_second, pe_second, half);
  }
  if (parts & 4) {
    // the scratch-merged region (TRegionMap::merge under key 0)
    feat::ShapeIn r;
    r.n = n0 + n1; r.border = P0->border + P1->border;
    for (int i = 0; i < 3; ++i) { r.lo[i] = P0->lo[i] < P1->lo[i] ? P0->lo[i] : P1->lo[i]; r.hi[i] = P0->hi[i] > P1->hi[i] ? P0->hi[i] : P1->hi[i]; }
    r.bn = B0->n + B1->n;
    for (int i = 0; i < GLIA_HMT_MAX_THRESH; ++i) r.thr[i] = B0->thr[i] + B1->thr[i];
    if (A0) { r.bn -= A0->n; for (int i = 0; i < GLIA_HMT_MAX_THRESH; ++i) r.thr[i] -= A0->thr[i]; }
    feat::region_feats_multi(c, r, src_of(2), o_merged, ar_m, pe_m, half);
  }
  if (parts & 8) {
    // shared boundary: counts from channel 0, image statistics from each boundary-list channel
    const EStats* sh0 = v.sh(0);
    auto srcSh = [&](int i) -> feat::ImgSrc {
      const EStats* sh = v.sh(c.bc[i]);
      return feat::ImgSrc{sh->hist, nullptr, nullptr, sh->n, sh->sum, sh->sq, sh->mn, sh->mx, pre ? pre + feat::pre_boundary(c, i) + 3 : nullptr};
    };
    if (swap) feat::boundary_feats_multi(c, sh0->n, sh0->thr, ar_second, pe_second, ar_first, pe_first, src_of(1), src_of(0), srcSh, pre, o_bf, half);
    else feat::boundary_feats_multi(c, sh0->n, sh0->thr, ar_first, pe_first, ar_second, pe_second, src_of(0), src_of(1), srcSh, pre, o_bf, half);
  }
  if (finish) feat::finish_features(c, out);
}
// ... from the global arrays (one thread; the shared boundary sets of all channels are formed first)
__device__ __forceinline__ void edge_features_global(const BcState& st, uint32_t first, uint32_t second, uint32_t rec, const float* ex, double* out) {
  EStats shv[kMaxChannels];
  for (int cc = 0; cc < st.cfg.K; ++cc) shared_boundary(st, cc, rec, shv[cc]);
  const GlobalView v{&st, first, second, rec, shv};
  edge_features(st.cfg, v, ex, out);
}

__device__ __forceinline__ int forest_vote(const DeviceForest& f, int tree, const double* x) {
  int k = f.root[tree];
  for (int step = 0; step < f.nrnodes; ++step) {        // bounded: a malformed tree cannot hang the device
    const PackedNode n = f.nodes[k];
    if (n.var < 0) return -1 - n.var;
    k = n.left + ((x[n.var] <= n.split) ? 0 : 1);        // left iff x[var] <= split
  }
  return 0;
}
// The walk over the three-levels-per-line layout (forest.hpp) -- a third of the dependent trips of a node-per-load walk; the line's seven
// feature values are requested from LDS together (slots below a terminal node name feature 0) -- by a PAIR of lanes (2 i, 2 i + 1):
// each loads half of the line's 96 bytes and the halves are swapped inside the quad
// (DPP), so a wave asks the texture path for 3 x 32 lines per step instead of 6 x 64 -- the walk of 255 trees on one compute unit is
// bound by that address rate, not by the trips.  Both lanes then decide (no divergence inside the pair) and end in the same state.
__device__ __forceinline__ uint4 swap_pair(const uint4 v) {
  return make_uint4(dpp_u32<0xB1, 0xf>(v.x), dpp_u32<0xB1, 0xf>(v.y), dpp_u32<0xB1, 0xf>(v.z), dpp_u32<0xB1, 0xf>(v.w));
}
__device__ __forceinline__ int forest_vote_triples_pair(const DeviceForest& f, int tree, const double* x, const bool odd) {
  int k = f.troot[tree];
  for (int step = 0; step < f.nrnodes; step += 3) {        // bounded: a malformed tree cannot hang the device
    const uint4* q = reinterpret_cast<const uint4*>(&f.triples[k]) + (odd ? 3 : 0);
    const uint4 m0 = q[0], m1 = q[1], m2 = q[2];
    const uint4 o0 = swap_pair(m0), o1 = swap_pair(m1), o2 = swap_pair(m2);
    const uint4 a = odd ? o0 : m0, b = odd ? o1 : m1, c = odd ? o2 : m2, d = odd ? m0 : o0, e = odd ? m1 : o1, g = odd ? m2 : o2;
    const int v0 = (int)(short)(d.z & 0xFFFFu), v1 = (int)(short)(d.z >> 16), v2 = (int)(short)(d.w & 0xFFFFu), v3 = (int)(short)(d.w >> 16);
    const int v4 = (int)(short)(e.x & 0xFFFFu), v5 = (int)(short)(e.x >> 16), v6 = (int)(short)(e.y & 0xFFFFu);
    if (v0 < 0) return -1 - v0;
    const double x0 = x[v0], x1 = x[v1 < 0 ? 0 : v1], x2 = x[v2 < 0 ? 0 : v2], x3 = x[v3 < 0 ? 0 : v3], x4 = x[v4 < 0 ? 0 : v4],
                 x5 = x[v5 < 0 ? 0 : v5], x6 = x[v6 < 0 ? 0 : v6];
    const double s0 = __hiloint2double((int)a.y, (int)a.x), s1 = __hiloint2double((int)a.w, (int)a.z);
    const double s2 = __hiloint2double((int)b.y, (int)b.x), s3 = __hiloint2double((int)b.w, (int)b.z);
    const double s4 = __hiloint2double((int)c.y, (int)c.x), s5 = __hiloint2double((int)c.w, (int)c.z);
    const double s6 = __hiloint2double((int)d.y, (int)d.x);
    const bool r0 = !(x0 <= s0);                            // left iff x[var] <= split
    const int va = r0 ? v2 : v1;
    if (va < 0) return -1 - va;
    const bool r1 = !((r0 ? x2 : x1) <= (r0 ? s2 : s1));
    const int vb = r0 ? (r1 ? v6 : v5) : (r1 ? v4 : v3);
    if (vb < 0) return -1 - vb;
    const double xb = r0 ? (r1 ? x6 : x5) : (r1 ? x4 : x3), sb = r0 ? (r1 ? s6 : s5) : (r1 ? s4 : s3);
    const int nx = r0 ? (r1 ? (int)g.y : (int)g.x) : (r1 ? (int)e.w : (int)e.z);
    k = nx + ((xb <= sb) ? 0 : 1);
  }
  return 0;
}
__device__ __forceinline__ int pick_model(const DeviceClassifier& c, const double* x) {
  if (c.n_models == 1) return 0;
  if (x[c.dim1] < c.threshold) return 0;                // type/function.hxx:80-84
  if (x[c.dim0] < c.threshold) return 1;
  return 2;
}
__device__ __forceinline__ double classify_serial(const DeviceClassifier& c, const double* x) {
  if (c.kind == 1) return 1.0 - x[c.stub_index];
  const DeviceForest& f = c.f[pick_model(c, x)];
  int votes = 0;
  for (int t = 0; t < f.ntree; ++t) votes += forest_vote(f, t, x);
  return (double)votes / (double)f.ntree;               // ml/rf/rf.hxx:366-369
}

// ---- initialisation kernels ---------------------------------------------------------------------------
// per channel c; the structural fields are written by the channel-0 launch
__global__ void bc_leaf_entries(BcState st, int c, const uint32_t* pa, const uint32_t* pb, const uint32_t* prec,
                                const uint32_t* rlabel, long long* partner, int bins, int nthr) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= st.P) return;
  if (c == 0) {
    const uint32_t a = pa[i], b = pb[i];
    const long long j = find_pair(pa, pb, st.P, b, a);
    partner[i] = j;
    st.le_mutual[i] = j >= 0;
    const uint32_t src = find_label(rlabel, st.R0, a), dst = find_label(rlabel, st.R0, b);
    st.le_src[i] = src; st.le_dst[i] = dst; st.le_next[i] = kNone;
    if (j < 0) atomicAdd(&st.nm_out[src], 1u);
  }
  const uint32_t* w = &prec[(size_t)i * kPairWords];
  EStats s;
  estats_clear(s);
  s.n = w[P_CNT];
  for (int t = 0; t < nthr; ++t) s.thr[t] = w[P_THR + t];
  s.mn = ord_float(~w[P_MIN]); s.mx = ord_float(w[P_MAX]);
  memcpy(&s.sum, &w[P_SUM], 8); memcpy(&s.sq, &w[P_SQ], 8);
  for (int k = 0; k < bins; ++k) s.hist[k] = w[P_HIST + k];
  st.ch[c].le_stats[i] = s;
}

// first out-entry of every leaf (pairs ascend by source label): lower bound of the leaf's label in pa
__global__ void bc_leaf_starts(BcState st, const uint32_t* pa, const uint32_t* rlabel) {
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > st.R0) return;
  if (r == st.R0) { st.le_start[r] = (uint32_t)st.P; return; }
  const uint32_t key = rlabel[r];
  long long lo = 0, hi = st.P;
  while (lo < hi) { long long mid = (lo + hi) >> 1; if (pa[mid] < key) lo = mid + 1; else hi = mid; }
  st.le_start[r] = (uint32_t)lo;
}

__global__ void bc_leaf_regions(BcState st, int c, const uint32_t* rrec, int bins) {
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= st.R0) return;
  const BcChan& ch = st.ch[c];
  const uint32_t* w = &rrec[(size_t)r * kRegionWords];
  PStats p;
  p.n = w[R_CNT]; p.border = w[R_BORDER];
  for (int d = 0; d < 3; ++d) { p.lo[d] = (int)(0x7fffffffu - w[R_LO + d]); p.hi[d] = (int)w[R_HI + d] - 1; }
  p.mn = ord_float(~w[R_MIN]); p.mx = ord_float(w[R_MAX]);
  memcpy(&p.sum, &w[R_SUM], 8); memcpy(&p.sq, &w[R_SQ], 8);
  for (int k = 0; k < GLIA_HMT_MAX_BINS; ++k) p.hist[k] = k < bins ? w[R_HIST + k] : 0;
  ch.pts[r] = p;
  EStats bn, bt;
  estats_clear(bn); estats_clear(bt);
  for (uint32_t i = st.le_start[r]; i < st.le_start[r + 1]; ++i) {
    estats_add(bt, ch.le_stats[i]);
    if (!st.le_mutual[i]) estats_add(bn, ch.le_stats[i]);
  }
  ch.Bn[r] = bn; ch.Bt[r] = bt; ch.Bmn[r] = bt.mn; ch.Bmx[r] = bt.mx;
  {
    // the same sums, in bin order from +0.0, as the scoring pass forms them (util/stats.hxx:145-152)
    double ep = 0.0, eb = 0.0;
    for (int k = 0; k < bins; ++k) { ep = ep - feat::entropy_term(p.hist[k], p.n, st.cfg.libm_log2); eb = eb - feat::entropy_term(bt.hist[k], bt.n, st.cfg.libm_log2); }
    ch.entP[r] = ep; ch.entB[r] = eb;
  }
  if (c == 0) st.parent[r] = r;
}

__global__ void bc_record_flags(BcState st, const uint32_t* pa, const uint32_t* pb, uint32_t* flag) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= st.P) return;
  // one record per unordered leaf pair: owned by the (a<b) entry when it exists, else by the lone (a>b) entry
  flag[i] = (pa[i] < pb[i] || !st.le_mutual[i]) ? 1u : 0u;
}

__global__ void bc_record_fill(BcState st, int c, const uint32_t* flag, const uint32_t* eidx, const long long* partner,
                               const uint32_t* rank, uint32_t* deg) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= st.P || !flag[i]) return;
  const BcChan& ch = st.ch[c];
  const uint32_t e = eidx[i];
  const uint32_t src = st.le_src[i], dst = st.le_dst[i];
  const uint32_t u = src < dst ? src : dst, v = src < dst ? dst : src;
  EStats A, NA;
  estats_clear(A); estats_clear(NA);
  float* d = &ch.e_dir[(size_t)e * 4];
  d[0] = d[2] = __builtin_inff(); d[1] = d[3] = -__builtin_inff();
  const long long j = partner[i];
  if (j >= 0) {            // mutual pair: i = (u -> v), j = (v -> u)
    const EStats& si = ch.le_stats[i];
    const EStats& sj = ch.le_stats[j];
    A = si; estats_add(A, sj);
    d[0] = si.mn; d[1] = si.mx; d[2] = sj.mn; d[3] = sj.mx;
  } else if (st.nm_out[dst]) NA = ch.le_stats[i];
  ch.e_A[e] = A; ch.e_NA[e] = NA;
  if (c != 0) return;
  // structure (once)
  st.e_u[e] = u; st.e_v[e] = v; st.e_alive[e] = 1;
  st.e_orient[e] = rank[u] < rank[v] ? 1 : 0;
  st.e_fhead[e] = st.e_ftail[e] = kNone;
  if (j >= 0) {
    st.e_table[e] = 1;
    st.pq.leaf_seq[e] = (unsigned long long)i + 1ull;      // lexicographic (u,v) order of the table edges
  } else {
    st.e_table[e] = 0;
    if (!st.nm_out[dst]) { st.e_fhead[e] = st.e_ftail[e] = (uint32_t)i; }
  }
  atomicAdd(&deg[u], 1u);
  atomicAdd(&deg[v], 1u);
}

__global__ void bc_adj_fill(BcState st, uint32_t E0, uint32_t* cursor) {
  uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E0) return;
  const uint32_t u = st.e_u[e], v = st.e_v[e];
  const uint32_t pu = atomicAdd(&cursor[u], 1u), pv = atomicAdd(&cursor[v], 1u);
  st.pool[st.adj_off[u] + pu] = e; st.e_posu[e] = pu;
  st.pool[st.adj_off[v] + pv] = e; st.e_posv[e] = pv;
  for (int c = 0; c < st.cfg.K; ++c) {
    const float* d = &st.ch[c].e_dir[(size_t)e * 4];
    st.ch[c].pool_dir[st.adj_off[u] + pu] = make_float2(d[0], d[1]);
    st.ch[c].pool_dir[st.adj_off[v] + pv] = make_float2(d[2], d[3]);
  }
}

// the helpers' copy of the leaves' list headers (BcState::hadj)
__global__ void bc_hadj_fill(BcState st) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < st.R0) st.hadj[r] = (unsigned long long)st.adj_off[r] | ((unsigned long long)st.adj_len[r] << 32);
}

__global__ void bc_init_dead(BcState st, uint32_t from) {
  uint32_t i = from + blockIdx.x * blockDim.x + threadIdx.x;
  if (i < st.Ecap) { st.pq.leaf_seq[i] = 0; st.pq.leaf_sal[i] = -__builtin_inf(); st.e_alive[i] = 0; st.e_table[i] = 0; }
}

// initFb + initFsal of every initial table edge (util/struct_merge_bc.hxx:18-27)
__global__ void bc_init_score(BcState st, uint32_t E0) {
  uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E0 || !st.e_table[e] || e % st.n_shards != st.shard) return;
  const uint32_t u = st.e_u[e], v = st.e_v[e];
  const uint32_t first = st.e_orient[e] ? u : v, second = st.e_orient[e] ? v : u;
  float ex[4 * kMaxChannels];
#pragma unroll
  for (int c = 0; c < kMaxChannels; ++c) {
    if (c < st.cfg.K) { excl_minmax(st, c, first, e, ex[4 * c + 0], ex[4 * c + 1]); excl_minmax(st, c, second, e, ex[4 * c + 2], ex[4 * c + 3]); }
    else { ex[4 * c + 0] = ex[4 * c + 1] = ex[4 * c + 2] = ex[4 * c + 3] = 0.f; }
  }
  double x[kMaxFeat];
  edge_features_global(st, first, second, e, ex, x);
  st.pq.leaf_sal[e] = classify_serial(st.clf, x);
}

// ---- the loop ----------------------------------------------------------------------------------------------
// The bins' terms of a sum are added in bin order, as the reference does: a chain over DPP row_shr:1 inside a group of 16
// lanes (lane l ends with t_0 .. t_l; lane 0's predecessor reads +0.0, the reference's start value); the lane of the last
// bin holds the sum.  Every lane of the 16 must be active.
__device__ __forceinline__ double bin_chain(double t, int bins, bool negate) {
  double acc = 0.0;
  for (int b = 0; b < bins; ++b) {
    const unsigned long long ab = (unsigned long long)__double_as_longlong(acc);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)ab, 0x111, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(ab >> 32), 0x111, 0xf, 0xf, true);
    const double prev = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    acc = negate ? prev - t : prev + t;
  }
  return acc;
}

constexpr int kPoolBytes = 112 * 1024;   // LDS workspace of the scoring phase, partitioned at run time (ws_layout)
// The loop's workgroup matches the neighbours of r0 and r1 in an LDS table at the END of the pool (keys: neighbour + 1; values: the
// record that reaches it from r0 / from r1, + 1); contractions with more incident entries use the global mark arrays.
constexpr uint32_t kMarkSlots = 2048, kMarkMax = 1408;
constexpr uint32_t kMarkBytes = 3u * kMarkSlots * 4u;
constexpr uint32_t kWsBytes = (uint32_t)kPoolBytes - kMarkBytes;      // what ws_layout and the per-record scratch of phase B may use
constexpr uint32_t kJobMax = 1u << 14;    // records per job = slots of the votes array and rows of hrec; a contraction with more scores locally
constexpr uint32_t kRowWords = 32;        // 8-byte words of a record's row in hrec
constexpr uint32_t kFlagReps = 16, kFlagStride = 64;     // copies of the job flag, 256 bytes apart: 255 pollers on ONE word queue up at its memory channel
constexpr uint32_t kJobBufs = 4;          // job descriptors in flight (slot = sequence % kJobBufs; see bc_helper_loop)
constexpr uint32_t kJobWords = 4;          // ne0 | cnt << 32, r2 | newcount << 32, sequence, -
constexpr uint32_t kJob2Words = 1 + 5 * kMaxChannels;
constexpr uint32_t kHelpChunk = 8;        // records a helper scores per round (two fill one pass of a 255-tree forest)
struct RecHdr { uint32_t rec, rs, own, on, fhead, off, len; int model; };
struct BcShared {
  uint32_t r0, r1, e, stop, len0, len1, off0, off1, newcount;
  // per channel: ord(value) << 32 | record over r2's new records (their r2 -> rs extremes), best and runner-up
  unsigned long long best_mn[kMaxChannels], best_mx[kMaxChannels], second_mn[kMaxChannels], second_mx[kMaxChannels];
  uint32_t ex[kMaxChannels][4];
  int votes[kChunk];
  // the best of the records the last contraction inserted: they are in the tree's worklist but not propagated yet -- that
  // happens while the helpers score the NEXT contraction's records -- so a pop compares the tree's root with this candidate
  Key cand; Key cand_part[kBcThreads / 64];
  unsigned long long t2[kBcThreads / 64][kMaxChannels][4];      // per wave: (best, second) of the min keys, of the complemented max keys
  uint32_t nlog;                 // slots of the full vector that take a logarithm
  uint32_t lost;                 // a helper did not answer in time
  // what the next pop will need if the tree's root stays the best edge: fetched by an idle thread while the helpers score (kNone = nothing)
  uint32_t pre_e, pre_r0, pre_r1, pre_len0, pre_len1, pre_off0, pre_off1;
  uint32_t job_seq, job_ne0, job_cnt, job_r2, job_newcount, job_ok;     // helper side: the job being worked on
  uint32_t job2_ok;              // helper side: part 2 of the job has arrived (0 = gave up waiting)
  float r2bm[kMaxChannels][2];   // helper side: min / max of B(r2) from part 2 (patched into every staged copy of r2)
  // loop side: what the contraction keeps of the region it creates (its statistics are stored write-through for the helpers;
  // reading them back would go to memory): histograms and counts of pts[r2] / Bt[r2], extremes of Bn[r2]
  uint32_t r2hist[kMaxChannels][2][GLIA_HMT_MAX_BINS]; uint32_t r2n[kMaxChannels][2]; float r2bn[kMaxChannels][2];
  alignas(16) uint64_t log2tab[glibc::kLog2TabWords];   // glibc's log2 tables (glibc_math.hpp): head | tab | tab2
  uint16_t logpos[feat::kMaxLogSlots];
  PqWork pq;
  __attribute__((aligned(16))) unsigned char pool[kPoolBytes];
};
// workspace of one scoring round of up to `cap` records: the staged copy of the region being created (per channel) | the records'
// staged inputs (per record and channel) | feature vectors | precomputed entropies / distances | headers
struct ScoreWs { uint32_t cap; int fstride, npre; R2In* r2; RecIn* in; double* feat; double* fx; RecHdr* hdr; };
__device__ __forceinline__ ScoreWs ws_layout(const BcCfg& c, unsigned char* pool, uint32_t cap_limit = (uint32_t)kChunk, uint32_t* used = nullptr) {
  ScoreWs W;
  const uint32_t K = (uint32_t)BC_K(c);
  W.fstride = bc_full_dim(c); W.npre = feat::pre_count(c);
  const uint32_t per = (uint32_t)sizeof(RecHdr) + (uint32_t)sizeof(RecIn) * K + 8u * (uint32_t)W.fstride + 8u * (uint32_t)W.npre;
  const uint32_t cap = (kWsBytes - (uint32_t)sizeof(R2In) * K) / per;
  W.cap = cap < cap_limit ? cap : cap_limit;
  W.r2 = reinterpret_cast<R2In*>(pool);
  W.in = reinterpret_cast<RecIn*>(pool + sizeof(R2In) * K);
  W.feat = reinterpret_cast<double*>(W.in + (size_t)W.cap * K);
  W.fx = W.feat + (size_t)W.cap * W.fstride;
  W.hdr = reinterpret_cast<RecHdr*>(W.fx + (size_t)W.cap * W.npre);
  if (used) *used = ((uint32_t)(reinterpret_cast<unsigned char*>(W.hdr + W.cap) - pool) + 15u) & ~15u;
  return W;
}

constexpr unsigned long long kHelperSpinLimit = 1ull << 27;     // polls (with s_sleep) before a side gives up: ~60 s

// per channel: extremes of r2's boundary set with all / all but one of its new records (the top two of the contraction)
__device__ __forceinline__ void r2_extremes_of(const BcShared& s, uint32_t newcount, int c, uint32_t rec, float bnmn, float bnmx, float& mn, float& mx) {
  const float best_mn = newcount ? ord_float((uint32_t)(s.best_mn[c] >> 32)) : __builtin_inff();
  const float best_mx = newcount ? ord_float((uint32_t)(s.best_mx[c] >> 32)) : -__builtin_inff();
  const float second_mn = (s.second_mn[c] != ~0ull) ? ord_float((uint32_t)(s.second_mn[c] >> 32)) : __builtin_inff();
  const float second_mx = (s.second_mx[c] != 0ull) ? ord_float((uint32_t)(s.second_mx[c] >> 32)) : -__builtin_inff();
  const uint32_t arg_mn = (uint32_t)(s.best_mn[c] & 0xFFFFFFFFull), arg_mx = (uint32_t)(s.best_mx[c] & 0xFFFFFFFFull);
  mn = fminf(bnmn, rec == arg_mn ? second_mn : best_mn);
  mx = fmaxf(bnmx, rec == arg_mx ? second_mx : best_mx);
}

#ifdef GLIA_HMT_PROFILE
#define SPH(i) do { if (prof && threadIdx.x == 0) { const unsigned long long tn = __builtin_readcyclecounter(); atomicAdd(&g_pqprof[32 + (i)], tn - tsp); tsp = tn; } } while (0)
#else
#define SPH(i) do {} while (0)
#endif
// One scoring round = the feature vectors (and the ensemble member) of up to W.cap records, all edges (rs, r2) of the region r2
// just created: a STAGING step copies what the vectors need into LDS (stage_local in the loop's own workgroup, from the arrays;
// stage_rows + stage_regions in a helper, from the job's rows, with agent-scope loads), score_chunk computes from LDS alone apart
// from the neighbours' incident lists.  s.best_* / s.second_* hold the top two of r2's new records.
// Region-side words of a staged set (RecIn for region rs, R2In for r2): lane w < 35 of 64 fetches ONE word -- w < 15 of pts[r],
// w < 29 of Bt[r], 29..32 the four extremes (4 bytes), 33 / 34 the entropies.  The address is selected, the loads are
// unconditional (one 8-byte and one 4-byte load per lane, all in flight together): a load inside each branch of an if-chain is a
// round trip per branch (measured: 6 k cycles for this step instead of 1.5 k).
template <bool AG>
__device__ __forceinline__ unsigned long long stage_region_word(const BcChan& ch, uint32_t r, uint32_t w, unsigned long long* dst8, float* dst4, uint32_t ext4, uint32_t ent8,
                                                                const unsigned long long* lane35 = nullptr) {
  const unsigned long long* p8 = reinterpret_cast<const unsigned long long*>(&ch.pts[r]);
  if (w < 15u) p8 += w;
  else if (w < 29u) p8 = reinterpret_cast<const unsigned long long*>(&ch.Bt[r]) + (w - 15u);
  else if (w == 33u) p8 = reinterpret_cast<const unsigned long long*>(&ch.entP[r]);
  else if (w == 34u) p8 = reinterpret_cast<const unsigned long long*>(&ch.entB[r]);
  else if (w == 35u && lane35) p8 = lane35;          // (the caller's own word: returned, not stored)
  const float* p4 = w == 29u ? &ch.Bn[r].mn : w == 30u ? &ch.Bn[r].mx : w == 31u ? &ch.Bmn[r] : &ch.Bmx[r];
  const unsigned long long v8 = ldm<AG>(p8);
  const float v4 = ldm<AG>(p4);
  if (w < 15u) dst8[w] = v8;
  else if (w < 29u) dst8[16u + (w - 15u)] = v8;
  else if (w < 33u) dst4[ext4 + (w - 29u)] = v4;
  else if (w < 35u) dst8[ent8 + (w - 33u)] = v8;
  return v8;
}
// staged copy of region r2 (K x 64 lanes)
template <bool AG>
__device__ __forceinline__ void stage_r2_word(const BcState& st, const ScoreWs& W, uint32_t r2, uint32_t t) {
  const int c = (int)(t >> 6);
  stage_region_word<AG>(chan_of(st, c), r2, t & 63u, reinterpret_cast<unsigned long long*>(&W.r2[c]), reinterpret_cast<float*>(&W.r2[c]), 60u, 32u);
}
// the neighbour-side statistics of record j on channel c
template <bool AG>
__device__ __forceinline__ unsigned long long stage_rs_word(const BcState& st, const ScoreWs& W, uint32_t j, int c, uint32_t w, int K, const unsigned long long* lane35 = nullptr) {
  RecIn& in = W.in[j * K + c];
  return stage_region_word<AG>(chan_of(st, c), W.hdr[j].rs, w, reinterpret_cast<unsigned long long*>(&in), reinterpret_cast<float*>(&in), 116u, 62u, lane35);
}
static_assert(offsetof(RecIn, bnmn) == 464 && offsetof(RecIn, entP) == 496 && offsetof(R2In, bnmn) == 240 && offsetof(R2In, entP) == 256, "staging layout");
// loop's own workgroup: headers and record statistics from the arrays it has just written (plain loads).  The caller has filled
// hdr[j].rec and put a barrier; ends with a barrier.
__device__ __forceinline__ void stage_local(const BcState& st, const ScoreWs& W, uint32_t n, uint32_t r2) {
  const int tid = threadIdx.x;
  const int K = BC_K(st.cfg);
  if ((uint32_t)tid < n) {
    RecHdr h;
    h.rec = W.hdr[tid].rec;
    h.rs = st.e_u[h.rec]; h.on = st.e_table[h.rec]; h.own = st.e_posu[h.rec]; h.fhead = st.e_fhead[h.rec];
    h.off = st.adj_off[h.rs];
    const uint32_t len = st.adj_len[h.rs];
    h.len = h.on ? len : 0u;
    h.model = -1;
    W.hdr[tid] = h;
  }
  for (uint32_t t = tid; t < (uint32_t)K * 64u; t += kBcThreads) stage_r2_word<false>(st, W, r2, t);
  __syncthreads();
  for (uint32_t t = tid; t < n * (uint32_t)K * 64u; t += kBcThreads) {
    const uint32_t j = t / ((uint32_t)K * 64u); const int c = (int)((t >> 6) % (uint32_t)K); const uint32_t w = t & 63u;
    const RecHdr h = W.hdr[j];
    if (!h.on) continue;
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(&W.in[j * K + c]);
    if (w < 35u) stage_rs_word<false>(st, W, j, c, w, K);
    else {
      const unsigned long long* pa = reinterpret_cast<const unsigned long long*>(&chan_of(st, c).e_A[h.rec]);
      const unsigned long long* pn = reinterpret_cast<const unsigned long long*>(&chan_of(st, c).e_NA[h.rec]);
      const unsigned long long v = *(w < 49u ? pa + (w - 35u) : pn + (w < 63u ? w - 49u : 0u));
      if (w < 49u) dst[30u + (w - 35u)] = v;
      else if (w < 63u) dst[44u + (w - 49u)] = v;
    }
  }
  __syncthreads();
}
// helper: the rows of the job's records pos[0 .. n) (positions in the job; clamped, a row past the job's end is read and ignored)
// into headers, RecIn::A and RecIn::sh (= e_NA until score_chunk completes it).  No barrier: the caller waits for its loads.
__device__ __forceinline__ void stage_rows(const BcState& st, const ScoreWs& W, uint32_t n, uint32_t h, uint32_t H, uint32_t i0, uint32_t ne0, int first_tid) {
  const int K = BC_K(st.cfg);
  const int t0 = (int)threadIdx.x - first_tid;
  if (t0 < 0) return;
  for (uint32_t t = (uint32_t)t0; t < n * (uint32_t)K * kRowWords; t += kBcThreads - (uint32_t)first_tid) {
    const uint32_t i = t / ((uint32_t)K * kRowWords); const int c = (int)((t / kRowWords) % (uint32_t)K); const uint32_t w = t % kRowWords;
    uint32_t pos = h + (i0 + i) * H;
    pos = pos < kJobMax ? pos : kJobMax - 1u;
    const unsigned long long v = ld_agent(st.hrec + ((size_t)pos * K + c) * kRowWords + w);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(&W.in[i * K + c]);
    if (w < 14u) dst[30u + w] = v;
    else if (w < 28u) dst[44u + (w - 14u)] = v;
    else if (c == 0 && w == 28u) { W.hdr[i].rs = (uint32_t)v; W.hdr[i].own = (uint32_t)(v >> 32); W.hdr[i].rec = ne0 + pos; W.hdr[i].model = -1; }
    else if (c == 0 && w == 29u) { W.hdr[i].on = (uint32_t)v; W.hdr[i].fhead = (uint32_t)(v >> 32); W.hdr[i].off = 0u; W.hdr[i].len = 0u; }
  }
}
// helper: neighbour-side statistics, list headers, and the staged copy of r2 (one round trip).  Ends with a barrier.
__device__ __forceinline__ void stage_regions(const BcState& st, const ScoreWs& W, uint32_t n, uint32_t r2) {
  const int tid = threadIdx.x;
  const int K = BC_K(st.cfg);
  for (uint32_t t = tid; t < (n + 1u) * (uint32_t)K * 64u; t += kBcThreads) {
    if (t < (uint32_t)K * 64u) { stage_r2_word<true>(st, W, r2, t); continue; }
    const uint32_t u = t - (uint32_t)K * 64u;
    const uint32_t j = u / ((uint32_t)K * 64u); const int c = (int)((u >> 6) % (uint32_t)K); const uint32_t w = u & 63u;
    if (!W.hdr[j].on || w > 35u) continue;
    const unsigned long long a = stage_rs_word<true>(st, W, j, c, w, K, &st.hadj[W.hdr[j].rs]);      // lane 35: the list header of rs
    if (w == 35u && c == 0) { W.hdr[j].off = (uint32_t)a; W.hdr[j].len = (uint32_t)(a >> 32); }
  }
  __syncthreads();
}

// ---- fragile entries of a record's shared boundary, helper side ----------------------------------------------------------------
// A record's fragile entries are a linked list, and each needs leaf_alive(): a merge-forest find plus a look at the leaf's mutual
// entries -- four to six DEPENDENT loads.  One lane per (record, channel) doing both is ~4 k cycles per entry; at 1024^3 a job nearly
// always holds a record with 4 .. 15 entries (+15 .. 60 k cycles on the answer the loop waits for), and a few thousand records late
// in the run hold hundreds to thousands (millions of cycles each: profiles/r04z_bc1024_fragile.txt).  A helper's LAST WAVE therefore
// owns the shared sets: lane i only WALKS the list of (record, channel) pair i (one load per entry), appending to ONE work list in
// LDS; then the wave's 64 lanes take the list's entries, decide aliveness and add the statistics to the pair's accumulator with LDS
// atomics; long lists go round in batches of kFragCap.  All of it inside one wave -- no workgroup barrier -- while the other waves
// scan the neighbour lists.  The helpers' copy of the mark table's LDS (never used by a helper) holds the lists.
constexpr uint32_t kFragCap = 4096;        // work-list entries per batch
constexpr uint32_t kFragItems = kHelpChunk * kMaxChannels;
struct FragAcc { uint32_t n, thr[GLIA_HMT_MAX_THRESH], mn_ord, mx_ord; double sum, sq; uint32_t hist[GLIA_HMT_MAX_BINS]; };
static_assert(sizeof(FragAcc) == sizeof(EStats), "same fields, extremes as ordered integers");
struct FragWs { FragAcc* acc; uint32_t* ids; unsigned char* tag; uint32_t* total; };
static_assert(kFragItems * sizeof(FragAcc) + kFragCap * 5u + 16u <= kMarkBytes, "fits the mark table's LDS");
static_assert(kFragItems <= 64u, "one lane per (record, channel) pair");
__device__ __forceinline__ FragWs frag_ws(unsigned char* pool) {
  FragWs f;
  f.acc = reinterpret_cast<FragAcc*>(pool + kWsBytes);
  f.ids = reinterpret_cast<uint32_t*>(f.acc + kFragItems);
  f.total = f.ids + kFragCap;
  f.tag = reinterpret_cast<unsigned char*>(f.total + 4);
  return f;
}
__device__ __forceinline__ void frag_clear(FragAcc& a) {
  a.n = 0; a.mn_ord = 0xFFFFFFFFu; a.mx_ord = 0u; a.sum = 0.0; a.sq = 0.0;
  for (int i = 0; i < GLIA_HMT_MAX_THRESH; ++i) a.thr[i] = 0;
  for (int i = 0; i < GLIA_HMT_MAX_BINS; ++i) a.hist[i] = 0;
}
// LDS traffic between lanes of ONE wave: the LDS keeps a wave's operations in order; the compiler must not move them across this
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Every thread of the workgroup calls, after a staging step.  AG: the caller is a helper workgroup -- what is still read from
// global memory (incident lists, fragile-entry chains, the merge forest) is read with agent-scope loads.
//   S2 shared boundary sets (one lane per record and channel) and the neighbours' "all but this record" extremes (16 lanes per
//   record)   S3 entropies and histogram distances (one lane per bin)   S4 the vector, one of its four blocks per wave
//   S5 logarithms, selection, model
// mid(): called by every thread between S3 and S4 (a helper waits there for part 2 of its job); returns false to abandon the round
template <bool AG, class Mid>
__device__ __forceinline__ bool score_chunk(const BcState& st, BcShared& s, const ScoreWs& W, uint32_t n, uint32_t newcount, Mid mid, bool prof = false) {
  const int tid = threadIdx.x;
  const int K = BC_K(st.cfg);
  const BcCfg& cf = st.cfg;
#ifdef GLIA_HMT_PROFILE
  unsigned long long tsp = __builtin_readcyclecounter();
#endif
  // ---- S3, as a function of the task (called from the places below) ----
  // entropies and histogram distances, one lane per bin: the log2 and divisions of a vector are by far its longest serial
  // stretch.  A TASK = 16 lanes on one group of sums of one record -- region / label list entry i: entropy of the merged voxel
  // set + L1 + chi-square; boundary list entry i: entropy of the merged boundary set; the same entry: entropy of the shared
  // boundary -- so the (usually three) logarithm passes of a record run side by side.  The bins' terms are added IN BIN ORDER, as
  // the reference does (bin_chain).
  // (consecutive tasks go to different WAVES: sixteen-lane groups of one wave would run their logarithms one after the other)
  const uint32_t l16 = (uint32_t)tid & 15u;
  const uint32_t nrl = (uint32_t)(BC_NR(cf) + BC_NL(cf)), NBq = (uint32_t)BC_NB(cf), G = nrl + 2u * NBq;
  auto s3_task = [&](const uint32_t task) __attribute__((always_inline)) {
      const uint32_t j = task / G, g = task - j * G;
      const bool on = W.hdr[j].on != 0;              // uniform over the 16 lanes
      const RecIn* in = &W.in[j * K];
      double* fx = W.fx + (size_t)j * W.npre;
      if (g < nrl) {
        const int kind = g < (uint32_t)BC_NR(cf) ? 0 : 1;
        const int i = kind ? (int)g - BC_NR(cf) : (int)g;
        const int cc = kind ? cf.lc[i] : cf.rc[i];
        const int bins = cf.cbins[cc];
        double t2 = 0.0, tl = 0.0, tx = 0.0;
        const PStats* P0 = &in[cc].P; const PStats* P1 = &W.r2[cc].P;
        const uint32_t h0 = P0->hist[l16], h1 = P1->hist[l16], pn0 = P0->n, pn1 = P1->n;
        const double e0 = in[cc].entP;                  // filed when rs was created (BcChan::entP)
        const double e1 = W.r2[cc].entP;                // r2's: worked out once per contraction
        if (on && (int)l16 < bins) {
          t2 = feat::entropy_term(h0 + h1, pn0 + pn1, cf.libm_log2, s.log2tab);
          feat::dist_terms(h0, pn0, h1, pn1, tl, tx);
        }
        const double e2 = bin_chain(t2, bins, true);
        const double dl = bin_chain(tl, bins, false), dx = bin_chain(tx, bins, false);
        if (on && (int)l16 == bins - 1) { double* q = fx + feat::pre_region(cf, kind, i); q[0] = e0; q[1] = e1; q[2] = e2; q[3] = dl; q[4] = dx; }
      } else {
        const uint32_t gb = g - nrl;
        const int i = (int)(gb >> 1); const bool shared = (gb & 1u) != 0u;
        const int cc = cf.bc[i];
        const int bins = cf.cbins[cc];
        const EStats* B0 = &in[cc].B; const EStats* B1 = &W.r2[cc].B;
        const EStats* A = &in[cc].A;
        const EStats* sh = &in[cc].sh;
        const uint32_t g0 = B0->hist[l16], g1 = B1->hist[l16], ga = A->hist[l16], bn0 = B0->n, bn1 = B1->n, an = A->n;
        const uint32_t gs = sh->hist[l16], sn = sh->n;
        const uint32_t cq = shared ? gs : g0 + g1 - ga, nq = shared ? sn : bn0 + bn1 - an;
        double t = 0.0;
        if (on && (int)l16 < bins) t = feat::entropy_term(cq, nq, cf.libm_log2, s.log2tab);
        const double en = bin_chain(t, bins, true);
        if (on && (int)l16 == bins - 1) {
          double* q = fx + feat::pre_boundary(cf, i);
          if (shared) q[3] = en;
          else { q[0] = in[cc].entB; q[1] = W.r2[cc].entB; q[2] = en; }
        }
      }
  };
  // ---- S2 ----
  for (uint32_t w = tid; w < (AG ? 0u : n * (uint32_t)K); w += kBcThreads) {       // (a helper: its last wave, below)
    // shared boundary (getBoundary / boundaryWith): mutual entries + always-alive non-mutual ones + the fragile ones whose target
    // leaf still owns an un-cancelled entry
    const uint32_t j = w / (uint32_t)K; const int c = (int)(w % (uint32_t)K);
    const RecHdr h = W.hdr[j];
    if (!h.on) continue;
    RecIn& in = W.in[j * K + c];
    EStats sh = in.A;
    estats_add(sh, in.sh);
    for (uint32_t f = h.fhead; f != kNone; f = ldm<AG>(&st.le_next[f])) if (leaf_alive<AG>(st, st.le_dst[f])) estats_add(sh, chan_of(st, c).le_stats[f]);
    in.sh = sh;
  }
  {
    // excl_minmax of every record's neighbour region, 16 lanes per record: unconditional, batched loads (a load behind a branch
    // is a round trip of its own), the list four entries per lane at a time.  A clamped index re-reads the last entry (harmless
    // for min / max), the record's own slot is masked out; dead entries hold (+inf, -inf).
    const uint32_t sub = (uint32_t)tid >> 4, l16 = (uint32_t)tid & 15u;
    for (uint32_t j = sub; j < n; j += kBcThreads / 16) {
      const RecHdr h = W.hdr[j];
      float mn[kMaxChannels], mx[kMaxChannels];
#pragma unroll
      for (int c = 0; c < kMaxChannels; ++c) { mn[c] = __builtin_inff(); mx[c] = -__builtin_inff(); }
      for (uint32_t i0 = 0; i0 < h.len; i0 += 64) {
#pragma unroll
        for (int c = 0; c < kMaxChannels; ++c) {
          if (c < K) {
            float2 d[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const uint32_t i = i0 + q * 16 + l16;
              d[q] = ldm<AG>(&chan_of(st, c).pool_dir[h.off + (i < h.len ? i : h.len - 1u)]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const uint32_t i = i0 + q * 16 + l16;
              if (i < h.len && i != h.own) { mn[c] = fminf(mn[c], d[q].x); mx[c] = fmaxf(mx[c], d[q].y); }
            }
          }
        }
      }
      if (h.on && l16 == 0) {
#pragma unroll
        for (int c = 0; c < kMaxChannels; ++c) if (c < K) { mn[c] = fminf(mn[c], W.in[j * K + c].bnmn); mx[c] = fmaxf(mx[c], W.in[j * K + c].bnmx); }
      }
#pragma unroll
      for (int c = 0; c < kMaxChannels; ++c) {
        if (c < K) {
          float a = mn[c], b = mx[c];
#pragma unroll
          for (int o = 8; o >= 1; o >>= 1) { a = fminf(a, __shfl_xor(a, o, 16)); b = fmaxf(b, __shfl_xor(b, o, 16)); }
          if (l16 == 0 && h.on) { W.in[j * K + c].exmn = a; W.in[j * K + c].exmx = b; }
        }
      }
    }
  }
  if (AG && tid >= (int)kBcThreads - 64) {
    // the shared sets, by the last wave alone (see FragWs)
    const FragWs F = frag_ws(s.pool);
    const uint32_t lane = (uint32_t)tid & 63u, items = n * (uint32_t)K;
    const bool mine = lane < items && W.hdr[lane / (uint32_t)K].on != 0u;
    uint32_t f = mine ? W.hdr[lane / (uint32_t)K].fhead : kNone;
    for (;;) {
      while (f != kNone) {
        const uint32_t pos = atomicAdd(F.total, 1u);
        if (pos >= kFragCap) break;                       // the list is full: this entry opens the next batch
        F.ids[pos] = f; F.tag[pos] = (unsigned char)lane;
        f = ldm<AG>(&st.le_next[f]);
      }
      wave_lds_fence();
      const uint32_t tot = *F.total < kFragCap ? *F.total : kFragCap;
      const bool more = __ballot(f != kNone) != 0ull;
      for (uint32_t k = lane; k < tot; k += 64u) {
        const uint32_t e = F.ids[k], q = F.tag[k];
        if (!leaf_alive<AG>(st, st.le_dst[e])) continue;
        const EStats es = chan_of(st, (int)(q % (uint32_t)K)).le_stats[e];
        FragAcc& a = F.acc[q];
        atomicAdd(&a.n, es.n);
#pragma unroll
        for (int i = 0; i < GLIA_HMT_MAX_THRESH; ++i) if (es.thr[i]) atomicAdd(&a.thr[i], es.thr[i]);
        atomicMin(&a.mn_ord, float_ord(es.mn)); atomicMax(&a.mx_ord, float_ord(es.mx));
        atomicAdd(&a.sum, es.sum); atomicAdd(&a.sq, es.sq);
#pragma unroll
        for (int i = 0; i < GLIA_HMT_MAX_BINS; ++i) if (es.hist[i]) atomicAdd(&a.hist[i], es.hist[i]);
      }
      wave_lds_fence();
      if (lane == 0u) *F.total = 0u;
      wave_lds_fence();
      if (!more) break;
    }
    if (mine) {
      RecIn& in = W.in[lane];                              // (W.in is indexed [record * K + channel], as the pairs are)
      EStats sh = in.A;
      estats_add(sh, in.sh);
      FragAcc& a = F.acc[lane];
      if (a.n) {
        sh.n += a.n; sh.sum += a.sum; sh.sq += a.sq;
        sh.mn = fminf(sh.mn, ord_float(a.mn_ord)); sh.mx = fmaxf(sh.mx, ord_float(a.mx_ord));
        for (int i = 0; i < GLIA_HMT_MAX_THRESH; ++i) sh.thr[i] += a.thr[i];
        for (int i = 0; i < GLIA_HMT_MAX_BINS; ++i) sh.hist[i] += a.hist[i];
        frag_clear(a);
      }
      in.sh = sh;
    }
    // ... and the entropies of the shared sets (S3's tasks that need them), four at a time
    wave_lds_fence();
    for (uint32_t t3 = lane >> 4; t3 < n * NBq; t3 += 4u) { const uint32_t j = t3 / NBq, i = t3 - j * NBq; s3_task(j * G + nrl + 2u * i + 1u); }
  } else if (AG && tid >= 128) {
    // S3's other tasks need nothing of this stage: the waves 2 .. 6 run them now, beside the list scans of the waves 0 and 1 and the
    // last wave's shared sets (consecutive tasks on different waves)
    const uint32_t G2 = nrl + NBq, sub2 = (((uint32_t)tid >> 6) - 2u) + 5u * (((uint32_t)tid & 63u) >> 4);
    for (uint32_t t2 = sub2; t2 < n * G2; t2 += 20u) {
      const uint32_t j = t2 / G2, g2 = t2 - j * G2;
      s3_task(j * G + (g2 < nrl ? g2 : nrl + 2u * (g2 - nrl)));
    }
  }
  if (!AG) __syncthreads();       // (a helper's S4 lies behind the barrier that ends S3)
  SPH(2);
  // ---- S3 ----
  if (!AG) {
    const uint32_t sub = ((uint32_t)tid >> 6) + (kBcThreads / 64) * (((uint32_t)tid & 63u) >> 4);
    for (uint32_t task = sub; task < n * G; task += kBcThreads / 16) s3_task(task);
  }
  __syncthreads();
  SPH(3);
  if (!mid()) return false;
  // ---- S4 ----
  {
    // one block of the vector per wave: waves 4g..4g+3 write the four blocks of the records 64g..64g+63 -- or, with at most 64
    // records (every helper round), one HALF of a block per wave: a vector is one lane's serial chain of divisions, and a helper
    // usually has one record
    const int wave = tid >> 6, part = wave & 3;
    const bool halves = n <= 64u;
    const uint32_t slot = halves ? (uint32_t)(tid & 63) : (uint32_t)(wave >> 2) * 64u + (uint32_t)(tid & 63);
    if (slot < n && W.hdr[slot].on) {
      const uint32_t rec = W.hdr[slot].rec;
      const RecIn* in = &W.in[slot * K];
      float ex[4 * kMaxChannels];
#pragma unroll
      for (int c = 0; c < kMaxChannels; ++c) {
        float a = 0.f, b = 0.f, cm = 0.f, dm = 0.f;
        if (c < K) { a = in[c].exmn; b = in[c].exmx; r2_extremes_of(s, newcount, c, rec, W.r2[c].bnmn, W.r2[c].bnmx, cm, dm); }
        ex[4 * c + 0] = a; ex[4 * c + 1] = b; ex[4 * c + 2] = cm; ex[4 * c + 3] = dm;
      }
      const StagedView v{in, W.r2};
      edge_features(cf, v, ex, &W.feat[slot * W.fstride], W.fx + (size_t)slot * W.npre, 1 << part, false,      // updateFb passes (rs, r2)
                    halves ? 1 + (wave >> 2) : 0);
    }
  }
  __syncthreads();
  SPH(4);
  // ---- S5 ----
  if (BC_LOG(cf)) {
    // feat.hxx:46-52, 103-106: the logarithms, one (record, slot) pair per thread
    const uint32_t nlog = s.nlog;
    for (uint32_t i = tid; i < n * nlog; i += kBcThreads) {
      const uint32_t j = i / nlog;
      if (!W.hdr[j].on) continue;
      double* q = &W.feat[j * W.fstride + s.logpos[i - j * nlog]];
      *q = feat::slog(*q, 0.0, cf.libm_log);
    }
    __syncthreads();
  }
  if ((uint32_t)tid < n && W.hdr[tid].on) {
    double* x = &W.feat[tid * W.fstride];
    feat::simple_selection(cf, x);
    W.hdr[tid].model = st.clf.kind == 1 ? 0 : pick_model(st.clf, x);
  }
  __syncthreads();
  SPH(5);
  return true;
}

// the forest's votes for the n vectors of a scoring round into s.votes (every thread calls; ends with a barrier)
__device__ __forceinline__ void forest_chunk(const BcState& st, BcShared& s, const ScoreWs& W, uint32_t n) {
  const int tid = threadIdx.x;
  if ((uint32_t)tid < n) s.votes[tid] = 0;
  __syncthreads();
  int ntree = st.clf.f[0].ntree;          // ensemble members may differ in size: iterate over the largest
  for (int m = 1; m < st.clf.n_models; ++m) ntree = st.clf.f[m].ntree > ntree ? st.clf.f[m].ntree : ntree;
  // a pair of lanes per (vector, tree) walk (forest_vote_triples_pair); both lanes of a pair take every branch together
  for (uint32_t i = (uint32_t)tid >> 1; i < n * (uint32_t)ntree; i += kBcThreads / 2) {
    const uint32_t j = i / (uint32_t)ntree, t = i % (uint32_t)ntree;
    const int m = W.hdr[j].model;
    if (m < 0 || (int)t >= st.clf.f[m].ntree) continue;
    const int vote = forest_vote_triples_pair(st.clf.f[m], (int)t, &W.feat[j * W.fstride], (tid & 1) != 0);
    if (vote && (tid & 1) == 0) atomicAdd(&s.votes[j], 1);
  }
  __syncthreads();
}

// Helper workgroups: wait for a job, score the records h, h + H, ... of it from start to finish (staging, extremes, shared sets,
// entropies, vector, forest), answer with one tagged 64-bit word per record.  The job descriptor is one of kJobBufs slots; a
// helper that had no record in the last jobs may find its slot being rewritten: the descriptor carries its sequence number and
// the flag is read again after it -- a helper WITH records in job v always finds it intact, because the loop's workgroup does
// not publish v + 1 before it has every answer of v.
__device__ __forceinline__ void bc_helper_loop(const BcState& st, BcShared& s) {
  const int tid = threadIdx.x;
  // helper number: workgroups go to the eight XCDs round-robin, the loop's own (0) to the first -- the workgroups 8, 16, ... share
  // its L2 and get the lowest numbers (a job's records go to the helpers 0, 1, ...)
  // (Measured: keeping a job WITH those 31 -- two to four records each, every stage is lane-parallel over the records -- is slower
  // than dealing it out across the XCDs one record per helper.)
  // The helpers of the OTHER XCDs are numbered XCD by XCD (1, 2, ... 7: the ping-pong latency from XCD 0 grows in that order,
  // tools/micro/xcd_pingpong), not round-robin: a job of ~94 records then stays on three XCDs, whose L2s keep the forest's lines --
  // spread over seven, each XCD walks too rarely to hold the deep ones.
  const uint32_t H = gridDim.x - 1u, b = blockIdx.x;
  uint32_t h;
  {
    const uint32_t x = b & 7u, q = b >> 3;                        // XCD, position among the XCD's workgroups
    if (x == 0u) h = q - 1u;                                      // (workgroup 0 is the loop's own)
    else {
      h = H >> 3;                                                 // the helpers of XCD 0
      for (uint32_t y = 1; y < x; ++y) h += y <= H ? (H - y) / 8u + 1u : 0u;      // workgroups y, y + 8, ... <= H
      h += q;
    }
  }
  uint32_t used = 0;
  const ScoreWs W = ws_layout(st.cfg, s.pool, kHelpChunk, &used);
  const uint32_t cap = W.cap;
  if (tid == 0) s.nlog = (uint32_t)feat::log_slots(st.cfg, s.logpos);
  if ((uint32_t)tid < kFragItems) frag_clear(frag_ws(s.pool).acc[tid]);
  if (tid == 0) *frag_ws(s.pool).total = 0u;
  for (int i = tid; i < glibc::kLog2TabWords; i += kBcThreads) s.log2tab[i] = i < 18 ? glibc::kLog2Head[i] : i < 18 + 128 ? glibc::kLog2Tab[i - 18] : glibc::kLog2Tab2[i - 18 - 128];
  uint32_t last = 0;
  for (;;) {
    __syncthreads();
    if (tid == 0) {
      uint32_t v = last;
      // poll relaxed, one lane, with a sleep (255 workgroups poll this word)
      for (unsigned long long spins = 0; spins < kHelperSpinLimit; ++spins) {
        v = ld_relaxed(&st.hctl[(h % kFlagReps) * kFlagStride]);
        if (v != last) break;
        __builtin_amdgcn_s_sleep(2);
      }
      s.job_seq = v;
    }
    __syncthreads();
    const uint32_t v = s.job_seq;
    if (v == last || v == 0xFFFFFFFFu) return;          // gave up waiting / the loop is over
    after_flag();
#ifdef GLIA_HMT_PROFILE
    const bool prof = h == 0;
    unsigned long long tsp = __builtin_readcyclecounter();
#endif
    const unsigned long long* jb = st.hjob + (size_t)(v % kJobBufs) * kJobWords;
    unsigned long long wv = 0;
    if ((uint32_t)tid < kJobWords) wv = ld_agent(jb + tid);
    // in the same round trip, before the job's size is known: the rows of this helper's first records (waves 1..)
    stage_rows(st, W, cap, h, H, 0u, 0u, 64);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the descriptor has arrived before the flag is read again (all its lanes are in wave 0)
    if (tid == 0) { s.job_ne0 = (uint32_t)wv; s.job_cnt = (uint32_t)(wv >> 32); }
    if (tid == 1) { s.job_r2 = (uint32_t)wv; s.job_newcount = (uint32_t)(wv >> 32); }
    if (tid == 2) {
      const uint32_t f2 = ld_relaxed(&st.hctl[(h % kFlagReps) * kFlagStride]);
      s.job_ok = ((uint32_t)wv == v && f2 != 0xFFFFFFFFu && f2 - v < kJobBufs - 1u) ? 1u : 0u;
    }
    __syncthreads();
    last = v;
    SPH(6);
    if (!s.job_ok) continue;                            // torn descriptor: this helper had nothing in that job anyway
    const uint32_t cnt = s.job_cnt, ne0 = s.job_ne0, r2 = s.job_r2, newcount = s.job_newcount;
    if (h >= cnt) continue;
    const uint32_t m = (cnt - h + H - 1u) / H;          // records at the job's positions h, h + H, ...
    bool got2 = false;                                  // part 2 of this job is in LDS (a register, the same in every thread: an LDS
                                                        // flag read here and rewritten below without a barrier in between would let a
                                                        // late wave skip the barriers the early ones wait at)
    for (uint32_t i0 = 0; i0 < m; i0 += cap) {
      const uint32_t n = m - i0 < cap ? m - i0 : cap;
      if (i0) { stage_rows(st, W, n, h, H, i0, 0u, 0); __syncthreads(); }
      if ((uint32_t)tid < n) W.hdr[tid].rec += ne0;     // (stage_rows left the position in the job)
      __syncthreads();
      SPH(0);
      stage_regions(st, W, n, r2);
      SPH(1);
      // S4 needs part 2 of the job: the top two of r2's extremes and min / max of B(r2).  The first round waits for it (bounded), every
      // round patches the staged copy of r2 with it.
      auto mid = [&]() __attribute__((always_inline)) -> bool {
        if (!got2) {
          if (tid == 0) {
            uint32_t ok = 0u;
            for (unsigned long long spins = 0; spins < kHelperSpinLimit; ++spins) {
              const uint32_t f2 = ld_relaxed(&st.hctl2[(h % kFlagReps) * kFlagStride]);
              if (f2 == v) { ok = 1u; break; }
              if (ld_relaxed(&st.hctl[(h % kFlagReps) * kFlagStride]) == 0xFFFFFFFFu) break;      // the loop is over
              __builtin_amdgcn_s_sleep(1);
            }
            s.job2_ok = ok;
          }
          __syncthreads();
          if (!s.job2_ok) return false;                      // (read by every thread behind the barrier; rewritten by the next job's wait, barriers later)
          after_flag();
          const unsigned long long* jb2 = st.hjob2 + (size_t)(v % kJobBufs) * kJob2Words;
          if ((uint32_t)tid >= 1u && (uint32_t)tid < kJob2Words) {
            const unsigned long long w2 = ld_agent(jb2 + tid);
            const int c = (tid - 1) / 5, q = (tid - 1) % 5;
            if (q == 4) { s.r2bm[c][0] = __uint_as_float((uint32_t)w2); s.r2bm[c][1] = __uint_as_float((uint32_t)(w2 >> 32)); }
            else (q == 0 ? s.best_mn : q == 1 ? s.best_mx : q == 2 ? s.second_mn : s.second_mx)[c] = w2;
          }
          __syncthreads();
          got2 = true;
        }
        if (tid < BC_K(st.cfg)) { W.r2[tid].bmn = s.r2bm[tid][0]; W.r2[tid].bmx = s.r2bm[tid][1]; }
        __syncthreads();
        return true;
      };
#ifdef GLIA_HMT_PROFILE
      const bool went = score_chunk<true>(st, s, W, n, newcount, mid, prof);
      tsp = __builtin_readcyclecounter();
#else
      const bool went = score_chunk<true>(st, s, W, n, newcount, mid);
#endif
      if (!went) return;                                    // part 2 never came: the loop's workgroup notices the missing answers
      forest_chunk(st, s, W, n);
      SPH(7);
      if ((uint32_t)tid < n && W.hdr[tid].on)
        st_agent(&st.hvotes[W.hdr[tid].rec - ne0], ((unsigned long long)v << 32) | ((unsigned long long)(uint32_t)W.hdr[tid].model << 24) | (unsigned long long)(uint32_t)s.votes[tid]);
      __syncthreads();
      SPH(8);
#ifdef GLIA_HMT_PROFILE
      if (prof && tid == 0) atomicAdd(&g_pqprof[32 + 9], (unsigned long long)n);
#endif
    }
  }
}

// The two smallest of a set of distinct 64-bit keys over a wave: every lane brings its own (best, second); a butterfly over
// disjoint lane sets (quad_perm xor 1, xor 2, row_ror 4, row_ror 8 inside a row, then xor 16 / 32 across rows), so a key is never
// merged with itself.  All 64 lanes must call.  (Sixty lanes updating ONE 64-bit LDS word with atomicMin are a
// compare-and-swap loop of sixty rounds: measured 7-9 k cycles per contraction for the four extremes.)
__device__ __forceinline__ void top2_merge(unsigned long long& b, unsigned long long& s2, unsigned long long ob, unsigned long long os) {
  const unsigned long long nb = ob < b ? ob : b, mx = ob < b ? b : ob;
  unsigned long long ns = s2 < os ? s2 : os;
  ns = mx < ns ? mx : ns;
  b = nb; s2 = ns;
}
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp64(unsigned long long v) {
  const uint32_t lo = dpp_u32<CTRL, 0xf>((uint32_t)v), hi = dpp_u32<CTRL, 0xf>((uint32_t)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long xor64(unsigned long long v, int m) {
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ void wave_top2_min(unsigned long long& b, unsigned long long& s2) {
  top2_merge(b, s2, dpp64<0xB1>(b), dpp64<0xB1>(s2));
  top2_merge(b, s2, dpp64<0x4E>(b), dpp64<0x4E>(s2));
  top2_merge(b, s2, dpp64<0x124>(b), dpp64<0x124>(s2));
  top2_merge(b, s2, dpp64<0x128>(b), dpp64<0x128>(s2));
  top2_merge(b, s2, xor64(b, 16), xor64(s2, 16));
  top2_merge(b, s2, xor64(b, 32), xor64(s2, 32));
}

// The state is a by-value kernel argument and every function that takes it is inlined: its pointers are then known to be global
// pointers (see chan_of).  (Round 2 passed a pointer to the state because a by-value argument whose address escapes to a
// NON-inlined function is copied to private memory; the price was flat memory instructions throughout.)
__global__ __launch_bounds__(kBcThreads) void greedy_bc_kernel(const BcState st) {
  __shared__ BcShared s;
  if (blockIdx.x != 0) { bc_helper_loop(st, s); return; }
  const int tid = threadIdx.x;
  uint32_t hseq = 0;
  unsigned long long k = st.ctrl[0], ne = st.ctrl[1], pool_used = st.ctrl[2];
  uint32_t status = ST_RUN;
  const int fdim = st.cfg.fdim;
  const int K = BC_K(st.cfg);
  const ScoreWs W = ws_layout(st.cfg, s.pool);
  if (tid == 0) { s.pq.wln[0] = s.pq.wln[1] = 0; s.pq.ovf = 0; s.pq.spill = 0; s.lost = 0; s.pre_e = kNone; s.nlog = (uint32_t)feat::log_slots(st.cfg, s.logpos); s.cand.sal = -__builtin_inf(); s.cand.seq = 0; s.cand.arg = 0; }
  bool deferred = false;         // inserts of the last contraction are waiting in the worklist (uniform)
  for (int i = tid; i < glibc::kLog2TabWords; i += blockDim.x) s.log2tab[i] = i < 18 ? glibc::kLog2Head[i] : i < 18 + 128 ? glibc::kLog2Tab[i - 18] : glibc::kLog2Tab2[i - 18 - 128];
  for (int i = tid; i < kSetSlots; i += blockDim.x) { s.pq.set[0][i] = 0; s.pq.set[1][i] = 0; }
  for (uint32_t i = tid; i < 3u * kMarkSlots; i += kBcThreads) reinterpret_cast<uint32_t*>(s.pool + kWsBytes)[i] = 0u;      // the neighbour table
  __syncthreads();
  pq_top<kBcThreads>(st.pq, s.pq, tid);      // the root lives in LDS: rebuilt at every launch

#ifdef GLIA_HMT_PROFILE
  unsigned long long tph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define PH(i) do { if (tid == 0) { unsigned long long tn = __builtin_readcyclecounter(); tph[i] += tn - tlast; tlast = tn; } } while (0)
#else
#define PH(i) do {} while (0)
#endif
  for (unsigned long long it = 0; it < st.max_iters; ++it) {
    PH(7);
    if (tid == 0) {
      Key root = pq_root<kBcThreads>(s.pq);
      if (better(s.cand, root)) root = s.cand;          // (an empty candidate has seq 0 and -inf: never better)
      s.cand.sal = -__builtin_inf(); s.cand.seq = 0; s.cand.arg = 0;
      s.stop = ST_RUN; s.newcount = 0;
      for (int c = 0; c < kMaxChannels; ++c) {
        s.best_mn[c] = s.second_mn[c] = ~0ull; s.best_mx[c] = s.second_mx[c] = 0ull;
        s.ex[c][0] = s.ex[c][2] = 0xFFFFFFFFu; s.ex[c][1] = s.ex[c][3] = 0u;
      }
      const bool forced = st.forced != nullptr;
      if (s.lost) s.stop = ST_BAD_SALIENCY;
      else if (forced ? (k >= st.forced_n) : (root.seq == 0)) s.stop = ST_DONE;
      else {
        const uint32_t e = forced ? kNone : root.arg;
        s.e = e;
        if (!forced && s.pre_e == e) {                   // (two dependent round trips of one thread, already made)
          s.r0 = s.pre_r0; s.r1 = s.pre_r1; s.len0 = s.pre_len0; s.len1 = s.pre_len1; s.off0 = s.pre_off0; s.off1 = s.pre_off1;
        } else {
          if (forced) { s.r0 = st.forced[2 * k]; s.r1 = st.forced[2 * k + 1]; }
          else { s.r0 = st.e_u[e]; s.r1 = st.e_v[e]; }
          s.len0 = st.adj_len[s.r0]; s.len1 = st.adj_len[s.r1];
          s.off0 = st.adj_off[s.r0]; s.off1 = st.adj_off[s.r1];
        }
        s.pre_e = kNone;
        const unsigned long long tot = (unsigned long long)s.len0 + s.len1;
        if (ne + tot > st.Ecap) s.stop = ST_NEED_EDGES;
        else if (pool_used + tot > st.pool_cap) s.stop = ST_NEED_POOL;
        else {
          st.order[3 * k + 0] = s.r0; st.order[3 * k + 1] = s.r1; st.order[3 * k + 2] = st.R0 + (uint32_t)k;
          st.sal_out[k] = forced ? 0.0 : root.sal;
        }
      }
    }
    __syncthreads();
    if (s.stop != ST_RUN) { status = s.stop; break; }
    const bool forced = st.forced != nullptr;
    if (forced) {      // the record between the two regions, if any
      for (uint32_t i = tid; i < s.len0; i += kBcThreads) {
        const uint32_t eid = st.pool[s.off0 + i];
        if (!st.e_alive[eid]) continue;
        const uint32_t u = st.e_u[eid], v = st.e_v[eid];
        if ((u == s.r0 && v == s.r1) || (u == s.r1 && v == s.r0)) s.e = eid;
      }
      __syncthreads();
    }
    const uint32_t r0 = s.r0, r1 = s.r1, e = s.e, len0 = s.len0, len1 = s.len1, off0 = s.off0, off1 = s.off1;
    const uint32_t r2 = st.R0 + (uint32_t)k;
    const uint32_t total = len0 + len1;
    const uint32_t r2off = (uint32_t)pool_used;

    // ---- optional: the feature vector the popped edge was scored with (main_merge_order_bc.cxx:148-157) ----
    if (st.feats_out) {
      for (uint32_t i = tid; i < total; i += kBcThreads) {
        const bool side1 = i >= len0;
        const uint32_t r = side1 ? r1 : r0;
        const uint32_t eid = st.pool[side1 ? off1 + (i - len0) : off0 + i];
        if (eid == e || !st.e_alive[eid]) continue;
        const int sel = st.e_u[eid] == r ? 0 : 2;
        for (int c = 0; c < K; ++c) {
          const float* d = &chan_of(st, c).e_dir[(size_t)eid * 4 + sel];
          atomicMin(&s.ex[c][side1 ? 2 : 0], float_ord(d[0]));
          atomicMax(&s.ex[c][side1 ? 3 : 1], float_ord(d[1]));
        }
      }
      __syncthreads();
      // The vector, as the scoring rounds build it: the shared boundary sets one lane per channel, then every HALF of a block
      // (bc_features.hpp) by the first lane of one wave, straight into the idle scoring workspace -- one thread with the vector in a
      // private array (scratch memory) took 40 of the 70 microseconds a bc_feat row cost.
      EStats* shv = reinterpret_cast<EStats*>(W.in);                 // [K] (a RecIn holds four of them)
      static_assert(sizeof(RecIn) >= kMaxChannels * sizeof(EStats), "shared sets fit the first staging slot");
      if (tid < K) shared_boundary(st, tid, e, shv[tid]);
      __syncthreads();
      if ((tid & 63) == 0) {
        const bool keep = forced || st.e_orient[e];        // bc_feat: (x0, x1) as given
        float ex[4 * kMaxChannels];
#pragma unroll
        for (int c = 0; c < kMaxChannels; ++c) {
          float m0 = 0.f, x0 = 0.f, m1 = 0.f, x1 = 0.f;
          if (c < K) {
            m0 = fminf(chan_of(st, c).Bn[r0].mn, ord_float(s.ex[c][0])); x0 = fmaxf(chan_of(st, c).Bn[r0].mx, ord_float(s.ex[c][1]));
            m1 = fminf(chan_of(st, c).Bn[r1].mn, ord_float(s.ex[c][2])); x1 = fmaxf(chan_of(st, c).Bn[r1].mx, ord_float(s.ex[c][3]));
          }
          ex[4 * c + 0] = keep ? m0 : m1; ex[4 * c + 1] = keep ? x0 : x1; ex[4 * c + 2] = keep ? m1 : m0; ex[4 * c + 3] = keep ? x1 : x0;
        }
        const int wave = tid >> 6;
        const GlobalView v{&st, keep ? r0 : r1, keep ? r1 : r0, e, shv};
        edge_features(st.cfg, v, ex, W.feat, nullptr, 1 << (wave & 3), false, 1 + (wave >> 2));
      }
      __syncthreads();
      if (tid == 0) feat::finish_features(st.cfg, W.feat);
      __syncthreads();
      for (int i = tid; i < fdim; i += kBcThreads) st.feats_out[(size_t)k * fdim + i] = W.feat[i];
      __syncthreads();
    }

    PH(0);
    // ---- the merged region (TRegionMap::merge, type/region_map.hxx:113-118) ----
    // one (channel, statistics set) pair per wave 1.. lane: each is two loads and a store, all in flight together.  Region and
    // record statistics, list headers and set extremes are what the helper workgroups read: stored write-through (st_agent).
    if (tid >= 64 && tid < 64 + 3 * K) {
      const int c = (tid - 64) / 3, kind = (tid - 64) % 3;
      const BcChan ch = chan_of(st, c);
      if (kind == 0) {
        PStats p = ch.pts[r0];
        const PStats q = ch.pts[r1];
        pstats_add(p, q);
        st_agent_struct(&ch.pts[r2], p);
        for (int b = 0; b < GLIA_HMT_MAX_BINS; ++b) s.r2hist[c][0][b] = p.hist[b];
        s.r2n[c][0] = p.n;
      } else if (kind == 1) {
        EStats bn = ch.Bn[r0];
        const EStats q = ch.Bn[r1];
        estats_add(bn, q);
        st_agent_struct(&ch.Bn[r2], bn);
        s.r2bn[c][0] = bn.mn; s.r2bn[c][1] = bn.mx;
      } else {
        EStats bt = ch.Bt[r0];
        const EStats q = ch.Bt[r1];
        const EStats a = ch.e_A[e != kNone ? e : 0u];
        estats_add(bt, q);
        if (e != kNone) estats_sub_additive(bt, a);
        st_agent_struct(&ch.Bt[r2], bt);
        for (int b = 0; b < GLIA_HMT_MAX_BINS; ++b) s.r2hist[c][1][b] = bt.hist[b];
        s.r2n[c][1] = bt.n;
      }
    }
    if (tid == 0) {
      st_agent(&st.parent[r0], r2); st_agent(&st.parent[r1], r2); st_agent(&st.parent[r2], r2);
      if (e != kNone) { st.e_alive[e] = 0; if (!forced) { st.pq.leaf_seq[e] = 0; pq_touch(st.pq, s.pq, 0, 0, e);   /* the root is the maximum of its node */ } }
    }
    // ---- phase A: mark the neighbours of r0 / r1 with the record that reaches them ----
    const bool rows = !forced && st.clf.kind == 0 && st.n_helpers != 0 && total <= kJobMax;      // helpers score this contraction
    const bool small = total <= kMarkMax;                 // neighbour table in LDS
    uint32_t* mk = reinterpret_cast<uint32_t*>(s.pool + kWsBytes);
    uint32_t* mv0 = mk + kMarkSlots; uint32_t* mv1 = mv0 + kMarkSlots;
    // (what the first pass of this loop reads about its list entry, phase B's first pass needs again: kept in registers, two round
    // trips less in front of the new records)
    uint32_t k_eid = kNone, k_u = 0, k_v = 0, k_pu = 0, k_pv = 0;
    for (uint32_t i = tid; i < total; i += kBcThreads) {
      const bool side1 = i >= len0;
      const uint32_t eid = st.pool[side1 ? off1 + (i - len0) : off0 + i];
      const uint8_t alive = st.e_alive[eid];           // unconditional loads: one round trip (see greedy_common.hpp)
      const uint32_t u = st.e_u[eid], v = st.e_v[eid];
      if (i == (uint32_t)tid) { k_eid = eid; k_u = u; k_v = v; k_pu = st.e_posu[eid]; k_pv = st.e_posv[eid]; }
      if (eid == e || !alive) continue;
      const uint32_t r = side1 ? r1 : r0;
      const uint32_t rs = (u == r) ? v : u;
      if (small) {
        uint32_t h = (rs * 2654435761u) >> 21;
        for (;;) {
          const uint32_t old = atomicCAS(&mk[h], 0u, rs + 1u);
          if (old == 0u || old == rs + 1u) break;
          h = (h + 1u) & (kMarkSlots - 1u);
        }
        (side1 ? mv1 : mv0)[h] = eid + 1u;
      } else (side1 ? st.mark1 : st.mark0)[rs] = eid + 1u;
    }
    __syncthreads();

    PH(1);
    // The loop's own LDS workspace is idle while it builds records: it keeps, per new record, (rs, queue category, table flag)
    // and the r2 -> rs extremes of every channel there, for the top-two pass and for the queue inserts (no global re-reads).
    const uint32_t ldsCap = kWsBytes / (8u + 8u * (uint32_t)K);
    unsigned long long* smeta = reinterpret_cast<unsigned long long*>(s.pool);
    float2* sdd = reinterpret_cast<float2*>(s.pool + (size_t)ldsCap * 8u);
    // ---- phase B: one new record (rs, r2) per distinct neighbour ----
    // Three unconditional round trips (pool entry; the record; everything about the record, its partner and the neighbour): a load
    // behind a branch costs a round trip of its own, so the partner's statistics are fetched whether there is a partner or not
    // (without one they are the record's own: a cache hit).
    for (uint32_t i = tid; i < total; i += kBcThreads) {
      const bool side1 = i >= len0;
      uint32_t eid = k_eid, u = k_u, v = k_v, pu = k_pu, pv = k_pv;
      if (i != (uint32_t)tid) {                          // (a list of more than 512 entries: the later passes read again)
        eid = st.pool[side1 ? off1 + (i - len0) : off0 + i];
        u = st.e_u[eid]; v = st.e_v[eid]; pu = st.e_posu[eid]; pv = st.e_posv[eid];
      }
      if (eid == e) continue;
      const uint32_t r = side1 ? r1 : r0;
      if (u != r && v != r) continue;
      const uint32_t rs = (u == r) ? v : u;
      uint32_t mk0 = 0u, mk1 = 0u;
      if (small) {
        uint32_t h = (rs * 2654435761u) >> 21;
        for (;;) {
          const uint32_t key = mk[h];
          if (key == rs + 1u) { mk0 = mv0[h]; mk1 = mv1[h]; break; }
          if (key == 0u) break;                       // (a dead record's neighbour may be in nobody's list)
          h = (h + 1u) & (kMarkSlots - 1u);
        }
      } else { mk0 = st.mark0[rs]; mk1 = st.mark1[rs]; }
      // the record itself is the first (and maybe only) parent of the new record; a common neighbour is handled from
      // the r0 side, where the r1-side record is the second parent
      uint32_t partner = kNone;
      if (!side1) {
        if (mk0 != eid + 1u) continue;
        if (mk1) partner = mk1 - 1u;
      } else {
        if (mk1 != eid + 1u) continue;
        if (mk0 != 0u) continue;
      }
      const bool both = partner != kNone;
      const uint32_t a1 = both ? partner : eid;
      const uint32_t idx = atomicAdd(&s.newcount, 1u);
      const uint32_t newE = (uint32_t)ne + idx;
      const uint32_t posRs = (u == rs) ? pu : pv;
      const uint32_t offRs = st.adj_off[rs];
      const uint32_t u1 = st.e_u[a1], pu1 = st.e_posu[a1], pv1 = st.e_posv[a1];
      const uint32_t fh1 = st.e_fhead[eid], ft1 = st.e_ftail[eid], fh2 = st.e_fhead[a1], ft2 = st.e_ftail[a1];
      const uint8_t tb1 = st.e_table[eid], tb2 = st.e_table[a1];
      const uint32_t top1 = st.pq.lv[0].arg[eid / kFan], top2 = st.pq.lv[0].arg[a1 / kFan];
      // rs held two entries when it touched both r0 and r1: the new record reuses one, the other one is dead from now on
      const uint32_t posDead = both ? ((u1 == rs) ? pu1 : pv1) : kNone;
      // image statistics of the new record, channel by channel
      for (int c = 0; c < K; ++c) {
        const BcChan ch = chan_of(st, c);
        const EStats A1 = ch.e_A[eid], N1 = ch.e_NA[eid];
        const EStats A2 = ch.e_A[a1], N2 = ch.e_NA[a1];
        const float4 d1 = *reinterpret_cast<const float4*>(&ch.e_dir[(size_t)eid * 4]);
        const float4 d2 = *reinterpret_cast<const float4*>(&ch.e_dir[(size_t)a1 * 4]);
        EStats A, NA;
        estats_clear(A); estats_clear(NA);
        estats_add(A, A1); estats_add(NA, N1);
        if (both) { estats_add(A, A2); estats_add(NA, N2); }
        float d[4] = {__builtin_inff(), -__builtin_inff(), __builtin_inff(), -__builtin_inff()};   // rs->r2, r2->rs
        {
          const bool rsIsU = u == rs;                   // d1.x, d1.y = u->v
          d[0] = fminf(d[0], rsIsU ? d1.x : d1.z); d[1] = fmaxf(d[1], rsIsU ? d1.y : d1.w);
          d[2] = fminf(d[2], rsIsU ? d1.z : d1.x); d[3] = fmaxf(d[3], rsIsU ? d1.w : d1.y);
        }
        if (both) {
          const bool rsIsU = u1 == rs;
          d[0] = fminf(d[0], rsIsU ? d2.x : d2.z); d[1] = fmaxf(d[1], rsIsU ? d2.y : d2.w);
          d[2] = fminf(d[2], rsIsU ? d2.z : d2.x); d[3] = fmaxf(d[3], rsIsU ? d2.w : d2.y);
        }
        ch.e_A[newE] = A; ch.e_NA[newE] = NA;
        if (rows) {          // the helpers' copy: one packed row per record and channel
          unsigned long long* row = st.hrec + ((size_t)idx * K + c) * kRowWords;
          unsigned long long wa[14], wn[14];
          __builtin_memcpy(wa, &A, 112); __builtin_memcpy(wn, &NA, 112);
#pragma unroll
          for (int q = 0; q < 14; ++q) { st_agent(row + q, wa[q]); st_agent(row + 14 + q, wn[q]); }
        }
        if (idx < ldsCap) sdd[idx * K + c] = make_float2(d[2], d[3]);
        *reinterpret_cast<float4*>(&ch.e_dir[(size_t)newE * 4]) = make_float4(d[0], d[1], d[2], d[3]);
        st_agent(&ch.pool_dir[offRs + posRs], make_float2(d[0], d[1]));
        st_agent(&ch.pool_dir[r2off + idx], make_float2(d[2], d[3]));
        if (posDead != kNone) st_agent(&ch.pool_dir[offRs + posDead], make_float2(__builtin_inff(), -__builtin_inff()));
      }
      // the fragile-entry chains of the parents, concatenated in (r0 side, r1 side) order
      uint32_t fh = kNone, ft = kNone;
      if (fh1 != kNone) { fh = fh1; ft = ft1; }
      if (both && fh2 != kNone) {
        if (fh == kNone) { fh = fh2; ft = ft2; }
        else { st_agent(&st.le_next[ft], fh2); ft = ft2; }
      }
      // the record is the r1-side parent exactly when it was found from r1
      const bool t0 = !side1 && tb1, t1 = side1 ? (tb1 != 0) : (both && tb2);
      st.e_alive[eid] = 0;
      if (tb1 && !forced) { st.pq.leaf_seq[eid] = 0; if (top1 == eid) pq_touch(st.pq, s.pq, 0, 0, eid); }
      if (both) {
        st.e_alive[partner] = 0;
        if (tb2 && !forced) { st.pq.leaf_seq[partner] = 0; if (top2 == partner) pq_touch(st.pq, s.pq, 0, 0, partner); }
      }
      const uint32_t on = (t0 || t1) ? 1u : 0u;
      st.e_u[newE] = rs; st.e_v[newE] = r2; st.e_posu[newE] = posRs;
      st.e_alive[newE] = 1; st.e_table[newE] = (uint8_t)on; st.e_orient[newE] = 1;
      st.e_fhead[newE] = fh; st.e_ftail[newE] = ft;
      // queue position (only meaningful for table edges): reference visit order, see greedy.hip; the category rides in
      // posv's upper bits until the record has been scored
      const uint32_t cat = rs < r0 ? 0u : (t0 ? 1u : 2u);
      if (rows) {
        unsigned long long* row = st.hrec + (size_t)idx * K * kRowWords;
        st_agent(row + 28, (unsigned long long)rs | ((unsigned long long)posRs << 32));
        st_agent(row + 29, (unsigned long long)on | ((unsigned long long)fh << 32));
      }
      if (idx < ldsCap) smeta[idx] = (unsigned long long)rs | ((unsigned long long)(cat | (on << 2)) << 32);
      st.pq.leaf_seq[newE] = 0;
      st.pq.leaf_sal[newE] = -__builtin_inf();
      st.pool[offRs + posRs] = newE;
      st.pool[r2off + idx] = newE;
      st.e_posv[newE] = (rows && idx < ldsCap) ? idx : (idx | (cat << 30));      // (the helper route reads the category from smeta)
    }
    // r2's own histogram entropies (BcChan::entP / entB), worked out once -- every new record needs them -- by the last wave,
    // which like most of the workgroup has little to do in this phase: one lane per bin, 16 lanes per (channel, set)
    if (tid >= (int)kBcThreads - 64) {
      const uint32_t g = ((uint32_t)tid & 63u) >> 4, l16 = (uint32_t)tid & 15u;
      for (uint32_t wq = g; wq < 2u * (uint32_t)K; wq += 4u) {
        const int cc = (int)(wq >> 1); const bool isB = (wq & 1u) != 0u;
        const int bins = st.cfg.cbins[cc];
        const uint32_t cnt = s.r2hist[cc][isB ? 1 : 0][l16], n = s.r2n[cc][isB ? 1 : 0];
        const double t = (int)l16 < bins ? feat::entropy_term(cnt, n, st.cfg.libm_log2, s.log2tab) : 0.0;
        const double en = bin_chain(t, bins, true);
        if ((int)l16 == bins - 1) { const BcChan chq = chan_of(st, cc); st_agent(&(isB ? chq.entB : chq.entP)[r2], en); }
      }
    }
    __syncthreads();
    PH(2);
    const uint32_t newcount = s.newcount;
    const bool job = rows && newcount != 0u;
    if (job) {
      // Part 1 of the job -- records ne .. ne + newcount of region r2 -- goes out NOW: the helpers' first stages (rows, region
      // statistics, shared sets, entropies) need nothing of the top-two pass below and run beside it.
      hseq += 1u;
      unsigned long long* jb = st.hjob + (size_t)(hseq % kJobBufs) * kJobWords;
      if (tid == 0) st_agent(jb + 0, (unsigned long long)(uint32_t)ne | ((unsigned long long)newcount << 32));
      if (tid == 1) st_agent(jb + 1, (unsigned long long)r2 | ((unsigned long long)newcount << 32));
      if (tid == 2) st_agent(jb + 2, (unsigned long long)hseq);
    }
    stores_done();          // every wave: the records, rows and statistics of this contraction are acknowledged ...
    __syncthreads();        // ... before the flag
    if (job && (uint32_t)tid < kFlagReps) st_agent(&st.hctl[(uint32_t)tid * kFlagStride], hseq);
    // the last thread publishes r2's list header and boundary extremes after the pass below; it requests what it needs
    // from global memory now (Bn(r2) was written two barriers ago), not behind its own stores
    float pre_mn[kMaxChannels], pre_mx[kMaxChannels];
#pragma unroll
    for (int c = 0; c < kMaxChannels; ++c) { pre_mn[c] = 0.f; pre_mx[c] = 0.f; }
    if (tid == kBcThreads - 1) {
#pragma unroll
      for (int c = 0; c < kMaxChannels; ++c) if (c < K) { pre_mn[c] = s.r2bn[c][0]; pre_mx[c] = s.r2bn[c][1]; }
    }
    // r2's mutual boundary extremes (its entries r2 -> rs) per channel: the best and the runner-up of the new records, for B(r2)
    // and the "all but this record" queries of the vectors
    {
      unsigned long long kb[kMaxChannels][2], ks[kMaxChannels][2];
#pragma unroll
      for (int c = 0; c < kMaxChannels; ++c) { kb[c][0] = kb[c][1] = ~0ull; ks[c][0] = ks[c][1] = ~0ull; }
      for (uint32_t j = tid; j < newcount; j += kBcThreads) {
        const uint32_t rec = (uint32_t)ne + j;
        const bool lds = j < ldsCap;
        if (!small) { const uint32_t rs = lds ? (uint32_t)smeta[j] : st.e_u[rec]; st.mark0[rs] = 0; st.mark1[rs] = 0; }
#pragma unroll
        for (int c = 0; c < kMaxChannels; ++c) {
          if (c < K) {
            float2 nd;
            if (lds) nd = sdd[j * K + c];
            else { const float* g = &chan_of(st, c).e_dir[(size_t)rec * 4]; nd = make_float2(g[2], g[3]); }
            top2_merge(kb[c][0], ks[c][0], ((unsigned long long)float_ord(nd.x) << 32) | rec, ~0ull);
            top2_merge(kb[c][1], ks[c][1], ~(((unsigned long long)float_ord(nd.y) << 32) | rec), ~0ull);      // the largest = the smallest complement
          }
        }
      }
#pragma unroll
      for (int c = 0; c < kMaxChannels; ++c) {
        if (c < K) {
          wave_top2_min(kb[c][0], ks[c][0]);
          wave_top2_min(kb[c][1], ks[c][1]);
          if ((tid & 63) == 0) { s.t2[tid >> 6][c][0] = kb[c][0]; s.t2[tid >> 6][c][1] = ks[c][0]; s.t2[tid >> 6][c][2] = kb[c][1]; s.t2[tid >> 6][c][3] = ks[c][1]; }
        }
      }
    }
    PH(9);
    if (small) for (uint32_t i = tid; i < 3u * kMarkSlots; i += kBcThreads) mk[i] = 0u;      // the neighbour table is empty again
    __syncthreads();
    PH(10);
    if (tid < K) {
      unsigned long long b0 = ~0ull, s0 = ~0ull, b1 = ~0ull, s1 = ~0ull;
#pragma unroll
      for (int w = 0; w < kBcThreads / 64; ++w) { top2_merge(b0, s0, s.t2[w][tid][0], s.t2[w][tid][1]); top2_merge(b1, s1, s.t2[w][tid][2], s.t2[w][tid][3]); }
      s.best_mn[tid] = b0; s.second_mn[tid] = s0; s.best_mx[tid] = ~b1; s.second_mx[tid] = ~s1;      // (empty: ~0 / ~0 / 0 / 0, as the pop initialised them)
    }
    __syncthreads();
    if (tid == kBcThreads - 1) {
      st.adj_off[r2] = r2off; st.adj_len[r2] = newcount;
      st_agent(&st.hadj[r2], (unsigned long long)r2off | ((unsigned long long)newcount << 32));
#pragma unroll
      for (int c = 0; c < kMaxChannels; ++c)
        if (c < K) {
          float mn, mx;
          r2_extremes_of(s, newcount, c, kNone, pre_mn[c], pre_mx[c], mn, mx);
          st_agent(&chan_of(st, c).Bmn[r2], mn); st_agent(&chan_of(st, c).Bmx[r2], mx);
          if (job) st_agent(st.hjob2 + (size_t)(hseq % kJobBufs) * kJob2Words + 1 + 5 * c + 4, (unsigned long long)__float_as_uint(mn) | ((unsigned long long)__float_as_uint(mx) << 32));
        }
    }
    if (job) {
      // part 2 of the job: what the vector assembly needs of the top-two pass (the other words went out before it)
      unsigned long long* jb2 = st.hjob2 + (size_t)(hseq % kJobBufs) * kJob2Words;
      if (tid == 0) st_agent(jb2, (unsigned long long)hseq);
      if ((uint32_t)tid >= 1u && (uint32_t)tid < kJob2Words) {
        const int c = (tid - 1) / 5, q = (tid - 1) % 5;
        if (q < 4 && c < K) st_agent(jb2 + tid, (q == 0 ? s.best_mn : q == 1 ? s.best_mx : q == 2 ? s.second_mn : s.second_mx)[c]);
      }
    }
    PH(3);
    stores_done();          // every wave: what this contraction wrote since part 1 is acknowledged ...
    __syncthreads();        // ... before anybody (this workgroup's staging pass, a helper after the flag) reads it
    if (job && (uint32_t)tid < kFlagReps) st_agent(&st.hctl2[(uint32_t)tid * kFlagStride], hseq);

    PH(8);
    // ---- score the new table edges ----
    if (job) {
      // Helper workgroups score whole records (at most kJobMax: larger contractions take the branch below).
      PH(5);
      // while the helpers work: the priority tree is brought up to date for the removals of this contraction
      pq_propagate<kBcThreads>(st.pq, s.pq, tid);
      PH(6);
      // the tree's root is final now (this contraction's own records are the candidate `s.cand`): if it stays the best edge, the next
      // pop needs its regions and their lists -- the last thread, idle here, fetches them (read by thread 0 behind the next barrier)
      if (tid == kBcThreads - 1) {
        const Key rt = pq_root<kBcThreads>(s.pq);
        if (rt.seq != 0) {
          const uint32_t pe = rt.arg, pu = st.e_u[pe], pv = st.e_v[pe];
          s.pre_r0 = pu; s.pre_r1 = pv; s.pre_len0 = st.adj_len[pu]; s.pre_len1 = st.adj_len[pv]; s.pre_off0 = st.adj_off[pu]; s.pre_off1 = st.adj_off[pv];
          s.pre_e = pe;
        }
      }
      Key mine; mine.sal = -__builtin_inf(); mine.seq = 0; mine.arg = 0;
      for (uint32_t j = tid; j < newcount; j += kBcThreads) {
        const uint32_t rec = (uint32_t)ne + j;
        const bool lds = j < ldsCap;
        const unsigned long long meta = lds ? smeta[j] : 0ull;
        if (!(lds ? (uint32_t)((meta >> 34) & 1ull) : (uint32_t)st.e_table[rec])) continue;
        unsigned long long w = 0, spins = 0;
        for (;;) {
          w = ld_agent(&st.hvotes[j]);
          if ((uint32_t)(w >> 32) == hseq) break;
          __builtin_amdgcn_s_sleep(1);
          if (++spins > kHelperSpinLimit) { s.lost = 1u; break; }     // helpers lost: reported as a failed run at the next pop
        }
        const int model = (int)((w >> 24) & 0xFFu);
        const double sal = (double)(uint32_t)(w & 0xFFFFFFu) / (double)st.clf.f[model < st.clf.n_models ? model : 0].ntree;
        const uint32_t cat = lds ? (uint32_t)((meta >> 32) & 3ull) : st.e_posv[rec] >> 30;
        Key c; c.sal = sal; c.seq = ((k + 1ull) << 32) | ((unsigned long long)cat << 30) | (lds ? (uint32_t)meta : st.e_u[rec]); c.arg = rec;
        st.pq.leaf_sal[rec] = sal;
        st.pq.leaf_seq[rec] = c.seq;
        pq_leaf_added(st.pq, s.pq, rec);
        if (better(c, mine)) mine = c;
      }
      mine = wave_max(mine);
      if ((tid & 63) == 0) s.cand_part[tid >> 6] = mine;
      __syncthreads();
      if (tid == 0) {
        Key b = s.cand_part[0];
#pragma unroll
        for (int w = 1; w < kBcThreads / 64; ++w) if (better(s.cand_part[w], b)) b = s.cand_part[w];
        s.cand = b;
      }
      deferred = true;
      PH(11);
    } else if (!forced && newcount) {
      // the loop's own workgroup scores (no helpers, or the stub scorer of the tests), W.cap records per round
      for (uint32_t c0 = 0; c0 < newcount; c0 += W.cap) {
        const uint32_t cn = newcount - c0 < W.cap ? newcount - c0 : W.cap;
        if ((uint32_t)tid < cn) W.hdr[tid].rec = (uint32_t)ne + c0 + (uint32_t)tid;
        __syncthreads();
        stage_local(st, W, cn, r2);
        (void)score_chunk<false>(st, s, W, cn, newcount, []() { return true; });
        PH(4);
        if (st.clf.kind == 0) forest_chunk(st, s, W, cn);
        PH(5);
        if ((uint32_t)tid < cn && W.hdr[tid].model >= 0) {
          const uint32_t rec = W.hdr[tid].rec;
          const double sal = st.clf.kind == 1 ? 1.0 - W.feat[tid * W.fstride + st.clf.stub_index]
                                              : (double)s.votes[tid] / (double)st.clf.f[W.hdr[tid].model].ntree;
          const uint32_t cat = st.e_posv[rec] >> 30;
          st.pq.leaf_sal[rec] = sal;
          st.pq.leaf_seq[rec] = ((k + 1ull) << 32) | ((unsigned long long)cat << 30) | W.hdr[tid].rs;
          pq_leaf_added(st.pq, s.pq, rec);
        }
        __syncthreads();
      }
    }
    for (uint32_t j = (rows ? ldsCap : 0u) + (uint32_t)tid; j < newcount; j += kBcThreads) st.e_posv[(uint32_t)ne + j] &= 0x3FFFFFFFu;
    PH(3);
    // the tree is brought up to date here unless the records just inserted can wait for the next contraction's helper round
    // (their best is s.cand; the next pop looks at it)
    if (!forced && !(job && deferred)) pq_propagate<kBcThreads>(st.pq, s.pq, tid);
    else __syncthreads();
    if (!job) deferred = false;
    PH(6);
    k += 1; ne += newcount; pool_used += total;
  }
  if (deferred) pq_propagate<kBcThreads>(st.pq, s.pq, tid);      // the worklist lives in LDS: nothing may be left in it
  if ((uint32_t)tid < kFlagReps && st.n_helpers) st_release(&st.hctl[(uint32_t)tid * kFlagStride], 0xFFFFFFFFu);
  if (tid == 0) { st.ctrl[0] = k; st.ctrl[1] = ne; st.ctrl[2] = pool_used; st.ctrl[3] = status; }
#ifdef GLIA_HMT_PROFILE
  if (tid == 0) printf("[bc profile] pq propagations by dirty level-0 nodes (<=8, <=16, more): %llu %llu %llu\n", g_pqprof[28], g_pqprof[29], g_pqprof[30]);
  if (tid == 0) printf("[bc profile] helper 0 (cumulative cycles): descriptor %llu  S0 headers %llu  S1 words %llu  S2 shared+extremes %llu  S3 entropies %llu  S4 assembly %llu  S5 finish %llu  forest %llu  publish %llu  records %llu\n",
                       g_pqprof[38], g_pqprof[32], g_pqprof[33], g_pqprof[34], g_pqprof[35], g_pqprof[36], g_pqprof[37], g_pqprof[39], g_pqprof[40], g_pqprof[41]);
  if (tid == 0) printf("[bc profile] merges %llu: pop+feats_out %llu  region+mark %llu  build %llu  top2 %llu  store drain %llu  (top2: second-best pass %llu  table clear + barrier %llu)  local scoring %llu  forest/publish %llu  pq %llu  loop-top %llu  wait-for-votes %llu (cycles)\n", k, tph[0], tph[1], tph[2], tph[3] + tph[9] + tph[10], tph[8], tph[9], tph[10], tph[4], tph[5], tph[6], tph[7], tph[11]);
#endif
}

}  // namespace

// first voxel index of every leaf (the accumulation pass records its complement, R_FIRST) as a sort key
namespace {
__global__ void bc_first_keys(const uint32_t* rrec, uint32_t R, unsigned long long* keys, uint32_t* leaf) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R) return;
  const uint2 f = *reinterpret_cast<const uint2*>(&rrec[(size_t)i * kRegionWords + R_FIRST]);
  keys[i] = ~(((unsigned long long)f.y << 32) | f.x);
  leaf[i] = i;
}
}  // namespace

#ifndef GLIA_BC_ENTRY
#define GLIA_BC_ENTRY greedy_bc_generic        // (see hmt_internal.hpp: the Makefile builds two more instances with GLIA_LIBM_FIXED)
#endif
int GLIA_BC_ENTRY(const RagArrays& rag, const BcCfg& cfg, const DeviceClassifier& clf, hipStream_t stream,
              uint32_t* h_order, double* h_sal, double* h_feats, int64_t capacity, int64_t* n_merges,
              double* ms_table, double* ms_init, double* ms_loop, int64_t* n_scored, bool init_only,
              const uint32_t* h_forced, int64_t n_forced, int shard, int n_shards, double* h_scores) {
  const long long P = rag.P;
  const uint32_t R = (uint32_t)rag.R;
  *n_merges = 0;
  if (R == 0 || P == 0) return GLIA_HMT_OK;
  // the vector is assembled at FULL length (bc_full_dim) and the simple selection compacted in place afterwards: the full length
  // is what the per-thread buffers (double x[kMaxFeat]) hold, not cfg.fdim
  if (bc_full_dim(cfg) > kMaxFeat || cfg.fdim > kMaxFeat) {
    set_error("merge_order_bc: feature vector too long (" + std::to_string(bc_full_dim(cfg)) + " columns before --simpf, limit " + std::to_string(kMaxFeat) + ")");
    return GLIA_HMT_ERR_ARG;
  }
  hipEvent_t ev[4];
  for (auto& e : ev) GLIA_HIP_TRY(hipEventCreate(&e));
  GLIA_HIP_TRY(hipEventRecord(ev[0], stream));
  DeviceBuffers buf;
  int rc;
  BcState st;
  memset(&st, 0, sizeof(st));
  st.R0 = R; st.P = P; st.cfg = cfg; st.clf = clf;
  st.shard = (uint32_t)shard; st.n_shards = (uint32_t)(n_shards > 0 ? n_shards : 1);

  // reference region-map iteration order (rmap_order.cpp): leaves sorted by first voxel on the device, the hashtable replay on the host
  std::vector<uint32_t> lab(R), by_first(R), rank;
  uint32_t* h_stage = nullptr;                // page-locked: [by_first | labels | rank]
  {
    unsigned long long* k0; unsigned long long* k1; uint32_t* v0; uint32_t* v1;
    if ((rc = buf.get(&k0, R, false, stream))) return rc;
    if ((rc = buf.get(&k1, R, false, stream))) return rc;
    if ((rc = buf.get(&v0, R, false, stream))) return rc;
    if ((rc = buf.get(&v1, R, false, stream))) return rc;
    hipLaunchKernelGGL(bc_first_keys, dim3((R + 255) / 256), dim3(256), 0, stream, rag.d_rrec, R, k0, v0);
    size_t tmp = 0;
    GLIA_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp, k0, k1, v0, v1, (size_t)R, 0, 64, stream));
    char* d_tmp;
    if ((rc = buf.get(&d_tmp, tmp ? tmp : 16, false, stream))) return rc;
    GLIA_HIP_TRY(rocprim::radix_sort_pairs((void*)d_tmp, tmp, k0, k1, v0, v1, (size_t)R, 0, 64, stream));
    if ((rc = PinnedHost::mine().get((void**)&h_stage, sizeof(uint32_t) * 3 * (size_t)R))) return rc;
    GLIA_HIP_TRY(hipMemcpyAsync(h_stage, v1, sizeof(uint32_t) * R, hipMemcpyDeviceToHost, stream));
    GLIA_HIP_TRY(hipMemcpyAsync(h_stage + R, rag.d_rlabel, sizeof(uint32_t) * R, hipMemcpyDeviceToHost, stream));
    GLIA_HIP_TRY(hipStreamSynchronize(stream));
    std::copy(h_stage, h_stage + R, by_first.begin());
    std::copy(h_stage + R, h_stage + 2 * (size_t)R, lab.begin());
  }
  const bool trace = option("GLIA_HMT_TRACE");
  const auto tr0 = std::chrono::steady_clock::now();
  auto tr_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr0).count(); };
  rmap_ranks_ordered(lab, by_first, &rank);
  if (trace) fprintf(stderr, "[trace] greedy_bc: region-map order replay %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr0).count());
  uint32_t* d_rank;
  if ((rc = buf.get(&d_rank, R, false, stream))) return rc;
  std::copy(rank.begin(), rank.end(), h_stage + 2 * (size_t)R);
  GLIA_HIP_TRY(hipMemcpyAsync(d_rank, h_stage + 2 * (size_t)R, sizeof(uint32_t) * R, hipMemcpyHostToDevice, stream));
  if (trace) { fprintf(stderr, "[trace] greedy_bc: rank upload queued at %.2f ms\n", tr_ms()); (void)hipStreamSynchronize(stream); fprintf(stderr, "[trace] greedy_bc: rank upload done at %.2f ms\n", tr_ms()); }

  if ((rc = buf.get(&st.le_src, P, false, stream))) return rc;
  if ((rc = buf.get(&st.le_dst, P, false, stream))) return rc;
  for (int c = 0; c < cfg.K; ++c) if ((rc = buf.get(&st.ch[c].le_stats, P, false, stream))) return rc;
  if ((rc = buf.get(&st.le_next, P, false, stream))) return rc;
  if ((rc = buf.get(&st.le_mutual, P, false, stream))) return rc;
  if ((rc = buf.get(&st.le_start, (size_t)R + 1, false, stream))) return rc;
  if ((rc = buf.get(&st.nm_out, R, true, stream))) return rc;
  long long* partner; uint32_t* flag; uint32_t* eidx;
  if ((rc = buf.get(&partner, P, false, stream))) return rc;
  if ((rc = buf.get(&flag, P + 1, true, stream))) return rc;
  if ((rc = buf.get(&eidx, P + 1, false, stream))) return rc;
  if (trace) { (void)hipStreamSynchronize(stream); fprintf(stderr, "[trace] greedy_bc: leaf buffers taken (and zeroed) at %.2f ms\n", tr_ms()); }
  const unsigned gP = (unsigned)((P + 255) / 256);
  for (int c = 0; c < cfg.K; ++c)
    hipLaunchKernelGGL(bc_leaf_entries, dim3(gP), dim3(256), 0, stream, st, c, rag.d_pa, rag.d_pb, rag.c_prec[c], rag.d_rlabel, partner,
                       cfg.cbins[c], cfg.T);
  hipLaunchKernelGGL(bc_leaf_starts, dim3((R + 256) / 256), dim3(256), 0, stream, st, rag.d_pa, rag.d_rlabel);
  hipLaunchKernelGGL(bc_record_flags, dim3(gP), dim3(256), 0, stream, st, rag.d_pa, rag.d_pb, flag);
  {
    size_t tmp = 0;
    GLIA_HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, flag, eidx, 0u, (size_t)(P + 1), rocprim::plus<uint32_t>(), stream));
    char* d_tmp;
    if ((rc = buf.get(&d_tmp, tmp ? tmp : 16, false, stream))) return rc;
    GLIA_HIP_TRY(rocprim::exclusive_scan((void*)d_tmp, tmp, flag, eidx, 0u, (size_t)(P + 1), rocprim::plus<uint32_t>(), stream));
  }
  uint32_t E0 = 0;
  GLIA_HIP_TRY(hipMemcpyAsync(&E0, eidx + P, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  GLIA_HIP_TRY(hipStreamSynchronize(stream));
  if (trace) fprintf(stderr, "[trace] greedy_bc: leaf entries + record flags done at %.2f ms\n", tr_ms());
  if (E0 == 0) return GLIA_HMT_OK;

  st.Ecap = (uint32_t)std::min<unsigned long long>(0xFFFFFF00ull, (unsigned long long)E0 * 6ull + (1u << 16));
  st.pool_cap = (unsigned long long)E0 * 12ull + (1u << 16);
  const size_t R2 = 2 * (size_t)R;
  for (int c = 0; c < cfg.K; ++c) {
    BcChan& ch = st.ch[c];
    if ((rc = buf.get(&ch.pts, R2, false, stream))) return rc;
    if ((rc = buf.get(&ch.Bn, R2, false, stream))) return rc;
    if ((rc = buf.get(&ch.Bt, R2, false, stream))) return rc;
    if ((rc = buf.get(&ch.Bmn, R2, false, stream))) return rc;
    if ((rc = buf.get(&ch.Bmx, R2, false, stream))) return rc;
    if ((rc = buf.get(&ch.entP, R2, false, stream))) return rc;
    if ((rc = buf.get(&ch.entB, R2, false, stream))) return rc;
    if ((rc = buf.get(&ch.e_A, st.Ecap, false, stream))) return rc;
    if ((rc = buf.get(&ch.e_NA, st.Ecap, false, stream))) return rc;
    if ((rc = buf.get(&ch.e_dir, (size_t)st.Ecap * 4, false, stream))) return rc;
    if ((rc = buf.get(&ch.pool_dir, (size_t)st.pool_cap, false, stream))) return rc;
  }
  if ((rc = buf.get(&st.parent, R2, false, stream))) return rc;
  if ((rc = buf.get(&st.adj_off, R2, true, stream))) return rc;
  if ((rc = buf.get(&st.adj_len, R2 + 1, true, stream))) return rc;
  if ((rc = buf.get(&st.pool, st.pool_cap, false, stream))) return rc;
  if ((rc = buf.get(&st.e_u, st.Ecap, false, stream))) return rc;
  if ((rc = buf.get(&st.e_v, st.Ecap, false, stream))) return rc;
  if ((rc = buf.get(&st.e_posu, st.Ecap, false, stream))) return rc;
  if ((rc = buf.get(&st.e_posv, st.Ecap, false, stream))) return rc;
  if ((rc = buf.get(&st.e_alive, st.Ecap, false, stream))) return rc;
  if ((rc = buf.get(&st.e_table, st.Ecap, false, stream))) return rc;
  if ((rc = buf.get(&st.e_orient, st.Ecap, false, stream))) return rc;
  if ((rc = buf.get(&st.e_fhead, st.Ecap, false, stream))) return rc;
  if ((rc = buf.get(&st.e_ftail, st.Ecap, false, stream))) return rc;
  st.pq.nleaves = st.Ecap;
  if ((rc = buf.get(&st.pq.leaf_sal, st.Ecap, false, stream))) return rc;
  if ((rc = buf.get(&st.pq.leaf_seq, st.Ecap, false, stream))) return rc;
  if ((rc = buf.get(&st.mark0, R2, true, stream))) return rc;
  if ((rc = buf.get(&st.mark1, R2, true, stream))) return rc;
  if ((rc = buf.get(&st.order, 3 * (size_t)R, false, stream))) return rc;
  if ((rc = buf.get(&st.sal_out, (size_t)R, false, stream))) return rc;
  if (h_feats) { if ((rc = buf.get(&st.feats_out, (size_t)R * cfg.fdim, false, stream))) return rc; }
  if ((rc = buf.get(&st.hctl, (size_t)kFlagReps * kFlagStride, true, stream))) return rc;
  if ((rc = buf.get(&st.hrec, (size_t)kJobMax * cfg.K * kRowWords, false, stream))) return rc;
  if ((rc = buf.get(&st.hadj, R2, true, stream))) return rc;
  if ((rc = buf.get(&st.hjob, (size_t)kJobBufs * kJobWords, true, stream))) return rc;
  if ((rc = buf.get(&st.hctl2, (size_t)kFlagReps * kFlagStride, true, stream))) return rc;
  if ((rc = buf.get(&st.hjob2, (size_t)kJobBufs * kJob2Words, true, stream))) return rc;
  if ((rc = buf.get(&st.hvotes, kJobMax, true, stream))) return rc;
  {
    // helper workgroups that score whole records (only a real forest in a scoring run needs them): one per compute unit
    // beside the loop's; GLIA_HMT_HELPERS overrides
    std::string env;
    int nh = option("GLIA_HMT_HELPERS", &env) ? atoi(env.c_str()) : 255;
    if (nh < 0) nh = 0;
    if (nh > 1023) nh = 1023;
    // the loop's workgroup and its helpers talk through polled flags, so they must all be resident at once: never ask for
    // more workgroups than the device can hold of this kernel (a helper that still does not answer -- the device is shared --
    // is noticed by the spin limit and reported as a failed run)
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    GLIA_HIP_TRY(hipGetDevice(&dev));
    GLIA_HIP_TRY(hipGetDeviceProperties(&prop, dev));
    GLIA_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, greedy_bc_kernel, (int)kBcThreads, 0));
    const int resident = per_cu * prop.multiProcessorCount;
    if (resident < 1) { set_error("merge_order_bc: the loop kernel does not fit this device"); return GLIA_HMT_ERR_HIP; }
    if (nh > resident - 1) nh = resident - 1;
    st.n_helpers = (clf.kind == 0 && !h_forced && !init_only) ? (uint32_t)nh : 0u;
  }
  if ((rc = buf.get(&st.ctrl, 8, true, stream))) return rc;
  uint32_t* cursor;
  if ((rc = buf.get(&cursor, R2, true, stream))) return rc;

  if (trace) fprintf(stderr, "[trace] greedy_bc: buffers taken at %.2f ms\n", tr_ms());
  hipLaunchKernelGGL(bc_init_dead, dim3((st.Ecap + 255) / 256), dim3(256), 0, stream, st, 0u);
  for (int c = 0; c < cfg.K; ++c) {
    hipLaunchKernelGGL(bc_leaf_regions, dim3((R + 255) / 256), dim3(256), 0, stream, st, c, rag.c_rrec[c], cfg.cbins[c]);
    hipLaunchKernelGGL(bc_record_fill, dim3(gP), dim3(256), 0, stream, st, c, flag, eidx, partner, d_rank, st.adj_len);
  }
  {
    size_t tmp = 0;
    GLIA_HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, st.adj_len, st.adj_off, 0u, (size_t)R, rocprim::plus<uint32_t>(), stream));
    char* d_tmp;
    if ((rc = buf.get(&d_tmp, tmp ? tmp : 16, false, stream))) return rc;
    GLIA_HIP_TRY(rocprim::exclusive_scan((void*)d_tmp, tmp, st.adj_len, st.adj_off, 0u, (size_t)R, rocprim::plus<uint32_t>(), stream));
  }
  hipLaunchKernelGGL(bc_adj_fill, dim3((E0 + 255) / 256), dim3(256), 0, stream, st, E0, cursor);
  hipLaunchKernelGGL(bc_hadj_fill, dim3((R + 255) / 256), dim3(256), 0, stream, st);
  GLIA_HIP_TRY(hipGetLastError());
  GLIA_HIP_TRY(hipEventRecord(ev[1], stream));
  if (!h_forced) hipLaunchKernelGGL(bc_init_score, dim3((E0 + 127) / 128), dim3(128), 0, stream, st, E0);
  if (h_forced) {
    uint32_t* d_forced;
    if ((rc = buf.get(&d_forced, (size_t)2 * n_forced + 2, false, stream))) return rc;
    GLIA_HIP_TRY(hipMemcpyAsync(d_forced, h_forced, sizeof(uint32_t) * 2 * (size_t)n_forced, hipMemcpyHostToDevice, stream));
    st.forced = d_forced; st.forced_n = (unsigned long long)n_forced;
  }
  GLIA_HIP_TRY(hipGetLastError());
  if ((rc = pq_setup(buf, st.pq, stream))) return rc;
  unsigned long long ctrl[4] = {0, E0, 2ull * E0, ST_RUN};
  GLIA_HIP_TRY(hipMemcpyAsync(st.ctrl, ctrl, sizeof(ctrl), hipMemcpyHostToDevice, stream));
  GLIA_HIP_TRY(hipEventRecord(ev[2], stream));

  if (trace) fprintf(stderr, "[trace] greedy_bc: kernels launched at %.2f ms\n", tr_ms());
  if (trace) { (void)hipStreamSynchronize(stream); fprintf(stderr, "[trace] greedy_bc: set-up + initial scores %.2f ms since the replay started; %zu of %zu device blocks (%.1f MB) were not in the block cache\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr0).count(), buf.misses, buf.all.size(), (double)buf.miss_bytes / 1048576.0); }
  if (init_only) {     // features + scores of the initial table edges only (TBoundaryTable::init)
    uint32_t* d_cnt;
    if ((rc = buf.get(&d_cnt, 1, true, stream))) return rc;
    GLIA_HIP_TRY(hipEventSynchronize(ev[2]));
    float t01 = 0, t12 = 0;
    (void)hipEventElapsedTime(&t01, ev[0], ev[1]);
    (void)hipEventElapsedTime(&t12, ev[1], ev[2]);
    for (auto& e : ev) (void)hipEventDestroy(e);
    *ms_table = t01; *ms_init = t12; *ms_loop = 0;
    std::vector<uint8_t> tab(E0);
    if ((rc = copy_to_host_staged(tab.data(), st.e_table, E0))) return rc;
    int64_t nt = 0;
    for (uint8_t t : tab) nt += t;
    *n_scored = nt;
    *n_merges = (int64_t)E0;            // init-only calls report the number of records here
    if (h_scores && (rc = copy_to_host_staged(h_scores, st.pq.leaf_sal, sizeof(double) * E0))) return rc;
    return GLIA_HMT_OK;
  }
  st.max_iters = 1ull << 14;
  while (true) {
    GLIA_HIP_TRY(hipMemsetAsync(st.hctl, 0, (size_t)kFlagReps * kFlagStride * sizeof(uint32_t), stream));
    GLIA_HIP_TRY(hipMemsetAsync(st.hjob, 0, (size_t)kJobBufs * kJobWords * sizeof(unsigned long long), stream));
    GLIA_HIP_TRY(hipMemsetAsync(st.hctl2, 0, (size_t)kFlagReps * kFlagStride * sizeof(uint32_t), stream));
    GLIA_HIP_TRY(hipMemsetAsync(st.hjob2, 0, (size_t)kJobBufs * kJob2Words * sizeof(unsigned long long), stream));
    GLIA_HIP_TRY(hipMemsetAsync(st.hvotes, 0, (size_t)kJobMax * sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(greedy_bc_kernel, dim3(1 + st.n_helpers), dim3(kBcThreads), 0, stream, st);
    GLIA_HIP_TRY(hipGetLastError());
    GLIA_HIP_TRY(hipMemcpyAsync(ctrl, st.ctrl, sizeof(ctrl), hipMemcpyDeviceToHost, stream));
    GLIA_HIP_TRY(hipStreamSynchronize(stream));
    if (ctrl[3] == ST_RUN) continue;
    if (ctrl[3] == ST_DONE) break;
    if (ctrl[3] == ST_NEED_POOL) {
      const unsigned long long ncap = st.pool_cap * 2;
      if ((rc = buf.grow(&st.pool, (size_t)st.pool_cap, (size_t)ncap, stream))) return rc;
      for (int c = 0; c < cfg.K; ++c) if ((rc = buf.grow(&st.ch[c].pool_dir, (size_t)st.pool_cap, (size_t)ncap, stream))) return rc;
      st.pool_cap = ncap;
    } else if (ctrl[3] == ST_NEED_EDGES) {
      if (st.Ecap >= 0xFFFFFF00u) { set_error("greedy: more than 2^32 edge slots needed"); return GLIA_HMT_ERR_ARG; }
      const uint32_t ocap = st.Ecap;
      const uint32_t ncap = (uint32_t)std::min<unsigned long long>(0xFFFFFF00ull, (unsigned long long)ocap * 2ull);
      if ((rc = buf.grow(&st.e_u, ocap, ncap, stream))) return rc;
      if ((rc = buf.grow(&st.e_v, ocap, ncap, stream))) return rc;
      if ((rc = buf.grow(&st.e_posu, ocap, ncap, stream))) return rc;
      if ((rc = buf.grow(&st.e_posv, ocap, ncap, stream))) return rc;
      if ((rc = buf.grow(&st.e_alive, ocap, ncap, stream))) return rc;
      if ((rc = buf.grow(&st.e_table, ocap, ncap, stream))) return rc;
      if ((rc = buf.grow(&st.e_orient, ocap, ncap, stream))) return rc;
      for (int c = 0; c < cfg.K; ++c) {
        if ((rc = buf.grow(&st.ch[c].e_A, ocap, ncap, stream))) return rc;
        if ((rc = buf.grow(&st.ch[c].e_NA, ocap, ncap, stream))) return rc;
        if ((rc = buf.grow(&st.ch[c].e_dir, (size_t)ocap * 4, (size_t)ncap * 4, stream))) return rc;
      }
      if ((rc = buf.grow(&st.e_fhead, ocap, ncap, stream))) return rc;
      if ((rc = buf.grow(&st.e_ftail, ocap, ncap, stream))) return rc;
      if ((rc = buf.grow(&st.pq.leaf_sal, ocap, ncap, stream))) return rc;
      if ((rc = buf.grow(&st.pq.leaf_seq, ocap, ncap, stream))) return rc;
      st.Ecap = ncap; st.pq.nleaves = ncap;
      hipLaunchKernelGGL(bc_init_dead, dim3((ncap - ocap + 255) / 256), dim3(256), 0, stream, st, ocap);
      if ((rc = pq_setup(buf, st.pq, stream))) return rc;
    }
    unsigned long long zero = ST_RUN;
    GLIA_HIP_TRY(hipMemcpyAsync(st.ctrl + 3, &zero, sizeof(zero), hipMemcpyHostToDevice, stream));
  }
  GLIA_HIP_TRY(hipEventRecord(ev[3], stream));
  GLIA_HIP_TRY(hipEventSynchronize(ev[3]));
  float t01 = 0, t12 = 0, t23 = 0;
  (void)hipEventElapsedTime(&t01, ev[0], ev[1]);
  (void)hipEventElapsedTime(&t12, ev[1], ev[2]);
  (void)hipEventElapsedTime(&t23, ev[2], ev[3]);
  for (auto& e : ev) (void)hipEventDestroy(e);
  *ms_table = t01; *ms_init = t12; *ms_loop = t23;
  const int64_t n = (int64_t)ctrl[0];
  *n_scored = (int64_t)ctrl[1];
  if (n > capacity) { set_error("merge_order_bc: output capacity too small"); return GLIA_HMT_ERR_CAPACITY; }
  if (n) {
    if ((rc = copy_to_host_staged(h_order, st.order, sizeof(uint32_t) * 3 * n))) return rc;
    if ((rc = copy_to_host_staged(h_sal, st.sal_out, sizeof(double) * n))) return rc;
    if (h_feats && (rc = copy_to_host_staged(h_feats, st.feats_out, sizeof(double) * (size_t)n * cfg.fdim))) return rc;
  }
  *n_merges = n;
  return GLIA_HMT_OK;
}

}  // namespace glia
