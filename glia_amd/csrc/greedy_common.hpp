// glia_amd/csrc/greedy_common.hpp -- pieces shared by the greedy merge kernels (pb-mean and classifier linkage):
// the 64-ary tournament tree used as priority queue and small host helpers.
#pragma once
#include <cstring>
#include <mutex>

#include "hmt_internal.hpp"
#include "skew.hpp"       // (after every library header: the skew build wraps the workgroup barriers of OUR kernels only)

namespace glia {

#ifndef GLIA_PQ_LANE_CHILDREN
#define GLIA_PQ_LANE_CHILDREN 4
#endif
constexpr int kLaneChildren = GLIA_PQ_LANE_CHILDREN;   // children of a node per lane
constexpr int kFan = 64 * kLaneChildren;              // fewer, fatter levels: every level is a dependent round trip
constexpr int kMaxLevels = 6;
constexpr uint32_t kTopMax = 4096;   // the last stored level has at most this many nodes: the whole workgroup reduces it in one step
#ifndef GLIA_GREEDY_THREADS
#define GLIA_GREEDY_THREADS 512
#endif
constexpr int kGreedyThreads = GLIA_GREEDY_THREADS;
constexpr int kWorkCap = 2048;       // >= kSetSlots: every list entry owns a set entry, so the set fills up first
constexpr uint32_t kNone = 0xFFFFFFFFu;

struct PqLevel {
  double* sal;
  unsigned long long* seq;
  uint32_t* arg;
  uint32_t* dirty;
  uint32_t size;
};


enum { ST_RUN = 0, ST_DONE = 1, ST_NEED_EDGES = 2, ST_NEED_POOL = 3, ST_BAD_SALIENCY = 4, ST_NEED_VALUES = 5, ST_NEED_TREE = 6, ST_INTERNAL = 7, ST_REBASE = 8 };

struct Key { double sal; unsigned long long seq; uint32_t arg; };

__device__ __forceinline__ bool better(const Key& a, const Key& b) {
  return (a.sal > b.sal) | ((a.sal == b.sal) & (a.seq > b.seq));      // no short-circuit: three compares, no branches
}

// 64-lane maximum by (saliency, seq), returned to every lane.  Hand-written over DPP: a generic reduction of the 20-byte
// key selects between two structs through private memory (a scratch round trip per step, ~4000 cycles per call);
// here every step is five v_mov_dpp plus compares and selects on registers.  max is idempotent, so the lanes a
// row_bcast step does not write (row_mask) simply combine with themselves.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ void key_max_step(Key& k) {
  const unsigned long long sb = (unsigned long long)__double_as_longlong(k.sal);
  const uint32_t s_lo = dpp_u32<CTRL, ROW_MASK>((uint32_t)sb), s_hi = dpp_u32<CTRL, ROW_MASK>((uint32_t)(sb >> 32));
  const uint32_t q_lo = dpp_u32<CTRL, ROW_MASK>((uint32_t)k.seq), q_hi = dpp_u32<CTRL, ROW_MASK>((uint32_t)(k.seq >> 32));
  const uint32_t a = dpp_u32<CTRL, ROW_MASK>(k.arg);
  const double osal = __longlong_as_double((long long)(((unsigned long long)s_hi << 32) | s_lo));
  const unsigned long long oseq = ((unsigned long long)q_hi << 32) | q_lo;
  const bool b = (osal > k.sal) | ((osal == k.sal) & (oseq > k.seq));
  k.sal = b ? osal : k.sal;
  k.seq = b ? oseq : k.seq;
  k.arg = b ? a : k.arg;
}
__device__ __forceinline__ Key wave_max(Key k) {
  key_max_step<0xB1>(k);          // quad_perm [1,0,3,2]
  key_max_step<0x4E>(k);          // quad_perm [2,3,0,1]
  key_max_step<0x124>(k);         // row_ror 4
  key_max_step<0x128>(k);         // row_ror 8: every lane holds its row's maximum
  key_max_step<0x142, 0xa>(k);    // row_bcast 15 into rows 1 and 3
  key_max_step<0x143, 0xc>(k);    // row_bcast 31 into rows 2 and 3: lane 63 holds the wave's maximum
  const unsigned long long sb = (unsigned long long)__double_as_longlong(k.sal);
  const uint32_t s_lo = __builtin_amdgcn_readlane((int)(uint32_t)sb, 63), s_hi = __builtin_amdgcn_readlane((int)(uint32_t)(sb >> 32), 63);
  const uint32_t q_lo = __builtin_amdgcn_readlane((int)(uint32_t)k.seq, 63), q_hi = __builtin_amdgcn_readlane((int)(uint32_t)(k.seq >> 32), 63);
  Key out;
  out.sal = __longlong_as_double((long long)(((unsigned long long)s_hi << 32) | s_lo));
  out.seq = ((unsigned long long)q_hi << 32) | q_lo;
  out.arg = (uint32_t)__builtin_amdgcn_readlane((int)k.arg, 63);
  return out;
}

// The same maximum, cheaper when saliency ties inside a wave are uncommon: reduce the order-preserving integer image of the
// saliency alone (two DPP moves, one 64-bit compare and two selects per step instead of five moves, three compares and five
// selects), then fetch the winner's seq and arg with v_readlane; only if several lanes share the largest saliency a
// second reduction over their seq decides.  Same result as wave_max for keys whose (sal, seq) pairs are distinct or empty.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ void u64_max_step(unsigned long long& v) {
  const uint32_t lo = dpp_u32<CTRL, ROW_MASK>((uint32_t)v), hi = dpp_u32<CTRL, ROW_MASK>((uint32_t)(v >> 32));
  const unsigned long long o = ((unsigned long long)hi << 32) | lo;
  v = o > v ? o : v;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  u64_max_step<0xB1>(v);
  u64_max_step<0x4E>(v);
  u64_max_step<0x124>(v);
  u64_max_step<0x128>(v);
  u64_max_step<0x142, 0xa>(v);
  u64_max_step<0x143, 0xc>(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ Key wave_max_sal_first(const Key& k) {
  unsigned long long ord = (unsigned long long)__double_as_longlong(k.sal);
  ord ^= (ord >> 63) ? ~0ull : 0x8000000000000000ull;                   // ascending doubles <-> ascending integers
  const unsigned long long m = wave_max_u64(ord);
  unsigned long long tied = __ballot(ord == m);
  if (__popcll(tied) > 1) {                                              // (uniform) equal saliencies: the largest seq among them
    const unsigned long long ms = wave_max_u64(ord == m ? k.seq : 0ull);
    tied = __ballot((ord == m) & (k.seq == ms));
  }
  const int src = (int)__builtin_ctzll(tied);                            // (at least one lane holds the maximum)
  const unsigned long long sb = (unsigned long long)__double_as_longlong(k.sal);
  Key out;
  const uint32_t s_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)sb, src), s_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(sb >> 32), src);
  const uint32_t q_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)k.seq, src), q_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(k.seq >> 32), src);
  out.sal = __longlong_as_double((long long)(((unsigned long long)s_hi << 32) | s_lo));
  out.seq = ((unsigned long long)q_hi << 32) | q_lo;
  out.arg = (uint32_t)__builtin_amdgcn_readlane((int)k.arg, src);
  return out;
}


// The tree: level 0 nodes are the parents of the leaves (edge slots); the last stored level (<= kTopMax nodes) has no
// stored parent -- pq_top() reduces it with the whole workgroup into PqWork::part (LDS), from where pop reads the root.
struct PqTree {
  uint32_t nleaves;
  double* leaf_sal;
  unsigned long long* leaf_seq;   // 0 = dead
  int nlevels;
  PqLevel lv[kMaxLevels];
  // big contractions dirty more nodes than the LDS worklist holds: the rest goes to these lists (deduplicated by the
  // levels' dirty flags), [2] = current / next level
  uint32_t* glist[2];
  uint32_t* gcount;
};

// recompute node j of level l with one wave (all 64 lanes must call); returns true when the node's key changed
// topk (optional): the last stored level kept in LDS for the lifetime of a launch (pq_top_load / pq_top_store); nodes of
// that level are then read and written there
__device__ __forceinline__ bool pq_recompute_node(const PqTree& t, int l, uint32_t j, int lane, bool force = false, Key* topk = nullptr) {
  Key k;
  k.sal = -__builtin_inf(); k.seq = 0; k.arg = 0;
  // every lane takes kLaneChildren children.  The loads are unconditional (index clamped, validity applied afterwards):
  // a load inside `if (ci < size)` gets its own basic block and its own s_waitcnt -- the "one round trip" became one per
  // child (measured: 4.6 k cycles per node instead of ~1.5 k).
  Key kk[kLaneChildren];
  if (l == 0) {
    unsigned long long q[kLaneChildren];
    double v[kLaneChildren];
#pragma unroll
    for (int c = 0; c < kLaneChildren; ++c) {
      const uint32_t ci = j * kFan + c * 64 + lane, cj = ci < t.nleaves ? ci : t.nleaves - 1u;
      q[c] = t.leaf_seq[cj]; v[c] = t.leaf_sal[cj];
    }
#pragma unroll
    for (int c = 0; c < kLaneChildren; ++c) {
      const uint32_t ci = j * kFan + c * 64 + lane;
      const bool ok = ci < t.nleaves && q[c] != 0;
      kk[c].seq = ok ? q[c] : 0ull; kk[c].sal = ok ? v[c] : -__builtin_inf(); kk[c].arg = ci < t.nleaves ? ci : 0u;
    }
  } else {
    const PqLevel& cl = t.lv[l - 1];
#pragma unroll
    for (int c = 0; c < kLaneChildren; ++c) {
      const uint32_t ci = j * kFan + c * 64 + lane, cj = ci < cl.size ? ci : cl.size - 1u;
      kk[c].sal = cl.sal[cj]; kk[c].seq = cl.seq[cj]; kk[c].arg = cl.arg[cj];
    }
#pragma unroll
    for (int c = 0; c < kLaneChildren; ++c) {
      const uint32_t ci = j * kFan + c * 64 + lane;
      if (ci >= cl.size) { kk[c].sal = -__builtin_inf(); kk[c].seq = 0; kk[c].arg = 0; }
    }
  }
#pragma unroll
  for (int c = 0; c < kLaneChildren; ++c) if (better(kk[c], k)) k = kk[c];
  const PqLevel& d = t.lv[l];
  const bool in_lds = topk != nullptr && l == t.nlevels - 1;
  // the node's previous key, fetched alongside the children (same round trip)
  unsigned long long oseq;
  uint32_t oarg;
  if (in_lds) { oseq = topk[j].seq; oarg = topk[j].arg; }
  else { oseq = d.seq[j]; oarg = d.arg[j]; }
  k = wave_max(k);
  bool changed = false;
  if (lane == 0) {
    changed = oseq != k.seq || oarg != k.arg;       // (seq, arg) identify the item; its saliency never changes
    if (changed || force) {
      if (in_lds) topk[j] = k;
      else { d.sal[j] = k.sal; d.seq[j] = k.seq; d.arg[j] = k.arg; }
    }
  }
  return changed;     // meaningful in lane 0
}

constexpr int kSetSlots = 1024;
struct PqWork {            // lives in LDS
  uint32_t ovf;
  uint32_t spill;                 // some node went to the global lists
  uint32_t fast[2][16];           // small propagations: the node each wave handles on the next level
  Key part[16];                   // per-wave maxima of the last stored level (pq_top); their maximum is the root
  uint32_t wln[2];
  uint32_t wl[2][kWorkCap];
  uint32_t set[2][kSetSlots];     // membership of wl[which] (node + 1, 0 = empty): no global round trip to dedupe
};

// A leaf that DIES only matters to its ancestors if it is the current maximum of its level-0 node; a non-maximal
// leaf can be dropped without touching the tree (the node keys stay exact).  A NEW leaf always re-evaluates its node.
__device__ __forceinline__ void pq_touch(const PqTree& t, PqWork& w, int level, int which, uint32_t child);
__device__ __forceinline__ void pq_leaf_removed(const PqTree& t, PqWork& w, uint32_t leaf) {
  if (t.lv[0].arg[leaf / kFan] == leaf) pq_touch(t, w, 0, 0, leaf);
}
__device__ __forceinline__ void pq_leaf_added(const PqTree& t, PqWork& w, uint32_t leaf) { pq_touch(t, w, 0, 0, leaf); }

__device__ __forceinline__ void pq_touch(const PqTree& t, PqWork& w, int level, int which, uint32_t child) {
  const uint32_t p = child / kFan;
  uint32_t h = (p * 2654435761u) >> 22;        // 10 bits
  for (int probe = 0; probe < 32; ++probe) {
    const uint32_t old = atomicCAS(&w.set[which][h], 0u, p + 1u);
    if (old == p + 1u) return;                 // already queued
    if (old == 0u) {
      const uint32_t i = atomicAdd(&w.wln[which], 1u);
      if (i < kWorkCap) w.wl[which][i] = p; else w.ovf = 1;
      return;
    }
    h = (h + 1) & (kSetSlots - 1);
  }
  // LDS set crowded (kWorkCap >= kSetSlots, so the list itself cannot fill up first): spill to the global list
  w.spill = 1u;
  if (atomicExch(&t.lv[level].dirty[p], 1u) == 0u) t.glist[which][atomicAdd(&t.gcount[which], 1u)] = p;
}

#ifdef GLIA_HMT_PROFILE
__device__ unsigned long long g_pqprof[48];     // [l] cycles of level l, [8+l] nodes recomputed at level l, [16+l] spilled nodes
#endif
// the root: one round trip over the last stored level by the whole workgroup (every thread calls; the caller has put a
// barrier after the last store into that level; ends with a barrier)
template <int THREADS>
__device__ __forceinline__ void pq_top(const PqTree& t, PqWork& w, int tid, const Key* topk = nullptr) {
  const PqLevel& cl = t.lv[t.nlevels - 1];
  Key k;
  k.sal = -__builtin_inf(); k.seq = 0; k.arg = 0;
#ifdef GLIA_HMT_PROFILE
  const unsigned long long tt0 = __builtin_readcyclecounter();
#endif
  constexpr int kBatch = 4;
  if (topk) {
    for (uint32_t ci = tid; ci < cl.size; ci += THREADS) { const Key c = topk[ci]; if (better(c, k)) k = c; }
  } else
  for (uint32_t base = 0; base < cl.size; base += THREADS * kBatch) {
    Key kk[kBatch];
#pragma unroll
    for (int c = 0; c < kBatch; ++c) {       // unconditional loads, see pq_recompute_node
      const uint32_t ci = base + c * THREADS + tid, cj = ci < cl.size ? ci : cl.size - 1u;
      kk[c].sal = cl.sal[cj]; kk[c].seq = cl.seq[cj]; kk[c].arg = cl.arg[cj];
    }
#pragma unroll
    for (int c = 0; c < kBatch; ++c) {
      const uint32_t ci = base + c * THREADS + tid;
      if (ci >= cl.size) { kk[c].sal = -__builtin_inf(); kk[c].seq = 0; kk[c].arg = 0; }
      if (better(kk[c], k)) k = kk[c];
    }
  }
#ifdef GLIA_HMT_PROFILE
  if (k.arg == 0xFFFFFFF0u) w.ovf = 1;      // keep the loads before the timestamp
  const unsigned long long tt1 = __builtin_readcyclecounter();
#endif
  k = wave_max(k);
#ifdef GLIA_HMT_PROFILE
  const unsigned long long tt2 = __builtin_readcyclecounter();
#endif
  if ((tid & 63) == 0) w.part[tid >> 6] = k;
  __syncthreads();
#ifdef GLIA_HMT_PROFILE
  if (tid == 0) { g_pqprof[24] += tt1 - tt0; g_pqprof[25] += tt2 - tt1; g_pqprof[26] += __builtin_readcyclecounter() - tt2; g_pqprof[27] += 1; }
#endif
}
template <int THREADS>
__device__ __forceinline__ Key pq_root(const PqWork& w) {
  Key b = w.part[0];
#pragma unroll
  for (int j = 1; j < THREADS / 64; ++j) { const Key c = w.part[j]; if (better(c, b)) b = c; }
  return b;
}

// the last stored level <-> LDS (every thread calls; the caller puts the barriers)
template <int THREADS>
__device__ __forceinline__ void pq_top_load(const PqTree& t, Key* topk, int tid) {
  const PqLevel& d = t.lv[t.nlevels - 1];
  for (uint32_t i = tid; i < d.size; i += THREADS) { Key k; k.sal = d.sal[i]; k.seq = d.seq[i]; k.arg = d.arg[i]; topk[i] = k; }
}
template <int THREADS>
__device__ __forceinline__ void pq_top_store(const PqTree& t, const Key* topk, int tid) {
  const PqLevel& d = t.lv[t.nlevels - 1];
  for (uint32_t i = tid; i < d.size; i += THREADS) { const Key k = topk[i]; d.sal[i] = k.sal; d.seq[i] = k.seq; d.arg[i] = k.arg; }
}

// apply all pending leaf changes level by level; every thread of the workgroup must call (contains barriers)
template <int THREADS>
__device__ __forceinline__ void pq_propagate(const PqTree& t, PqWork& w, int tid, Key* topk = nullptr) {
  const int lane = tid & 63, wave = tid >> 6;
  constexpr int nwaves = THREADS / 64;
  static_assert(nwaves <= 16, "PqWork::fast");
  __syncthreads();
#ifdef GLIA_HMT_PROFILE
  if (tid == 0) { const uint32_t nd = w.wln[0]; g_pqprof[28 + (nd <= (uint32_t)nwaves ? 0 : (nd <= 2u * nwaves ? 1 : 2))] += 1; }
#endif
  if (w.wln[0] <= (uint32_t)nwaves && !w.spill && !w.ovf) {
    // The usual case -- at most one dirty node per wave: every wave walks its node up the tree, one barrier per
    // level; waves whose node did not change, or whose parent is taken by a lower wave, drop out.
    const uint32_t n0 = w.wln[0];
    uint32_t node = (uint32_t)wave < n0 ? w.wl[0][wave] : kNone;
    const uint32_t node0 = node;
    for (int l = 0; l < t.nlevels; ++l) {
      uint32_t parent = kNone;
#ifdef GLIA_HMT_PROFILE
      const unsigned long long tf0 = __builtin_readcyclecounter();
#endif
      if (node != kNone) {
        const bool changed = pq_recompute_node(t, l, node, lane, false, topk);
        const bool ch = __shfl((int)changed, 0) != 0;
        if (ch && l + 1 < t.nlevels) parent = node / kFan;
      }
#ifdef GLIA_HMT_PROFILE
      const unsigned long long tf1 = __builtin_readcyclecounter();
#endif
      if (lane == 0) w.fast[l & 1][wave] = parent;
      __syncthreads();
#ifdef GLIA_HMT_PROFILE
      if (tid == 0) { g_pqprof[l] += tf1 - tf0; g_pqprof[8 + l] += __builtin_readcyclecounter() - tf1; g_pqprof[16 + l] += (node != kNone); }
#endif
      if (parent != kNone) for (int j = 0; j < wave; ++j) if (w.fast[l & 1][j] == parent) { parent = kNone; break; }
      node = parent;
    }
    if (node0 != kNone && lane == 0) {      // consume the worklist entry's membership slot (off the loads' critical path)
      uint32_t h = (node0 * 2654435761u) >> 22;
      while (w.set[0][h] != node0 + 1u) h = (h + 1) & (kSetSlots - 1);
      w.set[0][h] = 0;
    }
    if (tid == 0) w.wln[0] = 0;
    pq_top<THREADS>(t, w, tid, topk);
    return;
  }
  int cur = 0;
  for (int l = 0; l < t.nlevels; ++l) {
#ifdef GLIA_HMT_PROFILE
    const unsigned long long tl0 = __builtin_readcyclecounter();
#endif
    __syncthreads();
    const bool ovf = w.ovf != 0;
    const uint32_t n = ovf ? t.lv[l].size : w.wln[cur];
    const uint32_t ng = (ovf || !w.spill) ? 0u : t.gcount[cur];
    for (uint32_t i = wave; i < n; i += nwaves) {
      const uint32_t j = ovf ? i : w.wl[cur][i];
      const bool changed = pq_recompute_node(t, l, j, lane, ovf, topk);
      if (lane == 0) {
        // an unchanged node cannot change its ancestors
        if (!ovf && changed && l + 1 < t.nlevels) pq_touch(t, w, l + 1, cur ^ 1, j);
      }
    }
    for (uint32_t i = wave; i < ng; i += nwaves) {
      const uint32_t j = t.glist[cur][i];
      const bool changed = pq_recompute_node(t, l, j, lane, false, topk);
      if (lane == 0) {
        t.lv[l].dirty[j] = 0u;
        if (changed && l + 1 < t.nlevels) pq_touch(t, w, l + 1, cur ^ 1, j);
      }
    }
    __syncthreads();
    if (tid == 0 && ng) t.gcount[cur] = 0u;
    // consume the list: empty it and its membership set
    if (ovf) { for (int i = tid; i < kSetSlots; i += THREADS) { w.set[0][i] = 0; w.set[1][i] = 0; } }
    else {
      for (uint32_t i = tid; i < n; i += THREADS) {
        const uint32_t p = w.wl[cur][i];
        uint32_t h = (p * 2654435761u) >> 22;
        while (w.set[cur][h] != p + 1u) h = (h + 1) & (kSetSlots - 1);
        w.set[cur][h] = 0;
      }
    }
    __syncthreads();
    if (tid == 0) w.wln[cur] = 0;
#ifdef GLIA_HMT_PROFILE
    if (tid == 0) { g_pqprof[l] += __builtin_readcyclecounter() - tl0; g_pqprof[8 + l] += n; g_pqprof[16 + l] += ng; }
#endif
    cur ^= 1;
  }
  __syncthreads();
  if (tid == 0) { w.ovf = 0; w.spill = 0; w.wln[0] = w.wln[1] = 0; }
  pq_top<THREADS>(t, w, tid, topk);
}

__global__ void pq_build_level_kernel(PqTree t, int l);
int pq_setup(struct DeviceBuffers& buf, PqTree& t, hipStream_t stream);

inline __device__ double sdivide(double l, double r, double d) { return fabs(r) >= 2.22e-16 ? l / r : d; }   // glia_base.hxx:77-78

// Device allocations owned by one call.  Freed blocks go to a small process-wide cache instead of back to the driver: a merge
// order at 1024^3 allocates ~5 GB in a few dozen blocks, and hipMalloc / hipFree of that cost more than the edge table
// and the feature kernel together (measured: ~80 ms of a 1.36 s step).  Every call ends synchronised, so a cached block is idle.
// Page-locked staging memory for the few host <-> device copies of a call (one buffer per thread, grown on demand, never
// freed): an asynchronous copy to or from pageable memory makes the runtime lock the pages first, which was measured at up to
// 20 ms for 1 MB when the pages came fresh from the allocator.
struct PinnedHost {
  void* p = nullptr; size_t cap = 0;
  int get(void** out, size_t bytes) {
    if (bytes > cap) {
      if (p) (void)hipHostFree(p);
      p = nullptr; cap = 0;
      const size_t want = bytes + bytes / 2 + 4096;
      GLIA_HIP_TRY(hipHostMalloc(&p, want, hipHostMallocDefault));
      cap = want;
    }
    *out = p;
    return GLIA_HMT_OK;
  }
  static PinnedHost& mine() { static thread_local PinnedHost h; return h; }
};
// device -> caller's host buffer through the page-locked staging area, in pieces.  Copying straight into pageable memory makes
// the runtime register those pages with the driver; when the caller frees them (a numpy array going out of scope), the
// driver's invalidation of that registration stalls the process's GPU queues for ~20 ms -- measured in bench.py as "the next
// call after a result array was freed is slow".
inline int copy_to_host_staged(void* h_dst, const void* d_src, size_t bytes) {
  constexpr size_t kPiece = 8u << 20;
  char* stage = nullptr;
  int rc = PinnedHost::mine().get((void**)&stage, bytes < kPiece ? bytes : kPiece);
  if (rc) return rc;
  for (size_t off = 0; off < bytes; off += kPiece) {
    const size_t n = bytes - off < kPiece ? bytes - off : kPiece;
    GLIA_HIP_TRY(hipMemcpy(stage, (const char*)d_src + off, n, hipMemcpyDeviceToHost));
    memcpy((char*)h_dst + off, stage, n);
  }
  return GLIA_HMT_OK;
}
struct BlockCache {
  struct Block { void* p; size_t bytes; int device; };
  std::vector<Block> free_blocks;
  std::mutex mu;
  size_t cached = 0;
  static constexpr size_t kMaxCached = 12ull << 30;      // parked at most; glia_hmt_release_cached_memory() / the last context's end return all of it
  static BlockCache& get() { static BlockCache c; return c; }
  void* take(size_t bytes, int device) {
    std::lock_guard<std::mutex> lock(mu);
    size_t best = (size_t)-1;
    for (size_t i = 0; i < free_blocks.size(); ++i) {
      const Block& b = free_blocks[i];
      if (b.device == device && b.bytes >= bytes && b.bytes <= bytes + bytes / 2 + 4096 && (best == (size_t)-1 || b.bytes < free_blocks[best].bytes)) best = i;
    }
    if (best == (size_t)-1) return nullptr;
    void* p = free_blocks[best].p;
    cached -= free_blocks[best].bytes;
    free_blocks.erase(free_blocks.begin() + (long)best);
    return p;
  }
  size_t trim() {                                        // every parked block back to the driver; returns the bytes released
    std::lock_guard<std::mutex> lock(mu);
    const size_t was = cached;
    for (const Block& b : free_blocks) (void)hipFree(b.p);
    free_blocks.clear();
    cached = 0;
    return was;
  }
  void give(void* p, size_t bytes, int device) {
    std::lock_guard<std::mutex> lock(mu);
    if (cached + bytes > kMaxCached) { (void)hipFree(p); return; }
    free_blocks.push_back({p, bytes, device});
    cached += bytes;
  }
};
struct DeviceBuffers {
  std::vector<void*> all;
  std::vector<size_t> sizes;
  int device = -1;
  size_t misses = 0, miss_bytes = 0;         // allocations the block cache could not serve (GLIA_HMT_TRACE reports them)
  ~DeviceBuffers() {
    for (size_t i = 0; i < all.size(); ++i) BlockCache::get().give(all[i], sizes[i], device);
  }
  int raw(void** p, size_t bytes) {
    if (device < 0) GLIA_HIP_TRY(hipGetDevice(&device));
    bytes = (bytes + 255) & ~(size_t)255;
    void* q = BlockCache::get().take(bytes, device);
    if (!q) {
      ++misses; miss_bytes += bytes;
      hipError_t e = hipMalloc(&q, bytes);
      if (e != hipSuccess) {                     // out of memory with blocks parked in the cache: release them and retry
        (void)BlockCache::get().trim();
        (void)hipGetLastError();
        GLIA_HIP_TRY(hipMalloc(&q, bytes));
      }
    }
    *p = q;
    all.push_back(q); sizes.push_back(bytes);
    return GLIA_HMT_OK;
  }
  template <typename T> int get(T** p, size_t n, bool zero, hipStream_t s) {
    int rc = raw((void**)p, sizeof(T) * (n ? n : 1));
    if (rc) return rc;
    if (zero) GLIA_HIP_TRY(hipMemsetAsync(*p, 0, sizeof(T) * (n ? n : 1), s));
    return GLIA_HMT_OK;
  }
  template <typename T> int grow(T** p, size_t old_n, size_t new_n, hipStream_t s) {
    T* q = nullptr;
    GLIA_HIP_TRY(hipMalloc((void**)&q, sizeof(T) * (new_n ? new_n : 1)));
    GLIA_HIP_TRY(hipMemcpyAsync(q, *p, sizeof(T) * old_n, hipMemcpyDeviceToDevice, s));
    GLIA_HIP_TRY(hipStreamSynchronize(s));
    for (size_t i = 0; i < all.size(); ++i) if (all[i] == (void*)*p) { (void)hipFree(all[i]); all[i] = q; sizes[i] = sizeof(T) * (new_n ? new_n : 1); }
    *p = q;
    return GLIA_HMT_OK;
  }
};

// binary searches over the sorted compact RAG
__device__ __forceinline__ long long find_pair(const uint32_t* pa, const uint32_t* pb, long long P, uint32_t a, uint32_t b) {
  long long lo = 0, hi = P;
  while (lo < hi) {
    long long mid = (lo + hi) >> 1;
    if (pa[mid] < a || (pa[mid] == a && pb[mid] < b)) lo = mid + 1; else hi = mid;
  }
  return (lo < P && pa[lo] == a && pb[lo] == b) ? lo : -1;
}
__device__ __forceinline__ uint32_t find_label(const uint32_t* lab, uint32_t R, uint32_t key) {
  uint32_t lo = 0, hi = R;
  while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (lab[mid] < key) lo = mid + 1; else hi = mid; }
  return lo;
}

}  // namespace glia
