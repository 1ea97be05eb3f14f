// glia_amd/csrc/greedy.hip -- K4b + K5: edge table and the greedy agglomeration loop, on the device.
//
// Reference semantics reproduced (all under /root/reference/code/):
//   * TBoundaryTable::init (type/boundary_table.hxx:91-114): one edge per unordered label pair, only if BOTH
//     directed boundaries exist; queue inserts happen in lexicographic (r0,r1) order.
//   * TBoundaryTable::top (:46-52) on a std::multimap<double,...>: largest saliency, and among equal
//     saliencies the MOST RECENTLY INSERTED item.  Here every queue item carries a 64-bit sequence number
//     that is order-isomorphic to the reference's insertion order, and the queue orders by (saliency, seq).
//   * TBoundaryTable::update (:121-167): the table scan visits the neighbours rs of the merged pair
//     (r0 < r1) in this order:  rs < r0 ascending;  neighbours of r0 with rs > r0 ascending;  neighbours of
//     r1 only, ascending.  seq = (merge# + 1) << 32 | category << 30 | rs encodes exactly that, so no sort
//     is needed.  pData0s is always the (r0,rs) item, pData1s the (r1,rs) item (:139-153).
//   * genMergeOrderGreedy (util/struct_merge.hxx:13-33) and the mean linkage (:62-76):
//     d2 = sdivide(m0*n0 + m1*n1, n0+n1, 0) with the reference's operation order (no FMA contraction).
//
// MI355X mapping.  The loop is a chain of R-1 dependent contractions, so it runs as ONE persistent
// workgroup: all state (edge records, adjacency pool, priority structure) stays in HBM/L2, each
// contraction is data-parallel over the neighbours of the merged pair.  The priority queue is a
// 64-ary tournament tree over edge slots (one wave recomputes one node with a 64-lane reduction);
// insertions and deletions of one contraction are applied level by level from a dirty-node worklist.
// Dense region ids: leaves 0..R-1 ascending by label, merged regions R+k; the map id -> key is monotone,
// so every key comparison of the reference is an id comparison here.
#include <atomic>
#include <cstring>
#include <vector>
#include <limits>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include "greedy_common.hpp"

namespace glia {

__global__ void pq_build_level_kernel(PqTree t, int l) {
  const int lane = threadIdx.x & 63;
  uint32_t j = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (j < t.lv[l].size) (void)pq_recompute_node(t, l, j, lane, true);   // fresh memory: always write
}

// (re)allocates the tree levels above t.nleaves leaves and builds them from the current leaf keys
int pq_setup(DeviceBuffers& buf, PqTree& t, hipStream_t stream) {
  t.nlevels = 0;
  uint32_t n = t.nleaves;
  while (true) {
    n = (n + kFan - 1) / kFan;
    if (t.nlevels >= kMaxLevels) { set_error("greedy: edge table too large"); return GLIA_HMT_ERR_ARG; }
    PqLevel& L = t.lv[t.nlevels++];
    L.size = n;
    int rc;
    if ((rc = buf.get(&L.sal, n, false, stream))) return rc;
    if ((rc = buf.get(&L.seq, n, false, stream))) return rc;
    if ((rc = buf.get(&L.arg, n, false, stream))) return rc;
    if ((rc = buf.get(&L.dirty, n, true, stream))) return rc;
    if (n <= kTopMax) break;
  }
  {
    int rc;
    if ((rc = buf.get(&t.glist[0], t.lv[0].size, false, stream))) return rc;
    if ((rc = buf.get(&t.glist[1], t.lv[0].size, false, stream))) return rc;
    if ((rc = buf.get(&t.gcount, 2, true, stream))) return rc;
  }
  for (int l = 0; l < t.nlevels; ++l) {
    unsigned long long threads = (unsigned long long)t.lv[l].size * 64ull;
    hipLaunchKernelGGL(pq_build_level_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, t, l);
  }
  GLIA_HIP_TRY(hipGetLastError());
  return GLIA_HMT_OK;
}

// A barrier behind which every global store and atomic of the workgroup has been performed.  (__syncthreads() is NOT that on
// gfx950: the workgroup-scope fence of a workgroup that is not split over CUs waits for lgkmcnt only -- found in round 3, when
// a merge order differed once in ~30 000 runs; the comments of rounds 1-2 that say "vmcnt(0) inside" were wishful.)
__device__ __forceinline__ void full_barrier(const int line = __builtin_LINE()) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); GLIA_SKEW_DELAY(line); }
struct GreedyState {
  uint32_t R0;
  uint32_t* adj_off;   // [2*R0] start of a region's incident-edge list in pool
  uint32_t* adj_len;   // [2*R0] slots in that list (live edges + tombstones)
  uint2* pool;         // incident-edge lists: (edge slot | kNone tombstone, the neighbour it leads to)
  unsigned long long pool_cap;
  uint32_t Ecap;
  uint32_t *e_u, *e_v, *e_posu, *e_posv;
  double* e_mean;                 // mean linkage: boundary mean; median linkage: the current median
  int* e_n;
  // median linkage (util/struct_merge.hxx:90-136): every edge owns a SORTED run of its boundary values in `vals`
  float* vals;
  unsigned long long vals_cap;
  unsigned long long* e_off;      // [Ecap] start of the edge's run
  unsigned long long* rbv;        // [2*R0] values held by the region's incident edges (capacity pre-check)
  int size_weight;                // ...AndMinSize linkage (util/struct_merge.hxx:141-185): saliency = -median * min(region sizes)
  PqTree pq;
  // pre_merge condition (gadget/main_pre_merge.cxx:27-76); cond_n == 0: f_true
  int cond_n; unsigned long long cond_t0, cond_t1; double cond_rpb;
  unsigned long long* rsz;        // [2*R0] region sizes (updateRegion = true)
  double* rsum;                   // [2*R0] sum of pb over the region's voxels
  uint32_t *mark0, *mark1;        // [2*R0], zero between contractions
  uint32_t* order;                // [R0][3] dense ids
  double* sal_out;
  unsigned long long* ctrl;       // [0] merges done, [1] edges used, [2] pool used, [3] status, [4] values used, [5] window queue: wlo
  unsigned long long max_iters;
};

// An incident-edge list entry of the window kernel: everything a contraction needs from the edge and from the
// neighbour, so that one 32-byte load replaces the second dependent round trip (edge record, neighbour's list offset).
// All of it is immutable for the lifetime of the edge / region.
struct __attribute__((aligned(16))) FatEntry {
  uint32_t eid;      // edge slot, kNone = tombstone
  uint32_t rs;       // the neighbour this entry leads to
  uint32_t n;        // boundary voxels of the edge
  uint32_t pos;      // position of the edge's other entry, in rs's list
  uint32_t off;      // adj_off[rs]
  uint32_t len;      // adj_len[rs]
  double mean;       // boundary mean of the edge
};


namespace {

constexpr uint32_t kMarkSlots = 2048;      // LDS neighbour table of one contraction
constexpr uint32_t kMarkMax = 1408;        // contractions with more incident entries use the global mark arrays
struct Shared {
  uint32_t r0, r1, e, stop, len0, len1, off0, off1, newcount, reject;
  PqWork pq;
  // neighbours of the contracted pair: key = neighbour + 1, values = (edge to r0) + 1, (edge to r1) + 1
  uint32_t mk[kMarkSlots], mv0[kMarkSlots], mv1[kMarkSlots];
  uint32_t items[kMarkMax], nitems;
};
constexpr uint32_t kMergeTile = 1024;      // outputs merged through LDS by one wave at a time
struct MedianJobs {               // median linkage: the value runs to merge in one batch of phase B
  uint32_t n;
  uint32_t newE[kGreedyThreads], e0[kGreedyThreads], e1[kGreedyThreads];
  unsigned long long off[kGreedyThreads + 1];    // output offset of job j (elements)
  uint32_t toff[kGreedyThreads + 1];             // first tile of job j
  uint32_t tjob[kGreedyThreads], ta0[kGreedyThreads], ta1[kGreedyThreads];   // tiles of the current round
  // the two input runs of job j (lengths, offsets in the value pool) and its median, kept here so that neither the tile
  // set-up nor the tiles nor the final pass go back to global memory for them
  uint32_t na[kGreedyThreads], nb[kGreedyThreads];
  unsigned long long oa[kGreedyThreads], ob[kGreedyThreads];
  float med[kGreedyThreads];
  float buf[kGreedyThreads / 64][2 * kMergeTile + 64];      // input pieces | output (padded: index + index / 16)
};
struct NoJobs {                   // mean linkage: never touched
  uint32_t n, newE[1], e0[1], e1[1], toff[2], tjob[1], ta0[1], ta1[1], na[1], nb[1];
  unsigned long long off[2], oa[1], ob[1];
  float med[1];
  float buf[kGreedyThreads / 64][2];
};

// number of elements of the sorted run a[0..n) that are < v (strict = true) or <= v
__device__ __forceinline__ uint32_t run_rank(const float* a, uint32_t n, float v, bool strict) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    const float x = a[mid];
    if (strict ? (x < v) : (x <= v)) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// merge path: how many elements of A are among the first d outputs of the stable merge (ties: A first)
__device__ __forceinline__ uint32_t merge_split(const float* A, uint32_t na, const float* B, uint32_t nb, uint32_t d) {
  uint32_t lo = d > nb ? d - nb : 0u, hi = d < na ? d : na;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (A[mid] <= B[d - 1u - mid]) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool MEDIAN>
__global__ __launch_bounds__(kGreedyThreads) void greedy_pb_kernel(GreedyState st) {
  __shared__ Shared s;
  __shared__ typename std::conditional<MEDIAN, MedianJobs, NoJobs>::type jobs;
  const int tid = threadIdx.x;
  unsigned long long k = st.ctrl[0], ne = st.ctrl[1], pool_used = st.ctrl[2], vals_used = st.ctrl[4];
  uint32_t status = ST_RUN;
  if (tid == 0) { s.pq.wln[0] = s.pq.wln[1] = 0; s.pq.ovf = 0; s.pq.spill = 0; s.nitems = 0; }
  for (int i = tid; i < kSetSlots; i += blockDim.x) { s.pq.set[0][i] = 0; s.pq.set[1][i] = 0; }
  for (uint32_t i = tid; i < kMarkSlots; i += blockDim.x) { s.mk[i] = 0; s.mv0[i] = 0; s.mv1[i] = 0; }
  const PqTree& pq = st.pq;
  // mean linkage: the last stored level of the tree (<= 4096 nodes) stays in LDS for the whole launch -- its nodes are
  // written by one step of the propagation and read by the next, which through global memory is a store round trip plus
  // a load round trip.  (The median kernel needs the LDS for its merge tiles.)
  constexpr uint32_t kTopLds = MEDIAN ? 1u : kTopMax;
  __shared__ Key s_topk[kTopLds];
  Key* topk = (!MEDIAN && pq.nlevels >= 2 && pq.lv[pq.nlevels - 1].size <= kTopLds) ? s_topk : nullptr;
  if (topk) pq_top_load<kGreedyThreads>(pq, topk, tid);
  full_barrier();
  pq_top<kGreedyThreads>(pq, s.pq, tid, topk);      // the root lives in LDS: rebuilt at every launch

#ifdef GLIA_HMT_PROFILE
  unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
  unsigned long long tb[4] = {0, 0, 0, 0}, nb[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0}, titer = tlast;
#define PH(i) do { if (tid == 0) { unsigned long long tn = __builtin_readcyclecounter(); tph[i] += tn - tlast; tlast = tn; } } while (0)
#else
#define PH(i) do {} while (0)
#endif
  for (unsigned long long it = 0; it < st.max_iters; ++it) {
    // ---- pop (TBoundaryTable::top) ----
    PH(5);
    if (tid == 0) {
      const Key root = pq_root<kGreedyThreads>(s.pq);
      s.stop = ST_RUN;
      s.newcount = 0;
      s.reject = 0;
      if (root.seq == 0) s.stop = ST_DONE;
      else {
        uint32_t e = root.arg;
        s.e = e; s.r0 = st.e_u[e]; s.r1 = st.e_v[e];
        if (st.cond_n > 0) {
          // TBoundaryTable::top(fcond) walks the queue from the best item down and returns the first one fcond accepts.
          // fcond depends only on the two regions, which cannot change while the item lives, so an item it rejects
          // is rejected for good: take it out of the queue (it stays in the table and is folded into later updates).
          unsigned long long sz0 = st.rsz[s.r0], sz1 = st.rsz[s.r1];
          double su0 = st.rsum[s.r0], su1 = st.rsum[s.r1];
          if (sz0 > sz1) { unsigned long long t = sz0; sz0 = sz1; sz1 = t; double d = su0; su0 = su1; su1 = d; }
          bool ok = sz0 < st.cond_t0;
          if (!ok && st.cond_n > 1) {
            if (sz0 < st.cond_t1 && sdivide(su0, (double)sz0, 0.0) > st.cond_rpb) ok = true;
            if (!ok && sz1 < st.cond_t1 && sdivide(su1, (double)sz1, 0.0) > st.cond_rpb) ok = true;
          }
          if (!ok) { s.reject = 1; pq.leaf_seq[e] = 0; pq_touch(pq, s.pq, 0, 0, e); /* e is the root, hence the maximum of its level-0 node */ }
        }
        // one round trip for everything the two regions contribute (loads first: they would queue behind the stores)
        const uint32_t len0 = st.adj_len[s.r0], len1 = st.adj_len[s.r1], off0 = st.adj_off[s.r0], off1 = st.adj_off[s.r1];
        const unsigned long long z0 = st.rsz[s.r0], z1 = st.rsz[s.r1];
        const double w0 = st.rsum[s.r0], w1 = st.rsum[s.r1];
        const unsigned long long b0 = MEDIAN ? st.rbv[s.r0] : 0ull, b1 = MEDIAN ? st.rbv[s.r1] : 0ull;
        const int en = MEDIAN ? st.e_n[e] : 0;
        s.len0 = len0; s.len1 = len1; s.off0 = off0; s.off1 = off1;
        unsigned long long tot = (unsigned long long)len0 + len1;
        if (s.reject) {}      // nothing is contracted: no capacity needed
        else if (ne + tot > st.Ecap) s.stop = ST_NEED_EDGES;
        else if (pool_used + tot > st.pool_cap) s.stop = ST_NEED_POOL;
        else if (MEDIAN && vals_used + b0 + b1 > st.vals_cap) s.stop = ST_NEED_VALUES;
        else {
          if (MEDIAN) st.rbv[st.R0 + (uint32_t)k] = b0 + b1 - 2ull * (unsigned long long)en;
          st.order[3 * k + 0] = s.r0; st.order[3 * k + 1] = s.r1; st.order[3 * k + 2] = st.R0 + (uint32_t)k;
          st.sal_out[k] = root.sal;
          st.rsz[st.R0 + (uint32_t)k] = z0 + z1;       // TRegionMap::merge (updateRegion)
          st.rsum[st.R0 + (uint32_t)k] = w0 + w1;
        }
      }
    }
    full_barrier();
    PH(0);
    if (s.stop != ST_RUN) { status = s.stop; break; }
    if (s.reject) { pq_propagate<kGreedyThreads>(pq, s.pq, tid, topk); continue; }
    const uint32_t r0 = s.r0, e = s.e, len0 = s.len0, len1 = s.len1, off0 = s.off0, off1 = s.off1;
    const uint32_t r2 = st.R0 + (uint32_t)k;
    const uint32_t total = len0 + len1;
    const uint32_t r2off = (uint32_t)pool_used;
    const bool small = total <= kMarkMax;          // the usual case: neighbour matching entirely in LDS

    // ---- phase A: one table entry per distinct neighbour, holding the edge(s) that reach it ----
    for (uint32_t i = tid; i < total; i += kGreedyThreads) {
      const bool side1 = i >= len0;
      const uint2 pe = st.pool[side1 ? off1 + (i - len0) : off0 + i];
      const uint32_t eid = pe.x, rs = pe.y;
      if (eid == e || eid == kNone) continue;        // the contracted edge / the dead twin of an earlier contraction
      if (small) {
        uint32_t h = (rs * 2654435761u) >> 21;
        while (true) {
          const uint32_t old = atomicCAS(&s.mk[h], 0u, rs + 1u);
          if (old == 0u) { s.items[atomicAdd(&s.nitems, 1u)] = h; break; }
          if (old == rs + 1u) break;
          h = (h + 1u) & (kMarkSlots - 1u);
        }
        (side1 ? s.mv1 : s.mv0)[h] = eid + 1u;
      } else (side1 ? st.mark1 : st.mark0)[rs] = eid + 1u;
    }
    full_barrier();
    PH(1);

    // ---- phase B: one new edge (rs, r2) per distinct neighbour (TBoundaryTable::update) ----
    bool bad = false;
    const uint32_t nwork = small ? s.nitems : total;
    for (uint32_t base = 0; base < nwork; base += kGreedyThreads) {
      if (MEDIAN) { if (tid == 0) jobs.n = 0; full_barrier(); }
      const uint32_t i = base + tid;
      do {
        if (i >= nwork) break;
        uint32_t rs, e0s, e1s;
        if (small) {
          const uint32_t h = s.items[i];
          rs = s.mk[h] - 1u;
          const uint32_t m0 = s.mv0[h], m1 = s.mv1[h];
          e0s = m0 ? m0 - 1u : kNone; e1s = m1 ? m1 - 1u : kNone;
          s.mk[h] = 0u; s.mv0[h] = 0u; s.mv1[h] = 0u;          // the table is clean again when the phase ends
        } else {
          const bool side1 = i >= len0;
          const uint2 pe = st.pool[side1 ? off1 + (i - len0) : off0 + i];
          const uint32_t eid = pe.x;
          rs = pe.y;
          if (eid == e || eid == kNone) break;
          if (!side1) {
            e0s = eid;
            const uint32_t m = st.mark1[rs];
            e1s = m ? m - 1u : kNone;
          } else {
            if (st.mark0[rs] != 0u) break;             // common neighbour: handled from the r0 side
            e0s = kNone; e1s = eid;
          }
        }
        const uint32_t idx = atomicAdd(&s.newcount, 1u);
        const uint32_t newE = (uint32_t)ne + idx;
        // Everything this record reads, requested up front and UNCONDITIONALLY (a missing side re-reads the other side's
        // slot): a load inside a branch gets its own basic block and its own s_waitcnt, i.e. its own memory round trip,
        // and a wave's loads queue behind its own earlier stores (vmcnt is in order).
        const bool h0 = e0s != kNone, h1 = e1s != kNone;
        const uint32_t a0 = h0 ? e0s : e1s, a1 = h1 ? e1s : e0s;
        const uint32_t u0 = st.e_u[a0], pu0 = st.e_posu[a0], pv0 = st.e_posv[a0];
        const uint32_t u1 = st.e_u[a1], pu1 = st.e_posu[a1], pv1 = st.e_posv[a1];
        const uint32_t offRs = st.adj_off[rs];
        const unsigned long long q0 = pq.leaf_seq[a0], q1 = pq.leaf_seq[a1];
        const uint32_t t0 = pq.lv[0].arg[a0 / kFan], t1 = pq.lv[0].arg[a1 / kFan];
        const int n0 = st.e_n[a0], n1 = st.e_n[a1];
        const double m0 = st.e_mean[a0], m1 = st.e_mean[a1];
        const unsigned long long eoff = MEDIAN ? st.e_off[a0] : 0ull, eoff1 = MEDIAN ? st.e_off[a1] : 0ull;
        const uint32_t posRs = (u0 == rs) ? pu0 : pv0;
        const unsigned long long seq0 = h0 ? q0 : 0ull, seq1 = h1 ? q1 : 0ull;
        const uint32_t top0 = h0 ? t0 : kNone, top1 = h1 ? t1 : kNone;
        const uint32_t pos1 = (h0 && h1) ? ((u1 == rs) ? pu1 : pv1) : kNone;
        double first = 0.0;
        int second = 0;
        if (!MEDIAN) {
          // util/struct_merge.hxx:62-76
          if (h0) { first += m0 * n0; second += n0; }
          if (h1) { first += m1 * n1; second += n1; }
          first = sdivide(first, (double)second, 0.0);
          if (first == -1.0) bad = true;               // DUMMY -> "invalid boundary saliency" (:78-79)
        } else {
          // util/struct_merge.hxx:118-127: the value lists are spliced; one list alone is moved (its run is reused)
          if (h0) second += n0;
          if (h1) second += n1;
          if (h0 && h1) {
            const uint32_t j = atomicAdd(&jobs.n, 1u);
            jobs.newE[j] = newE; jobs.e0[j] = e0s; jobs.e1[j] = e1s;
            jobs.na[j] = (uint32_t)n0; jobs.nb[j] = (uint32_t)n1; jobs.oa[j] = eoff; jobs.ob[j] = eoff1;
          } else { first = m0; st.e_off[newE] = eoff; }
        }
        // rs held two entries (to r0 and to r1): one is reused for the new edge, the other becomes a tombstone
        if (pos1 != kNone) st.pool[offRs + pos1] = make_uint2(kNone, 0u);
        const uint32_t cat = rs < r0 ? 0u : (e0s != kNone ? 1u : 2u);
        const unsigned long long seq = ((k + 1ull) << 32) | ((unsigned long long)cat << 30) | rs;
        st.e_u[newE] = rs; st.e_v[newE] = r2; st.e_posu[newE] = posRs; st.e_posv[newE] = idx;
        st.e_mean[newE] = first; st.e_n[newE] = second;
        pq.leaf_sal[newE] = (MEDIAN && st.size_weight) ? -first * (double)min(st.rsz[rs], st.rsz[r2]) : -first;
        pq.leaf_seq[newE] = seq;
        st.pool[offRs + posRs] = make_uint2(newE, r2);
        st.pool[r2off + idx] = make_uint2(newE, rs);
        pq_leaf_added(pq, s.pq, newE);
        // a dying leaf only matters to the tree if it is the current maximum of its level-0 node (see pq_leaf_removed)
        if (seq0) { pq.leaf_seq[e0s] = 0; if (top0 == e0s) pq_touch(pq, s.pq, 0, 0, e0s); }
        if (seq1) { pq.leaf_seq[e1s] = 0; if (top1 == e1s) pq_touch(pq, s.pq, 0, 0, e1s); }
      } while (false);
      if (MEDIAN) {
        full_barrier();
        const uint32_t J = jobs.n;
        if (J) {
          if (tid == 0) {
            unsigned long long o = 0;
            uint32_t to = 0;
            for (uint32_t j = 0; j < J; ++j) {
              const uint32_t n = jobs.na[j] + jobs.nb[j];
              jobs.off[j] = o; jobs.toff[j] = to;
              o += n; to += (n + kMergeTile - 1) / kMergeTile;
            }
            jobs.off[J] = o; jobs.toff[J] = to;
          }
          full_barrier();
          const unsigned long long tot = jobs.off[J];
          const uint32_t ntiles = jobs.toff[J];
          // stable merge of the two sorted runs (ties: the (r0,rs) run first).  Merge-path splits cut every job into
          // tiles of kMergeTile outputs; a wave stages a tile's two input pieces in LDS, places every element at
          // (own index + rank in the other piece) and streams the tile out.
          const int lane = tid & 63, wave = tid >> 6;
          for (uint32_t round0 = 0; round0 < ntiles; round0 += kGreedyThreads) {
            const uint32_t q = round0 + (uint32_t)tid;
            if (q < ntiles) {
              uint32_t lo = 0, hi = J;
              while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (jobs.toff[mid] <= q) lo = mid; else hi = mid; }
              const uint32_t na = jobs.na[lo], nb = jobs.nb[lo], n = na + nb;
              const float* A = st.vals + jobs.oa[lo];
              const float* B = st.vals + jobs.ob[lo];
              const uint32_t d0 = (q - jobs.toff[lo]) * kMergeTile, d1 = d0 + kMergeTile < n ? d0 + kMergeTile : n;
              jobs.tjob[tid] = lo;
              jobs.ta0[tid] = d0 == 0 ? 0u : merge_split(A, na, B, nb, d0);
              jobs.ta1[tid] = d1 == n ? na : merge_split(A, na, B, nb, d1);
            }
            full_barrier();
            const uint32_t cnt = ntiles - round0 < (uint32_t)kGreedyThreads ? ntiles - round0 : (uint32_t)kGreedyThreads;
            float* in = jobs.buf[wave];
            float* ob = in + kMergeTile;
            // a tile's two input pieces travel global memory -> registers -> LDS; the registers of the NEXT tile are
            // requested before the current one is merged, so the memory round trip overlaps the merge
            constexpr int kPer = (int)(kMergeTile / 64);
            float pre[kPer];
            auto fetch = [&](uint32_t t) __attribute__((always_inline)) {
              const uint32_t j = jobs.tjob[t], a0 = jobs.ta0[t], a1 = jobs.ta1[t];
              const uint32_t n = jobs.na[j] + jobs.nb[j];
              const uint32_t d0 = (round0 + t - jobs.toff[j]) * kMergeTile, d1 = d0 + kMergeTile < n ? d0 + kMergeTile : n;
              const uint32_t b0 = d0 - a0, la = a1 - a0, lt = d1 - d0;
              const float* A = st.vals + jobs.oa[j] + a0;
              const float* B = st.vals + jobs.ob[j] + b0;
#pragma unroll
              for (int k = 0; k < kPer; ++k) {          // unconditional loads: a clamped index re-reads the last element
                const uint32_t i = (uint32_t)lane + 64u * (uint32_t)k, ic = i < lt ? i : lt - 1u;
                const float* src = ic < la ? A + ic : B + (ic - la);      // one load through a selected address
                pre[k] = *src;
              }
            };
            if ((uint32_t)wave < cnt) fetch((uint32_t)wave);
            for (uint32_t t = wave; t < cnt; t += kGreedyThreads / 64) {
              const uint32_t j = jobs.tjob[t], a0 = jobs.ta0[t], a1 = jobs.ta1[t];
              const uint32_t n = jobs.na[j] + jobs.nb[j];
              const uint32_t d0 = (round0 + t - jobs.toff[j]) * kMergeTile, d1 = d0 + kMergeTile < n ? d0 + kMergeTile : n;
              const uint32_t b0 = d0 - a0, la = a1 - a0, lb = (d1 - a1) - b0, lt = la + lb;
              float* out = st.vals + vals_used + jobs.off[j] + d0;
#pragma unroll
              for (int k = 0; k < kPer; ++k) { const uint32_t i = (uint32_t)lane + 64u * (uint32_t)k; if (i < lt) in[i] = pre[k]; }
              if (t + kGreedyThreads / 64 < cnt) fetch(t + kGreedyThreads / 64);
              wave_lds_sync();
              {
                // every lane merges 16 consecutive outputs sequentially from its merge-path split (a rank search per
                // element cost ten dependent LDS reads each); the output index is padded against bank conflicts
                const float* TA = in;
                const float* TB = in + la;
                const uint32_t o0 = (uint32_t)lane * 16u < lt ? (uint32_t)lane * 16u : lt, o1 = o0 + 16u < lt ? o0 + 16u : lt;
                uint32_t ai = o0 == 0u ? 0u : (o0 >= lt ? la : merge_split(TA, la, TB, lb, o0));
                uint32_t bi = o0 - ai;
                float a = ai < la ? TA[ai] : 0.f, b = bi < lb ? TB[bi] : 0.f;
                for (uint32_t o = o0; o < o1; ++o) {
                  const bool ta = bi >= lb || (ai < la && a <= b);      // ties: the (r0, rs) run first
                  ob[o + (o >> 4)] = ta ? a : b;
                  if (ta) { ++ai; a = ai < la ? TA[ai] : 0.f; } else { ++bi; b = bi < lb ? TB[bi] : 0.f; }
                }
              }
              wave_lds_sync();
              for (uint32_t i = lane; i < lt; i += 64) out[i] = ob[i + (i >> 4)];
              const uint32_t mi = n / 2u;                               // util/stats.hxx:83-91
              if (lane == 0 && mi >= d0 && mi < d1) jobs.med[j] = ob[(mi - d0) + ((mi - d0) >> 4)];
              wave_lds_sync();
            }
            full_barrier();
          }
          full_barrier();
          if ((uint32_t)tid < J) {
            const uint32_t newE = jobs.newE[tid];
            const unsigned long long off = vals_used + jobs.off[tid];
            const double med = (double)jobs.med[tid];
            st.e_off[newE] = off; st.e_mean[newE] = med;
            pq.leaf_sal[newE] = st.size_weight ? -med * (double)min(st.rsz[st.e_u[newE]], st.rsz[r2]) : -med;
          }
          vals_used += tot;
          full_barrier();
        }
      }
        }
    if (tid == 0) { pq.leaf_seq[e] = 0; pq_touch(pq, s.pq, 0, 0, e);   /* the root is the maximum of its node: no need to look */ }
    if (__syncthreads_or(bad ? 1 : 0)) { status = ST_BAD_SALIENCY; break; }
    // (round 4 audit, DESIGN 3.3: s.nitems used to be cleared in front of this barrier -- the mean linkage has no barrier inside
    // phase B, so a wave that left the barrier behind phase A late could have read its item count after thread 0 had cleared it)
    if (tid == 0) s.nitems = 0;
    PH(2);

    // ---- phase C: publish r2's list; the rare big contraction resets the global marks it used ----
    const uint32_t newcount = s.newcount;
    if (!small) {
      for (uint32_t j = tid; j < newcount; j += kGreedyThreads) {
        const uint32_t rs = st.e_u[(uint32_t)ne + j];
        st.mark0[rs] = 0; st.mark1[rs] = 0;
      }
    }
    if (tid == 0) { st.adj_off[r2] = r2off; st.adj_len[r2] = newcount; }

    PH(3);
    // ---- priority structure: propagate dirty nodes level by level ----
    pq_propagate<kGreedyThreads>(pq, s.pq, tid, topk);
    PH(4);
#ifdef GLIA_HMT_PROFILE
    if (tid == 0) {
      const unsigned long long tn = __builtin_readcyclecounter();
      const int b = total <= 64 ? 0 : total <= 512 ? 1 : total <= kMarkMax ? 2 : 3;
      tb[b] += tn - titer; nb[b] += 1; db[b] += total; titer = tn;
    }
#endif
    k += 1; ne += newcount; pool_used += total;
  }
  full_barrier();
  if (topk) pq_top_store<kGreedyThreads>(pq, topk, tid);      // the next launch (or the host's rebuild) starts from global memory
  if (tid == 0) { st.ctrl[0] = k; st.ctrl[1] = ne; st.ctrl[2] = pool_used; st.ctrl[3] = status; st.ctrl[4] = vals_used; }
#ifdef GLIA_HMT_PROFILE
  if (tid == 0) printf("[greedy profile] pq propagations by dirty level-0 nodes (<=8, <=16, more): %llu %llu %llu\n", g_pqprof[28], g_pqprof[29], g_pqprof[30]);
  if (tid == 0) printf("[greedy profile] pq top: loads %llu wave_max %llu barrier %llu calls %llu\n", g_pqprof[24], g_pqprof[25], g_pqprof[26], g_pqprof[27]);
  if (tid == 0) printf("[greedy profile] pq levels (wave 0): recompute %llu %llu %llu %llu  barrier-wait %llu %llu %llu %llu  active %llu %llu %llu %llu\n", g_pqprof[0], g_pqprof[1], g_pqprof[2],
                       g_pqprof[3], g_pqprof[8], g_pqprof[9], g_pqprof[10], g_pqprof[11], g_pqprof[16], g_pqprof[17], g_pqprof[18], g_pqprof[19]);
  if (tid == 0) printf("[greedy profile] by degree (<=64, <=512, <=1408, more): merges %llu %llu %llu %llu  cycles %llu %llu %llu %llu  entries %llu %llu %llu %llu\n",
                       nb[0], nb[1], nb[2], nb[3], tb[0], tb[1], tb[2], tb[3], db[0], db[1], db[2], db[3]);
  if (tid == 0) printf("[greedy profile] merges %llu: pop %llu  mark %llu  build %llu  reset %llu  pq %llu  loop-top %llu (cycles)\n", k, tph[0], tph[1], tph[2], tph[3], tph[4], tph[5]);
#endif
}

// =====================================================================================================================
// The window queue: the priority queue of the pb-mean loop without a tree.
//
// The tournament tree above costs a contraction ~13 k of its ~24 k cycles: every new or dying edge dirties a 256-ary
// node somewhere in the slot space, each dirty node is a 4 KB gather, and three levels are three dependent round trips
// (plus two for the pop).  The queue only ever has to answer "largest (saliency, seq)", and popped saliencies fall
// (almost) monotonically, so the live items are kept in two places instead:
//   * GLOBAL, below a threshold key tau:
//       - the INITIAL edges in one array sorted by descending key (rocPRIM, once): consumed front to back by a pointer,
//         whatever the ties (a 1024^3 Q8 volume has tie groups of thousands of equal means);
//       - edges CREATED by contractions in singly linked lists, one per saliency CELL (a monotone quantisation of the
//         saliency, ~E0/4 cells).  Insert = atomicExch on the cell head + one store, nobody waits for it.
//       A per-cell counter holds the live items of both kinds; a dying edge only decrements it (dead array entries and
//       list nodes are skipped when their cell is loaded).
//   * LDS WINDOW, above tau: unordered, <= kWinCap entries carrying (saliency, seq, edge, both regions and their list
//     headers).  Its maximum is the maximum of the queue; it is found by one scan of the window per contraction, which
//     also applies the (rare) deaths of window items.  New edges above tau go straight into the window.  When the window
//     runs empty, tau moves down: whole cells while they fit, then a prefix of the next cell's sorted initial entries
//     (tau = key of the first entry left behind) plus that cell's list nodes above tau.
// Exactness: the order is (saliency, seq) with the very seq numbers of the tree kernel, so the result is bit-identical
// (gate: SHA-1 of the whole 1024^3 order, tools/pb_bench.py); no assumption about the linkage is made (a new edge may
// well beat the current maximum: it lands in the window).  A cell whose LIST part alone exceeds the window (massive
// exact ties among created edges) stops the kernel with ST_NEED_TREE and the host continues with the tree kernel from
// the same state (edge records are unpacked into its arrays).
// With the fat list entries (FatEntry) and the list headers carried in the window a contraction is ONE dependent global
// round trip -- the two incident-edge lists -- plus LDS work; its stores are fire-and-forget: the next contraction only
// waits for them when it touches a region whose list they rewrite (a bitmap of the touched regions decides).
// Edge state is one 64-byte record (EdgeRec) instead of twelve arrays: four wide stores per new edge, one base pointer.
// =====================================================================================================================
struct __attribute__((aligned(16))) EdgeRec {
  uint32_t u, v, posu, posv;                // regions (u < v) and the positions of the edge's entries in their lists
  double mean; int n; uint32_t next;        // linkage data; link of the cell list
  uint2 hu, hv;                             // (offset, length) of u's and v's incident-edge lists
  double sal; unsigned long long seq;       // queue key; seq == 0: not in the queue
};
static_assert(sizeof(EdgeRec) == 64, "EdgeRec layout");
struct WinState {
  EdgeRec* er; FatEntry* fpool;
  uint32_t* whead;                          // [wB] newest created edge of the cell's list (kNone = empty); atomics only
  uint32_t* wcnt;                           // [wB] live queue items of the cell below the threshold (sorted array + list); atomics only
  // the BASELINE: every queue item that was alive when it was taken (at the start: the initial edges; later: see
  // win_rebaseline), sorted by descending (saliency, seq), with its seq (a dead item's record no longer has it)
  const uint32_t* isort; const unsigned long long* isort_seq;
  const uint32_t* ige;                      // [wB + 1] baseline items whose cell is >= c
  const double* wrange;                     // [0] smallest initial saliency, [1] cells per unit of saliency
  uint32_t* order; double* sal_out; unsigned long long* ctrl;
  unsigned long long* rsz; double* rsum; uint32_t *mark0, *mark1, *adj_off, *adj_len;
  unsigned long long pool_cap, max_iters, cond_t0, cond_t1;
  double cond_rpb;
  uint32_t R0, Ecap, wB, nsort;             // nsort: items of the baseline
  unsigned long long ne_base, rebase_after; // edges that existed at the baseline; a new one is due after this many more
  uint32_t wcap, wbudget;                   // window slots in use (<= kWinCap) and the items a reload brings at most (tests shrink them: GLIA_HMT_WINCAP)
  int cond_n;
  // HORIZON (batch kernel): cells below wch are out of the queue's reach until the next baseline.  An edge created there is
  // neither linked into its cell's list nor counted, an edge dying there is not counted either: its record and its two list
  // entries are all that is written (the baseline is rebuilt from the records).  A reload that would have to go below the
  // horizon ends the launch with ST_REBASE instead.  0 = no horizon.
  uint32_t wch;
  // regions that have been merged away (batch kernel).  An edge that dies BELOW the horizon is not marked in its record (one
  // scattered store per dying edge less in the contraction's store stream): no reload can reach it before the next baseline,
  // and the baseline's collection pass recognises it by its dead region.
  uint8_t* rdead;
  // tests (GLIA_HMT_FORCE_TREE=k): hand the queue over to the tournament-tree kernel at the first empty window after k merges --
  // the path of ST_NEED_TREE, which no data set reaches by itself any more (oversized cells are split)
  unsigned long long force_tree;
};
constexpr uint32_t kWinCap = 1536;          // window slots (live items + holes)
constexpr uint32_t kWinBudget = 768;        // a reload stops before exceeding this many items ...
constexpr uint32_t kWinMinLoad = 192;       // ... and goes on to the next block of cells below this many
constexpr uint32_t kWinMinPartial = 96;     // a cell is split only if at least this much room is left
constexpr uint32_t kKillMax = 8;
constexpr int kNW = kGreedyThreads / 64;
constexpr int kWinPer = (int)(kWinCap / kGreedyThreads);
static_assert(kWinPer * kGreedyThreads == (int)kWinCap, "window capacity");
struct WinShared {
  double sal[kWinCap];
  unsigned long long seq[kWinCap];          // 0 = hole
  uint32_t e[kWinCap], u[kWinCap], v[kWinCap];
  uint2 hu[kWinCap], hv[kWinCap];           // (offset, length) of u's and v's incident-edge lists
  // threshold: an item is in the window iff cell(sal) > cthr, or cell(sal) == cthr and (sal, seq) > (tsal, tseq)
  int cthr; uint32_t iptr;                  // initial entries before iptr of the sorted array are consumed
  double tsal; unsigned long long tseq;
  alignas(16) uint32_t n;                   // slots in use   (n, nk, kovf, pad0: one 16-byte read in the scan)
  uint32_t nk, kovf, pad0;                  // edges that died in this contraction and sit in the window
  alignas(16) uint32_t kill[kKillMax];
  alignas(16) Key part[kNW];                // per-wave maxima of the last scan (arg = slot)
  uint32_t touched[2][64];                  // regions whose lists the previous / this contraction rewrites (bitmap over id mod 2048)
  uint32_t wsum[kNW];                       // block scan scratch
  uint32_t bcast, maxcell, err, need_tree;
  double psal; unsigned long long pseq;     // split of a cell: the list's contribution to tau
  unsigned long long spill_ord;             // image of the largest saliency that found the window full (0 = none): tau has to rise to it
};
struct WinWork {                            // the neighbour table of one contraction (small case)
  uint32_t mk[kMarkSlots], mv0[kMarkSlots], mv1[kMarkSlots];     // neighbour + 1, staged index + 1 of the (r0,rs) / (r1,rs) entry
  uint32_t items[kMarkMax], newidx[kMarkMax], nitems, newcount, bad;     // items[i]: table slot of neighbour i, then the pool position of its new entry
  FatEntry stage[kMarkMax];
};

// order-preserving image of a double: ascending doubles <-> ascending unsigned integers
__device__ __forceinline__ unsigned long long f64_ord(double d) {
  unsigned long long b = (unsigned long long)__double_as_longlong(d);
  return b ^ ((b >> 63) ? ~0ull : 0x8000000000000000ull);
}
__host__ __device__ __forceinline__ double f64_unord(unsigned long long o) {
  o ^= (o >> 63) ? 0x8000000000000000ull : ~0ull;
  return __builtin_bit_cast(double, o);
}
__device__ __forceinline__ void lds_barrier(const int line = __builtin_LINE()) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); GLIA_SKEW_DELAY(line); }   // global stores stay in flight

__host__ __device__ __forceinline__ uint32_t win_cell(double sal, double smin, double scale, uint32_t B) {
  double t = (sal - smin) * scale;          // monotone in sal (saliencies are never NaN: sdivide guards the division)
  t = t > 0.0 ? t : 0.0;
  return t >= (double)(B - 1u) ? B - 1u : (uint32_t)t;
}
__device__ __forceinline__ uint32_t ld_l2(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_l2(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool win_above(int cthr, double tsal, unsigned long long tseq, int cell, double sal, unsigned long long seq) {
  return cell > cthr || (cell == cthr && (sal > tsal || (sal == tsal && seq > tseq)));
}

// inclusive block scan of one value per thread (every thread calls; two barriers)
__device__ __forceinline__ uint32_t block_scan_incl(uint32_t v, uint32_t* wsum, int tid, uint32_t* total) {
  const int lane = tid & 63, wave = tid >> 6;
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)x, d); if (lane >= d) x += y; }
  full_barrier();
  if (lane == 63) wsum[wave] = x;
  full_barrier();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kNW; ++i) { const uint32_t s = wsum[i]; if (i < wave) base += s; tot += s; }
  *total = tot;
  return base + x;
}

__device__ __forceinline__ void win_put(WinShared& w, uint32_t slot, double sal, unsigned long long seq, uint32_t e, uint32_t u, uint32_t v, uint2 hu, uint2 hv) {
  w.sal[slot] = sal; w.seq[slot] = seq; w.e[slot] = e; w.u[slot] = u; w.v[slot] = v; w.hu[slot] = hu; w.hv[slot] = hv;
}
// edge e of the global storage into the window if it is alive
__device__ __forceinline__ void win_take(const WinState& st, WinShared& w, uint32_t e, const EdgeRec& r) {
  if (r.seq != 0) {
    const uint32_t slot = atomicAdd(&w.n, 1u);
    if (slot < st.wcap) win_put(w, slot, r.sal, r.seq, e, r.u, r.v, r.hu, r.hv); else w.err = 1;
  }
}

// One pass over the window: applies this contraction's deaths, completes the list header of the region just created
// (its length was not known when its edges were inserted), leaves the per-wave maxima in w.part.  Every LDS read is
// issued up front (a load inside a branch is a round trip of its own).  Ends with an LDS-only barrier.
__device__ __forceinline__ void win_scan(const WinState& st, WinShared& w, int tid, uint32_t r2, uint32_t r2len) {
  const uint4 hd = *reinterpret_cast<const uint4*>(&w.n);                 // n, nk, kovf
  const uint4 k0 = *reinterpret_cast<const uint4*>(&w.kill[0]), k1 = *reinterpret_cast<const uint4*>(&w.kill[4]);
  unsigned long long q[kWinPer]; uint32_t e[kWinPer], v[kWinPer]; double sl[kWinPer];
#pragma unroll
  for (int j = 0; j < kWinPer; ++j) { const uint32_t i = (uint32_t)tid + (uint32_t)j * kGreedyThreads; q[j] = w.seq[i]; e[j] = w.e[i]; v[j] = w.v[i]; sl[j] = w.sal[i]; }
  const uint32_t n = hd.x, nk = hd.y < kKillMax ? hd.y : kKillMax;
  const uint32_t kl[kKillMax] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
  Key k;
  k.sal = -__builtin_inf(); k.seq = 0; k.arg = 0;
#pragma unroll
  for (int j = 0; j < kWinPer; ++j) {
    const uint32_t i = (uint32_t)tid + (uint32_t)j * kGreedyThreads;
    bool live = i < n && q[j] != 0;
    bool dead = false;
    if (nk) {
#pragma unroll
      for (uint32_t t = 0; t < kKillMax; ++t) dead = dead || (t < nk && kl[t] == e[j]);
    }
    if (live && hd.z) dead = dead || st.er[e[j]].seq == 0;       // more deaths than the list holds (rare): ask the edge record
    if (live && dead) { w.seq[i] = 0; live = false; }
    if (live && v[j] == r2) w.hv[i].y = r2len;
    Key c; c.sal = live ? sl[j] : -__builtin_inf(); c.seq = live ? q[j] : 0ull; c.arg = i;
    if (better(c, k)) k = c;
  }
  k = wave_max(k);
  if ((tid & 63) == 0) w.part[tid >> 6] = k;
  lds_barrier();
  if (tid == 0) { w.nk = 0; w.kovf = 0; }
}
// the maximum of the per-wave maxima, in every lane: lanes 0..7 fetch one each, two quad steps leave the maxima of
// parts 0..3 / 4..7 in lanes 0 / 4, which are read out and compared as uniform values
__device__ __forceinline__ Key win_lane_key(const Key& k, int l) {
  const unsigned long long sb = (unsigned long long)__double_as_longlong(k.sal);
  Key out;
  out.sal = __longlong_as_double((long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(sb >> 32), l) << 32) |
                                              (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)sb, l)));
  out.seq = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(k.seq >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)k.seq, l);
  out.arg = (uint32_t)__builtin_amdgcn_readlane((int)k.arg, l);
  return out;
}
__device__ __forceinline__ Key win_root(const WinShared& w, int lane) {
  static_assert(kNW == 8, "win_root");
  Key k = w.part[lane & (kNW - 1)];
  key_max_step<0xB1>(k); key_max_step<0x4E>(k);          // quad_perm [1,0,3,2], [2,3,0,1]: every lane holds its quad's maximum
  const Key a = win_lane_key(k, 0), b = win_lane_key(k, 4);
  return better(b, a) ? b : a;
}

// squeeze the holes out (every thread calls); returns the number of live items -- the same value in every thread, from the scan:
// a caller that decides on it must NOT read w.n again (the first wave past the barrier may already be adding to it)
__device__ __forceinline__ uint32_t win_compact(WinShared& w, int tid, uint32_t cap = kWinCap) {
  double sal[kWinPer]; unsigned long long seq[kWinPer]; uint32_t e[kWinPer], u[kWinPer], v[kWinPer]; uint2 hu[kWinPer], hv[kWinPer];
  uint32_t live = 0;
  const uint32_t n = w.n < cap ? w.n : cap;
#pragma unroll
  for (int j = 0; j < kWinPer; ++j) {
    const uint32_t i = (uint32_t)tid * kWinPer + j;
    seq[j] = i < n ? w.seq[i] : 0ull;
    sal[j] = w.sal[i]; e[j] = w.e[i]; u[j] = w.u[i]; v[j] = w.v[i]; hu[j] = w.hu[i]; hv[j] = w.hv[i];
    live += seq[j] != 0;
  }
  uint32_t total;
  uint32_t o = block_scan_incl(live, w.wsum, tid, &total) - live;      // barriers inside: every read above is done
#pragma unroll
  for (int j = 0; j < kWinPer; ++j) if (seq[j] != 0) { win_put(w, o, sal[j], seq[j], e[j], u[j], v[j], hu[j], hv[j]); ++o; }
  if (tid == 0) w.n = total;
  full_barrier();
  return total;
}

__device__ __forceinline__ void win_push_global(const WinState& st, uint32_t e, uint32_t cell) {
  const uint32_t old = atomicExch(&st.whead[cell], e);
  st.er[e].next = old;
  atomicAdd(&st.wcnt[cell], 1u);
}

// every live window item into its cell's list (initial edges too: their place in the sorted array is gone); afterwards
// the window is empty and the threshold sits above everything (every thread calls)
__device__ __forceinline__ void win_flush(const WinState& st, WinShared& w, int tid) {
  const double smin = st.wrange[0], scale = st.wrange[1];
  if (tid == 0) w.maxcell = 0;
  full_barrier();
  const uint32_t n = w.n < st.wcap ? w.n : st.wcap;
  uint32_t mc = 0;
  for (uint32_t i = tid; i < n; i += kGreedyThreads) {
    if (w.seq[i] == 0) continue;
    const uint32_t c = win_cell(w.sal[i], smin, scale, st.wB);
    win_push_global(st, w.e[i], c);
    mc = mc > c + 1u ? mc : c + 1u;
  }
  if (mc) atomicMax(&w.maxcell, mc);
  full_barrier();
  if (tid == 0) {
    if (w.maxcell && (int)w.maxcell - 1 >= w.cthr) { w.cthr = (int)w.maxcell - 1; w.tsal = __builtin_inf(); w.tseq = ~0ull; }
    w.n = 0;
  }
  full_barrier();
}

// Items above tau found the window full and went to their cells' lists: tau rises to the largest of their saliencies and
// the window items at or below it follow them (every thread calls, after a scan has applied the pending deaths)
__device__ __forceinline__ void win_evict(const WinState& st, WinShared& w, int tid) {
  const double smin = st.wrange[0], scale = st.wrange[1];
  const double lim = f64_unord(w.spill_ord);
  const uint32_t n = w.n < st.wcap ? w.n : st.wcap;
  full_barrier();
  for (uint32_t i = tid; i < n; i += kGreedyThreads) {
    if (w.seq[i] == 0 || w.sal[i] > lim) continue;
    win_push_global(st, w.e[i], win_cell(w.sal[i], smin, scale, st.wB));
    w.seq[i] = 0;
  }
  full_barrier();
  if (tid == 0) { w.n = n; w.cthr = (int)win_cell(lim, smin, scale, st.wB); w.tsal = lim; w.tseq = ~0ull; w.spill_ord = 0; }
  full_barrier();
}

// initial entries [a, b) of the sorted array into the window (every thread calls; no barrier)
__device__ __forceinline__ void win_take_initial(const WinState& st, WinShared& w, uint32_t a, uint32_t b, int tid) {
  for (uint32_t i = a + (uint32_t)tid; i < b; i += kGreedyThreads) { const uint32_t e = st.isort[i]; const EdgeRec r = st.er[e]; win_take(st, w, e, r); }
}

// The window holds no live item: move the threshold down.  Returns 0 = loaded something (or made progress), 1 = the
// queue is empty, 2 = a cell's list does not fit the window, 3 = nothing left above the horizon (every thread calls;
// contains barriers)
constexpr uint32_t kSelMax = 384;           // list items a split cell hands over at most (bounded min-heap in LDS)
__device__ __forceinline__ int win_reload(const WinState& st, WinShared& w, int tid, double* sel_sal, unsigned long long* sel_seq) {
  full_barrier();                       // (vmcnt(0) inside) this workgroup's list pushes and counter updates are done
  if (tid == 0) { w.n = 0; w.need_tree = 0; }
  full_barrier();
  uint32_t c_hi = (uint32_t)(w.cthr + 1) < st.wB ? (uint32_t)(w.cthr + 1) : st.wB, loaded = 0, iptr = w.iptr;
  int result = 1;
  const uint32_t c_floor = st.wch < st.wB ? st.wch : 0u;      // the horizon: cells below it are not loaded
  while (c_hi > c_floor) {
    const bool valid = (uint32_t)tid < c_hi - c_floor;
    const uint32_t c = valid ? c_hi - 1u - (uint32_t)tid : 0u;
    const uint32_t cn = valid ? ld_l2(&st.wcnt[c]) : 0u;
    uint32_t total;
    const uint32_t incl = block_scan_incl(cn, w.wsum, tid, &total);
    const bool ok = valid && incl <= st.wbudget - loaded;
    const uint32_t m = (uint32_t)__syncthreads_count(ok ? 1 : 0);       // ok is monotone in tid: the first m cells fit whole
    const uint32_t nvalid = c_hi - c_floor < (uint32_t)kGreedyThreads ? c_hi - c_floor : (uint32_t)kGreedyThreads;
    if (m != 0) {
      const uint32_t c_lo = c_hi - m;
      if ((uint32_t)tid == m - 1u) w.bcast = incl;
      const uint32_t i_to = st.ige[c_lo];                                // initial entries with cell >= c_lo
      win_take_initial(st, w, iptr, i_to, tid);
      iptr = i_to > iptr ? i_to : iptr;
      if ((uint32_t)tid < m) {
        uint32_t e = ld_l2(&st.whead[c]);
        if (e != kNone) {
          st_l2(&st.whead[c], kNone);
          if (cn != 0) while (e != kNone) { const EdgeRec r = st.er[e]; win_take(st, w, e, r); e = r.next; }
        }
        if (cn != 0) st_l2(&st.wcnt[c], 0u);
      }
      full_barrier();
      loaded += w.bcast;
      c_hi = c_lo;
      if (loaded) result = 0;
    }
    if (m < nvalid) {
      // Cell c* = c_hi - 1 does not fit whole (a tie group of thousands of equal means, typically): the threshold moves
      // INTO the cell.  Its items are a sorted array segment (initial edges) and an unordered list (created edges); thread 0
      // finds the list's K largest keys with a bounded min-heap in LDS, the new tau is the larger of the heap's minimum and
      // the key of the array entry RA places ahead, and everything above tau moves: at most K - 1 + RA items.
      const uint32_t room = st.wbudget - loaded;
      const uint32_t cs = c_hi - 1u;
      if (room >= (kWinMinPartial < st.wbudget / 8u ? kWinMinPartial : st.wbudget / 8u) || loaded == 0) {
        const uint32_t K = (room / 2u < kSelMax ? room / 2u : kSelMax) > 1u ? (room / 2u < kSelMax ? room / 2u : kSelMax) : 2u, RA = room > K ? room - K : 1u;
        const uint32_t seg_end = st.ige[cs];
        const uint32_t before = w.n;
        if (tid == 0) {
          uint32_t hn = 0, nlive = 0;
          for (uint32_t e = ld_l2(&st.whead[cs]); e != kNone;) {
            const EdgeRec r = st.er[e];
            if (r.seq != 0) {
              ++nlive;
              if (hn < K) {                                                   // push, sift up (min-heap by key)
                uint32_t i = hn++;
                while (i > 0) {
                  const uint32_t p = (i - 1u) >> 1;
                  if (!(sel_sal[p] > r.sal || (sel_sal[p] == r.sal && sel_seq[p] > r.seq))) break;
                  sel_sal[i] = sel_sal[p]; sel_seq[i] = sel_seq[p]; i = p;
                }
                sel_sal[i] = r.sal; sel_seq[i] = r.seq;
              } else if (r.sal > sel_sal[0] || (r.sal == sel_sal[0] && r.seq > sel_seq[0])) {   // replace the minimum, sift down
                uint32_t i = 0;
                while (true) {
                  uint32_t c = 2u * i + 1u;
                  if (c >= hn) break;
                  if (c + 1u < hn && (sel_sal[c + 1u] < sel_sal[c] || (sel_sal[c + 1u] == sel_sal[c] && sel_seq[c + 1u] < sel_seq[c]))) ++c;
                  if (!(sel_sal[c] < r.sal || (sel_sal[c] == r.sal && sel_seq[c] < r.seq))) break;
                  sel_sal[i] = sel_sal[c]; sel_seq[i] = sel_seq[c]; i = c;
                }
                sel_sal[i] = r.sal; sel_seq[i] = r.seq;
              }
            }
            e = r.next;
          }
          // tau from the list: the heap's minimum if the list holds more than the heap
          w.psal = nlive > K ? sel_sal[0] : -__builtin_inf(); w.pseq = nlive > K ? sel_seq[0] : 0ull;
        }
        full_barrier();
        double tsal = w.psal; unsigned long long tseq = w.pseq;
        const uint32_t iA = seg_end > iptr ? (seg_end - iptr < RA ? seg_end : iptr + RA) : iptr;
        if (iA < seg_end) {                                                     // array entries stay behind: their first one bounds tau
          const uint32_t et = st.isort[iA]; const double as = st.er[et].sal; const unsigned long long aq = st.isort_seq[iA];
          if (as > tsal || (as == tsal && aq > tseq)) { tsal = as; tseq = aq; }
        }
        if (tid == 0) w.bcast = 0;
        full_barrier();
        // array entries above tau (a prefix of [iptr, iA): the array is sorted)
        uint32_t mine = 0;
        for (uint32_t i = iptr + (uint32_t)tid; i < iA; i += kGreedyThreads) {
          const uint32_t e = st.isort[i]; const EdgeRec r = st.er[e];
          const unsigned long long q = st.isort_seq[i];                         // (a dead entry's seq is gone from its record)
          if (r.sal > tsal || (r.sal == tsal && q > tseq)) { ++mine; win_take(st, w, e, r); }
        }
        if (mine) atomicAdd(&w.bcast, mine);
        full_barrier();
        iptr += w.bcast;
        if (tid == 0) {
          // list nodes above tau move, the others stay linked
          uint32_t keep_head = kNone, keep_tail = kNone;
          for (uint32_t e = ld_l2(&st.whead[cs]); e != kNone;) {
            const EdgeRec r = st.er[e];
            if (r.seq != 0) {
              if (r.sal > tsal || (r.sal == tsal && r.seq > tseq)) win_take(st, w, e, r);
              else { if (keep_head == kNone) keep_head = e; else st.er[keep_tail].next = e; keep_tail = e; }
            }
            e = r.next;
          }
          if (keep_tail != kNone) st.er[keep_tail].next = kNone;
          st_l2(&st.whead[cs], keep_head);
        }
        full_barrier();
        const uint32_t moved = w.n - before;
        if (moved == 0u && w.bcast == 0u) {
          // nothing above the new tau and no array entry passed: the cell is empty, its count was too high (counts are upper
          // bounds: the batch kernel does not discount an edge that dies with exactly tau's key) -- on to the cells below
          full_barrier();
          if (tid == 0) st_l2(&st.wcnt[cs], 0u);
          c_hi = cs;
          continue;
        }
        if (tid == 0) { if (moved) atomicSub(&st.wcnt[cs], moved); w.cthr = (int)cs; w.tsal = tsal; w.tseq = tseq; w.iptr = iptr; }
        if (moved || w.bcast) result = 0;
        full_barrier();
        return result;
      }
      break;
    }
    if (loaded >= (kWinMinLoad < st.wbudget / 4u ? kWinMinLoad : st.wbudget / 4u)) break;      // else: a whole block of (nearly) empty cells, go on below it
  }
  full_barrier();
  if (tid == 0 && result != 2) { w.cthr = (int)c_hi - 1; w.tsal = __builtin_inf(); w.tseq = ~0ull; w.iptr = iptr; }
  full_barrier();
  if (result == 1 && c_floor != 0u) result = 3;
  return result;
}

template <bool COND>
__global__ __launch_bounds__(kGreedyThreads) void greedy_window_kernel(WinState st) {
  __shared__ WinShared w;
  __shared__ WinWork s;
  const int tid = threadIdx.x, lane = tid & 63;
  unsigned long long k = st.ctrl[0], ne = st.ctrl[1], pool_used = st.ctrl[2];
  uint32_t status = ST_RUN;
  if (tid == 0) {
    w.n = 0; w.nk = 0; w.kovf = 0; w.err = 0; s.nitems = 0; s.newcount = 0; s.bad = 0;
    w.cthr = (int)(long long)st.ctrl[5]; w.tsal = __longlong_as_double((long long)st.ctrl[6]); w.tseq = st.ctrl[7]; w.iptr = (uint32_t)st.ctrl[8];
  }
  for (uint32_t i = tid; i < kMarkSlots; i += kGreedyThreads) { s.mk[i] = 0; s.mv0[i] = 0; s.mv1[i] = 0; }
  for (uint32_t i = tid; i < kWinCap; i += kGreedyThreads) { w.seq[i] = 0; w.e[i] = 0; w.v[i] = 0; w.sal[i] = 0.0; }
  if (tid < 128) w.touched[tid >> 6][tid & 63] = 0;
  if (tid < kNW) { w.part[tid].sal = -__builtin_inf(); w.part[tid].seq = 0; w.part[tid].arg = 0; }
  full_barrier();
  const double smin = st.wrange[0], scale = st.wrange[1];
  uint32_t r2prev = kNone;
#ifdef GLIA_HMT_PROFILE
  unsigned long long wph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wlast = __builtin_readcyclecounter(), wtiter = wlast;
  unsigned long long wtb[5] = {0, 0, 0, 0, 0}, wnb[5] = {0, 0, 0, 0, 0}, wdb[5] = {0, 0, 0, 0, 0}, wreloads = 0, wcompacts = 0, wloaded = 0, winwin = 0, wdeps = 0;
#define WPH(i) do { if (tid == 0) { unsigned long long tn = __builtin_readcyclecounter(); wph[i] += tn - wlast; wlast = tn; } } while (0)
#else
#define WPH(i) do {} while (0)
#endif

  for (unsigned long long it = 0; it < st.max_iters; ++it) {
    const Key root = win_root(w, lane);
    if (root.seq == 0) {
      WPH(5);
      if (st.force_tree && k >= st.force_tree) { status = ST_NEED_TREE; break; }
      const int r = win_reload(st, w, tid, reinterpret_cast<double*>(&s.stage[0]), reinterpret_cast<unsigned long long*>(&s.stage[0]) + kSelMax);
#ifdef GLIA_HMT_PROFILE
      wreloads += 1; wloaded += w.n;
#endif
      WPH(6);
      if (r == 1) { status = ST_DONE; break; }
      if (r == 2) { status = ST_NEED_TREE; break; }
      if (r == 3) { status = ST_REBASE; break; }
      win_scan(st, w, tid, kNone, 0);
      r2prev = kNone;                      // (the reload's barriers waited for every store)
      continue;
    }
    const uint32_t slot = root.arg;
    const uint32_t e = w.e[slot], r0 = w.u[slot], r1 = w.v[slot];
    const uint2 h0r = w.hu[slot], h1r = w.hv[slot];
    const uint32_t wn_now = w.n;
    const int cthr = w.cthr; const double tsal = w.tsal; const unsigned long long tseq = w.tseq;
    const uint32_t off0 = h0r.x, len0 = h0r.y, off1 = h1r.x, len1 = h1r.y;
    const uint32_t total = len0 + len1;
    if (k >= (unsigned long long)st.R0) { status = ST_INTERNAL; break; }      // more merges than regions: the state is corrupt, stop before writing past the outputs
    const uint32_t r2 = st.R0 + (uint32_t)k;
    const uint32_t r2off = (uint32_t)pool_used;
    const int par = (int)(k & 1ull);
    // does this contraction read a list the previous one is still writing?  (r2prev's list and its neighbours' lists)
    const uint32_t tb0 = w.touched[par ^ 1][(r0 >> 5) & 63u], tb1 = w.touched[par ^ 1][(r1 >> 5) & 63u];
    const bool dep = r2prev != kNone && (r1 == r2prev || r0 == r2prev || ((tb0 >> (r0 & 31u)) & 1u) || ((tb1 >> (r1 & 31u)) & 1u));
    if (dep) {
      full_barrier();                    // (vmcnt(0) inside) the previous contraction's stores are done
#ifdef GLIA_HMT_PROFILE
      wdeps += 1;
#endif
    }
    WPH(0);
    if (COND) {
      // pre_merge condition (gadget/main_pre_merge.cxx:27-76), see the tree kernel: a rejected item leaves the queue for good
      unsigned long long sz0 = st.rsz[r0], sz1 = st.rsz[r1]; double su0 = st.rsum[r0], su1 = st.rsum[r1];
      const unsigned long long z2 = sz0 + sz1; const double w2 = su0 + su1;
      if (sz0 > sz1) { const unsigned long long t = sz0; sz0 = sz1; sz1 = t; const double d = su0; su0 = su1; su1 = d; }
      bool ok = sz0 < st.cond_t0;
      if (!ok && st.cond_n > 1) {
        if (sz0 < st.cond_t1 && sdivide(su0, (double)sz0, 0.0) > st.cond_rpb) ok = true;
        if (!ok && sz1 < st.cond_t1 && sdivide(su1, (double)sz1, 0.0) > st.cond_rpb) ok = true;
      }
      if (!ok) {
        full_barrier();                 // every thread has read the slot
        if (tid == 0) { w.seq[slot] = 0; st.er[e].seq = 0; }
        full_barrier();
        win_scan(st, w, tid, kNone, 0);
        continue;
      }
      if (tid == 0) { st.rsz[r2] = z2; st.rsum[r2] = w2; }               // TRegionMap::merge (updateRegion)
    }
    if (ne + total > st.Ecap) { status = ST_NEED_EDGES; break; }
    if (pool_used + total > st.pool_cap) { status = ST_NEED_POOL; break; }
    if (tid == 0) {
      w.seq[slot] = 0;                                                   // popped (the other threads read the rest of the slot only)
      st.order[3 * k + 0] = r0; st.order[3 * k + 1] = r1; st.order[3 * k + 2] = r2;
      st.sal_out[k] = root.sal;
      st.er[e].seq = 0;
    }
    if (tid >= 64 && tid < 128) w.touched[par][tid & 63] = 0;            // this contraction's bitmap (last read at the pop of the previous one)
    const bool small = total <= kMarkMax;

    // ---- the one round trip: the two lists; one table entry per distinct neighbour ----
    for (uint32_t i = tid; i < total; i += kGreedyThreads) {
      const bool side1 = i >= len0;
      const FatEntry fe = st.fpool[side1 ? off1 + (i - len0) : off0 + i];
      if (fe.eid == e || fe.eid == kNone) continue;
      if (small) {
        s.stage[i] = fe;
        uint32_t h = (fe.rs * 2654435761u) >> 21;
        while (true) {
          const uint32_t old = atomicCAS(&s.mk[h], 0u, fe.rs + 1u);
          if (old == 0u) { s.items[atomicAdd(&s.nitems, 1u)] = h; break; }
          if (old == fe.rs + 1u) break;
          h = (h + 1u) & (kMarkSlots - 1u);
        }
        (side1 ? s.mv1 : s.mv0)[h] = i + 1u;
      } else (side1 ? st.mark1 : st.mark0)[fe.rs] = i + 1u;
    }
    full_barrier();     // full: a wave that loaded has waited for its loads anyway, so its older stores are done for free
    WPH(1);
    // room for every new edge that may land in the window (total bounds their number)
    if (wn_now + total > st.wcap) {
#ifdef GLIA_HMT_PROFILE
      wcompacts += 1;
#endif
      // (round 4, found by the wave-skew build: this test used to read w.n again behind win_compact's barrier -- a wave that came out of
      // it early had already started to append this contraction's edges, a late one then saw a fuller window, flushed ALONE, and the
      // workgroup hung at mismatched barriers.  The window kernel of pre_merge is this code.)
      const uint32_t wn_live = win_compact(w, tid, st.wcap);
      if (wn_live + total > st.wcap) {
        win_flush(st, w, tid);
        if (total > st.wcap) {             // a contraction wider than the window: nothing of it goes there
          if (tid == 0) { w.cthr = (int)st.wB; w.tsal = __builtin_inf(); w.tseq = ~0ull; }
          full_barrier();
        }
      }
    }
    const int cthr2 = (wn_now + total > st.wcap) ? w.cthr : cthr;
    const double tsal2 = (wn_now + total > st.wcap) ? w.tsal : tsal;
    const unsigned long long tseq2 = (wn_now + total > st.wcap) ? w.tseq : tseq;
    const uint32_t nwork = small ? s.nitems : total;
    WPH(2);

    // ---- one new edge (rs, r2) per distinct neighbour (TBoundaryTable::update) ----
    bool bad = false;
    uint32_t pend_e = kNone, pend_old = kNone;          // a list push whose link is stored later (nobody waits for the atomic)
    for (uint32_t base = 0; base < nwork; base += kGreedyThreads) {
      const uint32_t i = base + tid;
      if (i >= nwork) break;
      FatEntry f0, f1;
      bool h0, h1;
      uint32_t rs;
      if (small) {
        const uint32_t h = s.items[i];
        rs = s.mk[h] - 1u;
        const uint32_t m0 = s.mv0[h], m1 = s.mv1[h];
        s.mk[h] = 0u; s.mv0[h] = 0u; s.mv1[h] = 0u;
        h0 = m0 != 0; h1 = m1 != 0;
        f0 = s.stage[h0 ? m0 - 1u : m1 - 1u]; f1 = s.stage[h1 ? m1 - 1u : m0 - 1u];
      } else {
        const bool side1 = i >= len0;
        const FatEntry fe = st.fpool[side1 ? off1 + (i - len0) : off0 + i];
        if (fe.eid == e || fe.eid == kNone) continue;
        rs = fe.rs;
        if (!side1) {
          const uint32_t m = st.mark1[rs];
          h0 = true; h1 = m != 0; f0 = fe;
          f1 = h1 ? st.fpool[off1 + (m - 1u - len0)] : fe;
        } else {
          if (st.mark0[rs] != 0u) continue;               // common neighbour: handled from the r0 side
          h0 = false; h1 = true; f0 = fe; f1 = fe;
        }
      }
      const uint32_t idx = atomicAdd(&s.newcount, 1u);
      const uint32_t newE = (uint32_t)ne + idx;
      atomicOr(&w.touched[par][(rs >> 5) & 63u], 1u << (rs & 31u));
      // util/struct_merge.hxx:62-76
      double first = 0.0;
      int second = 0;
      if (h0) { first += f0.mean * (int)f0.n; second += (int)f0.n; }
      if (h1) { first += f1.mean * (int)f1.n; second += (int)f1.n; }
      first = sdivide(first, (double)second, 0.0);
      if (first == -1.0) bad = true;                      // DUMMY -> "invalid boundary saliency" (:78-79)
      const uint32_t offRs = f0.off, posRs = f0.pos, lenRs = f0.len;      // rs's entry of the (r0,rs) edge -- or of (r1,rs) alone -- is reused
      if (h0 && h1) st.fpool[offRs + f1.pos].eid = kNone; // rs held two entries: the other becomes a tombstone
      const uint32_t cat = rs < r0 ? 0u : (h0 ? 1u : 2u);
      const unsigned long long seq = ((k + 1ull) << 32) | ((unsigned long long)cat << 30) | rs;
      const double sal = -first;
      EdgeRec* pe = &st.er[newE];
      uint4* pq4 = reinterpret_cast<uint4*>(pe);
      pq4[0] = make_uint4(rs, r2, posRs, idx);
      const unsigned long long mb = (unsigned long long)__double_as_longlong(first);
      pq4[1] = make_uint4((uint32_t)mb, (uint32_t)(mb >> 32), (uint32_t)second, kNone);
      pq4[2] = make_uint4(offRs, lenRs, r2off, 0u);                        // r2's length: stored below
      const unsigned long long sbits = (unsigned long long)__double_as_longlong(sal);
      pq4[3] = make_uint4((uint32_t)sbits, (uint32_t)(sbits >> 32), (uint32_t)seq, (uint32_t)(seq >> 32));
      FatEntry* pa = &st.fpool[offRs + posRs];
      FatEntry a; a.eid = newE; a.rs = r2; a.n = (uint32_t)second; a.pos = idx; a.off = r2off; a.len = 0; a.mean = first;   // len: stored below
      *pa = a;
      FatEntry b; b.eid = newE; b.rs = rs; b.n = (uint32_t)second; b.pos = posRs; b.off = offRs; b.len = lenRs; b.mean = first;
      st.fpool[r2off + idx] = b;
      if (small) { s.items[i] = offRs + posRs; s.newidx[i] = idx; }      // (this thread comes back to them below)
      const uint32_t cell = win_cell(sal, smin, scale, st.wB);
      if (win_above(cthr2, tsal2, tseq2, (int)cell, sal, seq)) {
        const uint32_t sl = atomicAdd(&w.n, 1u);
        win_put(w, sl, sal, seq, newE, rs, r2, make_uint2(offRs, lenRs), make_uint2(r2off, 0u));
      } else {
        if (pend_e != kNone) st.er[pend_e].next = pend_old;
        pend_e = newE; pend_old = atomicExch(&st.whead[cell], newE);
        atomicAdd(&st.wcnt[cell], 1u);
      }
      // the replaced edges leave the queue
#pragma unroll
      for (int side = 0; side < 2; ++side) {
        const bool hs = side ? h1 : h0;
        if (!hs) continue;
        const uint32_t de = side ? f1.eid : f0.eid;
        const double dsal = -(side ? f1.mean : f0.mean);
        const uint32_t dc = win_cell(dsal, smin, scale, st.wB);
        unsigned long long dq = 1;
        const bool tie = (int)dc == cthr2 && dsal == tsal2;       // tie with tau: the seq decides where the edge lives
        if (COND || tie) dq = st.er[de].seq;   // COND: 0 = rejected earlier, out of the queue
        st.er[de].seq = 0;
        if (dq != 0) {
          if (win_above(cthr2, tsal2, tseq2, (int)dc, dsal, dq)) {
            const uint32_t j = atomicAdd(&w.nk, 1u); if (j < kKillMax) w.kill[j] = de; else w.kovf = 1;
          } else atomicSub(&st.wcnt[dc], 1u);
        }
      }
    }
    if (bad) s.bad = 1;
    if (small) lds_barrier(); else full_barrier();         // the stores of this phase stay in flight
    if (s.bad) { if (pend_e != kNone) st.er[pend_e].next = pend_old; status = ST_BAD_SALIENCY; break; }
    WPH(3);
    const uint32_t newcount = s.newcount;
    // r2's list length is known now: complete the headers that point at it
    if (small) {
      for (uint32_t i = tid; i < nwork; i += kGreedyThreads) { st.fpool[s.items[i]].len = newcount; st.er[(uint32_t)ne + s.newidx[i]].hv.y = newcount; }
    } else {
      for (uint32_t j = tid; j < newcount; j += kGreedyThreads) {
        const FatEntry fb = st.fpool[r2off + j];
        st.fpool[fb.off + fb.pos].len = newcount;
        st.er[(uint32_t)ne + j].hv.y = newcount;
        st.mark0[fb.rs] = 0; st.mark1[fb.rs] = 0;
      }
    }
    // Every wave has to have READ s.newcount (above) before thread 0 clears it for the next contraction.  Rounds 2-3 cleared it here
    // without a barrier in between: a wave that came out of the last barrier a few hundred cycles late read 0, completed its headers
    // with length 0 and -- worse -- went on with its private copy of `ne` short by this contraction's edges, so the edges it created
    // later overwrote records of live ones (the rare pre_merge failure of round 3; the wide path of the batch kernel
    // always had this barrier).  kovf: the scan will ask the edge records which window items died, those stores must be done as well.
    if (w.kovf) full_barrier(); else lds_barrier();
    if (tid == 0) { st.adj_off[r2] = r2off; st.adj_len[r2] = newcount; s.nitems = 0; s.newcount = 0; }
    win_scan(st, w, tid, r2, newcount);
    if (pend_e != kNone) st.er[pend_e].next = pend_old;     // (the atomic has long returned; only a reload reads the link, behind a full barrier)
    r2prev = r2;
    WPH(4);
#ifdef GLIA_HMT_PROFILE
    if (tid == 0) {
      const unsigned long long tn = __builtin_readcyclecounter();
      const int b = total <= 64 ? 0 : total <= 512 ? 1 : total <= kMarkMax ? 2 : total <= 8192 ? 3 : 4;
      wtb[b] += tn - wtiter; wnb[b] += 1; wdb[b] += total; wtiter = tn; winwin += w.n;
    }
#endif
    k += 1; ne += newcount; pool_used += total;
  }
#ifdef GLIA_HMT_PROFILE
  if (tid == 0) printf("[window profile] merges %llu: pop %llu  lists+table %llu  room %llu  build %llu  finish+scan %llu  loop-top %llu  reload %llu (cycles); reloads %llu (items %llu) compactions %llu dependent %llu; mean window fill %llu\n",
                       k, wph[0], wph[1], wph[2], wph[3], wph[4], wph[5], wph[6], wreloads, wloaded, wcompacts, wdeps, k ? winwin / k : 0ull);
  if (tid == 0) printf("[window profile] by width (<=64, <=512, <=1408, <=8192, more): merges %llu %llu %llu %llu %llu  cycles %llu %llu %llu %llu %llu  entries %llu %llu %llu %llu %llu\n",
                       wnb[0], wnb[1], wnb[2], wnb[3], wnb[4], wtb[0], wtb[1], wtb[2], wtb[3], wtb[4], wdb[0], wdb[1], wdb[2], wdb[3], wdb[4]);
#endif
  // leave through the global lists: the next launch (or the tree kernel) starts from them
  full_barrier();
  win_flush(st, w, tid);
  if (tid == 0) {
    st.ctrl[0] = k; st.ctrl[1] = ne; st.ctrl[2] = pool_used; st.ctrl[3] = w.err ? (unsigned long long)ST_INTERNAL : status; st.ctrl[9] = w.err; st.ctrl[10] = w.n;
    st.ctrl[5] = (unsigned long long)(long long)w.cthr; st.ctrl[6] = (unsigned long long)__double_as_longlong(w.tsal); st.ctrl[7] = w.tseq; st.ctrl[8] = w.iptr;
  }
}

// =====================================================================================================================
// Batched contractions on the window queue (pb-mean linkage without a condition).
//
// One contraction of two small regions keeps a single wave busy (a few dozen list entries) and costs ~13 k cycles of
// pure sequence: pop, one global round trip, neighbour matching, a division, a dozen stores, one scan of the window.
// The other seven waves wait.  The queue's top items, however, are mostly far apart in the volume, and the greedy
// order of FAR-APART top items is known before any of them is contracted:
//   let c0 > c1 > ... be the top items of the queue (exact keys).  After contracting c0 the next pop is c1 provided
//   (a) no edge created by c0 has a saliency >= c1's (a created edge is newer, so it wins a tie), and
//   (b) c1's two regions are neither c0's regions nor neighbours of them (then c1's lists are untouched by c0);
//   by induction over the batch, member j is merge number k + j, creates region R0 + k + j and its new edges carry
//   seq = (k + j + 1) << 32 | ... exactly as in the one-by-one loop.
// A round: every wave finds the best and the second-best item of its share of the window; the items that beat every
// second-best are the exact top of the queue, in order.  Wave j COMPUTES member j on its own (lists into registers,
// neighbour matching in a private LDS table, new means), publishes what it creates (count, largest new saliency, a
// bitmap of the regions it touches); every wave then evaluates (a) and (b) for the whole batch and the valid prefix
// COMMITS (stores, queue inserts, deaths) in parallel.  Nothing is speculated on memory: a member that fails the check
// has only read.  Contractions with more than 64 list entries take the whole workgroup, one at a time.
// The result is bit-identical to the sequential kernels (same gate: SHA-1 of the whole 1024^3 order).
// =====================================================================================================================
constexpr uint32_t kBatchKill = 32;
constexpr int kMemP = 3;                   // list entries per lane of a batch member ...
constexpr uint32_t kMemMax = 192;          // ... and their limit (the wave's 256-slot neighbour table stays under 3/4 full)
struct BatchShared {
  alignas(16) Key part1[kNW];               // per-wave best / second-best of the last scan
  alignas(16) Key part2[kNW];
  uint32_t nkill, kovf; alignas(16) uint32_t kill[kBatchKill];
  uint32_t byrank[kNW];                     // candidate (wave) index of the batch member of rank r
  uint32_t bitmap[kNW][64];                 // regions a member touches (id mod 2048): its own two and every neighbour
  uint32_t m_newcount[kNW], m_total[kNW], m_ok[kNW];
  double m_maxsal[kNW];
  uint32_t bad;
};

// one pass over the window: applies the deaths of the last round, leaves every wave's best and second-best item
#ifdef GLIA_HMT_PROFILE
__device__ unsigned long long g_scanprof[8];
#define SCAN_T(i) do { if (tid == 0) { const unsigned long long tn_ = __builtin_readcyclecounter(); g_scanprof[i] += tn_ - st_; st_ = tn_; } } while (0)
#else
#define SCAN_T(i) do {} while (0)
#endif
__device__ __forceinline__ void batch_scan(const WinState& st, WinShared& w, BatchShared& b, int tid) {
#ifdef GLIA_HMT_PROFILE
  unsigned long long st_ = __builtin_readcyclecounter();
  if (tid == 0) g_scanprof[7] += 1;
#endif
  const uint32_t n = w.n < st.wcap ? w.n : st.wcap, nk = b.nkill < kBatchKill ? b.nkill : kBatchKill, kovf = b.kovf;
  // Slot ownership is STRIPED over the waves (lane l of wave v scans the l-th slot of chunk (v + l) mod 8 in every block
  // of 512): a reload fills consecutive slots with consecutive keys, and the exact top of the queue is only as long as
  // the run of best items that sit with different waves.  (Bank pattern of a wave's reads: that of consecutive slots.)
  const uint32_t own = 64u * (uint32_t)(((tid >> 6) + (tid & 63)) & 7) + (uint32_t)(tid & 63);
  unsigned long long q[kWinPer]; uint32_t e[kWinPer]; double sl[kWinPer];
#pragma unroll
  for (int j = 0; j < kWinPer; ++j) { const uint32_t i = own + (uint32_t)j * kGreedyThreads; q[j] = w.seq[i]; e[j] = w.e[i]; sl[j] = w.sal[i]; }
  const uint4 ka = *reinterpret_cast<const uint4*>(&b.kill[0]), kb = *reinterpret_cast<const uint4*>(&b.kill[4]);
  const uint32_t kl[8] = {ka.x, ka.y, ka.z, ka.w, kb.x, kb.y, kb.z, kb.w};
  Key k1, k2;
  k1.sal = -__builtin_inf(); k1.seq = 0; k1.arg = 0; k2 = k1;
  SCAN_T(0);
  // deaths: branch-free for the first eight (a short-circuit || / && chain compiles to one branch per term), a uniform
  // loop over the rest of the list, and -- only when the list overflowed -- a look at the edge records
  uint32_t deadm[kWinPer];
#pragma unroll
  for (int j = 0; j < kWinPer; ++j) {
    uint32_t d = 0;
#pragma unroll
    for (uint32_t t = 0; t < 8; ++t) d |= (uint32_t)(t < nk) & (uint32_t)(kl[t] == e[j]);
    deadm[j] = d;
  }
  if (nk > 8u) {                                                         // (uniform)
    for (uint32_t t = 8; t < nk; ++t) {
      const uint32_t kt = b.kill[t];
#pragma unroll
      for (int j = 0; j < kWinPer; ++j) deadm[j] |= (uint32_t)(kt == e[j]);
    }
    if (kovf) {                                                          // more deaths than the list holds (rare): ask the edge records
      full_barrier();                                                    // (the stores that mark them are performed)
#pragma unroll
      for (int j = 0; j < kWinPer; ++j) {
        const uint32_t i = own + (uint32_t)j * kGreedyThreads;
        if (i < n && q[j] != 0) deadm[j] |= (uint32_t)(st.er[e[j]].seq == 0);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < kWinPer; ++j) {
    const uint32_t i = own + (uint32_t)j * kGreedyThreads;
    const bool was = (i < n) & (q[j] != 0);
    const bool live = was & (deadm[j] == 0u);
    if (was & !live) w.seq[i] = 0;
    Key c; c.sal = live ? sl[j] : -__builtin_inf(); c.seq = live ? q[j] : 0ull; c.arg = i;
    const bool b1 = better(c, k1), b2 = better(c, k2);
    // new best: the old best becomes second; else new second if it beats the old second (field by field: selecting whole
    // structs goes through private memory)
    k2.sal = b1 ? k1.sal : (b2 ? c.sal : k2.sal); k2.seq = b1 ? k1.seq : (b2 ? c.seq : k2.seq); k2.arg = b1 ? k1.arg : (b2 ? c.arg : k2.arg);
    k1.sal = b1 ? c.sal : k1.sal; k1.seq = b1 ? c.seq : k1.seq; k1.arg = b1 ? c.arg : k1.arg;
  }
  SCAN_T(1);
  const Key m1 = wave_max_sal_first(k1);
  SCAN_T(2);
  const bool mine = (k1.seq == m1.seq) & (k1.arg == m1.arg) & (m1.seq != 0);
  Key kk; kk.sal = mine ? k2.sal : k1.sal; kk.seq = mine ? k2.seq : k1.seq; kk.arg = mine ? k2.arg : k1.arg;
  const Key m2 = wave_max_sal_first(kk);
  SCAN_T(3);
  if ((tid & 63) == 0) { b.part1[tid >> 6] = m1; b.part2[tid >> 6] = m2; }
  full_barrier();                  // ... and are performed here, before the next round loads the lists they rewrote
  SCAN_T(4);
  if (tid == 0) { b.nkill = 0; b.kovf = 0; }
}

#ifdef GLIA_HMT_PROFILE
__device__ unsigned long long g_wideprof[8];
#define WIDE_T(i) do { if (tid == 0) { const unsigned long long tn_ = __builtin_readcyclecounter(); g_wideprof[i] += tn_ - wt_; wt_ = tn_; } } while (0)
#else
#define WIDE_T(i) do {} while (0)
#endif
// The whole workgroup contracts ONE edge (more than 64 list entries): the body of greedy_window_kernel.
__device__ __forceinline__ uint32_t batch_contract_wide(const WinState& st, WinShared& w, WinWork& s, BatchShared& b, int tid, uint32_t slot, double rootsal,
                                                      unsigned long long k, unsigned long long ne, unsigned long long pool_used, uint32_t* newcount_out) {
  const double smin = st.wrange[0], scale = st.wrange[1];
  const uint32_t e = w.e[slot], r0 = w.u[slot], r1 = w.v[slot];
  const uint2 h0r = w.hu[slot], h1r = w.hv[slot];
  const uint32_t wn_now = w.n;
  const uint32_t off0 = h0r.x, len0 = h0r.y, off1 = h1r.x, len1 = h1r.y;
  const uint32_t total = len0 + len1;
  const uint32_t r2 = st.R0 + (uint32_t)k;
  const uint32_t r2off = (uint32_t)pool_used;
#ifdef GLIA_HMT_PROFILE
  unsigned long long wt_ = __builtin_readcyclecounter();
#endif
  lds_barrier();                                                        // every thread has read the slot
  WIDE_T(0);
  if (tid == 0) {
    w.seq[slot] = 0;
    st.order[3 * k + 0] = r0; st.order[3 * k + 1] = r1; st.order[3 * k + 2] = r2;
    st.sal_out[k] = rootsal;
    st.er[e].seq = 0;
    st.rdead[r0] = 1; st.rdead[r1] = 1;
  }
  const bool small = total <= kMarkMax;
  for (uint32_t i = tid; i < total; i += kGreedyThreads) {
    const bool side1 = i >= len0;
    const FatEntry fe = st.fpool[side1 ? off1 + (i - len0) : off0 + i];
    if (fe.eid == e || fe.eid == kNone) continue;
    if (small) {
      s.stage[i] = fe;
      uint32_t h = (fe.rs * 2654435761u) >> 21;
      while (true) {
        const uint32_t old = atomicCAS(&s.mk[h], 0u, fe.rs + 1u);
        if (old == 0u) { s.items[atomicAdd(&s.nitems, 1u)] = h; break; }
        if (old == fe.rs + 1u) break;
        h = (h + 1u) & (kMarkSlots - 1u);
      }
      (side1 ? s.mv1 : s.mv0)[h] = i + 1u;
    } else (side1 ? st.mark1 : st.mark0)[fe.rs] = i + 1u;
  }
  if (small) lds_barrier(); else full_barrier();      // (the global mark arrays are read by other threads below)
  WIDE_T(1);
  if (wn_now + total > st.wcap && wn_now > st.wcap / 2u) win_compact(w, tid, st.wcap);      // (holes out; a full window spills, see win_evict)
  WIDE_T(2);
  const int cthr = w.cthr; const double tsal = w.tsal; const unsigned long long tseq = w.tseq;
  const uint32_t nwork = small ? s.nitems : total;
  const uint32_t lenR2 = small ? nwork : 0u;        // small case: every table item becomes exactly one new edge, so r2's list length is known here
  bool bad = false;
  uint32_t pend_e = kNone, pend_old = kNone;
  // Two instances of the loop: the LDS-table case must not share code with the one that loads from global memory -- where
  // the two meet the compiler waits for "every memory operation", and that counter includes the stores of earlier rounds.
  auto rounds = [&](auto small_tag) {
    constexpr bool SMALL = decltype(small_tag)::value;
    for (uint32_t base = 0; base < nwork; base += kGreedyThreads) {
      const uint32_t i = base + tid;
      if (i >= nwork) break;
      FatEntry f0, f1;
      bool h0, h1;
      uint32_t rs;
      if constexpr (SMALL) {
        const uint32_t h = s.items[i];
        rs = s.mk[h] - 1u;
        const uint32_t m0 = s.mv0[h], m1 = s.mv1[h];
        s.mk[h] = 0u; s.mv0[h] = 0u; s.mv1[h] = 0u;
        h0 = m0 != 0; h1 = m1 != 0;
        f0 = s.stage[h0 ? m0 - 1u : m1 - 1u]; f1 = s.stage[h1 ? m1 - 1u : m0 - 1u];
      } else {
        const bool side1 = i >= len0;
        const FatEntry fe = st.fpool[side1 ? off1 + (i - len0) : off0 + i];
        if (fe.eid == e || fe.eid == kNone) continue;
        rs = fe.rs;
        if (!side1) {
          const uint32_t m = st.mark1[rs];
          h0 = true; h1 = m != 0; f0 = fe;
          f1 = h1 ? st.fpool[off1 + (m - 1u - len0)] : fe;
        } else {
          if (st.mark0[rs] != 0u) continue;               // common neighbour: handled from the r0 side
          h0 = false; h1 = true; f0 = fe; f1 = fe;
        }
      }
      const uint32_t idx = atomicAdd(&s.newcount, 1u);
      const uint32_t newE = (uint32_t)ne + idx;
      double first = 0.0;                                  // util/struct_merge.hxx:62-76
      int second = 0;
      if (h0) { first += f0.mean * (int)f0.n; second += (int)f0.n; }
      if (h1) { first += f1.mean * (int)f1.n; second += (int)f1.n; }
      first = sdivide(first, (double)second, 0.0);
      if (first == -1.0) bad = true;                      // DUMMY -> "invalid boundary saliency" (:78-79)
      const uint32_t offRs = f0.off, posRs = f0.pos, lenRs = f0.len;
      if (h0 && h1) st.fpool[offRs + f1.pos].eid = kNone;
      const uint32_t cat = rs < r0 ? 0u : (h0 ? 1u : 2u);
      const unsigned long long seq = ((k + 1ull) << 32) | ((unsigned long long)cat << 30) | rs;
      const double sal = -first;
      uint4* pq4 = reinterpret_cast<uint4*>(&st.er[newE]);
      pq4[0] = make_uint4(rs, r2, posRs, idx);
      const unsigned long long mb = (unsigned long long)__double_as_longlong(first);
      pq4[1] = make_uint4((uint32_t)mb, (uint32_t)(mb >> 32), (uint32_t)second, kNone);
      pq4[2] = make_uint4(offRs, lenRs, r2off, lenR2);                     // (wide case: r2's length is stored below)
      const unsigned long long sbits = (unsigned long long)__double_as_longlong(sal);
      pq4[3] = make_uint4((uint32_t)sbits, (uint32_t)(sbits >> 32), (uint32_t)seq, (uint32_t)(seq >> 32));
      FatEntry a; a.eid = newE; a.rs = r2; a.n = (uint32_t)second; a.pos = idx; a.off = r2off; a.len = lenR2; a.mean = first;
      st.fpool[offRs + posRs] = a;
      FatEntry bb; bb.eid = newE; bb.rs = rs; bb.n = (uint32_t)second; bb.pos = posRs; bb.off = offRs; bb.len = lenRs; bb.mean = first;
      st.fpool[r2off + idx] = bb;
      const uint32_t cell = win_cell(sal, smin, scale, st.wB);
      uint32_t sl = kWinCap;
      const bool above = win_above(cthr, tsal, tseq, (int)cell, sal, seq);
      if (above) {
        sl = atomicAdd(&w.n, 1u);
        if (sl < st.wcap) win_put(w, sl, sal, seq, newE, rs, r2, make_uint2(offRs, lenRs), make_uint2(r2off, lenR2));
        else atomicMax(&w.spill_ord, f64_ord(sal));          // the window is full: tau will rise above this item
      }
      if (sl >= st.wcap && (above || cell >= st.wch)) {                   // (an item below the horizon is not queued at all)
        if (pend_e != kNone) st.er[pend_e].next = pend_old;
        pend_e = newE; pend_old = atomicExch(&st.whead[cell], newE);
        atomicAdd(&st.wcnt[cell], 1u);
      }
  #pragma unroll
      for (int side = 0; side < 2; ++side) {
        const bool hs = side ? h1 : h0;
        if (!hs) continue;
        const uint32_t de = side ? f1.eid : f0.eid;
        const double dsal = -(side ? f1.mean : f0.mean);
        const uint32_t dc = win_cell(dsal, smin, scale, st.wB);
        // A key equal to tau's in tau's cell: the seq would decide, and it sits in the edge record -- a load in the middle of
        // the store stream, which on this hardware makes the wave wait for every store before it (one counter for both).  Such
        // an edge is treated as a window item instead: a kill that matches nothing is harmless, and its cell's count stays one
        // too high until the next baseline (counts are upper bounds: a reload walks a list whenever its count is not zero).
        const bool tie = (int)dc == cthr && dsal == tsal;
        if (dc >= st.wch) {                                  // (below the horizon: WinState::rdead speaks for the edge)
          st.er[de].seq = 0;
          if (tie || win_above(cthr, tsal, tseq, (int)dc, dsal, 0ull)) {
            const uint32_t j = atomicAdd(&b.nkill, 1u); if (j < kBatchKill) b.kill[j] = de; else b.kovf = 1;
          } else atomicSub(&st.wcnt[dc], 1u);
        }
      }
    }
  };
  if (small) rounds(std::true_type{}); else rounds(std::false_type{});
  if (bad) b.bad = 1;
  WIDE_T(3);
  if (small) lds_barrier(); else full_barrier();      // (wide case: r2's new list entries are read back below, by other threads than wrote them)
  WIDE_T(4);
  const uint32_t newcount = s.newcount;
  if (!small) {
    for (uint32_t j = tid; j < newcount; j += kGreedyThreads) {
      const FatEntry fb = st.fpool[r2off + j];
      st.fpool[fb.off + fb.pos].len = newcount;
      st.er[(uint32_t)ne + j].hv.y = newcount;
      st.mark0[fb.rs] = 0; st.mark1[fb.rs] = 0;
    }
  }
  if (!small) for (uint32_t i = tid; i < (w.n < st.wcap ? w.n : st.wcap); i += kGreedyThreads) if (w.v[i] == r2) w.hv[i].y = newcount;      // window items of r2: its list length
  if (pend_e != kNone) st.er[pend_e].next = pend_old;
  if (tid == 0) { st.adj_off[r2] = r2off; st.adj_len[r2] = newcount; }
  lds_barrier();                   // (the scan that follows ends with the full barrier)
  WIDE_T(5);
  if (tid == 0) { s.nitems = 0; s.newcount = 0; }
  *newcount_out = newcount;
  return total;
}

__global__ __launch_bounds__(kGreedyThreads) void greedy_batch_kernel(WinState st) {
  __shared__ WinShared w;
  __shared__ WinWork s;
  __shared__ BatchShared b;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long k = st.ctrl[0], ne = st.ctrl[1], pool_used = st.ctrl[2];
  uint32_t status = ST_RUN;
  if (tid == 0) {
    w.n = 0; w.nk = 0; w.kovf = 0; w.err = 0; w.spill_ord = 0; s.nitems = 0; s.newcount = 0; s.bad = 0; b.nkill = 0; b.kovf = 0; b.bad = 0;
    w.cthr = (int)(long long)st.ctrl[5]; w.tsal = __longlong_as_double((long long)st.ctrl[6]); w.tseq = st.ctrl[7]; w.iptr = (uint32_t)st.ctrl[8];
  }
  for (uint32_t i = tid; i < kMarkSlots; i += kGreedyThreads) { s.mk[i] = 0; s.mv0[i] = 0; s.mv1[i] = 0; }
  for (uint32_t i = tid; i < kWinCap; i += kGreedyThreads) { w.seq[i] = 0; w.e[i] = 0; w.v[i] = 0; w.sal[i] = 0.0; }
  b.bitmap[wave][lane] = 0;
  if (tid < kNW) { Key z; z.sal = -__builtin_inf(); z.seq = 0; z.arg = 0; b.part1[tid] = z; b.part2[tid] = z; }
  full_barrier();
  const double smin = st.wrange[0], scale = st.wrange[1];
  uint32_t pend_e = kNone, pend_old = kNone;          // (per lane) a list push whose link is stored a round later
  constexpr uint32_t kTab = kMarkSlots / kNW;         // private neighbour table of a wave
  uint32_t* const tk = &s.mk[wave * kTab]; uint32_t* const t0 = &s.mv0[wave * kTab]; uint32_t* const t1 = &s.mv1[wave * kTab];
  // ... and what a slot's (r1, rs) entry carries for the lane that owns the (r0, rs) entry; lives in the staging area of
  // the whole-workgroup path (never active at the same time)
  static_assert(sizeof(s.stage) >= (size_t)kNW * kTab * 20, "per-wave staging");
  uint32_t* const sg_eid = reinterpret_cast<uint32_t*>(&s.stage[0]) + wave * kTab;
  uint32_t* const sg_n = reinterpret_cast<uint32_t*>(&s.stage[0]) + (kNW + wave) * kTab;
  uint32_t* const sg_pos = reinterpret_cast<uint32_t*>(&s.stage[0]) + (2 * kNW + wave) * kTab;
  double* const sg_mean = reinterpret_cast<double*>(reinterpret_cast<uint32_t*>(&s.stage[0]) + 3 * kNW * kTab) + wave * kTab;
#ifdef GLIA_HMT_PROFILE
  unsigned long long bph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, blast = __builtin_readcyclecounter(), brounds = 0, bmembers = 0, bvalid = 0, bwide = 0, bcut_sal = 0, bcut_dep = 0;
  unsigned long long bw_n[4] = {0, 0, 0, 0}, bw_cyc[4] = {0, 0, 0, 0}, bw_ent[4] = {0, 0, 0, 0}, bw_new[4] = {0, 0, 0, 0};
#define BPH(i) do { if (tid == 0) { unsigned long long tn = __builtin_readcyclecounter(); bph[i] += tn - blast; blast = tn; } } while (0)
#else
#define BPH(i) do {} while (0)
#endif

  for (unsigned long long it = 0; it < st.max_iters; ++it) {
    // lists grow garbage (dead nodes are only dropped when their cell is loaded): time for a new baseline?
    if (ne - st.ne_base > st.rebase_after) { status = ST_REBASE; break; }
    // ---- the exact top of the queue, in order: per-wave bests that beat every per-wave second-best ----
    const int gi = lane >> 3, gj = lane & 7;                             // an 8 x 8 grid of (i, j) comparisons per wave
    const Key A = b.part1[gi], B = b.part1[gj], C = b.part2[gi];
    const unsigned long long beats = __ballot(better(A, B)), under = __ballot(better(C, B));
    const unsigned long long col = 0x0101010101010101ull << gj;
    const uint32_t rank = (uint32_t)__popcll(beats & col);               // position of candidate gj in the order
    const bool cand_ok = B.seq != 0 && (under & col) == 0;               // it beats every second-best: part of the exact top
    if (gi == 0 && cand_ok) b.byrank[rank] = (uint32_t)gj;               // (every wave writes the same values)
    const unsigned long long okmask = __ballot(gi == 0 && cand_ok);      // (bit j = candidate j)
    const uint32_t M = (uint32_t)__popcll(okmask);
    if (M == 0) {
      // no live item in the window
      BPH(5);
      if (pend_e != kNone) { st.er[pend_e].next = pend_old; pend_e = kNone; }
      if (st.force_tree && k >= st.force_tree) { status = ST_NEED_TREE; break; }
      const int r = win_reload(st, w, tid, reinterpret_cast<double*>(&s.stage[0]), reinterpret_cast<unsigned long long*>(&s.stage[0]) + kSelMax);
      BPH(6);
      if (r == 1) { status = ST_DONE; break; }
      if (r == 2) { status = ST_NEED_TREE; break; }
      if (r == 3) { status = ST_REBASE; break; }                          // the queue continues below the horizon: new baseline
      batch_scan(st, w, b, tid);
      continue;
    }
    // the candidate this wave is responsible for: the one of rank `wave`
    const unsigned long long minemask = __ballot(gi == 0 && cand_ok && rank == (uint32_t)wave);
    const bool member = minemask != 0;                                   // (uniform per wave)
    const int cj = member ? (int)__builtin_ctzll(minemask) : 0;
    const Key me = b.part1[cj];
    const uint32_t slot = me.arg;
    // member data (every wave reads its own; waves without a member read a harmless slot)
    const uint32_t e = w.e[slot], r0 = w.u[slot], r1 = w.v[slot];
    const uint2 h0r = w.hu[slot], h1r = w.hv[slot];
    const uint32_t off0 = h0r.x, len0 = h0r.y, off1 = h1r.x, len1 = h1r.y;
    const uint32_t total = len0 + len1;
    if (k >= (unsigned long long)st.R0) { status = ST_INTERNAL; break; }      // more merges than regions: the state is corrupt
    // the best candidate decides: wide -> the whole workgroup takes it alone
    const unsigned long long firstmask = __ballot(gi == 0 && cand_ok && rank == 0u);
    const Key top = b.part1[__builtin_ctzll(firstmask)];
    const uint2 th0 = w.hu[top.arg], th1 = w.hv[top.arg];
    const uint32_t top_total = th0.y + th1.y;
    BPH(0);
    if (top_total > kMemMax) {
      if (ne + top_total > st.Ecap) { status = ST_NEED_EDGES; break; }
      if (pool_used + top_total > st.pool_cap) { status = ST_NEED_POOL; break; }
      uint32_t newcount = 0;
#ifdef GLIA_HMT_PROFILE
      const unsigned long long tw0 = __builtin_readcyclecounter();
#endif
      const uint32_t tt = batch_contract_wide(st, w, s, b, tid, top.arg, top.sal, k, ne, pool_used, &newcount);
      if (b.bad) { status = ST_BAD_SALIENCY; break; }
      k += 1; ne += newcount; pool_used += tt;
#ifdef GLIA_HMT_PROFILE
      bwide += 1;
      { const int cls = tt <= 512u ? 0 : tt <= kMarkMax ? 1 : tt <= 8192u ? 2 : 3; bw_n[cls] += 1; bw_cyc[cls] += __builtin_readcyclecounter() - tw0; bw_ent[cls] += tt; bw_new[cls] += newcount; }
#endif
      batch_scan(st, w, b, tid);
      if (w.spill_ord) { if (pend_e != kNone) { st.er[pend_e].next = pend_old; pend_e = kNone; } win_evict(st, w, tid); batch_scan(st, w, b, tid); }
      BPH(4);
      continue;
    }
    // ---- compute: wave j works out member j (rank order) on its own, up to kMemP list entries per lane; wider members end the batch ----
    const bool narrow = member && total <= kMemMax;
    b.bitmap[wave][lane] = 0;
    FatEntry fe[kMemP];
    bool act[kMemP], side1[kMemP];
    uint32_t h[kMemP];
#pragma unroll
    for (int p = 0; p < kMemP; ++p) {
      const uint32_t i = (uint32_t)lane + 64u * (uint32_t)p;
      const bool inlist = narrow && i < total;
      side1[p] = i >= len0;
      fe[p].eid = kNone; fe[p].rs = 0; fe[p].n = 0; fe[p].pos = 0; fe[p].off = 0; fe[p].len = 0; fe[p].mean = 0.0;
      if (inlist) fe[p] = st.fpool[side1[p] ? off1 + (i - len0) : off0 + i];
      act[p] = inlist;
    }
    if (pend_e != kNone) { st.er[pend_e].next = pend_old; pend_e = kNone; }      // (last round's atomic has returned with these loads)
#pragma unroll
    for (int p = 0; p < kMemP; ++p) {
      act[p] = act[p] && fe[p].eid != e && fe[p].eid != kNone;
      h[p] = (fe[p].rs * 2654435761u) >> 24;
      if (act[p]) {
        while (true) {
          const uint32_t old = atomicCAS(&tk[h[p]], 0u, fe[p].rs + 1u);
          if (old == 0u || old == fe[p].rs + 1u) break;
          h[p] = (h[p] + 1u) & (kTab - 1u);
        }
        if (side1[p]) {                                  // the (r1, rs) entry: its data waits in the table for the (r0, rs) entry's lane
          t1[h[p]] = 1u;
          sg_eid[h[p]] = fe[p].eid; sg_n[h[p]] = fe[p].n; sg_pos[h[p]] = fe[p].pos; sg_mean[h[p]] = fe[p].mean;
        } else t0[h[p]] = 1u;
        atomicOr(&b.bitmap[wave][(fe[p].rs >> 5) & 63u], 1u << (fe[p].rs & 31u));
      }
    }
    if (narrow && lane == 0) { atomicOr(&b.bitmap[wave][(r0 >> 5) & 63u], 1u << (r0 & 31u)); atomicOr(&b.bitmap[wave][(r1 >> 5) & 63u], 1u << (r1 & 31u)); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // (a wave's LDS operations execute in order)
    bool owner[kMemP], both[kMemP];
    uint32_t p_eid[kMemP], p_n[kMemP], p_pos[kMemP];
    double p_mean[kMemP];
#pragma unroll
    for (int p = 0; p < kMemP; ++p) {
      const uint32_t m0 = act[p] ? t0[h[p]] : 0u, m1 = act[p] ? t1[h[p]] : 0u;
      owner[p] = act[p] && (!side1[p] || m0 == 0u);
      both[p] = owner[p] && !side1[p] && m1 != 0u;
      p_eid[p] = sg_eid[h[p]]; p_n[p] = sg_n[h[p]]; p_pos[p] = sg_pos[h[p]]; p_mean[p] = sg_mean[h[p]];
    }
#pragma unroll
    for (int p = 0; p < kMemP; ++p) if (act[p]) { tk[h[p]] = 0u; t0[h[p]] = 0u; t1[h[p]] = 0u; }      // the table is clean again
    double first[kMemP]; int second[kMemP]; uint32_t idx[kMemP];
    bool bad = false;
    uint32_t newcount = 0;
    double mx = -__builtin_inf();
#pragma unroll
    for (int p = 0; p < kMemP; ++p) {
      const bool h0 = owner[p] && !side1[p], h1 = owner[p] && (side1[p] || both[p]);
      // util/struct_merge.hxx:62-76 (operand order: the (r0,rs) item first)
      double f = 0.0;
      int sc = 0;
      if (h0) { f += fe[p].mean * (int)fe[p].n; sc += (int)fe[p].n; }
      if (h1) { const double m = both[p] ? p_mean[p] : fe[p].mean; const int nn = both[p] ? (int)p_n[p] : (int)fe[p].n; f += m * nn; sc += nn; }
      f = owner[p] ? sdivide(f, (double)sc, 0.0) : 0.0;
      bad = bad || (owner[p] && f == -1.0);               // DUMMY -> "invalid boundary saliency" (:78-79)
      first[p] = f; second[p] = sc;
      const unsigned long long ownmask = __ballot(owner[p]);
      idx[p] = newcount + (uint32_t)__popcll(ownmask & ((1ull << lane) - 1ull));
      newcount += (uint32_t)__popcll(ownmask);
      mx = (owner[p] && -f > mx) ? -f : mx;              // the largest saliency this member creates
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const double o = __shfl_xor(mx, d); mx = o > mx ? o : mx; }
    if (lane == 0) { b.m_newcount[wave] = narrow ? newcount : 0u; b.m_total[wave] = narrow ? total : 0xFFFFFFFFu; b.m_maxsal[wave] = mx; }
    if (bad) b.bad = 1;
    // the window's fill, read by every wave BEFORE the barrier: behind it the committing waves raise w.n, and the "compact first?" test below
    // has to come out the same in every wave (it guards barriers) -- on its own late read of w.n a slow wave could take the branch alone
    const uint32_t wn_round = w.n;
    lds_barrier();                 // (LDS data only; the round's global stores are waited for at the END of the scan that follows the commit)
    BPH(1);
    if (b.bad) { status = ST_BAD_SALIENCY; break; }
    // ---- validate: the longest prefix of the batch whose order is certain (lane j checks member j) ----
    uint32_t V, ne_off, pool_off, my_ne, my_pool;
    {
      const uint32_t j = (uint32_t)lane & 7u;
      const Key kj = b.part1[b.byrank[j]];
      const uint32_t u = w.u[kj.arg], v = w.v[kj.arg];
      const uint32_t tj = b.m_total[j], nj = b.m_newcount[j];
      double pm = -__builtin_inf();
      uint32_t hit = 0, pre_n = 0, pre_t = 0;
#pragma unroll
      for (uint32_t i = 0; i < (uint32_t)kNW; ++i) {
        const double ms = b.m_maxsal[i];
        const uint32_t bu = b.bitmap[i][(u >> 5) & 63u], bv = b.bitmap[i][(v >> 5) & 63u];
        const uint32_t ti = b.m_total[i], ni = b.m_newcount[i];
        if (i < j) { pm = ms > pm ? ms : pm; hit |= ((bu >> (u & 31u)) | (bv >> (v & 31u))) & 1u; pre_n += ni; pre_t += ti; }
      }
      const bool okj = j < M && tj != 0xFFFFFFFFu && (j == 0u || (kj.sal > pm && hit == 0u));
      const uint32_t okbits = (uint32_t)(__ballot(lane < kNW && okj) & 0xFFull);
      V = (uint32_t)__builtin_ctz(~okbits);                               // members 0 .. V-1 are certain
#ifdef GLIA_HMT_PROFILE
      if (V < M) { const uint32_t cutsal = (uint32_t)(__ballot(lane < kNW && j == V && !(kj.sal > pm)) & 0xFFull); if (cutsal) bcut_sal += 1; else bcut_dep += 1; }
#endif
      // offsets of this wave's member, totals of the valid prefix
      my_ne = (uint32_t)__builtin_amdgcn_readlane((int)pre_n, wave); my_pool = (uint32_t)__builtin_amdgcn_readlane((int)pre_t, wave);
      const uint32_t last = V - 1u;
      ne_off = (uint32_t)__shfl((int)(pre_n + nj), (int)last); pool_off = (uint32_t)__shfl((int)(pre_t + tj), (int)last);
    }
    const uint32_t sum_tot = pool_off;
    if (ne + sum_tot > st.Ecap) { status = ST_NEED_EDGES; break; }
    if (pool_used + sum_tot > st.pool_cap) { status = ST_NEED_POOL; break; }
    // room in the window for everything the batch may insert (the popped items leave first: a flush must not see them)
    bool popped = false;
    if (wn_round + ne_off > st.wcap && wn_round > st.wcap / 2u) {          // holes out (a full window spills, see win_evict)
      if ((uint32_t)wave < V && lane == 0) w.seq[slot] = 0;
      popped = true;                                                     // (slot numbers are void after a compaction)
      lds_barrier();
      win_compact(w, tid, st.wcap);
    }
    const int cthr = w.cthr; const double tsal = w.tsal; const unsigned long long tseq = w.tseq;
    BPH(2);
    // ---- commit: the valid members, each by its own wave ----
    if ((uint32_t)wave < V) {
      const unsigned long long kk = k + (unsigned long long)wave;
      const uint32_t r2 = st.R0 + (uint32_t)kk;
      const uint32_t r2off = (uint32_t)pool_used + my_pool;
      if (lane == 0) {
        if (!popped) w.seq[slot] = 0;                                    // popped
        st.order[3 * kk + 0] = r0; st.order[3 * kk + 1] = r1; st.order[3 * kk + 2] = r2;
        st.sal_out[kk] = me.sal;
        st.er[e].seq = 0;
        st.rdead[r0] = 1; st.rdead[r1] = 1;
        st.adj_off[r2] = r2off; st.adj_len[r2] = newcount;
      }
#pragma unroll
      for (int p = 0; p < kMemP; ++p) {
        if (!owner[p]) continue;
        const bool h0 = !side1[p];
        const uint32_t rs = fe[p].rs;
        const uint32_t newE = (uint32_t)ne + my_ne + idx[p];
        const uint32_t offRs = fe[p].off, posRs = fe[p].pos, lenRs = fe[p].len;   // rs's entry of the (r0,rs) edge -- or of (r1,rs) alone -- is reused
        if (both[p]) st.fpool[offRs + p_pos[p]].eid = kNone;             // rs held two entries: the other becomes a tombstone
        const uint32_t cat = rs < r0 ? 0u : (h0 ? 1u : 2u);
        const unsigned long long seq = ((kk + 1ull) << 32) | ((unsigned long long)cat << 30) | rs;
        const double sal = -first[p];
        uint4* pq4 = reinterpret_cast<uint4*>(&st.er[newE]);
        pq4[0] = make_uint4(rs, r2, posRs, idx[p]);
        const unsigned long long mb = (unsigned long long)__double_as_longlong(first[p]);
        pq4[1] = make_uint4((uint32_t)mb, (uint32_t)(mb >> 32), (uint32_t)second[p], kNone);
        pq4[2] = make_uint4(offRs, lenRs, r2off, newcount);
        const unsigned long long sbits = (unsigned long long)__double_as_longlong(sal);
        pq4[3] = make_uint4((uint32_t)sbits, (uint32_t)(sbits >> 32), (uint32_t)seq, (uint32_t)(seq >> 32));
        FatEntry a; a.eid = newE; a.rs = r2; a.n = (uint32_t)second[p]; a.pos = idx[p]; a.off = r2off; a.len = newcount; a.mean = first[p];
        st.fpool[offRs + posRs] = a;
        FatEntry bb; bb.eid = newE; bb.rs = rs; bb.n = (uint32_t)second[p]; bb.pos = posRs; bb.off = offRs; bb.len = lenRs; bb.mean = first[p];
        st.fpool[r2off + idx[p]] = bb;
        const uint32_t cell = win_cell(sal, smin, scale, st.wB);
        uint32_t sl = kWinCap;
        const bool above = win_above(cthr, tsal, tseq, (int)cell, sal, seq);
        if (above) {
          sl = atomicAdd(&w.n, 1u);
          if (sl < st.wcap) win_put(w, sl, sal, seq, newE, rs, r2, make_uint2(offRs, lenRs), make_uint2(r2off, newcount));
          else atomicMax(&w.spill_ord, f64_ord(sal));      // the window is full: tau will rise above this item
        }
        if (sl >= st.wcap && (above || cell >= st.wch)) {               // (an item below the horizon is not queued at all)
          if (pend_e != kNone) st.er[pend_e].next = pend_old;
          pend_e = newE; pend_old = atomicExch(&st.whead[cell], newE);
          atomicAdd(&st.wcnt[cell], 1u);
        }
        // the replaced edges leave the queue
#pragma unroll
        for (int side = 0; side < 2; ++side) {
          const bool hs = side ? (side1[p] || both[p]) : h0;
          if (!hs) continue;
          const uint32_t de = (side && both[p]) ? p_eid[p] : fe[p].eid;
          const double dsal = -((side && both[p]) ? p_mean[p] : fe[p].mean);
          const uint32_t dc = win_cell(dsal, smin, scale, st.wB);
          const bool tie = (int)dc == cthr && dsal == tsal;                // (no look at the record's seq: see batch_contract_wide)
          if (dc >= st.wch) {                                // (below the horizon: WinState::rdead speaks for the edge)
            st.er[de].seq = 0;
            if (tie || win_above(cthr, tsal, tseq, (int)dc, dsal, 0ull)) {
              const uint32_t j = atomicAdd(&b.nkill, 1u); if (j < kBatchKill) b.kill[j] = de; else b.kovf = 1;
            } else atomicSub(&st.wcnt[dc], 1u);
          }
        }
      }
    }
    lds_barrier();                   // the commit's global stores stay in flight through the scan ...
    BPH(3);
#ifdef GLIA_HMT_PROFILE
    brounds += 1; bmembers += M; bvalid += V;
#endif
    k += V; ne += ne_off; pool_used += pool_off;
    batch_scan(st, w, b, tid);
    if (w.spill_ord) { if (pend_e != kNone) { st.er[pend_e].next = pend_old; pend_e = kNone; } win_evict(st, w, tid); batch_scan(st, w, b, tid); }
    BPH(4);
  }
#ifdef GLIA_HMT_PROFILE
  if (tid == 0) printf("[batch profile] merges %llu: select %llu  compute %llu  validate %llu  commit %llu  scan %llu  loop-top %llu  reload %llu (cycles); rounds %llu candidates %llu committed %llu (cut by saliency %llu, by adjacency %llu) wide %llu\n",
                       k, bph[0], bph[1], bph[2], bph[3], bph[4], bph[5], bph[6], brounds, bmembers, bvalid, bcut_sal, bcut_dep, bwide);
  if (tid == 0) printf("[batch profile] scan phases (cumulative cycles, wave 0): loads %llu  compare %llu  max1 %llu  max2 %llu  barrier %llu  calls %llu\n",
                       g_scanprof[0], g_scanprof[1], g_scanprof[2], g_scanprof[3], g_scanprof[4], g_scanprof[7]);
  if (tid == 0) printf("[batch profile] wide phases (cumulative cycles): entry-barrier %llu  lists+table %llu  compact %llu  main loop (wave 0) %llu  loop barrier %llu  tail %llu\n",
                       g_wideprof[0], g_wideprof[1], g_wideprof[2], g_wideprof[3], g_wideprof[4], g_wideprof[5]);
  if (tid == 0) printf("[batch profile] wide by entries (<=512, <=1408, <=8192, more): n %llu %llu %llu %llu  cycles %llu %llu %llu %llu  entries %llu %llu %llu %llu  new edges %llu %llu %llu %llu\n",
                       bw_n[0], bw_n[1], bw_n[2], bw_n[3], bw_cyc[0], bw_cyc[1], bw_cyc[2], bw_cyc[3], bw_ent[0], bw_ent[1], bw_ent[2], bw_ent[3], bw_new[0], bw_new[1], bw_new[2], bw_new[3]);
#endif
  if (pend_e != kNone) st.er[pend_e].next = pend_old;
  // leave through the global lists: the next launch (or the tree kernel) starts from them
  full_barrier();
  win_flush(st, w, tid);
  if (tid == 0) {
    st.ctrl[0] = k; st.ctrl[1] = ne; st.ctrl[2] = pool_used; st.ctrl[3] = w.err ? (unsigned long long)ST_INTERNAL : status; st.ctrl[9] = w.err; st.ctrl[10] = w.n;
    st.ctrl[5] = (unsigned long long)(long long)w.cthr; st.ctrl[6] = (unsigned long long)__double_as_longlong(w.tsal); st.ctrl[7] = w.tseq; st.ctrl[8] = w.iptr;
  }
}

__global__ void win_range_kernel(const double* sal, uint32_t E0, unsigned long long* mm) {
  unsigned long long lo = ~0ull, hi = 0ull;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < E0; i += gridDim.x * blockDim.x) {
    const unsigned long long b = f64_ord(sal[i]);
    lo = b < lo ? b : lo; hi = b > hi ? b : hi;
  }
  atomicMin(&mm[0], lo); atomicMax(&mm[1], hi);
}
__global__ void win_params_kernel(const unsigned long long* mm, uint32_t B, double* range) {
  auto back = [](unsigned long long o) { o ^= (o >> 63) ? 0x8000000000000000ull : ~0ull; return __longlong_as_double((long long)o); };
  const double smin = back(mm[0]), smax = back(mm[1]);
  range[0] = smin;
  range[1] = smax > smin ? (double)B / (smax - smin) : 0.0;
}
// ---- baseline: every live queue item, sorted by descending (saliency, seq) (whole-GPU kernels between launches) ----
__global__ void win_collect_kernel(const EdgeRec* er, uint32_t n_edges, const uint8_t* rdead, unsigned long long* kseq, uint32_t* vals, uint32_t* counter) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  const unsigned long long q = er[e].seq;
  if (q == 0) return;
  const uint2 uv = *reinterpret_cast<const uint2*>(&er[e].u);
  if (rdead[uv.x] | rdead[uv.y]) return;         // died below the horizon: its record was not touched (WinState::rdead)
  const uint32_t i = atomicAdd(counter, 1u);
  kseq[i] = q; vals[i] = e;
}
__global__ void win_salkey_kernel(const EdgeRec* er, const uint32_t* vals, uint32_t n, unsigned long long* ksal) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ksal[i] = f64_ord(er[vals[i]].sal);
}
__global__ void win_baseline_fill_kernel(WinState st, uint32_t n, uint32_t* isort_w, unsigned long long* isort_seq) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const EdgeRec r = st.er[isort_w[i]];
  isort_seq[i] = r.seq;
  atomicAdd(&st.wcnt[win_cell(r.sal, st.wrange[0], st.wrange[1], st.wB)], 1u);
}
// ige[c] = number of baseline items whose cell is >= c (c = 0..B): cell c's segment of the sorted array is [ige[c+1], ige[c])
__global__ void win_segments_kernel(WinState st, const uint32_t* isort, uint32_t n, uint32_t* ige) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > st.wB) return;
  uint32_t lo = 0, hi = n;                    // first index whose cell is < c
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (win_cell(st.er[isort[mid]].sal, st.wrange[0], st.wrange[1], st.wB) >= c) lo = mid + 1; else hi = mid;
  }
  ige[c] = c == 0 ? n : lo;
}
__global__ void adj_fill_fat(GreedyState g, WinState st, uint32_t E0, uint32_t* cursor) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E0) return;
  const uint32_t u = g.e_u[e], v = g.e_v[e];
  const uint32_t pu = atomicAdd(&cursor[u], 1u), pv = atomicAdd(&cursor[v], 1u);
  const uint32_t ou = g.adj_off[u], ov = g.adj_off[v], lu = g.adj_len[u], lv = g.adj_len[v];
  FatEntry a; a.eid = e; a.rs = v; a.n = (uint32_t)g.e_n[e]; a.pos = pv; a.off = ov; a.len = lv; a.mean = g.e_mean[e];
  FatEntry b = a; b.rs = u; b.pos = pu; b.off = ou; b.len = lu;
  st.fpool[ou + pu] = a; g.e_posu[e] = pu;
  st.fpool[ov + pv] = b; g.e_posv[e] = pv;
  EdgeRec r; r.u = u; r.v = v; r.posu = pu; r.posv = pv; r.mean = g.e_mean[e]; r.n = g.e_n[e]; r.next = kNone;
  r.hu = make_uint2(ou, lu); r.hv = make_uint2(ov, lv); r.sal = g.pq.leaf_sal[e]; r.seq = g.pq.leaf_seq[e];
  st.er[e] = r;
}
// ST_NEED_TREE: the tree kernel continues on its own arrays -- thin list entries and one array per edge field
__global__ void fat_to_thin(const FatEntry* f, uint2* out, unsigned long long n) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const FatEntry fe = f[i]; out[i] = make_uint2(fe.eid, fe.eid == kNone ? 0u : fe.rs); }
}
__global__ void edge_unpack(GreedyState g, const EdgeRec* er, const uint8_t* rdead, uint32_t n) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const EdgeRec r = er[e];
  g.e_u[e] = r.u; g.e_v[e] = r.v; g.e_posu[e] = r.posu; g.e_posv[e] = r.posv; g.e_mean[e] = r.mean; g.e_n[e] = r.n;
  g.pq.leaf_sal[e] = r.sal; g.pq.leaf_seq[e] = (rdead[r.u] | rdead[r.v]) ? 0ull : r.seq;
}

// ---- edge table construction --------------------------------------------------------------------------
__global__ void edge_flags(const uint32_t* pa, const uint32_t* pb, long long P, uint32_t* flag, long long* partner) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  uint32_t a = pa[i], b = pb[i];
  long long j = (a < b) ? find_pair(pa, pb, P, b, a) : -1;   // "boundaries have to be mutual" (boundary_table.hxx:99-102)
  flag[i] = j >= 0 ? 1u : 0u;
  partner[i] = j;
}

__global__ void edge_fill(const uint32_t* pa, const uint32_t* pb, const uint32_t* prec, long long P,
                          const uint32_t* flag, const uint32_t* eidx, const long long* partner,
                          const uint32_t* rlabel, uint32_t R, GreedyState st, uint32_t* deg) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P || !flag[i]) return;
  const uint32_t e = eidx[i];
  const long long j = partner[i];
  const uint32_t u = find_label(rlabel, R, pa[i]), v = find_label(rlabel, R, pb[i]);
  const uint32_t* wi = &prec[(size_t)i * kPairWords];
  const uint32_t* wj = &prec[(size_t)j * kPairWords];
  double si, sj;
  memcpy(&si, &wi[P_SUM], 8);
  memcpy(&sj, &wj[P_SUM], 8);
  // util/struct_merge.hxx:45-56: sum of pb over both directed boundaries / their voxel count
  const int n = (int)(wi[P_CNT] + wj[P_CNT]);
  const double mean = sdivide(si + sj, (double)n, 0.0);
  st.e_u[e] = u; st.e_v[e] = v; st.e_mean[e] = mean; st.e_n[e] = n;
  st.pq.leaf_sal[e] = -mean; st.pq.leaf_seq[e] = (unsigned long long)e + 1ull;
  atomicAdd(&deg[u], 1u);
  atomicAdd(&deg[v], 1u);
}

__global__ void adj_fill(GreedyState st, uint32_t E0, uint32_t* cursor) {
  uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E0) return;
  const uint32_t u = st.e_u[e], v = st.e_v[e];
  const uint32_t pu = atomicAdd(&cursor[u], 1u), pv = atomicAdd(&cursor[v], 1u);
  st.pool[st.adj_off[u] + pu] = make_uint2(e, v); st.e_posu[e] = pu;
  st.pool[st.adj_off[v] + pv] = make_uint2(e, u); st.e_posv[e] = pv;
}

__global__ void region_sizes(const uint32_t* rrec, uint32_t R, unsigned long long* rsz, double* rsum) {
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const uint32_t* w = &rrec[(size_t)r * kRegionWords];
  rsz[r] = w[R_CNT];
  double d; memcpy(&d, &w[R_SUM], 8); rsum[r] = d;
}

// median linkage, initFb (util/struct_merge.hxx:97-103): the pb value of every voxel on either directed boundary of a
// table edge.  The voxel's pair is re-derived with the neighbour rule of getContourTraits (type/neighbor.hxx:109-126).
__global__ void median_collect(VolumeRef vol, const uint32_t* pa, const uint32_t* pb, long long P, const uint32_t* flag,
                               const uint32_t* eidx, const unsigned long long* e_off, uint32_t* cursor, float* out) {
  const long long N = vol.nx * vol.ny * vol.nz;
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  const long long x = p % vol.nx, y = (p / vol.nx) % vol.ny, z = p / (vol.nx * vol.ny);
  const uint32_t t = vol.lab[p];
  if (t == kMaskedLabel) return;                          // masked-out centre (point-map mode)
  uint32_t nb = t;
  const long long sy = vol.nx, sz = vol.nx * vol.ny;
  const uint32_t* L = vol.lab_nb;                         // masked-out neighbours hold kMaskedLabel: invalid
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
  const long long i = find_pair(pa, pb, P, t < nb ? t : nb, t < nb ? nb : t);
  if (i < 0 || !flag[i]) return;                         // not a mutual boundary: no table edge
  const uint32_t e = eidx[i];
  out[e_off[e] + atomicAdd(&cursor[e], 1u)] = vol.pb[p];
}

// the same from a map that carries its boundary values per directed pair (slab route, RagArrays::d_pv): the run of a table edge
// = the runs of its two directions, one thread per value
__global__ void median_from_runs(const unsigned long long* pv_off, const float* pv, unsigned long long nV, const uint32_t* pa, const uint32_t* pb, long long P,
                                 const uint32_t* flag, const uint32_t* eidx, const unsigned long long* e_off, float* out) {
  const unsigned long long v = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nV) return;
  long long lo = 0, hi = P;                     // the pair whose run holds value v
  while (lo < hi) { const long long mid = (lo + hi) >> 1; if (pv_off[mid + 1] <= v) lo = mid + 1; else hi = mid; }
  const long long i = lo;
  const uint32_t a = pa[i], b = pb[i];
  // the (a < b) direction owns the edge slot when the boundary is mutual (edge_flags); its values come first
  const long long owner = a < b ? i : find_pair(pa, pb, P, b, a);
  if (owner < 0 || !flag[owner]) return;       // not a mutual boundary: no table edge
  const unsigned long long first = owner == i ? 0ull : pv_off[owner + 1] - pv_off[owner];
  out[e_off[eidx[owner]] + first + (v - pv_off[i])] = pv[v];
}

__global__ void median_init(GreedyState st, uint32_t E0) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E0) return;
  const uint32_t n = (uint32_t)st.e_n[e];
  const double med = (double)st.vals[st.e_off[e] + n / 2u];      // amedian, util/stats.hxx:83-91
  st.e_mean[e] = med;
  st.pq.leaf_sal[e] = st.size_weight ? -med * (double)min(st.rsz[st.e_u[e]], st.rsz[st.e_v[e]]) : -med;
  atomicAdd(&st.rbv[st.e_u[e]], (unsigned long long)n);
  atomicAdd(&st.rbv[st.e_v[e]], (unsigned long long)n);
}

__global__ void fill_leaves_dead(PqTree t, uint32_t from) {
  uint32_t i = from + blockIdx.x * blockDim.x + threadIdx.x;
  if (i < t.nleaves) { t.leaf_seq[i] = 0; t.leaf_sal[i] = -__builtin_inf(); }
}

}  // namespace

// Runs the pb-mean greedy merge on a compact RAG.  h_order receives dense ids (leaf i = i-th label ascending,
// merged region R+k); the caller maps them to keys.
static int greedy_mean_once(const RagArrays& rag, hipStream_t stream, uint32_t* h_order, double* h_sal, int64_t capacity,
                            int64_t* n_merges, double* ms_table, double* ms_loop, int64_t* n_scored, int cond_n,
                            const long long* cond_sizes, double cond_rpb, const VolumeRef* median_of, bool size_weight) {
  const long long P = rag.P;
  const uint32_t R = (uint32_t)rag.R;
  *n_merges = 0;
  if (R == 0 || P == 0) return GLIA_HMT_OK;
  hipEvent_t ev[3];
  for (auto& e : ev) GLIA_HIP_TRY(hipEventCreate(&e));
  GLIA_HIP_TRY(hipEventRecord(ev[0], stream));
  DeviceBuffers buf;
  int rc;
  uint32_t* flag; uint32_t* eidx; long long* partner;
  if ((rc = buf.get(&flag, P + 1, true, stream))) return rc;
  if ((rc = buf.get(&eidx, P + 1, false, stream))) return rc;
  if ((rc = buf.get(&partner, P, false, stream))) return rc;
  hipLaunchKernelGGL(edge_flags, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, rag.d_pa, rag.d_pb, P, flag, partner);
  {
    size_t tmp = 0;
    GLIA_HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, flag, eidx, 0u, (size_t)(P + 1), rocprim::plus<uint32_t>(), stream));
    void* d_tmp;
    if ((rc = buf.get((char**)&d_tmp, tmp ? tmp : 16, false, stream))) return rc;
    GLIA_HIP_TRY(rocprim::exclusive_scan(d_tmp, tmp, flag, eidx, 0u, (size_t)(P + 1), rocprim::plus<uint32_t>(), stream));
  }
  uint32_t E0 = 0;
  GLIA_HIP_TRY(hipMemcpyAsync(&E0, eidx + P, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  GLIA_HIP_TRY(hipStreamSynchronize(stream));
  if (E0 == 0) return GLIA_HMT_OK;

  GreedyState st;
  memset(&st, 0, sizeof(st));
  st.R0 = R;
  // created edges are never reused: a 1024^3 run ends at ~16.4 x E0 edge slots and ~33 x E0 list entries (measured) --
  // sized so that the usual run never stops to grow (growth = a relaunch plus a copy of gigabytes)
  st.Ecap = (uint32_t)std::min<unsigned long long>(0xFFFFFF00ull, (unsigned long long)E0 * 18ull + (1u << 16));
  st.pool_cap = (unsigned long long)E0 * 36ull + (1u << 16);
  if ((rc = buf.get(&st.adj_off, 2 * (size_t)R, true, stream))) return rc;
  if ((rc = buf.get(&st.adj_len, 2 * (size_t)R + 1, true, stream))) return rc;
  // pb-mean linkage (with or without the pre_merge condition) runs on the window queue; GLIA_HMT_PB_WINDOW=0 keeps the
  // tournament tree (kernel experiments, parity gate: both must give byte-identical results)
  std::string o_window, o_batch, o_txt;
  const bool batch_off = option("GLIA_HMT_PB_BATCH", &o_batch) && o_batch[0] == '0';      // one contraction at a time on the window queue (same result)
  bool window = !median_of && !size_weight && !(option("GLIA_HMT_PB_WINDOW", &o_window) && o_window[0] == '0');
  WinState ws;
  memset(&ws, 0, sizeof(ws));
  if (window) {
    if ((rc = buf.get(&ws.fpool, st.pool_cap, false, stream))) return rc;
    if ((rc = buf.get(&ws.er, st.Ecap, false, stream))) return rc;
  }
  else if ((rc = buf.get(&st.pool, st.pool_cap, false, stream))) return rc;
  if ((rc = buf.get(&st.e_u, st.Ecap, false, stream))) return rc;
  if ((rc = buf.get(&st.e_v, st.Ecap, false, stream))) return rc;
  if ((rc = buf.get(&st.e_posu, st.Ecap, false, stream))) return rc;
  if ((rc = buf.get(&st.e_posv, st.Ecap, false, stream))) return rc;
  if ((rc = buf.get(&st.e_mean, st.Ecap, false, stream))) return rc;
  if ((rc = buf.get(&st.e_n, st.Ecap, true, stream))) return rc;
  st.pq.nleaves = st.Ecap;
  if ((rc = buf.get(&st.pq.leaf_sal, st.Ecap, false, stream))) return rc;
  if ((rc = buf.get(&st.pq.leaf_seq, st.Ecap, false, stream))) return rc;
  if ((rc = buf.get(&st.rsz, 2 * (size_t)R, true, stream))) return rc;
  if ((rc = buf.get(&st.rsum, 2 * (size_t)R, true, stream))) return rc;
  st.cond_n = cond_n; st.cond_rpb = cond_rpb; st.size_weight = size_weight ? 1 : 0;
  st.cond_t0 = cond_n > 0 ? (unsigned long long)cond_sizes[0] : 0; st.cond_t1 = cond_n > 1 ? (unsigned long long)cond_sizes[1] : 0;
  hipLaunchKernelGGL(region_sizes, dim3((R + 255) / 256), dim3(256), 0, stream, rag.d_rrec, R, st.rsz, st.rsum);
  if ((rc = buf.get(&st.mark0, 2 * (size_t)R, true, stream))) return rc;
  if ((rc = buf.get(&st.mark1, 2 * (size_t)R, true, stream))) return rc;
  // (+ kNW entries: a loop whose state is corrupt is stopped when k reaches R, at most one round of the batch kernel later)
  if ((rc = buf.get(&st.order, 3 * ((size_t)R + 16), false, stream))) return rc;
  if ((rc = buf.get(&st.sal_out, (size_t)R + 16, false, stream))) return rc;
  if ((rc = buf.get(&st.ctrl, 16, true, stream))) return rc;
  uint32_t* cursor;
  if ((rc = buf.get(&cursor, 2 * (size_t)R, true, stream))) return rc;

  hipLaunchKernelGGL(fill_leaves_dead, dim3((st.Ecap - E0 + 255) / 256), dim3(256), 0, stream, st.pq, E0);
  hipLaunchKernelGGL(edge_fill, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, rag.d_pa, rag.d_pb, rag.d_prec, P,
                     flag, eidx, partner, rag.d_rlabel, R, st, st.adj_len);
  {
    // adjacency offsets = exclusive scan of the degrees (adj_len[0..R) holds them, the rest is zero)
    size_t tmp = 0;
    GLIA_HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, st.adj_len, st.adj_off, 0u, (size_t)R, rocprim::plus<uint32_t>(), stream));
    void* d_tmp;
    if ((rc = buf.get((char**)&d_tmp, tmp ? tmp : 16, false, stream))) return rc;
    GLIA_HIP_TRY(rocprim::exclusive_scan(d_tmp, tmp, st.adj_len, st.adj_off, 0u, (size_t)R, rocprim::plus<uint32_t>(), stream));
  }
  if (window) hipLaunchKernelGGL(adj_fill_fat, dim3((E0 + 255) / 256), dim3(256), 0, stream, st, ws, E0, cursor);
  else hipLaunchKernelGGL(adj_fill, dim3((E0 + 255) / 256), dim3(256), 0, stream, st, E0, cursor);
  GLIA_HIP_TRY(hipGetLastError());
  unsigned long long n_values = 0;
  if (median_of) {
    // sorted value runs: offsets = scan of the edges' voxel counts, one scatter pass over the volume, segmented sort
    if ((rc = buf.get(&st.e_off, st.Ecap, false, stream))) return rc;
    if ((rc = buf.get(&st.rbv, 2 * (size_t)R, true, stream))) return rc;
    {
      size_t tmp = 0;
      GLIA_HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, st.e_n, st.e_off, 0ull, (size_t)E0 + 1, rocprim::plus<unsigned long long>(), stream));
      void* d_tmp;
      if ((rc = buf.get((char**)&d_tmp, tmp ? tmp : 16, false, stream))) return rc;
      GLIA_HIP_TRY(rocprim::exclusive_scan(d_tmp, tmp, st.e_n, st.e_off, 0ull, (size_t)E0 + 1, rocprim::plus<unsigned long long>(), stream));
    }
    GLIA_HIP_TRY(hipMemcpyAsync(&n_values, st.e_off + E0, sizeof(n_values), hipMemcpyDeviceToHost, stream));
    GLIA_HIP_TRY(hipStreamSynchronize(stream));
    if (n_values >= 0xFFFFFFFFull) { set_error("merge_order_pb: more than 2^32 boundary voxels (median linkage)"); return GLIA_HMT_ERR_ARG; }
    st.vals_cap = n_values * 4ull + (1ull << 20);
    float* unsorted;
    if ((rc = buf.get(&st.vals, (size_t)st.vals_cap, false, stream))) return rc;
    if ((rc = buf.get(&unsorted, (size_t)n_values, false, stream))) return rc;
    uint32_t* vcursor;
    if ((rc = buf.get(&vcursor, (size_t)E0, true, stream))) return rc;
    if (median_of->lab) {
      const long long N = median_of->nx * median_of->ny * median_of->nz;
      hipLaunchKernelGGL(median_collect, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, *median_of, rag.d_pa, rag.d_pb, P,
                         flag, eidx, st.e_off, vcursor, unsorted);
    } else {
      if (!rag.d_pv_off) { set_error("merge_order_pb: median linkage needs the volume or the map's boundary values"); return GLIA_HMT_ERR_UNSUPPORTED; }
      if (rag.nV) hipLaunchKernelGGL(median_from_runs, dim3((unsigned)((rag.nV + 255) / 256)), dim3(256), 0, stream, rag.d_pv_off, rag.d_pv, rag.nV, rag.d_pa, rag.d_pb,
                                     (long long)P, flag, eidx, st.e_off, unsorted);
    }
    GLIA_HIP_TRY(hipGetLastError());
    {
      size_t tmp = 0;
      GLIA_HIP_TRY(rocprim::segmented_radix_sort_keys(nullptr, tmp, unsorted, st.vals, (unsigned)n_values, (unsigned)E0, st.e_off,
                                                      st.e_off + 1, 0, 32, stream));
      void* d_tmp;
      if ((rc = buf.get((char**)&d_tmp, tmp ? tmp : 16, false, stream))) return rc;
      GLIA_HIP_TRY(rocprim::segmented_radix_sort_keys(d_tmp, tmp, unsorted, st.vals, (unsigned)n_values, (unsigned)E0, st.e_off,
                                                      st.e_off + 1, 0, 32, stream));
    }
    hipLaunchKernelGGL(median_init, dim3((E0 + 255) / 256), dim3(256), 0, stream, st, E0);
    GLIA_HIP_TRY(hipGetLastError());
  }
  unsigned long long ctrl[11] = {0, E0, 2ull * E0, ST_RUN, n_values, 0, 0, 0, 0, 0, 0};      // [9], [10]: diagnostics of ST_INTERNAL
  // buffers of the window queue's baseline (see win_rebaseline below)
  unsigned long long *rb_kseq = nullptr, *rb_kseq2 = nullptr, *rb_ksal = nullptr, *rb_ksal2 = nullptr, *rb_iseq = nullptr;
  uint32_t *rb_vals = nullptr, *rb_vals2 = nullptr, *rb_isort = nullptr, *rb_ige = nullptr, *rb_counter = nullptr;
  void* rb_tmp = nullptr; size_t rb_tmp_bytes = 0;
  // A baseline = all live queue items sorted by descending (saliency, seq): two stable radix sorts (by seq, then by the
  // saliency's order-preserving image), per-cell segments and live counters; the cell lists start empty.  Taken at the start
  // (the initial edges) and whenever a launch ends before the queue is empty: the lists only shed their dead nodes when
  // their cell is loaded, so after a few million created edges walking them dominates; a re-sort is ~1 ms of whole-GPU work.
  double h_range[2] = {0.0, 0.0};
  long long prev_top_cell = -1;
  unsigned long long prev_ne = 0;
  double horizon_factor = 0.0;                                           // 0 = no horizon (set below for the batch kernel)
  auto win_rebaseline = [&](uint32_t n_edges) -> int {
    GLIA_HIP_TRY(hipMemsetAsync(rb_counter, 0, sizeof(uint32_t), stream));
    hipLaunchKernelGGL(win_collect_kernel, dim3((n_edges + 255) / 256), dim3(256), 0, stream, ws.er, n_edges, ws.rdead, rb_kseq, rb_vals, rb_counter);
    uint32_t n = 0;
    GLIA_HIP_TRY(hipMemcpyAsync(&n, rb_counter, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    GLIA_HIP_TRY(hipStreamSynchronize(stream));
    if (n > E0) { set_error("greedy: more live edges than initial edges (internal error)"); return GLIA_HMT_ERR_INTERNAL; }
    if (n) {
      size_t tmp = rb_tmp_bytes;
      GLIA_HIP_TRY(rocprim::radix_sort_pairs_desc(rb_tmp, tmp, rb_kseq, rb_kseq2, rb_vals, rb_vals2, (size_t)n, 0, 64, stream));
      hipLaunchKernelGGL(win_salkey_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, ws.er, rb_vals2, n, rb_ksal);
      tmp = rb_tmp_bytes;
      GLIA_HIP_TRY(rocprim::radix_sort_pairs_desc(rb_tmp, tmp, rb_ksal, rb_ksal2, rb_vals2, rb_isort, (size_t)n, 0, 64, stream));
    }
    GLIA_HIP_TRY(hipMemsetAsync(ws.whead, 0xFF, sizeof(uint32_t) * ws.wB, stream));
    GLIA_HIP_TRY(hipMemsetAsync(ws.wcnt, 0, sizeof(uint32_t) * ws.wB, stream));
    if (n) hipLaunchKernelGGL(win_baseline_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, ws, n, rb_isort, rb_iseq);
    hipLaunchKernelGGL(win_segments_kernel, dim3((ws.wB + 1 + 255) / 256), dim3(256), 0, stream, ws, rb_isort, n, rb_ige);
    GLIA_HIP_TRY(hipGetLastError());
    ws.nsort = n;
    std::string renv;                                                      // created edges between baselines (tuning)
    ws.rebase_after = option("GLIA_HMT_REBASE", &renv) ? strtoull(renv.c_str(), nullptr, 10) : std::max<unsigned long long>(800000ull, (unsigned long long)n / 2ull);
    // the horizon (WinState::wch): the top of the queue sank by d cells while the last interval's edges were created; the next
    // interval is given horizon_factor times that (scaled to its planned length; at least 1/512 of the cells) before a reload
    // would run into the horizon
    ws.wch = 0;
    if (horizon_factor > 0.0 && n > 4096) {
      unsigned long long topkey = 0;
      GLIA_HIP_TRY(hipMemcpyAsync(&topkey, rb_ksal2, sizeof(topkey), hipMemcpyDeviceToHost, stream));
      GLIA_HIP_TRY(hipStreamSynchronize(stream));
      const long long top_cell = (long long)win_cell(f64_unord(topkey), h_range[0], h_range[1], ws.wB);
      if (prev_top_cell >= 0 && n_edges > prev_ne) {
        const double d = (double)std::max<long long>(0, prev_top_cell - top_cell);
        const double delta = std::max(horizon_factor * d * (double)ws.rebase_after / (double)(n_edges - prev_ne), (double)ws.wB / 512.0);
        ws.wch = (double)top_cell > delta ? (uint32_t)((double)top_cell - delta) : 0u;
      }
      prev_top_cell = top_cell; prev_ne = n_edges;
    }
    ws.ne_base = n_edges;
    const double inf = std::numeric_limits<double>::infinity();
    ctrl[5] = (unsigned long long)(long long)(ws.wB - 1);                  // threshold: everything is below it
    memcpy(&ctrl[6], &inf, 8); ctrl[7] = ~0ull; ctrl[8] = 0;
    return GLIA_HMT_OK;
  };
  if (window) {
    // saliency cells: ~4 initial edges per cell on average
    uint32_t B = 256;
    while (B < E0 / 4 && B < (1u << 22)) B <<= 1;
    ws.wB = B; ws.R0 = R;
    ws.wcap = kWinCap; ws.wbudget = kWinBudget;
    ws.force_tree = 0;
    if (option("GLIA_HMT_FORCE_TREE", &o_txt)) ws.force_tree = strtoull(o_txt.c_str(), nullptr, 10);
    if (option("GLIA_HMT_WINCAP", &o_txt)) {                             // tests: a tiny window makes spills, evictions and cell splits routine
      const uint32_t c = (uint32_t)strtoul(o_txt.c_str(), nullptr, 10);
      if (c >= 16 && c <= kWinCap) { ws.wcap = c; ws.wbudget = c / 2; }
    }
    unsigned long long* mm; double* range;
    if ((rc = buf.get(&ws.rdead, 2 * (size_t)R, true, stream))) return rc;
    if ((rc = buf.get(&ws.whead, B, false, stream))) return rc;
    if ((rc = buf.get(&ws.wcnt, B, true, stream))) return rc;
    if ((rc = buf.get(&rb_isort, E0, false, stream))) return rc;
    if ((rc = buf.get(&rb_iseq, E0, false, stream))) return rc;
    if ((rc = buf.get(&rb_ige, (size_t)B + 1, false, stream))) return rc;
    if ((rc = buf.get(&mm, 2, false, stream))) return rc;
    if ((rc = buf.get(&range, 2, false, stream))) return rc;
    if ((rc = buf.get(&rb_kseq, E0, false, stream))) return rc;
    if ((rc = buf.get(&rb_kseq2, E0, false, stream))) return rc;
    if ((rc = buf.get(&rb_ksal, E0, false, stream))) return rc;
    if ((rc = buf.get(&rb_ksal2, E0, false, stream))) return rc;
    if ((rc = buf.get(&rb_vals, E0, false, stream))) return rc;
    if ((rc = buf.get(&rb_vals2, E0, false, stream))) return rc;
    if ((rc = buf.get(&rb_counter, 1, true, stream))) return rc;
    GLIA_HIP_TRY(rocprim::radix_sort_pairs_desc(nullptr, rb_tmp_bytes, rb_kseq, rb_kseq2, rb_vals, rb_vals2, (size_t)E0, 0, 64, stream));
    if ((rc = buf.get((char**)&rb_tmp, rb_tmp_bytes ? rb_tmp_bytes : 16, false, stream))) return rc;
    ws.wrange = range; ws.isort = rb_isort; ws.isort_seq = rb_iseq; ws.ige = rb_ige;
    ws.order = st.order; ws.sal_out = st.sal_out; ws.ctrl = st.ctrl; ws.rsz = st.rsz; ws.rsum = st.rsum;
    ws.mark0 = st.mark0; ws.mark1 = st.mark1; ws.adj_off = st.adj_off; ws.adj_len = st.adj_len;
    ws.pool_cap = st.pool_cap; ws.Ecap = st.Ecap;
    ws.cond_n = st.cond_n; ws.cond_t0 = st.cond_t0; ws.cond_t1 = st.cond_t1; ws.cond_rpb = st.cond_rpb;
    const unsigned long long mm0[2] = {~0ull, 0ull};
    GLIA_HIP_TRY(hipMemcpyAsync(mm, mm0, sizeof(mm0), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(win_range_kernel, dim3(256), dim3(256), 0, stream, st.pq.leaf_sal, E0, mm);
    hipLaunchKernelGGL(win_params_kernel, dim3(1), dim3(1), 0, stream, mm, B, range);
    GLIA_HIP_TRY(hipGetLastError());
    GLIA_HIP_TRY(hipMemcpyAsync(h_range, range, sizeof(h_range), hipMemcpyDeviceToHost, stream));
    GLIA_HIP_TRY(hipStreamSynchronize(stream));
    // the horizon: 0 = off; else the factor on the measured descent (swept 0.05 .. 8 at 1024^3: flat from 0.1 to 0.5, +1 % at 2, +2 % at 4)
    if (cond_n <= 0 && !batch_off) horizon_factor = option("GLIA_HMT_HORIZON", &o_txt) ? atof(o_txt.c_str()) : 0.5;
    if ((rc = win_rebaseline(E0))) return rc;
  } else if ((rc = pq_setup(buf, st.pq, stream))) return rc;
  GLIA_HIP_TRY(hipMemcpyAsync(st.ctrl, ctrl, sizeof(ctrl), hipMemcpyHostToDevice, stream));
  GLIA_HIP_TRY(hipEventRecord(ev[1], stream));

  // ---- the loop, in bounded launches so a contraction budget can be re-negotiated between them ----
  st.max_iters = window ? 1ull << 22 : 1ull << 16;
  const bool trace = option("GLIA_HMT_TRACE");
  if (option("GLIA_HMT_MAXITERS", &o_txt)) st.max_iters = strtoull(o_txt.c_str(), nullptr, 10);      // tests: launches that end early
  while (true) {
    if (window) {
      ws.max_iters = st.max_iters;
      if (cond_n > 0) hipLaunchKernelGGL(greedy_window_kernel<true>, dim3(1), dim3(kGreedyThreads), 0, stream, ws);
      else if (batch_off) hipLaunchKernelGGL(greedy_window_kernel<false>, dim3(1), dim3(kGreedyThreads), 0, stream, ws);
      else hipLaunchKernelGGL(greedy_batch_kernel, dim3(1), dim3(kGreedyThreads), 0, stream, ws);
    } else if (median_of) hipLaunchKernelGGL(greedy_pb_kernel<true>, dim3(1), dim3(kGreedyThreads), 0, stream, st);
    else hipLaunchKernelGGL(greedy_pb_kernel<false>, dim3(1), dim3(kGreedyThreads), 0, stream, st);
    GLIA_HIP_TRY(hipGetLastError());
    GLIA_HIP_TRY(hipMemcpyAsync(ctrl, st.ctrl, sizeof(ctrl), hipMemcpyDeviceToHost, stream));
    GLIA_HIP_TRY(hipStreamSynchronize(stream));
    if (trace) fprintf(stderr, "[trace] merge loop launch ended: status %llu, merges %llu of %u regions, edges %llu, list entries %llu\n", ctrl[3], ctrl[0], R, ctrl[1], ctrl[2]);
    if (ctrl[3] == ST_RUN && !window) continue;
    if (ctrl[3] == ST_DONE) break;
    if (ctrl[3] == ST_BAD_SALIENCY) { set_error("Error: invalid boundary saliency..."); return GLIA_HMT_ERR_SALIENCY; }
    if (ctrl[3] == ST_INTERNAL) {
      char msg[256];
      snprintf(msg, sizeof(msg), "greedy: window queue overflow or more merges than regions (internal error: merges %llu of %u regions, %llu edges of %u initial, "
               "window %llu%s, %s kernel)", ctrl[0], R, ctrl[1], E0, ctrl[10], ctrl[9] ? " overflowed" : "", !window ? "tree" : cond_n > 0 ? "window" : "batch");
      set_error(msg);
      return GLIA_HMT_ERR_INTERNAL;
    }
    if (ctrl[3] == ST_NEED_TREE) {
      // a saliency cell with more live items than the window holds (massive exact ties): the tournament tree takes over
      // from the same state -- leaf keys are the ground truth of both queues, the lists get their thin entries
      if ((rc = buf.get(&st.pool, st.pool_cap, false, stream))) return rc;
      hipLaunchKernelGGL(fat_to_thin, dim3((unsigned)((ctrl[2] + 255) / 256)), dim3(256), 0, stream, ws.fpool, st.pool, ctrl[2]);
      hipLaunchKernelGGL(edge_unpack, dim3((unsigned)((ctrl[1] + 255) / 256)), dim3(256), 0, stream, st, ws.er, ws.rdead, (uint32_t)ctrl[1]);
      GLIA_HIP_TRY(hipGetLastError());
      if ((rc = pq_setup(buf, st.pq, stream))) return rc;
      window = false;
      st.max_iters = 1ull << 16;
    } else if (ctrl[3] == ST_NEED_POOL) {
      unsigned long long ncap = st.pool_cap * 2;
      if (window) { if ((rc = buf.grow(&ws.fpool, (size_t)st.pool_cap, (size_t)ncap, stream))) return rc; }
      else if ((rc = buf.grow(&st.pool, (size_t)st.pool_cap, (size_t)ncap, stream))) return rc;
      st.pool_cap = ncap; ws.pool_cap = ncap;
    } else if (ctrl[3] == ST_NEED_VALUES) {
      unsigned long long ncap = st.vals_cap * 2;
      if ((rc = buf.grow(&st.vals, (size_t)ctrl[4], (size_t)ncap, stream))) return rc;
      st.vals_cap = ncap;
    } else if (ctrl[3] == ST_NEED_EDGES) {
      if (st.Ecap >= 0xFFFFFF00u) { set_error("greedy: more than 2^32 edge slots needed"); return GLIA_HMT_ERR_ARG; }
      uint32_t ocap = st.Ecap;
      uint32_t ncap = (uint32_t)std::min<unsigned long long>(0xFFFFFF00ull, (unsigned long long)ocap * 2ull);
      if ((rc = buf.grow(&st.e_u, ocap, ncap, stream))) return rc;
      if ((rc = buf.grow(&st.e_v, ocap, ncap, stream))) return rc;
      if ((rc = buf.grow(&st.e_posu, ocap, ncap, stream))) return rc;
      if ((rc = buf.grow(&st.e_posv, ocap, ncap, stream))) return rc;
      if ((rc = buf.grow(&st.e_mean, ocap, ncap, stream))) return rc;
      if ((rc = buf.grow(&st.e_n, ocap, ncap, stream))) return rc;
      if (median_of && (rc = buf.grow(&st.e_off, ocap, ncap, stream))) return rc;
      if ((rc = buf.grow(&st.pq.leaf_sal, ocap, ncap, stream))) return rc;
      if ((rc = buf.grow(&st.pq.leaf_seq, ocap, ncap, stream))) return rc;
      if (window && (rc = buf.grow(&ws.er, ocap, ncap, stream))) return rc;
      st.Ecap = ncap; st.pq.nleaves = ncap; ws.Ecap = ncap;
      hipLaunchKernelGGL(fill_leaves_dead, dim3((ncap - ocap + 255) / 256), dim3(256), 0, stream, st.pq, ocap);
      if (!window && (rc = pq_setup(buf, st.pq, stream))) return rc;
    }
    if (window) {
      // the launch left through the lists (everything alive sits there): a fresh baseline, an empty window
      if ((rc = win_rebaseline((uint32_t)ctrl[1]))) return rc;
      GLIA_HIP_TRY(hipMemcpyAsync(st.ctrl + 5, ctrl + 5, 4 * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
    }
    unsigned long long zero = ST_RUN;
    GLIA_HIP_TRY(hipMemcpyAsync(st.ctrl + 3, &zero, sizeof(zero), hipMemcpyHostToDevice, stream));
  }
  GLIA_HIP_TRY(hipEventRecord(ev[2], stream));
  GLIA_HIP_TRY(hipEventSynchronize(ev[2]));
  float t01 = 0, t12 = 0;
  (void)hipEventElapsedTime(&t01, ev[0], ev[1]);
  (void)hipEventElapsedTime(&t12, ev[1], ev[2]);
  for (auto& e : ev) (void)hipEventDestroy(e);
  *ms_table = t01; *ms_loop = t12;
  const int64_t n = (int64_t)ctrl[0];
  *n_scored = (int64_t)ctrl[1];
  if (n > capacity) { set_error("merge_order: output capacity too small"); return GLIA_HMT_ERR_CAPACITY; }
  if (n) {
    if ((rc = copy_to_host_staged(h_order, st.order, sizeof(uint32_t) * 3 * n))) return rc;
    if ((rc = copy_to_host_staged(h_sal, st.sal_out, sizeof(double) * n))) return rc;
  }
  *n_merges = n;
  return GLIA_HMT_OK;
}


// Every merge of a correct order joins two regions that still exist and creates region R + k (util/struct_merge.hxx:19-31: the loop
// appends (r0, r1, key++) and erases both regions' items).  The pb / pre_merge loops replay their order against this rule on the host
// before they return it -- an O(R) pass, ~1 ms at 262 144 regions -- and a violation is an ERROR (GLIA_HMT_ERR_INTERNAL), never a
// silent second run: round 3 re-ran such calls (a net under the window kernel's race on its edge counter), which let a
// kernel defect pass every test.  glia_hmt_internal_errors() counts the calls that ended this way.
static std::atomic<unsigned long long> g_internal_errors{0};
unsigned long long internal_errors() { return g_internal_errors.load(); }
void count_internal_error() { g_internal_errors.fetch_add(1); }
bool merge_order_is_consistent(const uint32_t* o, int64_t n, uint32_t R, int64_t* first_bad) {
  std::vector<uint8_t> gone(2 * (size_t)R + 1, 0);
  for (int64_t k = 0; k < n; ++k) {
    const uint32_t a = o[3 * k], b = o[3 * k + 1], c = o[3 * k + 2];
    if (k >= (int64_t)R || c != R + (uint32_t)k || a >= c || b >= c || a == b || gone[a] || gone[b]) { if (first_bad) *first_bad = k; return false; }
    gone[a] = gone[b] = 1;
  }
  return true;
}
int greedy_mean(const RagArrays& rag, hipStream_t stream, uint32_t* h_order, double* h_sal, int64_t capacity,
                int64_t* n_merges, double* ms_table, double* ms_loop, int64_t* n_scored, int cond_n,
                const long long* cond_sizes, double cond_rpb, const VolumeRef* median_of, bool size_weight) {
  int rc = greedy_mean_once(rag, stream, h_order, h_sal, capacity, n_merges, ms_table, ms_loop, n_scored, cond_n, cond_sizes, cond_rpb, median_of, size_weight);
  int64_t bad = -1;
  if (rc == GLIA_HMT_OK && !merge_order_is_consistent(h_order, *n_merges, (uint32_t)rag.R, &bad)) {
    set_error("greedy: merge " + std::to_string(bad) + " of " + std::to_string(*n_merges) + " joins a region that does not exist (any more) or creates the wrong one (internal error)");
    rc = GLIA_HMT_ERR_INTERNAL;
  }
  if (rc == GLIA_HMT_ERR_INTERNAL) { count_internal_error(); (void)hipStreamSynchronize(stream); }
  return rc;
}

}  // namespace glia
