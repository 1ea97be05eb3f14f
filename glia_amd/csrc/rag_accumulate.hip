// glia_amd/csrc/rag_accumulate.hip -- K1+K2+K3: one streaming pass over the label volume and one
// float image that produces the region adjacency structure AND every sufficient statistic of it.
//
// Reference semantics reproduced (all under /root/reference/code/):
//   * per voxel: getContourTraits (type/neighbor.hxx:109-126) -- neighbours visited -x,+x,-y,+y,-z,+z
//     (type/neighbor.hxx:78-88); the FIRST valid neighbour with a different label names the single directed
//     boundary (own -> nbr) this voxel belongs to (util/struct.hxx:100-106,137-142); a voxel with no
//     differing neighbour and fewer than 2*D valid neighbours is a border voxel.
//   * per label: genCountMap (util/struct.hxx:61-74), bounding box (alg/geometry.hxx:21-39),
//     ImageRealFeats sums (type/feat.hxx:724-736), histc (util/image_stats.hxx:12-37).
//   * per directed pair: the same moments/histogram over its boundary voxels plus the thresholded
//     counts of type/feat.hxx:493-501,574-588.
//
// MI355X mapping (round 3 rewrite).  HBM-bound by bytes -- 8 algorithmic bytes per voxel (4 label + 4 image) --, but
// instruction-issue bound in practice, so the design minimises wave instructions per voxel and keeps four waves per SIMD:
//   * Tile = 64(x) x 32(y) x 32(z) voxels per 1024-thread workgroup (one per CU, it owns the 160 KiB of LDS).  A wave is
//     one ROW of 64 voxels (lane = x, one 256-byte load per row), it marches the z COLUMN of its row through the tile's
//     planes, then the column of a second row.  z-1 / z / z+1 labels of a column roll through registers, y+-1 rows are
//     re-read through L1/L2, x+-1 come from the neighbouring lanes with one DPP move each (wave_shr / wave_shl, the halo
//     label of lanes 0 / 63 rides in the `old` operand).  Loads run four planes ahead of the arithmetic.
//   * Supervoxels are spatially coherent, so a lane sees long RUNS of one label / one directed pair down its column.  It
//     reduces a run in registers (f64 sums with the square folded into an FMA -- v*v is exact in f64 --, float min / max,
//     8-bit packed histogram counters in one 64-bit register, 8-bit packed threshold counters).  Everything positional
//     (bounding box, first voxel, voxel count) follows from the run's first and last plane and costs nothing per voxel.
//   * A finished run is appended to a per-wave LDS ring (three 16-byte stores, regions and pairs in one batch).  When the
//     ring cannot take the next batch the whole wave DRAINS it -- lane j owns entry j -- into the workgroup's LDS hash
//     tables (find-or-insert + LDS atomics; two 32-bit counters per 64-bit atomic, bounding boxes as OR-ed occupancy
//     masks).  After the march the workgroup folds its LDS tables into the global hash tables.
//   * All reductions are integer adds, ORs, unsigned max, or f64 adds: exact, hence order-independent and
//     bit-reproducible, whenever the image is a multiple of 2^-k (Q8 pb); otherwise within ~1e-15 relative.
#include <tuple>
#include <type_traits>
#include <utility>

#include "hmt_internal.hpp"
#include "skew.hpp"

// the pass's own cycle counters (ring waits of the marching waves, busy cycles of the drainers): profiling build, or make accvar
// ACCFLAGS=-DGLIA_ACC_STATS (printed by rag_build when GLIA_HMT_DEBUG has bit 32)
#if defined(GLIA_HMT_PROFILE) || defined(GLIA_ACC_STATS)
#define GLIA_ACC_COUNTERS 1
#endif

namespace glia {

namespace {

__device__ __forceinline__ uint32_t hash64(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
  return (uint32_t)k;
}
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

static_assert(kTileX == 64 && kTileY <= 32 && kTZ == 32 && kTileX * kTileY * kTZ < 65536, "64 x (<= 32) x 32 tiles: y occupancy in 32 bits, 16-bit counters");

// ---- geometry ----------------------------------------------------------------------------------------
// marching : draining waves.  Round 4 (prefetch repaired, combining drain): the marching waves alone need 4.6 ms at 12 waves, the
// four drainers could not keep up (8.7 ms); the split is a build parameter of the experiments (make accvar) -- see DESIGN 3.1.
#ifndef GLIA_ACC_MARCHERS
#define GLIA_ACC_MARCHERS 12
#endif
#ifndef GLIA_ACC_DRAINERS
#define GLIA_ACC_DRAINERS 4
#endif
template <int BINS> struct Geo {
  static constexpr int kMarchers = BINS <= 8 ? GLIA_ACC_MARCHERS : GLIA_ACC_MARCHERS / 2;     // waves that march columns (16-bin records need the LDS for tables: half the waves)
  static constexpr int kDrainers = BINS <= 8 ? GLIA_ACC_DRAINERS : (GLIA_ACC_DRAINERS + 1) / 2;   // waves that only empty the rings of the others
  static constexpr int kWaves = kMarchers + kDrainers;
  static constexpr int kThreadsT = kWaves * 64;
  static constexpr int kPasses = kTileY / kMarchers;        // rows a wave marches one after the other
  static constexpr int kRegSlots = 256;
  static constexpr int kPairSlots = 1024;
  static constexpr int kEntryWords = BINS <= 8 ? 12 : 16;   // ring entry
  static_assert(kPasses * kMarchers == kTileY, "rows of a tile = marching waves x passes");
  static_assert(kWaves <= 16 && kMarchers <= 3 * kDrainers, "at most 1024 threads; a drainer serves at most three rings");
};
#ifndef GLIA_ACC_DRAINAT
#define GLIA_ACC_DRAINAT 48          // (round 4, with the combining drain: 16 -> 9.02 ms, 32 -> 8.81, 48 -> 8.78)
#endif
constexpr int kDrainAt = GLIA_ACC_DRAINAT;            // a drainer gathers at least this many entries unless a wave of its waits or is done
constexpr int kLdsProbes = 32;
constexpr int kGlobalProbes = 512;
#ifndef GLIA_ACC_AHEAD
#define GLIA_ACC_AHEAD 4
#endif
constexpr int kAhead = GLIA_ACC_AHEAD;   // planes the loads run ahead of the arithmetic (= buffers per stream); a power of two

// ---- LDS record layouts (words); all-zero = empty.  Histogram / threshold counters are 16 bits wide, two per word: a tile
// has fewer than 65536 voxels, and a 64-bit LDS atomic then adds four of them at once.
// region: 0 cnt | 1 border | 2,3 x occupancy (u64, tile-relative) | 4 y occupancy | 5 z occupancy | 6 ~ord(min) | 7 ord(max) |
//         8 sum(f64) | 10 sq(f64) | 12 0xFFFFF - first(tile-rel z<<11|y<<6|x) | 14.. hist
// pair:   0 cnt | 1 ~ord(min) | 2,3 thr | 4 ord(max) | 6 sum | 8 sq | 10.. hist
constexpr int LR_CNT = 0, LR_BORDER = 1, LR_XMASK = 2, LR_YMASK = 4, LR_ZMASK = 5, LR_MIN = 6, LR_MAX = 7, LR_SUM = 8, LR_SQ = 10,
              LR_FIRST = 12, LR_HIST = 14;
constexpr int LP_CNT = 0, LP_MIN = 1, LP_THR = 2, LP_MAX = 4, LP_SUM = 6, LP_SQ = 8, LP_HIST = 10;
static_assert(kTileX * kTileY * kTZ < 65536, "16-bit counters in the LDS records");
constexpr int kRingR = 64, kRingP = 64;      // entries of a marching wave's region / pair ring (powers of two)

// what the drain and the fold need from the kernel arguments; copied to LDS once: read back from there they do not occupy
// scalar registers during the march
struct TableParams {
  uint32_t* rkeys; uint32_t* rrec; unsigned long long* pkeys; uint32_t* prec; uint32_t* flags;
  uint32_t rmask, pmask;
  int64_t nx, ny;
  int64_t x0, y0, z0;     // tile origin (global coordinates)
};

// control words of a marching wave's rings
constexpr int C_TAIL_R = 0, C_TAIL_P = 1, C_STATE = 2, C_HEAD_R = 4, C_HEAD_P = 5;     // 0..3: written by the marching wave (one 16-byte read for the drainer)
constexpr uint32_t kRingWaiting = 1u, kRingDone = 2u;

template <int BINS>
struct Lds {
  using G = Geo<BINS>;
  static constexpr int kRegWords = LR_HIST + BINS / 2;      // 18 / 22
  static constexpr int kPairWordsL = LP_HIST + BINS / 2;    // 14 / 18
  unsigned long long rkey[G::kRegSlots];
  unsigned long long pkey[G::kPairSlots];
  uint32_t rrec[G::kRegSlots * kRegWords];
  uint32_t prec[G::kPairSlots * kPairWordsL];
  uint32_t ringR[G::kMarchers][kRingR * G::kEntryWords];   // circular, per marching wave; reused as gslot[] by the final fold
  uint32_t ringP[G::kMarchers][kRingP * G::kEntryWords];
  alignas(16) uint32_t ctl[G::kMarchers][8];
  TableParams tp;
};
static_assert(sizeof(Lds<8>) <= 160 * 1024 && sizeof(Lds<16>) <= 160 * 1024, "LDS budget of one CU");

// find-or-insert in an LDS key table.  The hash costs two quarter-rate multiplies; the slot index comes from a 24-bit multiply.
__device__ __forceinline__ int lds_slot(unsigned long long* keys, int nslots, unsigned long long key) {
  uint32_t m = (uint32_t)key ^ ((uint32_t)(key >> 32) * 0x9E3779B1u);
  m ^= m >> 16; m *= 0x7feb352dU; m ^= m >> 15;
  uint32_t h = __umul24(m >> 16, (uint32_t)nslots) >> 16;      // 16 bits x 10 bits: no overflow
  for (int i = 0; i < kLdsProbes; ++i) {
    unsigned long long cur = keys[h];
    if (cur == key) return (int)h;
    if (cur == 0) {
      unsigned long long old = atomicCAS(&keys[h], 0ull, key);
      if (old == 0 || old == key) return (int)h;
    }
    h = (h + 1 == (uint32_t)nslots) ? 0 : h + 1;
  }
  return -1;
}
template <typename P>
__device__ __forceinline__ int global_region_slot(const P& p, uint32_t key) {
  uint32_t h = hash32(key) & p.rmask;
  for (int i = 0; i < kGlobalProbes; ++i) {
    uint32_t cur = __hip_atomic_load(&p.rkeys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == key) return (int)h;
    if (cur == 0) {
      uint32_t old = atomicCAS(&p.rkeys[h], 0u, key);
      if (old == 0 || old == key) return (int)h;
    }
    h = (h + 1) & p.rmask;
  }
  atomicOr(&p.flags[0], 1u);
  return -1;
}
template <typename P>
__device__ __forceinline__ int global_pair_slot(const P& p, unsigned long long key) {
  uint32_t h = hash64(key) & p.pmask;
  for (int i = 0; i < kGlobalProbes; ++i) {
    unsigned long long cur = __hip_atomic_load(&p.pkeys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == key) return (int)h;
    if (cur == 0) {
      unsigned long long old = atomicCAS(&p.pkeys[h], 0ull, key);
      if (old == 0 || old == key) return (int)h;
    }
    h = (h + 1) & p.pmask;
  }
  atomicOr(&p.flags[1], 1u);
  return -1;
}

// ---- ring entry (dwords) ---------------------------------------------------------------------------
//  0 key lo (region: label+1; pair: b+1)     1 key hi (region: 0; pair: a+1)
//  2 region: x | y<<6 | zfirst<<11 | zlast<<16 | border<<21 (tile-relative)     pair: cnt
//  3 region: cnt (only read with a mask: without one cnt = zlast - zfirst + 1)  pair: 4 x 8-bit threshold counters
//  4,5 sum (f64)   6,7 sq (f64)   8 min (float)   9 max (float)   10,11 hist bins 0-7 (8-bit packed)  [12,13 bins 8-15]
// (0 .. kAhead-1 as a tuple of integral constants: the plane buffers are indexed at compile time)
template <int... I> constexpr auto ahead_seq(std::integer_sequence<int, I...>) { return std::tuple<std::integral_constant<int, I>...>{}; }
using kAheadSeq = decltype(ahead_seq(std::make_integer_sequence<int, kAhead>{}));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((address_space(3))) lds_u32x4;
typedef lds_u32x4* lds_u4_ptr;

// control words of the rings: plain LDS loads / stores the compiler may neither cache nor move.  The LDS executes the
// operations of a wave in order, so entries written before the tail are visible to whoever reads the new tail.
__device__ __forceinline__ uint32_t ctl_load(const uint32_t* w) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void ctl_store(uint32_t* w, uint32_t v) {
  asm volatile("" ::: "memory");
  __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}
// four 8-bit counters -> two words of two 16-bit counters
__device__ __forceinline__ unsigned long long widen4(uint32_t w) {
  return (unsigned long long)__builtin_amdgcn_perm(w, w, 0x0c010c00u) | ((unsigned long long)__builtin_amdgcn_perm(w, w, 0x0c030c02u) << 32);
}
__device__ __forceinline__ uint32_t hist_byte(const u32x4& c, const u32x4& d, int k) {
  const uint32_t w = (k < 4) ? c.z : (k < 8) ? c.w : (k < 12) ? d.x : d.y;
  return (w >> ((k & 3) * 8)) & 0xFF;
}

// ---- combining a batch before it touches the tables (round 4) --------------------------------------------------------------
// Lanes of a marching wave that sit in the same supervoxel finish their runs together: a batch (lane = entry) holds RUNS of adjacent
// lanes with one key -- thirteen on average -- and every LDS atomic on one address is serialised by the hardware (f64 adds worst).
// With the prefetch pipeline repaired the drainers became the pass's critical path (ablations of round 4: march alone 2.9 ms,
// + ring writes 4.6 ms, + drain 8.4 ms, + fold 9.7 ms; drain without same-address conflicts 6.5 ms).  So a batch is first reduced by
// key inside the wave: a segmented inclusive scan over each row of sixteen lanes (DPP row_shr 1, 2, 4, 8; a segment = adjacent
// lanes with equal keys), after which only the LAST lane of every segment goes to the tables -- a fifth of the lookups and atomics,
// nearly all of them on distinct addresses.  Every field of an entry reduces by an integer add, an OR, an unsigned max or an f64 add
// (exact for Q8 images, 1e-15 otherwise, like the table atomics they replace), and zero is the identity of all of them, which is what a
// lane without a source in its row receives.
template <int D> __device__ __forceinline__ uint32_t row_shr(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + D, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t row_shl1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true); }
struct SegAdd { __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a + b; } };
struct SegOr { __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a | b; } };
struct SegMax { __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; } };
template <int D, class Op> __device__ __forceinline__ void seg_u32(uint32_t& v, const uint32_t m, Op op) { v = op(v, m & row_shr<D>(v)); }
template <int D> __device__ __forceinline__ void seg_f64(double& v, const uint32_t m) {
  const uint32_t lo = m & row_shr<D>((uint32_t)__double2loint(v)), hi = m & row_shr<D>((uint32_t)__double2hiint(v));
  v += __hiloint2double((int)hi, (int)lo);            // (a masked-out contribution is +0.0)
}
#ifndef GLIA_ACC_COMBINE
#define GLIA_ACC_COMBINE 2       // (swept in round 4 at 1024^3: none 9.78 ms, 1 level 9.24, 2 levels 8.81, 3 levels 8.96, 4 levels 9.14)
#endif
constexpr int kCombineLevels = GLIA_ACC_COMBINE;
// head flags of a batch: 1 where a segment starts (first lane of a row, inactive lane, key differs from the lane before)
__device__ __forceinline__ uint32_t seg_heads(const uint32_t klo, const uint32_t khi, const bool act, const int lane) {
  const uint32_t plo = row_shr<1>(klo), phi = row_shr<1>(khi);
  return (!act || (lane & ((1 << kCombineLevels) - 1)) == 0 || plo != klo || phi != khi) ? 1u : 0u;
}
// ... and whether a lane is the LAST of its segment (the lane that goes to the tables).  LEVELS scan steps combine groups of
// 2^LEVELS lanes: a segment is cut at every multiple of that.
template <int LEVELS>
__device__ __forceinline__ bool seg_tail(const uint32_t head, const bool act, const int lane) {
  constexpr int G = (1 << LEVELS) - 1;
  const uint32_t next_head = row_shl1(head);      // (fetched by EVERY lane, in front of the test: a DPP move under a divergent condition cannot read the lanes the condition switched off)
  return act && ((lane & G) == G || next_head != 0u);      // (an inactive lane is a head; no lane behind the row's last: handled by the first test)
}

// One batch of finished REGION runs (lane = entry) goes into the workgroup's tables
template <int BINS, bool MASK, class Release>
__device__ __forceinline__ void drain_regions(Lds<BINS>& s, const uint32_t entryAddr, const bool act, const int lane, Release release) {
  using G = Geo<BINS>;
  const lds_u4_ptr e = (lds_u4_ptr)(uintptr_t)entryAddr;
  u32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0}, c = {0, 0, 0, 0}, d = {0, 0, 0, 0};
  if (act) { a = e[0]; b = e[1]; c = e[2]; if (BINS > 8) d = e[3]; }
  // the entries are in registers: their ring slots are free from here on (the heads move BEFORE the tables are touched -- a
  // marching wave that waits for room waits a few hundred cycles less per batch)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  release();
  double sum = __hiloint2double((int)b.y, (int)b.x), sq = __hiloint2double((int)b.w, (int)b.z);
  const uint32_t cmn = c.x, cmx = c.y;      // (__builtin_bit_cast of a vector ELEMENT reads element 0 whatever the index)
  uint32_t omin = act ? ~float_ord(__builtin_bit_cast(float, cmn)) : 0u, omax = act ? float_ord(__builtin_bit_cast(float, cmx)) : 0u;
  const uint32_t meta = a.z;
  const uint32_t rx = meta & 63, ry = (meta >> 6) & 31, zs = (meta >> 11) & 31, zl = (meta >> 16) & 31;
  uint32_t border = (meta >> 21) & 63;
  uint32_t rcnt = act ? (MASK ? a.w : zl - zs + 1u) : 0u;
  uint32_t zmask = act ? (2u << zl) - (1u << zs) : 0u;      // planes zs..zl (2u << 31 wraps to 0: the subtraction still gives the bits)
  uint32_t xlo = act ? (rx < 32u ? 1u << rx : 0u) : 0u, xhi = act ? (rx >= 32u ? 1u << (rx - 32u) : 0u) : 0u, ymask = act ? 1u << ry : 0u;
  uint32_t first = act ? 0xFFFFFu - ((zs << 11) | (meta & 0x7FFu)) : 0u;
  unsigned long long hw[BINS / 4];
  hw[0] = widen4(c.z); hw[1] = widen4(c.w);
  if (BINS > 8) { hw[BINS / 4 - 2] = widen4(d.x); hw[BINS / 4 - 1] = widen4(d.y); }
  uint32_t h[BINS / 2];
#pragma unroll
  for (int k = 0; k < BINS / 4; ++k) { h[2 * k] = (uint32_t)hw[k]; h[2 * k + 1] = (uint32_t)(hw[k] >> 32); }
#ifndef GLIA_ACC_NOCOMBINE
  const uint32_t head = seg_heads(a.x, 0u, act, lane);
  uint32_t f = head;
  auto level = [&](auto DT) __attribute__((always_inline)) {
    constexpr int D = decltype(DT)::value;
    const uint32_t m = f ? 0u : 0xFFFFFFFFu;
    seg_u32<D>(rcnt, m, SegAdd{}); seg_u32<D>(border, m, SegAdd{});
    seg_u32<D>(xlo, m, SegOr{}); seg_u32<D>(xhi, m, SegOr{}); seg_u32<D>(ymask, m, SegOr{}); seg_u32<D>(zmask, m, SegOr{});
    seg_u32<D>(first, m, SegMax{}); seg_u32<D>(omin, m, SegMax{}); seg_u32<D>(omax, m, SegMax{});
    seg_f64<D>(sum, m); seg_f64<D>(sq, m);
#pragma unroll
    for (int k = 0; k < BINS / 2; ++k) seg_u32<D>(h[k], m, SegAdd{});
    f |= row_shr<D>(f);
  };
  level(std::integral_constant<int, 1>{}); level(std::integral_constant<int, 2>{});
  if (kCombineLevels > 2) level(std::integral_constant<int, 4>{});
  if (kCombineLevels > 3) level(std::integral_constant<int, 8>{});
  if (!seg_tail<kCombineLevels>(head, act, lane)) return;
#else
  if (!act) return;
#endif
#ifdef GLIA_ACC_NOLOOKUP
  int slot = (int)((a.x * 2654435761u) >> 24);
#else
  int slot = lds_slot(s.rkey, G::kRegSlots, (unsigned long long)a.x);
#endif
#ifdef GLIA_ACC_NOATOM
  if (slot >= 0) { if (rcnt == 0xFFFFFFFFu) s.rrec[slot] = rcnt + first + omin + omax + xlo + xhi + ymask + zmask + h[0] + h[1] + h[2] + h[3] + (uint32_t)sum + (uint32_t)sq; return; }
#endif
#ifdef GLIA_ACC_NOCONFLICT
  if (slot >= 0) slot = (int)((entryAddr / 48u) % (uint32_t)G::kRegSlots);
#endif
  if (slot >= 0) {
    uint32_t* rec = &s.rrec[slot * Lds<BINS>::kRegWords];
    atomicAdd((unsigned long long*)&rec[LR_CNT], (unsigned long long)rcnt | ((unsigned long long)border << 32));
    atomicAdd((double*)&rec[LR_SUM], sum);
    atomicAdd((double*)&rec[LR_SQ], sq);
    atomicMax(&rec[LR_MIN], omin);
    atomicMax(&rec[LR_MAX], omax);
#pragma unroll
    for (int k = 0; k < BINS / 4; ++k) atomicAdd((unsigned long long*)&rec[LR_HIST + 2 * k], (unsigned long long)h[2 * k] | ((unsigned long long)h[2 * k + 1] << 32));
    atomicOr((unsigned long long*)&rec[LR_XMASK], (unsigned long long)xlo | ((unsigned long long)xhi << 32));
    atomicOr((unsigned long long*)&rec[LR_YMASK], (unsigned long long)ymask | ((unsigned long long)zmask << 32));
    atomicMax(&rec[LR_FIRST], first);
    return;
  }
  // LDS table saturated (tile with very many tiny supervoxels): straight to the global tables (slow, exact).  The host
  // watches the count and gives the next pass shallower tiles.
  const TableParams& p = s.tp;
  atomicAdd(&p.flags[7], 1u);
  const int g = global_region_slot(p, a.x);
  if (g < 0) return;
  uint32_t* r = &p.rrec[(size_t)g * kRegionWords];
  atomicAdd(&r[R_CNT], rcnt);
  if (border) atomicAdd(&r[R_BORDER], border);
  atomicAdd((double*)&r[R_SUM], sum); atomicAdd((double*)&r[R_SQ], sq);
  atomicMax(&r[R_MIN], omin); atomicMax(&r[R_MAX], omax);
  {
    // the combined run's extent from its occupancy masks (tile-relative), its first voxel from `first`
    const unsigned long long xm = (unsigned long long)xlo | ((unsigned long long)xhi << 32);
    const uint32_t gx0 = (uint32_t)p.x0 + (uint32_t)__builtin_ctzll(xm), gx1 = (uint32_t)p.x0 + 63u - (uint32_t)__builtin_clzll(xm);
    const uint32_t gy0 = (uint32_t)p.y0 + (uint32_t)__builtin_ctz(ymask), gy1 = (uint32_t)p.y0 + 31u - (uint32_t)__builtin_clz(ymask);
    const uint32_t gz0 = (uint32_t)p.z0 + (uint32_t)__builtin_ctz(zmask), gz1 = (uint32_t)p.z0 + 31u - (uint32_t)__builtin_clz(zmask);
    atomicMax(&r[R_LO + 0], 0x7fffffffu - gx0); atomicMax(&r[R_HI + 0], gx1 + 1u);
    atomicMax(&r[R_LO + 1], 0x7fffffffu - gy0); atomicMax(&r[R_HI + 1], gy1 + 1u);
    atomicMax(&r[R_LO + 2], 0x7fffffffu - gz0); atomicMax(&r[R_HI + 2], gz1 + 1u);
    const uint32_t fv = 0xFFFFFu - first;
    const unsigned long long fidx = (unsigned long long)((p.z0 + (fv >> 11)) * p.ny * p.nx + (p.y0 + ((fv >> 6) & 31)) * p.nx + (p.x0 + (fv & 63)));
    atomicMax((unsigned long long*)&r[R_FIRST], ~fidx);
  }
#pragma unroll
  for (int k = 0; k < BINS; ++k) {
    const uint32_t hk = (h[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
    if (hk) atomicAdd(&r[R_HIST + k], hk);
  }
}

// One batch of finished PAIR runs
template <int BINS, class Release>
__device__ __forceinline__ void drain_pairs(Lds<BINS>& s, const uint32_t entryAddr, const bool act, const int lane, Release release) {
  using G = Geo<BINS>;
  const lds_u4_ptr e = (lds_u4_ptr)(uintptr_t)entryAddr;
  u32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0}, c = {0, 0, 0, 0}, d = {0, 0, 0, 0};
  if (act) { a = e[0]; b = e[1]; c = e[2]; if (BINS > 8) d = e[3]; }
  // the entries are in registers: their ring slots are free from here on (the heads move BEFORE the tables are touched -- a
  // marching wave that waits for room waits a few hundred cycles less per batch)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  release();
  double sum = __hiloint2double((int)b.y, (int)b.x), sq = __hiloint2double((int)b.w, (int)b.z);
  const uint32_t cmn = c.x, cmx = c.y;      // (__builtin_bit_cast of a vector ELEMENT reads element 0 whatever the index)
  uint32_t omin = act ? ~float_ord(__builtin_bit_cast(float, cmn)) : 0u, omax = act ? float_ord(__builtin_bit_cast(float, cmx)) : 0u;
  const unsigned long long key = ((unsigned long long)a.y << 32) | a.x;
  uint32_t cnt = a.z;
  const unsigned long long tw = widen4(a.w);
  uint32_t t0 = (uint32_t)tw, t1 = (uint32_t)(tw >> 32);
  unsigned long long hw[BINS / 4];
  hw[0] = widen4(c.z); hw[1] = widen4(c.w);
  if (BINS > 8) { hw[BINS / 4 - 2] = widen4(d.x); hw[BINS / 4 - 1] = widen4(d.y); }
  uint32_t h[BINS / 2];
#pragma unroll
  for (int k = 0; k < BINS / 4; ++k) { h[2 * k] = (uint32_t)hw[k]; h[2 * k + 1] = (uint32_t)(hw[k] >> 32); }
#ifndef GLIA_ACC_NOCOMBINE
  const uint32_t head = seg_heads(a.x, a.y, act, lane);
  uint32_t f = head;
  auto level = [&](auto DT) __attribute__((always_inline)) {
    constexpr int D = decltype(DT)::value;
    const uint32_t m = f ? 0u : 0xFFFFFFFFu;
    seg_u32<D>(cnt, m, SegAdd{}); seg_u32<D>(t0, m, SegAdd{}); seg_u32<D>(t1, m, SegAdd{});
    seg_u32<D>(omin, m, SegMax{}); seg_u32<D>(omax, m, SegMax{});
    seg_f64<D>(sum, m); seg_f64<D>(sq, m);
#pragma unroll
    for (int k = 0; k < BINS / 2; ++k) seg_u32<D>(h[k], m, SegAdd{});
    f |= row_shr<D>(f);
  };
  level(std::integral_constant<int, 1>{}); level(std::integral_constant<int, 2>{});
  if (kCombineLevels > 2) level(std::integral_constant<int, 4>{});
  if (kCombineLevels > 3) level(std::integral_constant<int, 8>{});
  if (!seg_tail<kCombineLevels>(head, act, lane)) return;
#else
  if (!act) return;
#endif
#ifdef GLIA_ACC_NOLOOKUP
  int slot = (int)(((a.x ^ (a.y * 0x9E3779B1u)) * 2654435761u) >> 22);
#else
  int slot = lds_slot(s.pkey, G::kPairSlots, key);
#endif
#ifdef GLIA_ACC_NOATOM
  if (slot >= 0) { if (cnt == 0xFFFFFFFFu) s.prec[slot] = cnt + t0 + t1 + omin + omax + h[0] + h[1] + h[2] + h[3] + (uint32_t)sum + (uint32_t)sq; return; }
#endif
#ifdef GLIA_ACC_NOCONFLICT
  if (slot >= 0) slot = (int)((entryAddr / 48u) % (uint32_t)G::kPairSlots);
#endif
  if (slot >= 0) {
    uint32_t* rec = &s.prec[slot * Lds<BINS>::kPairWordsL];
    atomicAdd(&rec[LP_CNT], cnt);
    atomicAdd((unsigned long long*)&rec[LP_THR], (unsigned long long)t0 | ((unsigned long long)t1 << 32));
    atomicAdd((double*)&rec[LP_SUM], sum);
    atomicAdd((double*)&rec[LP_SQ], sq);
    atomicMax(&rec[LP_MIN], omin);
    atomicMax(&rec[LP_MAX], omax);
#pragma unroll
    for (int k = 0; k < BINS / 4; ++k) atomicAdd((unsigned long long*)&rec[LP_HIST + 2 * k], (unsigned long long)h[2 * k] | ((unsigned long long)h[2 * k + 1] << 32));
    return;
  }
  const TableParams& p = s.tp;
  atomicAdd(&p.flags[7], 1u);
  const int g = global_pair_slot(p, key);
  if (g < 0) return;
  uint32_t* r = &p.prec[(size_t)g * kPairWords];
  atomicAdd(&r[P_CNT], cnt);
  atomicAdd((double*)&r[P_SUM], sum); atomicAdd((double*)&r[P_SQ], sq);
  atomicMax(&r[P_MIN], omin); atomicMax(&r[P_MAX], omax);
#pragma unroll
  for (int k = 0; k < GLIA_HMT_MAX_THRESH; ++k) {
    const uint32_t tk = ((k < 2 ? t0 : t1) >> ((k & 1) * 16)) & 0xFFFFu;
    if (tk) atomicAdd(&r[P_THR + k], tk);
  }
#pragma unroll
  for (int k = 0; k < BINS; ++k) {
    const uint32_t hk = (h[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
    if (hk) atomicAdd(&r[P_HIST + k], hk);
  }
}

// a drainer wave: empties the rings of the marching waves d, d + kDrainers, d + 2 kDrainers until they are done.  A batch
// gathers up to 64 entries of ONE kind from the three rings (lane = entry): full lanes, no region / pair divergence.
template <int BINS, bool MASK>
__device__ __forceinline__ void drainer_loop(Lds<BINS>& s, const int d, const int lane) {
  using G = Geo<BINS>;
  constexpr uint32_t EB = G::kEntryWords * 4;
  // the rings of the marching waves d, d + kDrainers, d + 2 kDrainers (those that exist: one to three)
  const int w0 = d, w1 = d + G::kDrainers, w2 = d + 2 * G::kDrainers;
  const bool v1 = w1 < G::kMarchers, v2 = w2 < G::kMarchers;                        // (wave-uniform)
  const int x1 = v1 ? w1 : w0, x2 = v2 ? w2 : w0;                                   // (a ring that does not exist: ring 0's words, never taken from)
  const uint32_t share = v2 ? 21u : v1 ? 32u : 64u;                                 // a balanced take
  const uint32_t bR0 = (uint32_t)(uintptr_t)s.ringR[w0], bR1 = (uint32_t)(uintptr_t)s.ringR[x1], bR2 = (uint32_t)(uintptr_t)s.ringR[x2];
  const uint32_t bP0 = (uint32_t)(uintptr_t)s.ringP[w0], bP1 = (uint32_t)(uintptr_t)s.ringP[x1], bP2 = (uint32_t)(uintptr_t)s.ringP[x2];
  uint32_t hR0 = 0, hR1 = 0, hR2 = 0, hP0 = 0, hP1 = 0, hP2 = 0;      // entries consumed (wave-uniform)
#ifndef GLIA_ACC_DPRIO
#define GLIA_ACC_DPRIO 3
#endif
  __builtin_amdgcn_s_setprio(GLIA_ACC_DPRIO);     // the marching waves of this SIMD always have work: without priority the drainer starves
#ifdef GLIA_ACC_COUNTERS
  unsigned long long pc[6] = {0, 0, 0, 0, 0, 0};     // region batches, entries, cycles; pair batches, entries, cycles
  uint32_t polls = 0;
  const unsigned long long tstart = __builtin_readcyclecounter();
#define DPROF(x) x
#else
#define DPROF(x)
#endif
  for (;;) {
    // tails and state of a wave in one 16-byte read ("done" is stored after the last tails, and the LDS keeps a wave's order)
    u32x4 q0 = *(const volatile lds_u32x4*)(uintptr_t)(uint32_t)(uintptr_t)&s.ctl[w0][0];
    u32x4 q1 = *(const volatile lds_u32x4*)(uintptr_t)(uint32_t)(uintptr_t)&s.ctl[x1][0];
    u32x4 q2 = *(const volatile lds_u32x4*)(uintptr_t)(uint32_t)(uintptr_t)&s.ctl[x2][0];
    const uint32_t st0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)q0.z), st1 = v1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)q1.z) : kRingDone,
                   st2 = v2 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)q2.z) : kRingDone;
    const uint32_t tR0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)q0.x), tR1 = v1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)q1.x) : hR1,
                   tR2 = v2 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)q2.x) : hR2;
    const uint32_t tP0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)q0.y), tP1 = v1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)q1.y) : hP1,
                   tP2 = v2 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)q2.y) : hP2;
    const bool urgent = ((st0 | st1 | st2) & kRingWaiting) != 0u || st0 == kRingDone || (v1 && st1 == kRingDone) || (v2 && st2 == kRingDone);
    bool any = false;
    uint32_t left;
    {
      const uint32_t a0 = tR0 - hR0, a1 = tR1 - hR1, a2 = tR2 - hR2;
      const uint32_t total = a0 + a1 + a2;
      left = total;
      if (total >= (uint32_t)kDrainAt || (urgent && total)) {
        // a balanced take: entries of one wave repeat few keys (LDS atomics on one address serialise), those of three waves
        // rows apart do not
        uint32_t c0 = a0 < share ? a0 : share, c1 = a1 < share ? a1 : share, c2 = a2 < share ? a2 : share;
        {
          uint32_t room = 64u - (c0 + c1 + c2);
          const uint32_t e0 = a0 - c0 < room ? a0 - c0 : room; c0 += e0; room -= e0;
          const uint32_t e1 = a1 - c1 < room ? a1 - c1 : room; c1 += e1; room -= e1;
          const uint32_t e2 = a2 - c2 < room ? a2 - c2 : room; c2 += e2;
        }
        const uint32_t j = (uint32_t)lane;
        const bool in0 = j < c0, in1 = j < c0 + c1;
        const uint32_t base = in0 ? bR0 : in1 ? bR1 : bR2;
        const uint32_t idx = in0 ? hR0 + j : in1 ? hR1 + (j - c0) : hR2 + (j - c0 - c1);
        DPROF(const unsigned long long tq = __builtin_readcyclecounter();)
        hR0 += c0; hR1 += c1; hR2 += c2;
        drain_regions<BINS, MASK>(s, base + (idx & (uint32_t)(kRingR - 1)) * EB, j < c0 + c1 + c2, lane, [&]() __attribute__((always_inline)) {
          if (c0) ctl_store(&s.ctl[w0][C_HEAD_R], hR0);
          if (c1) ctl_store(&s.ctl[x1][C_HEAD_R], hR1);
          if (c2) ctl_store(&s.ctl[x2][C_HEAD_R], hR2);
        });
        DPROF(asm volatile("s_waitcnt lgkmcnt(0)"); pc[0] += 1; pc[1] += c0 + c1 + c2; pc[2] += __builtin_readcyclecounter() - tq;)
        left = total - (c0 + c1 + c2);
        any = true;
      }
    }
    {
      const uint32_t a0 = tP0 - hP0, a1 = tP1 - hP1, a2 = tP2 - hP2;
      const uint32_t total = a0 + a1 + a2;
      left += total;
      if (total >= (uint32_t)kDrainAt || (urgent && total)) {
        // a balanced take: entries of one wave repeat few keys (LDS atomics on one address serialise), those of three waves
        // rows apart do not
        uint32_t c0 = a0 < share ? a0 : share, c1 = a1 < share ? a1 : share, c2 = a2 < share ? a2 : share;
        {
          uint32_t room = 64u - (c0 + c1 + c2);
          const uint32_t e0 = a0 - c0 < room ? a0 - c0 : room; c0 += e0; room -= e0;
          const uint32_t e1 = a1 - c1 < room ? a1 - c1 : room; c1 += e1; room -= e1;
          const uint32_t e2 = a2 - c2 < room ? a2 - c2 : room; c2 += e2;
        }
        const uint32_t j = (uint32_t)lane;
        const bool in0 = j < c0, in1 = j < c0 + c1;
        const uint32_t base = in0 ? bP0 : in1 ? bP1 : bP2;
        const uint32_t idx = in0 ? hP0 + j : in1 ? hP1 + (j - c0) : hP2 + (j - c0 - c1);
        DPROF(const unsigned long long tq = __builtin_readcyclecounter();)
        hP0 += c0; hP1 += c1; hP2 += c2;
        drain_pairs<BINS>(s, base + (idx & (uint32_t)(kRingP - 1)) * EB, j < c0 + c1 + c2, lane, [&]() __attribute__((always_inline)) {
          if (c0) ctl_store(&s.ctl[w0][C_HEAD_P], hP0);
          if (c1) ctl_store(&s.ctl[x1][C_HEAD_P], hP1);
          if (c2) ctl_store(&s.ctl[x2][C_HEAD_P], hP2);
        });
        DPROF(asm volatile("s_waitcnt lgkmcnt(0)"); pc[3] += 1; pc[4] += c0 + c1 + c2; pc[5] += __builtin_readcyclecounter() - tq;)
        left -= c0 + c1 + c2;
        any = true;
      }
    }
    if (st0 == kRingDone && st1 == kRingDone && st2 == kRingDone && left == 0u) break;
    DPROF(polls += any ? 0u : 1u;)
    if (!any) __builtin_amdgcn_s_sleep(2);
  }
#ifdef GLIA_ACC_COUNTERS
  if (lane == 0) {
    unsigned long long* g = reinterpret_cast<unsigned long long*>(s.tp.flags + 16);
    for (int k = 0; k < 6; ++k) atomicAdd(&g[k], pc[k]);
    atomicAdd(&g[6], (unsigned long long)polls);
    atomicAdd(&g[7], __builtin_readcyclecounter() - tstart);
  }
#endif
}

struct Tile { int64_t x0, y0, z0; };

// per-lane run accumulators (registers)
template <int BINS>
struct Run {
  uint32_t k0, k1;              // region: label, -      pair: own label a, neighbour label b
  uint32_t pos;                 // region: x | y<<6 | zfirst<<11 (tile-relative)
  uint32_t cnt;                 // region: voxels (kept only with a mask)   pair: voxels
  uint32_t aux;                 // region: border voxels                    pair: 4 x 8-bit threshold counters
  uint32_t last;                // region, with a mask: plane of the most recent voxel
  double sum, sq;
  float mn, mx;
  unsigned long long h[BINS / 8];   // 8-bit packed histogram counters (a run has at most 32 voxels)
};

// x-1 / x+1 labels of a row held one voxel per lane: DPP shifts over the whole wave; the lane without a source keeps `edge`
__device__ __forceinline__ uint32_t from_left(uint32_t edge, uint32_t v) { return __builtin_amdgcn_update_dpp(edge, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t from_right(uint32_t edge, uint32_t v) { return __builtin_amdgcn_update_dpp(edge, v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t lanes_below(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// float min / max as the bare instructions (fminf / fmaxf canonicalise both operands first: three instructions each)
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// rows are fetched with buffer loads: a wave-uniform base (resource, scalar registers) + a wave-uniform byte offset (scalar) + the
// lane's constant byte offset -- no per-lane address arithmetic at all
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)0xFFFFFFFF, 0x00020000);
}
__device__ __forceinline__ uint32_t bload(rsrc_t r, uint32_t voff, uint32_t soff) { return __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0); }
template <int BINS, bool MASK>
__global__ __launch_bounds__(Geo<BINS>::kThreadsT) void rag_accumulate_kernel(const AccParams p) {
  using G = Geo<BINS>;
  __shared__ __attribute__((aligned(16))) Lds<BINS> s;
  constexpr int EW = G::kEntryWords;
  constexpr int kThreadsT = G::kThreadsT;
  const int tid = threadIdx.x;
  {
    uint4* w = reinterpret_cast<uint4*>(&s);
    const uint4 z4 = {0, 0, 0, 0};
    // keys + records only; the rings need no initialisation
    constexpr int n16 = (int)((sizeof(s.rkey) + sizeof(s.pkey) + sizeof(s.rrec) + sizeof(s.prec)) / 16);
    for (int i = tid; i < n16; i += kThreadsT) w[i] = z4;
    if (tid < G::kMarchers * 8) (&s.ctl[0][0])[tid] = 0u;
  }

  // tile coordinates: blocks that share blockIdx % 8 share an XCD (L2); give each XCD a contiguous
  // run of tiles so y/z halo rows of neighbouring tiles are served by the same L2.
  const uint32_t nb = (uint32_t)p.nbx * p.nby * p.nbz;
  uint32_t bid = blockIdx.x;
  {
    const uint32_t q = nb / 8, r = nb % 8, xcd = bid % 8, k = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int bx = bid % p.nbx, by = (bid / p.nbx) % p.nby, bz = bid / (p.nbx * p.nby);
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // the wave index as a scalar
  const int64_t nx = p.nx, ny = p.ny, nz = p.nz;
  Tile tile;
  tile.x0 = (int64_t)bx * kTileX; tile.y0 = (int64_t)by * kTileY; tile.z0 = p.zb + (int64_t)bz * p.tz;
  const int64_t x = tile.x0 + lane;
  const int64_t z1 = (tile.z0 + p.tz < p.ze) ? tile.z0 + p.tz : p.ze;
  const int n = (int)(z1 - tile.z0);            // planes of this tile
  const int64_t gz0 = p.gz0, gnz = p.gnz;
  const int64_t sz = nx * ny;
  const bool is3d = p.dim == 3;
  const int nfull = 2 * p.dim;
  const bool marcher = wave < G::kMarchers;
  uint32_t* ctl = s.ctl[marcher ? wave : 0];
  // LDS byte addresses of the wave's rings (the low half of the generic pointers)
  const uint32_t ringAddrR = (uint32_t)(uintptr_t)s.ringR[marcher ? wave : 0], ringAddrP = (uint32_t)(uintptr_t)s.ringP[marcher ? wave : 0];
  uint32_t tailR = 0, seenR = 0, tailP = 0, seenP = 0;     // wave-uniform: entries written / entries known to be consumed
#ifdef GLIA_HMT_PROFILE
  const uint32_t dbg = p.debug;       // ablation switches (GLIA_HMT_DEBUG), profiling builds only
#else
#ifndef GLIA_ACC_ABL
#define GLIA_ACC_ABL 0
#endif
  constexpr uint32_t dbg = GLIA_ACC_ABL;      // (kernel experiments: the same switches fixed at compile time, make accvar; 0 in the shipped build)
#endif
  if (tid == 0) {
    TableParams tp;
    tp.rkeys = p.rkeys; tp.rrec = p.rrec; tp.pkeys = p.pkeys; tp.prec = p.prec; tp.flags = p.flags;
    tp.rmask = p.rmask; tp.pmask = p.pmask; tp.nx = p.nx; tp.ny = p.ny;
    tp.x0 = tile.x0; tp.y0 = tile.y0; tp.z0 = tile.z0 + p.gz0;     // global coordinates for bbox / first-voxel index
    s.tp = tp;
  }
  __syncthreads();

  const bool sepCentre = MASK && p.lab_c != p.lab;     // contour-only mode: the centre keeps its label under the mask

  // thresholds: wave-uniform, kept in vector registers (scalar registers are the scarcer kind in the march)
  float fb[BINS];
#pragma unroll
  for (int k = 0; k < BINS; ++k) { fb[k] = p.hist.fb[k]; asm volatile("" : "+v"(fb[k])); }
  float lo_f = p.hist.lo_f, hi_f = p.hist.hi_f;
  asm volatile("" : "+v"(lo_f), "+v"(hi_f));
  const int nbins = p.hist.bins;
  float t0 = p.thr_f[0], t1 = p.thr_f[1], t2 = p.thr_f[2], t3 = p.thr_f[3];   // +inf beyond nthr
  asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));

  Run<BINS> rr;
  rr.k0 = rr.k1 = 0u; rr.pos = 0u; rr.cnt = 0u; rr.aux = 0u; rr.last = 0u; rr.sum = 0.0; rr.sq = 0.0; rr.mn = 0.f; rr.mx = 0.f;
#pragma unroll
  for (int k = 0; k < BINS / 8; ++k) rr.h[k] = 0ull;
  Run<BINS> pr = rr;
  bool rhas = false, phas = false;     // the lane holds an unfinished region / pair run

  auto put = [&](const uint32_t entryAddr, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, const Run<BINS>& r) __attribute__((always_inline)) {
    lds_u4_ptr e = (lds_u4_ptr)(uintptr_t)entryAddr;
    u32x4 a, b, c;
    a.x = w0; a.y = w1; a.z = w2; a.w = w3;
    b.x = (uint32_t)__double2loint(r.sum); b.y = (uint32_t)__double2hiint(r.sum);
    b.z = (uint32_t)__double2loint(r.sq); b.w = (uint32_t)__double2hiint(r.sq);
    c.x = __builtin_bit_cast(uint32_t, r.mn); c.y = __builtin_bit_cast(uint32_t, r.mx);
    c.z = (uint32_t)r.h[0]; c.w = (uint32_t)(r.h[0] >> 32);
    e[0] = a; e[1] = b; e[2] = c;
    if (BINS > 8) { u32x4 d; d.x = (uint32_t)r.h[BINS / 8 - 1]; d.y = (uint32_t)(r.h[BINS / 8 - 1] >> 32); d.z = 0; d.w = 0; e[3] = d; }
  };
#ifdef GLIA_ACC_COUNTERS
  unsigned long long waitCycles = 0; uint32_t waits = 0;
  const unsigned long long tmarch0 = __builtin_readcyclecounter();
#endif
  // both tails in one 8-byte store (every few planes, and before the wave waits or ends)
  auto publish = [&]() __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    *(volatile unsigned long long __attribute__((address_space(3)))*)(uintptr_t)(uint32_t)(uintptr_t)&ctl[C_TAIL_R] =
        (unsigned long long)tailR | ((unsigned long long)tailP << 32);
    asm volatile("" ::: "memory");
  };
  // room for n more entries in a ring of the wave (waits for its drainer when there is none)
  auto reserve = [&](const uint32_t tail, uint32_t& seen, const int n, const int cap, const int headWord) __attribute__((always_inline)) {
#ifdef GLIA_ACC_NODRAIN
    return;
#endif
    if (__builtin_expect(tail - seen + (uint32_t)n > (uint32_t)cap, 0)) {
      seen = ctl_load(&ctl[headWord]);
      if (tail - seen + (uint32_t)n > (uint32_t)cap) {
        publish();                                  // what is written must be seen, or nobody makes room
        ctl_store(&ctl[C_STATE], kRingWaiting);
#ifdef GLIA_ACC_COUNTERS
        const unsigned long long tw0 = __builtin_readcyclecounter();
#endif
        do {
          __builtin_amdgcn_s_sleep(1);
          seen = ctl_load(&ctl[headWord]);
        } while (tail - seen + (uint32_t)n > (uint32_t)cap);
#ifdef GLIA_ACC_COUNTERS
        waitCycles += __builtin_readcyclecounter() - tw0; waits += 1;
#endif
        ctl_store(&ctl[C_STATE], 0u);
      }
    }
  };
  // append the finished runs of the lanes in evR / evP to the wave's rings.
  // zlast: plane of the last voxel of a finished region run (wave-uniform without a mask)
  auto enqueue = [&](bool evR, bool evP, uint32_t zlast) __attribute__((always_inline)) {
    const unsigned long long mR = __ballot(evR), mP = __ballot(evP);
    const int nR = __popcll(mR), nP = __popcll(mP);
    if (nR) {
      const uint32_t zl = MASK ? rr.last : zlast;
      const uint32_t rmeta = rr.pos | (zl << 16) | (rr.aux << 21);
      reserve(tailR, seenR, nR, kRingR, C_HEAD_R);
      if (evR) put(ringAddrR + ((tailR + lanes_below(mR)) & (uint32_t)(kRingR - 1)) * (uint32_t)(EW * 4), rr.k0 + 1u, 0u, rmeta, rr.cnt, rr);
      tailR += (uint32_t)nR;
    }
    if (nP) {
      const uint32_t rank = lanes_below(mP);
      if (__builtin_expect(nP > kRingP, 0)) {      // more finished pair runs in one plane than their ring holds: two batches
        reserve(tailP, seenP, kRingP, kRingP, C_HEAD_P);
        if (evP && rank < (uint32_t)kRingP)
          put(ringAddrP + ((tailP + rank) & (uint32_t)(kRingP - 1)) * (uint32_t)(EW * 4), pr.k1 + 1u, pr.k0 + 1u, pr.cnt, pr.aux, pr);
        tailP += (uint32_t)kRingP;
        reserve(tailP, seenP, nP - kRingP, kRingP, C_HEAD_P);
        if (evP && rank >= (uint32_t)kRingP)
          put(ringAddrP + ((tailP + rank - (uint32_t)kRingP) & (uint32_t)(kRingP - 1)) * (uint32_t)(EW * 4), pr.k1 + 1u, pr.k0 + 1u, pr.cnt, pr.aux, pr);
        tailP += (uint32_t)(nP - kRingP);
      } else {
        reserve(tailP, seenP, nP, kRingP, C_HEAD_P);
        if (evP) put(ringAddrP + ((tailP + rank) & (uint32_t)(kRingP - 1)) * (uint32_t)(EW * 4), pr.k1 + 1u, pr.k0 + 1u, pr.cnt, pr.aux, pr);
        tailP += (uint32_t)nP;
      }
    }
    if (nR | nP) publish();
  };

  // INNER (a workgroup-uniform compile-time tag): the tile and its one-voxel halo lie inside the volume and there is no mask,
  // so every voxel is valid, every neighbour exists and no voxel is a border voxel -- the validity logic folds away.
  const bool inner_tile = !MASK && is3d && (dbg & 4) == 0 && tile.x0 > 0 && tile.x0 + kTileX < nx && tile.y0 > 0 && tile.y0 + kTileY < ny &&
                          tile.z0 + gz0 > 0 && z1 + gz0 < gnz && tile.z0 > 0 && z1 < nz;

  // one column: the row yrel of the tile, planes tile.z0 .. z1-1, one voxel per lane and plane
  // WHOLE (workgroup-uniform): the tile has a whole number of groups of kAhead planes, at least two -- every request of the
  // prefetch pipeline is then unconditional and the wait counts are the same on every path (see `step`)
  auto column = [&](auto INNER_T, auto WHOLE_T, const int yrel) __attribute__((always_inline)) {
    constexpr bool INNER = decltype(INNER_T)::value;
    constexpr bool WHOLE = decltype(WHOLE_T)::value;
    const int64_t y = tile.y0 + yrel;
    if (!INNER && y >= ny) return;                                  // wave-uniform
    const bool laneOk = INNER || x < nx;
    const bool ymv = INNER || y > 0, ypv = INNER || y + 1 < ny;    // wave-uniform
    const bool xmv = INNER || x > 0, xpv = INNER || x + 1 < nx;
    const int nv_lane = (int)xmv + (int)xpv;
    // wave-uniform element offset of (x0-1, y-1, z0-1): every row the column needs lies at a non-negative offset from it
    const int64_t rowoff = (tile.z0 * ny + y) * nx + tile.x0 - (sz + nx + 1);
    const uint32_t szb = (uint32_t)sz * 4u, nxb = (uint32_t)nx * 4u;     // the host keeps 11 planes below 4 GiB
    const uint32_t oc0 = szb + nxb + 4u;                                    // byte offset of (x0, y, plane 0 of the current base)
    // per-lane byte offsets.  The halo load reads relative to x0-1: lane 0 fetches x0-1, lane 63 fetches x0+64, the others
    // their own voxel again (the same cache lines).
    // Lanes beyond the volume's last column (edge tiles) read the last valid voxel of the row instead: every load of the pipeline is
    // then UNCONDITIONAL -- no branch, no exec mask around it -- and what an invalid lane or an invalid neighbour row delivers is
    // ignored by the validity flags of `step`.  (A skipped load is a path with fewer loads in flight: the wait-count pass then
    // waits for everything on every path.)
    const uint32_t lc = INNER ? (uint32_t)lane : (uint32_t)(x < nx ? lane : (int)(nx - 1 - tile.x0));
    const uint32_t lx4 = lc * 4u;
    uint32_t hx4 = lc * 4u + 4u;
    if (lane == 0 && xmv) hx4 = 0u;
    if (lane == 63 && xpv) hx4 = 65u * 4u;
    const uint32_t upb = ymv ? nxb : 0u, dnb = ypv ? nxb : 0u;         // a row that does not exist: the centre row again
    const uint32_t yx = ((uint32_t)yrel << 6) | (uint32_t)lane;

    uint32_t bLn[kAhead], bUp[kAhead], bDn[kAhead], bH[kAhead], bC[kAhead];
    float bV[kAhead];
    // request the rows of plane index t; oc = byte offset of its centre row from the resources' base.  (oc passes through an
    // empty asm so that the offsets of the five rows are formed from it with one scalar add each: left to itself the compiler
    // hoists twenty loop-invariant offset constants into scalar registers it does not have.)
    auto issue = [&](const int t, uint32_t oc, const rsrc_t rL, const rsrc_t rV, const rsrc_t rC, auto J) __attribute__((always_inline)) {
      constexpr int j = decltype(J)::value;
      asm volatile("" : "+s"(oc));
      if (INNER) {
        bLn[j] = bload(rL, lx4, oc + szb); bUp[j] = bload(rL, lx4, oc - nxb); bDn[j] = bload(rL, lx4, oc + nxb);
        bH[j] = bload(rL, hx4, oc - 4u);
        bV[j] = __builtin_bit_cast(float, bload(rV, lx4, oc));
      } else {
        const uint32_t dz = (tile.z0 + t + 1 < nz) ? szb : 0u;         // the plane above the volume's last: this plane again
        bLn[j] = bload(rL, lx4, oc + dz); bUp[j] = bload(rL, lx4, oc - upb); bDn[j] = bload(rL, lx4, oc + dnb);
        bH[j] = bload(rL, hx4, oc - 4u);
        bV[j] = __builtin_bit_cast(float, bload(rV, lx4, oc));
        if (MASK) bC[j] = bload(rC, lx4, oc);
      }
    };
    uint32_t Lp = 0u, Lc = 0u;
    {
      const rsrc_t rL = make_rsrc(p.lab + rowoff), rV = make_rsrc(p.img + rowoff), rC = make_rsrc(p.lab_c + rowoff);
      Lp = bload(rL, lx4, oc0 - ((INNER || tile.z0 > 0) ? szb : 0u));
      Lc = bload(rL, lx4, oc0);
      auto first = [&](auto... J) __attribute__((always_inline)) {
        ((void)((WHOLE || decltype(J)::value < n) ? (issue(decltype(J)::value, oc0 + (uint32_t)decltype(J)::value * szb, rL, rV, rC, J), 0) : 0), ...);
      };
      std::apply(first, kAheadSeq{});
    }
    if (INNER) {     // every voxel is valid: the region run of the column's first voxel starts here, `rhas` is constant
      rr.k0 = Lc; rr.pos = yx; rr.cnt = 0; rr.aux = 0; rr.sum = 0.0; rr.sq = 0.0; rr.mn = __builtin_inff(); rr.mx = -__builtin_inff();
#pragma unroll
      for (int k = 0; k < BINS / 8; ++k) rr.h[k] = 0ull;
      rhas = true;
    }

    // one voxel per lane: neighbour rule, run bookkeeping, accumulation
    // ISSUE: 1 = the rows of plane t + kAhead are requested unconditionally, 2 = never, 0 = if that plane exists.  The conditional
    // form costs the prefetch: where the two paths meet the compiler's wait-count pass must assume the path WITHOUT the newer
    // loads, so every step waited for the loads it had just issued (s_waitcnt vmcnt(4) right behind five buffer loads, vmcnt(0)
    // at the top of every group -- a full memory round trip per plane, found in the ISA in round 4).  The main loop below
    // issues always, its last group never; only ragged columns take the conditional form.
    auto step = [&](const int t, const int tb, const rsrc_t rL, const rsrc_t rV, const rsrc_t rC, auto J, auto ISSUE_T) __attribute__((always_inline)) {
      constexpr int j = decltype(J)::value;
      constexpr int ISSUE = decltype(ISSUE_T)::value;
      const uint32_t Ln = bLn[j], Up = bUp[j], Dn = bDn[j], Hh = bH[j];
      const float v = bV[j];
      const uint32_t xm = from_left(Hh, Lc), xp = from_right(Hh, Lc);
      const uint32_t L = (MASK && sepCentre) ? bC[j] : Lc;
      const int64_t zg = tile.z0 + t + gz0;
      bool zmv = INNER || (is3d && zg > 0), zpv = INNER || (is3d && zg + 1 < gnz);     // wave-uniform
      bool vxm = xmv, vxp = xpv, vym = ymv, vyp = ypv;
      const bool ok = INNER || (laneOk && !(dbg & 4) && (!MASK || L != kMaskedLabel));
      int nvalid = 0;
      if (MASK) {
        vxm = vxm && xm != kMaskedLabel; vxp = vxp && xp != kMaskedLabel;
        vym = vym && Up != kMaskedLabel; vyp = vyp && Dn != kMaskedLabel;
        zmv = zmv && Lp != kMaskedLabel; zpv = zpv && Ln != kMaskedLabel;
        nvalid = (int)vxm + (int)vxp + (int)vym + (int)vyp + (int)zmv + (int)zpv;
      } else if (!INNER) {
        nvalid = nv_lane + (int)ymv + (int)ypv + (int)zmv + (int)zpv;
      }
      uint32_t b = L;
      b = ((INNER || zpv) && Ln != L) ? Ln : b;
      b = ((INNER || zmv) && Lp != L) ? Lp : b;
      b = ((INNER || vyp) && Dn != L) ? Dn : b;
      b = ((INNER || vym) && Up != L) ? Up : b;
      b = ((INNER || vxp) && xp != L) ? xp : b;
      b = ((INNER || vxm) && xm != L) ? xm : b;
      const bool boundary = ok && (b != L);
      const bool border = !INNER && ok && !boundary && nvalid < nfull;
      // reference bin rule (util/image_stats.hxx:24-35) with float-exact thresholds
      int c = 0;
#pragma unroll
      for (int k = 0; k < BINS; ++k) c += (v >= fb[k]) ? 1 : 0;
      const bool inside = (v > lo_f) && (v < hi_f);
      const int edge = (v <= lo_f) ? 0 : nbins - 1;
      const int bin = inside ? c : edge;
      const bool keep = !inside || c < nbins;
      const unsigned long long hinc = (unsigned long long)(keep ? 1u : 0u) << ((bin & 7) * 8);
      const double dv = (double)v;

      // ---- finished runs ----
      const bool fresh = INNER ? (L != rr.k0) : (ok && (!rhas || L != rr.k0));
      const bool diff = boundary && (!phas || b != pr.k1 || L != pr.k0);
      enqueue((dbg & 1) ? false : (INNER ? fresh : (fresh && rhas)), (dbg & 2) ? false : (diff && phas), (uint32_t)(t - 1));
      if (fresh) {
        rr.k0 = L; rr.pos = ((uint32_t)t << 11) | yx; rr.cnt = 0; rr.aux = 0; rr.sum = 0.0; rr.sq = 0.0;
        rr.mn = __builtin_inff(); rr.mx = -__builtin_inff();
#pragma unroll
        for (int k = 0; k < BINS / 8; ++k) rr.h[k] = 0ull;
      }
      if (!INNER) rhas = rhas || fresh;
      if (diff) {
        pr.k0 = L; pr.k1 = b; pr.cnt = 0; pr.aux = 0; pr.sum = 0.0; pr.sq = 0.0;
        pr.mn = __builtin_inff(); pr.mx = -__builtin_inff();
#pragma unroll
        for (int k = 0; k < BINS / 8; ++k) pr.h[k] = 0ull;
      }
      phas = phas || diff;
      // ---- accumulation ----
      if (ok) {
        if (MASK) { rr.cnt += 1; rr.last = (uint32_t)t; }
        if (!INNER) rr.aux += border ? 1u : 0u;
        rr.sum += dv; rr.sq = __builtin_fma(dv, dv, rr.sq);         // dv*dv is exact (24-bit x 24-bit significands)
        rr.mn = vmin(rr.mn, v); rr.mx = vmax(rr.mx, v);
        if (BINS <= 8) rr.h[0] += hinc;
        else { rr.h[0] += (bin < 8) ? hinc : 0ull; rr.h[BINS / 8 - 1] += (bin < 8) ? 0ull : hinc; }
      }
      if (boundary) {
        pr.cnt += 1;
        pr.aux += ((v >= t0) ? 1u : 0u) | ((v >= t1) ? 0x100u : 0u) | ((v >= t2) ? 0x10000u : 0u) | ((v >= t3) ? 0x1000000u : 0u);
        pr.sum += dv; pr.sq = __builtin_fma(dv, dv, pr.sq);
        pr.mn = vmin(pr.mn, v); pr.mx = vmax(pr.mx, v);
        if (BINS <= 8) pr.h[0] += hinc;
        else { pr.h[0] += (bin < 8) ? hinc : 0ull; pr.h[BINS / 8 - 1] += (bin < 8) ? 0ull : hinc; }
      }
      // Lc takes Ln's VALUE through an opaque move: left to coalesce Lc with the load's destination register, the compiler gives the
      // NEXT load of this buffer another register and copies it back at the loop's back edge -- a wait for loads just issued
      Lp = Lc;
      asm("v_mov_b32 %0, %1" : "=v"(Lc) : "v"(Ln));
      if (ISSUE == 1 || (ISSUE == 0 && t + kAhead < n)) issue(t + kAhead, oc0 + (uint32_t)(t + kAhead - tb) * szb, rL, rV, rC, J);
    };

    const uint32_t* baseL = p.lab + rowoff; const float* baseV = p.img + rowoff; const uint32_t* baseC = p.lab_c + rowoff;
    const int64_t stride4 = (int64_t)kAhead * sz;
    // a group of kAhead planes from base plane tb
    auto group = [&](const int tb, auto ISSUE_T) __attribute__((always_inline)) {
      constexpr int ISSUE = decltype(ISSUE_T)::value;
      // (through an empty asm: otherwise the 64-bit products that form the bases are recomputed in every plane)
      asm volatile("" : "+s"(baseL), "+s"(baseV));
      if (MASK) asm volatile("" : "+s"(baseC));
      const rsrc_t rL = make_rsrc(baseL), rV = make_rsrc(baseV), rC = make_rsrc(baseC);
      baseL += stride4; baseV += stride4; baseC += stride4;
      auto steps = [&](auto... J) __attribute__((always_inline)) {
        ((void)((ISSUE != 0 || tb + decltype(J)::value < n) ? (step(tb + decltype(J)::value, tb, rL, rV, rC, J, ISSUE_T), 0) : 0), ...);
      };
      std::apply(steps, kAheadSeq{});
    };
    if (WHOLE) {
      int tb = 0;
#pragma unroll 1
      for (; tb + 2 * kAhead <= n; tb += kAhead) group(tb, std::integral_constant<int, 1>{});
      group(tb, std::integral_constant<int, 2>{});
    } else {
#pragma unroll 1
      for (int tb = 0; tb < n; tb += kAhead) group(tb, std::integral_constant<int, 0>{});
    }
    // the column ends: every lane hands in what it holds
#ifndef GLIA_ACC_WHOLEFLUSH
    // in two halves: all 64 lanes at once need an EMPTY region ring, i.e. a full round trip through the drainer
    enqueue(((dbg & 1) ? false : rhas) && lane < 32, ((dbg & 2) ? false : phas) && lane < 32, (uint32_t)(n - 1));
    enqueue(((dbg & 1) ? false : rhas) && lane >= 32, ((dbg & 2) ? false : phas) && lane >= 32, (uint32_t)(n - 1));
#else
    enqueue((dbg & 1) ? false : rhas, (dbg & 2) ? false : phas, (uint32_t)(n - 1));
#endif
    publish();
    rhas = false; phas = false;
  };

  if (!marcher) {
#ifndef GLIA_ACC_NODRAIN
    drainer_loop<BINS, MASK>(s, wave - G::kMarchers, lane);
#endif
  } else {
    const bool whole = n >= 2 * kAhead && (n & (kAhead - 1)) == 0;
    auto passes = [&](auto INNER_T, auto WHOLE_T) __attribute__((always_inline)) {
#pragma unroll 1
      for (int pass = 0; pass < G::kPasses; ++pass) column(INNER_T, WHOLE_T, wave + pass * G::kMarchers);
    };
    if (inner_tile) { if (whole) passes(std::true_type{}, std::true_type{}); else passes(std::true_type{}, std::false_type{}); }
    else { if (whole) passes(std::false_type{}, std::true_type{}); else passes(std::false_type{}, std::false_type{}); }
    ctl_store(&ctl[C_STATE], kRingDone);
#ifdef GLIA_ACC_COUNTERS
    if (lane == 0) {
      unsigned long long* g = reinterpret_cast<unsigned long long*>(s.tp.flags + 16);
      atomicAdd(&g[8], waitCycles); atomicAdd(&g[9], (unsigned long long)waits); atomicAdd(&g[10], __builtin_readcyclecounter() - tmarch0);
    }
#endif
  }
  __syncthreads();
  if (dbg & 8) return;

  // ---- fold the workgroup's LDS tables into the global tables ----
  // (the table pointers come back from LDS: holding them in scalar registers through the march costs spills there)
  // 1. the used slots into a dense list (the rings are idle now: their space holds it); 2. one probe of the global table per
  // key, all in one round; 3. the records word by word -- consecutive lanes add to DIFFERENT records, i.e. different cache
  // lines: the L2 applies the atomics of one line one after the other.
  const TableParams& gp = s.tp;
  constexpr int kRegSlots = G::kRegSlots, kPairSlots = G::kPairSlots;
  uint32_t* const fold = &s.ringR[0][0];
  uint32_t* const nused = fold;                      // [0]
  unsigned short* const used = reinterpret_cast<unsigned short*>(fold + 4);            // slot ids, regions first come first
  int* const gslot = reinterpret_cast<int*>(fold + 4 + (kRegSlots + kPairSlots) / 2);  // global slot of used[j]
  static_assert(sizeof(s.ringR) >= (4 + (kRegSlots + kPairSlots) / 2 + (kRegSlots + kPairSlots)) * sizeof(uint32_t), "the fold's lists do not fit in the rings");
  if (tid == 0) *nused = 0u;
  __syncthreads();
  for (int i = tid; i < kRegSlots + kPairSlots; i += kThreadsT) {
    const unsigned long long k = i < kRegSlots ? s.rkey[i] : s.pkey[i - kRegSlots];
    if (k) used[atomicAdd(nused, 1u)] = (unsigned short)i;
  }
  __syncthreads();
  const int nu = (int)*nused;
  for (int j = tid; j < nu; j += kThreadsT) {
    const int i = used[j];
    gslot[j] = i < kRegSlots ? global_region_slot(gp, (uint32_t)s.rkey[i]) : global_pair_slot(gp, s.pkey[i - kRegSlots]);
  }
  __syncthreads();
  if (dbg & 64) return;      // (ablation: the global slots are resolved, nothing is added)
  constexpr int RW = Lds<BINS>::kRegWords, PW = Lds<BINS>::kPairWordsL;
  static_assert(RW >= PW, "the word loop runs over the longer record");
  const uint32_t tx = (uint32_t)gp.x0, ty = (uint32_t)gp.y0, tz = (uint32_t)gp.z0;
  uint32_t* const g_rrec = gp.rrec; uint32_t* const g_prec = gp.prec;
  const int64_t fsy = gp.nx, fsz = gp.nx * gp.ny;
#pragma unroll 1
  for (int w = 0; w < RW; ++w) {
    for (int j = tid; j < nu; j += kThreadsT) {
      const int i = used[j], g = gslot[j];
      if (g < 0 || (dbg & 128)) continue;
      if (i < kRegSlots) {
        uint32_t* dst = &g_rrec[(size_t)g * kRegionWords];
        const uint32_t* src = &s.rrec[i * RW];
        const uint32_t val = src[w];
        if (w == LR_SUM || w == LR_SQ) {
          double dd = *reinterpret_cast<const double*>(&src[w]);
          if (dd != 0.0) atomicAdd(reinterpret_cast<double*>(&dst[w == LR_SUM ? R_SUM : R_SQ]), dd);
        } else if (w == LR_CNT) {      // count and border count: adjacent in both records
          atomicAdd(reinterpret_cast<unsigned long long*>(&dst[R_CNT]), (unsigned long long)val | ((unsigned long long)src[LR_BORDER] << 32));
        } else if (w == LR_XMASK) {
          const unsigned long long m = *reinterpret_cast<const unsigned long long*>(&src[LR_XMASK]);
          atomicMax(&dst[R_LO + 0], 0x7fffffffu - (tx + (uint32_t)__builtin_ctzll(m)));
          atomicMax(&dst[R_HI + 0], tx + 64u - (uint32_t)__builtin_clzll(m));
        } else if (w == LR_YMASK) {
          atomicMax(&dst[R_LO + 1], 0x7fffffffu - (ty + (uint32_t)__builtin_ctz(val)));
          atomicMax(&dst[R_HI + 1], ty + 32u - (uint32_t)__builtin_clz(val));
        } else if (w == LR_ZMASK) {
          atomicMax(&dst[R_LO + 2], 0x7fffffffu - (tz + (uint32_t)__builtin_ctz(val)));
          atomicMax(&dst[R_HI + 2], tz + 32u - (uint32_t)__builtin_clz(val));
        } else if (w == LR_MIN) { atomicMax(&dst[R_MIN], val); }
        else if (w == LR_MAX) { atomicMax(&dst[R_MAX], val); }
        else if (w == LR_FIRST) {
          const uint32_t first = 0xFFFFFu - val;
          const unsigned long long fidx = (unsigned long long)(((int64_t)tz + (first >> 11)) * fsz + ((int64_t)ty + ((first >> 6) & 31)) * fsy +
                                                               ((int64_t)tx + (first & 63)));
          atomicMax(reinterpret_cast<unsigned long long*>(&dst[R_FIRST]), ~fidx);
        } else if (w >= LR_HIST) {     // two 16-bit counters -> two adjacent 32-bit counters of the record, one 64-bit add
          const int k = 2 * (w - LR_HIST);
          if (val) atomicAdd(reinterpret_cast<unsigned long long*>(&dst[R_HIST + k]), (unsigned long long)(val & 0xFFFFu) | ((unsigned long long)(val >> 16) << 32));
        }
      } else if (w < PW) {
        uint32_t* dst = &g_prec[(size_t)g * kPairWords];
        const uint32_t* src = &s.prec[(i - kRegSlots) * PW];
        const uint32_t val = src[w];
        if (w == LP_SUM || w == LP_SQ) {
          double dd = *reinterpret_cast<const double*>(&src[w]);
          if (dd != 0.0) atomicAdd(reinterpret_cast<double*>(&dst[w == LP_SUM ? P_SUM : P_SQ]), dd);
        } else if (w == LP_CNT) { atomicAdd(&dst[P_CNT], val); }
        else if (w == LP_MIN) { atomicMax(&dst[P_MIN], val); }
        else if (w == LP_MAX) { atomicMax(&dst[P_MAX], val); }
        else if (w == LP_THR || w == LP_THR + 1) {
          const int k = 2 * (w - LP_THR);
          if (val) atomicAdd(reinterpret_cast<unsigned long long*>(&dst[P_THR + k]), (unsigned long long)(val & 0xFFFFu) | ((unsigned long long)(val >> 16) << 32));
        } else if (w >= LP_HIST) {
          const int k = 2 * (w - LP_HIST);
          if (val) atomicAdd(reinterpret_cast<unsigned long long*>(&dst[P_HIST + k]), (unsigned long long)(val & 0xFFFFu) | ((unsigned long long)(val >> 16) << 32));
        }
      }
    }
  }
}

}  // namespace

template <int BINS>
static void launch_b(const AccParams& p, uint32_t nb, hipStream_t stream) {
  if (p.masked) hipLaunchKernelGGL((rag_accumulate_kernel<BINS, true>), dim3(nb), dim3(Geo<BINS>::kThreadsT), 0, stream, p);
  else hipLaunchKernelGGL((rag_accumulate_kernel<BINS, false>), dim3(nb), dim3(Geo<BINS>::kThreadsT), 0, stream, p);
}

int launch_accumulate(const AccParams& p, hipStream_t stream) {
  const uint32_t nb = (uint32_t)p.nbx * p.nby * p.nbz;
  if (p.hist.bins <= 8) launch_b<8>(p, nb, stream); else launch_b<16>(p, nb, stream);
  GLIA_HIP_TRY(hipGetLastError());
  return GLIA_HMT_OK;
}

namespace {
__global__ void mask_fold_kernel(const uint32_t* lab, const uint32_t* mask, uint32_t* out, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = mask[i] != 0u /* MASK_OUT_VAL, glia_image.hxx:28 */ ? lab[i] : kMaskedLabel;
}
}  // namespace

int launch_mask_fold(const uint32_t* lab, const uint32_t* mask, uint32_t* out, int64_t n, hipStream_t stream) {
  hipLaunchKernelGGL(mask_fold_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, lab, mask, out, (long long)n);
  GLIA_HIP_TRY(hipGetLastError());
  return GLIA_HMT_OK;
}

}  // namespace glia
