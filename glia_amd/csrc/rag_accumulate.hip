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
// MI355X mapping.  HBM-bound: 8 algorithmic bytes per voxel (4 label + 4 image).
//   * Tile = 64(x) x 32(y) x 32(z) voxels per 512-thread workgroup; a wave covers 4 rows of 64 voxels
//     (16 lanes x one 16-byte load per row) and marches the 32 planes, keeping the z-1 / z / z+1 label rows
//     in registers; y+-1 rows are re-read through L1/L2.  Tiles are dealt to XCDs in contiguous runs so halo
//     rows of neighbouring tiles hit the same L2.
//   * Supervoxels are spatially coherent, so a lane sees long RUNS of one label / one directed pair.  It
//     reduces a run in registers (f64 sums, 8-bit packed histogram / threshold counters; planes are walked
//     in serpentine x order so a lane that straddles a wall changes key once per plane, not twice).
//   * A finished run is not flushed by its (single, divergent) lane: it is appended to a per-wave LDS ring
//     (three 16-byte stores).  When the ring cannot take the next batch, the whole wave DRAINS it -- lane j
//     owns entry j -- into the workgroup's LDS hash tables (find-or-insert + LDS atomics) at near-full lane
//     utilisation.  After the march the workgroup folds its LDS tables into the global hash tables.
//   * All reductions are integer adds, unsigned max, or f64 adds: exact, hence order-independent and
//     bit-reproducible, whenever the image is a multiple of 2^-k (Q8 pb); otherwise within ~1e-15 relative.
#include <type_traits>

#include "hmt_internal.hpp"

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

// ---- LDS layout ------------------------------------------------------------------------------------
// region slot (words): 0 cnt | 1 border | 2..7 bbox (tile-relative, max-encoded: 63-xlo, xhi+1, 31-ylo, yhi+1,
// 31-zlo, zhi+1; 63/31 stand for kTileX-1 / kTileY-1) | 8 sum(f64) | 10 sq(f64) | 12 ~ord(min) | 13 ord(max) | 14 0xFFFFF - first(tile-rel) | 16.. hist
// pair slot (words):   0 cnt | 1 ~ord(min) | 2 ord(max) | 4..7 thr | 8 sum | 10 sq | 12.. hist
constexpr int LR_CNT = 0, LR_BORDER = 1, LR_BOX = 2, LR_SUM = 8, LR_SQ = 10, LR_MIN = 12, LR_MAX = 13, LR_FIRST = 14,
              LR_HIST = 16;
constexpr int LP_CNT = 0, LP_MIN = 1, LP_MAX = 2, LP_THR = 4, LP_SUM = 8, LP_SQ = 10, LP_HIST = 12;
constexpr int kXB = kTileX == 64 ? 6 : 8;      // bits of a tile-relative x
constexpr int kYB = 11 - kXB;                  // bits of a tile-relative y (kTileX * kTileY == 2048)
static_assert((1 << kXB) == kTileX && (1 << kYB) == kTileY && kTZ == 32, "tile-relative packing");
constexpr uint32_t kXM = kTileX - 1, kYM = kTileY - 1;
// LDS table sizes (one workgroup per CU owns the whole 160 KiB): 8-bin records allow 1024 pair slots, 16-bin 768
template <int BINS> struct Slots { static constexpr int kReg = 256; static constexpr int kPair = BINS <= 8 ? 1024 : 768; };
constexpr int kRingEntries = 64;
constexpr int kLdsProbes = 32;
constexpr int kGlobalProbes = 512;

// what drain_ring needs from the kernel arguments; copied to LDS once so that the non-inlined drain never forces
// the by-value kernel argument onto the stack
struct TableParams {
  uint32_t* rkeys; uint32_t* rrec; unsigned long long* pkeys; uint32_t* prec; uint32_t* flags;
  uint32_t rmask, pmask;
  int64_t nx, ny;
  int64_t x0, y0, z0;     // tile origin
};

template <int BINS>
struct Lds {
  static constexpr int kRegWords = LR_HIST + BINS;        // 24 / 32
  static constexpr int kPairWordsL = LP_HIST + BINS;      // 20 / 28
  static constexpr int kEntryWords = BINS <= 8 ? 12 : 16; // ring entry
  static constexpr int kRegSlots = Slots<BINS>::kReg, kPairSlots = Slots<BINS>::kPair;
  unsigned long long rkey[kRegSlots];
  unsigned long long pkey[kPairSlots];
  uint32_t rrec[kRegSlots * kRegWords];
  uint32_t prec[kPairSlots * kPairWordsL];
  uint32_t ring[kTileWaves][kRingEntries * kEntryWords];   // reused as gslot[] by the final fold
  TableParams tp;
};

__device__ __forceinline__ int lds_slot(unsigned long long* keys, int nslots, unsigned long long key) {
  uint32_t h = (uint32_t)(((unsigned long long)hash64(key) * (unsigned long long)nslots) >> 32);
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
//  2 region: cnt | border<<8 | first_rel<<16     pair: cnt
//  3 region: (63-xlo) | xhi<<6 | yrel<<12 | zhi<<17   pair: 4 x 8-bit threshold counters
//  4,5 sum (f64)   6,7 sq (f64)   8 ord(min)   9 ord(max)   10,11 hist bins 0-7 (8-bit packed)  [12,13 bins 8-15]
struct Tile { int64_t x0, y0, z0; };

// Deliberately NOT inlined: it is called from every enqueue site but runs rarely; inlining sixteen copies
// blows the register budget of the streaming loop.
template <int BINS>
__device__ __attribute__((noinline)) void drain_ring(Lds<BINS>& s, uint32_t* ring, int count, int lane) {
  constexpr int EW = Lds<BINS>::kEntryWords;
  if (lane >= count) return;
  const TableParams& p = s.tp;
  Tile t; t.x0 = p.x0; t.y0 = p.y0; t.z0 = p.z0;
  const uint4* e = reinterpret_cast<const uint4*>(ring + lane * EW);
  const uint4 a = e[0], b = e[1], c = e[2];
  uint4 d = {0, 0, 0, 0};
  if (BINS > 8) d = e[3];
  const bool isRegion = a.y == 0;
  const unsigned long long key = ((unsigned long long)a.y << 32) | a.x;
  const uint32_t cnt = a.z & 0xFF;
  double sum = __hiloint2double((int)b.y, (int)b.x), sq = __hiloint2double((int)b.w, (int)b.z);
  const uint32_t omin = ~c.x, omax = c.y;
  int slot = isRegion ? lds_slot(s.rkey, Lds<BINS>::kRegSlots, key) : lds_slot(s.pkey, Lds<BINS>::kPairSlots, key);
#ifdef GLIA_HMT_PROFILE
  atomicAdd(&p.flags[isRegion ? 2 : 3], 1u); if (slot < 0) atomicAdd(&p.flags[isRegion ? 4 : 5], 1u);
#endif
  if (slot >= 0) {
    uint32_t* rec = isRegion ? &s.rrec[slot * Lds<BINS>::kRegWords] : &s.prec[slot * Lds<BINS>::kPairWordsL];
    const int oCnt = isRegion ? LR_CNT : LP_CNT, oSum = isRegion ? LR_SUM : LP_SUM, oSq = isRegion ? LR_SQ : LP_SQ;
    const int oMin = isRegion ? LR_MIN : LP_MIN, oMax = isRegion ? LR_MAX : LP_MAX, oHist = isRegion ? LR_HIST : LP_HIST;
    atomicAdd(&rec[oCnt], cnt);
    atomicAdd((double*)&rec[oSum], sum);
    atomicAdd((double*)&rec[oSq], sq);
    atomicMax(&rec[oMin], omin);
    atomicMax(&rec[oMax], omax);
#pragma unroll
    for (int k = 0; k < BINS; ++k) {
      uint32_t w = (k < 4) ? c.z : (k < 8) ? c.w : (k < 12) ? d.x : d.y;
      uint32_t h = (w >> ((k & 3) * 8)) & 0xFF;
      if (h) atomicAdd(&rec[oHist + k], h);
    }
    if (isRegion) {
      const uint32_t border = (a.z >> 8) & 0xFF, first = a.z >> 16;
      const uint32_t xlo_c = a.w & kXM, xhi = (a.w >> kXB) & kXM, yrel = (a.w >> (2 * kXB)) & kYM, zhi = (a.w >> (2 * kXB + kYB)) & 31;
      const uint32_t zlo = first >> 11;
      if (border) atomicAdd(&rec[LR_BORDER], border);
      atomicMax(&rec[LR_BOX + 0], xlo_c);
      atomicMax(&rec[LR_BOX + 1], xhi + 1);
      atomicMax(&rec[LR_BOX + 2], kYM - yrel);
      atomicMax(&rec[LR_BOX + 3], yrel + 1);
      atomicMax(&rec[LR_BOX + 4], 31 - zlo);
      atomicMax(&rec[LR_BOX + 5], zhi + 1);
      atomicMax(&rec[LR_FIRST], 0xFFFFFu - first);
    } else {
#pragma unroll
      for (int k = 0; k < GLIA_HMT_MAX_THRESH; ++k) {
        uint32_t h = (a.w >> (8 * k)) & 0xFF;
        if (h) atomicAdd(&rec[LP_THR + k], h);
      }
    }
    return;
  }
  // LDS table saturated (tile with very many tiny supervoxels): straight to the global tables (slow, exact).  The host
  // watches the count and gives the next pass shallower tiles.
  atomicAdd(&p.flags[7], 1u);
  if (isRegion) {
    int g = global_region_slot(p, a.x);
    if (g < 0) return;
    uint32_t* r = &p.rrec[(size_t)g * kRegionWords];
    const uint32_t border = (a.z >> 8) & 0xFF, first = a.z >> 16;
    const uint32_t xlo = kXM - (a.w & kXM), xhi = (a.w >> kXB) & kXM, yrel = (a.w >> (2 * kXB)) & kYM, zhi = (a.w >> (2 * kXB + kYB)) & 31;
    const uint32_t zlo = first >> 11, fy = (first >> kXB) & kYM, fx = first & kXM;
    atomicAdd(&r[R_CNT], cnt);
    if (border) atomicAdd(&r[R_BORDER], border);
    atomicAdd((double*)&r[R_SUM], sum); atomicAdd((double*)&r[R_SQ], sq);
    atomicMax(&r[R_MIN], omin); atomicMax(&r[R_MAX], omax);
    atomicMax(&r[R_LO + 0], 0x7fffffffu - (uint32_t)(t.x0 + xlo)); atomicMax(&r[R_HI + 0], (uint32_t)(t.x0 + xhi) + 1u);
    atomicMax(&r[R_LO + 1], 0x7fffffffu - (uint32_t)(t.y0 + yrel)); atomicMax(&r[R_HI + 1], (uint32_t)(t.y0 + yrel) + 1u);
    atomicMax(&r[R_LO + 2], 0x7fffffffu - (uint32_t)(t.z0 + zlo)); atomicMax(&r[R_HI + 2], (uint32_t)(t.z0 + zhi) + 1u);
    unsigned long long fidx = (unsigned long long)((t.z0 + zlo) * p.ny * p.nx + (t.y0 + fy) * p.nx + (t.x0 + fx));
    atomicMax((unsigned long long*)&r[R_FIRST], ~fidx);
#pragma unroll
    for (int k = 0; k < BINS; ++k) {
      uint32_t w = (k < 4) ? c.z : (k < 8) ? c.w : (k < 12) ? d.x : d.y;
      uint32_t h = (w >> ((k & 3) * 8)) & 0xFF;
      if (h) atomicAdd(&r[R_HIST + k], h);
    }
  } else {
    int g = global_pair_slot(p, key);
    if (g < 0) return;
    uint32_t* r = &p.prec[(size_t)g * kPairWords];
    atomicAdd(&r[P_CNT], cnt);
    atomicAdd((double*)&r[P_SUM], sum); atomicAdd((double*)&r[P_SQ], sq);
    atomicMax(&r[P_MIN], omin); atomicMax(&r[P_MAX], omax);
#pragma unroll
    for (int k = 0; k < GLIA_HMT_MAX_THRESH; ++k) {
      uint32_t h = (a.w >> (8 * k)) & 0xFF;
      if (h) atomicAdd(&r[P_THR + k], h);
    }
#pragma unroll
    for (int k = 0; k < BINS; ++k) {
      uint32_t w = (k < 4) ? c.z : (k < 8) ? c.w : (k < 12) ? d.x : d.y;
      uint32_t h = (w >> ((k & 3) * 8)) & 0xFF;
      if (h) atomicAdd(&r[P_HIST + k], h);
    }
  }
}

// per-lane run accumulators (registers)
struct Run {
  uint32_t klo, khi;        // key; klo == 0 -> empty
  uint32_t cnt;             // voxels
  uint32_t aux;             // region: border count; pair: 4 x 8-bit threshold counters
  double sum, sq;
  uint32_t omin, omax;      // order-preserving uint images of the float min / max
  uint32_t h0, h1, h2, h3;  // 8-bit packed histogram counters
  uint32_t first;           // region: tile-relative index of the first voxel
  uint32_t last;            // region: tile-relative index of the most recent voxel (its plane = zhi)
  uint32_t xbox;            // region: (63-xlo) | xhi << 16, packed-max
};

struct U4 { uint32_t v[4]; };
struct F4 { float v[4]; };
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));

template <int BINS, bool VEC, bool MASK>
__global__ __launch_bounds__(kThreads, 2) void rag_accumulate_kernel(const AccParams p) {
  __shared__ __attribute__((aligned(16))) Lds<BINS> s;
  constexpr int EW = Lds<BINS>::kEntryWords;
  const int tid = threadIdx.x;
  {
    uint4* w = reinterpret_cast<uint4*>(&s);
    const uint4 z4 = {0, 0, 0, 0};
    // keys + records only; the rings need no initialisation
    constexpr int n16 = (int)((sizeof(s.rkey) + sizeof(s.pkey) + sizeof(s.rrec) + sizeof(s.prec)) / 16);
    for (int i = tid; i < n16; i += kThreads) w[i] = z4;
  }
  __syncthreads();

  // tile coordinates: blocks that share blockIdx % 8 share an XCD (L2); give each XCD a contiguous
  // run of tiles so y/z halo rows of neighbouring tiles are served by the same L2.
  const uint32_t nb = (uint32_t)p.nbx * p.nby * p.nbz;
  uint32_t bid = blockIdx.x;
  {
    const uint32_t q = nb / 8, r = nb % 8, xcd = bid % 8, k = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int bx = bid % p.nbx, by = (bid / p.nbx) % p.nby, bz = bid / (p.nbx * p.nby);
  const int lane = tid & 63, wave = tid >> 6;
  const int64_t nx = p.nx, ny = p.ny, nz = p.nz;
  Tile tile;
  tile.x0 = (int64_t)bx * kTileX; tile.y0 = (int64_t)by * kTileY; tile.z0 = p.zb + (int64_t)bz * p.tz;
  const int xrel0 = (lane % kLanesPerRow) * kVX;
  const int yrel = wave * kRowsPerWave + (lane / kLanesPerRow);
  const int64_t x0 = tile.x0 + xrel0;
  const int64_t y = tile.y0 + yrel;
  const int64_t z1 = (tile.z0 + p.tz < p.ze) ? tile.z0 + p.tz : p.ze;
  const int64_t gz0 = p.gz0, gnz = p.gnz;
  const bool rowOk = (y < ny) && (x0 < nx);
  const int64_t sy = nx, sz = nx * ny;
  const bool is3d = p.dim == 3;
  const int nfull = 2 * p.dim;
  uint32_t* ring = s.ring[wave];
  int ringCount = 0;   // wave-uniform
#ifdef GLIA_HMT_PROFILE
  const uint32_t dbg = p.debug;       // ablation switches (GLIA_HMT_DEBUG), profiling builds only
#else
  constexpr uint32_t dbg = 0;
#endif
  if (tid == 0) {
    TableParams tp;
    tp.rkeys = p.rkeys; tp.rrec = p.rrec; tp.pkeys = p.pkeys; tp.prec = p.prec; tp.flags = p.flags;
    tp.rmask = p.rmask; tp.pmask = p.pmask; tp.nx = p.nx; tp.ny = p.ny;
    tp.x0 = tile.x0; tp.y0 = tile.y0; tp.z0 = tile.z0 + p.gz0;     // global coordinates for bbox / first-voxel index
    s.tp = tp;
  }
  __syncthreads();

  auto loadLabFrom = [&](const uint32_t* base, int64_t yy, int64_t zz, bool ok) __attribute__((always_inline)) -> U4 {
    U4 r;
    r.v[0] = r.v[1] = r.v[2] = r.v[3] = 0;
    if (ok && rowOk) {
      const uint32_t* q = base + zz * sz + yy * sy + x0;
      if (VEC) {
        uint4 t = *reinterpret_cast<const uint4*>(q);
        r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
      } else {
#pragma unroll
        for (int i = 0; i < kVX; ++i) if (x0 + i < nx) r.v[i] = q[i];
      }
    }
    return r;
  };
  auto loadLab = [&](int64_t yy, int64_t zz, bool ok) __attribute__((always_inline)) -> U4 { return loadLabFrom(p.lab, yy, zz, ok); };
  const bool sepCentre = MASK && p.lab_c != p.lab;     // contour-only mode: the centre keeps its label under the mask
  auto loadImg = [&](int64_t yy, int64_t zz) __attribute__((always_inline)) -> F4 {
    F4 r;
    r.v[0] = r.v[1] = r.v[2] = r.v[3] = 0.f;
    if (rowOk) {
      const float* q = p.img + zz * sz + yy * sy + x0;
      if (VEC) {
        float4 t = *reinterpret_cast<const float4*>(q);
        r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
      } else {
#pragma unroll
        for (int i = 0; i < kVX; ++i) if (x0 + i < nx) r.v[i] = q[i];
      }
    }
    return r;
  };

  Run rr, pr;
  rr.klo = rr.khi = 0; rr.cnt = 0; rr.aux = 0; rr.sum = 0.0; rr.sq = 0.0; rr.omin = 0xFFFFFFFFu; rr.omax = 0;
  rr.h0 = rr.h1 = rr.h2 = rr.h3 = 0; rr.first = 0; rr.last = 0; rr.xbox = 0;
  pr = rr;

  // append finished runs of the lanes in `ev` to the wave's ring (draining it first when it cannot take them)
  auto enqueue = [&](bool ev, const Run& r, bool isRegion) __attribute__((always_inline)) {
    const unsigned long long m = __ballot(ev);
    if (m == 0) return;
    const int n = __popcll(m);
    if (ringCount + n > kRingEntries) {
      drain_ring<BINS>(s, ring, ringCount, lane);
      ringCount = 0;
    }
    if (ev) {
      const int pos = ringCount + (int)__popcll(m & ((1ull << lane) - 1ull));
      uint4* e = reinterpret_cast<uint4*>(ring + pos * EW);
      uint4 a, b, c;
      a.x = r.klo; a.y = r.khi;
      if (isRegion) {
        a.z = r.cnt | (r.aux << 8) | (r.first << 16);
        a.w = (r.xbox & kXM) | ((r.xbox >> 16) << kXB) | ((uint32_t)yrel << (2 * kXB)) | ((r.last >> 11) << (2 * kXB + kYB));
      } else { a.z = r.cnt; a.w = r.aux; }
      b.x = (uint32_t)__double2loint(r.sum); b.y = (uint32_t)__double2hiint(r.sum);
      b.z = (uint32_t)__double2loint(r.sq); b.w = (uint32_t)__double2hiint(r.sq);
      c.x = r.omin; c.y = r.omax; c.z = r.h0; c.w = r.h1;
      e[0] = a; e[1] = b; e[2] = c;
      if (BINS > 8) { uint4 d; d.x = r.h2; d.y = r.h3; d.z = 0; d.w = 0; e[3] = d; }
    }
    ringCount += n;
  };

  // thresholds to registers
  float fb[BINS];
#pragma unroll
  for (int k = 0; k < BINS; ++k) fb[k] = p.hist.fb[k];
  const float lo_f = p.hist.lo_f, hi_f = p.hist.hi_f;
  const int nbins = p.hist.bins;
  const float t0 = p.thr_f[0], t1 = p.thr_f[1], t2 = p.thr_f[2], t3 = p.thr_f[3];   // +inf beyond nthr

  // one voxel: neighbour rule, run bookkeeping, accumulation
  // INNER (a workgroup-uniform compile-time tag): the tile and its one-voxel halo lie inside the volume and there is no mask,
  // so every voxel is valid, every neighbour exists and no voxel is a border voxel -- the validity logic (six flags, their
  // count, the border test) folds away: ~25 of ~220 vector instructions per voxel.
  auto voxel = [&](auto INNER_T, auto I, const U4& Lp, const U4& Lc, const U4& Cc, const U4& Ln, const U4& Up, const U4& Dn, const F4& V,
                   uint32_t left, uint32_t right, int zrel, bool zmv, bool zpv, bool ymv, bool ypv)
                   __attribute__((always_inline)) {
    constexpr int i = decltype(I)::value;
    constexpr bool INNER = decltype(INNER_T)::value;
    const int64_t x = x0 + i;
    const uint32_t L = MASK ? Cc.v[i] : Lc.v[i];
    const bool ok = INNER || (rowOk && (VEC || x < nx) && !(dbg & 4) && (!MASK || L != kMaskedLabel));
    const uint32_t xm = (i == 0) ? left : Lc.v[i > 0 ? i - 1 : 0];
    const uint32_t xp = (i == kVX - 1) ? right : Lc.v[i < kVX - 1 ? i + 1 : kVX - 1];
    bool xmv = INNER || x > 0, xpv = INNER || x + 1 < nx;
    if (INNER) { zmv = zpv = ymv = ypv = true; }
    if (MASK) {
      xmv = xmv && xm != kMaskedLabel; xpv = xpv && xp != kMaskedLabel;
      ymv = ymv && Up.v[i] != kMaskedLabel; ypv = ypv && Dn.v[i] != kMaskedLabel;
      zmv = zmv && Lp.v[i] != kMaskedLabel; zpv = zpv && Ln.v[i] != kMaskedLabel;
    }
    uint32_t b = L;
    b = (zpv && Ln.v[i] != L) ? Ln.v[i] : b;
    b = (zmv && Lp.v[i] != L) ? Lp.v[i] : b;
    b = (ypv && Dn.v[i] != L) ? Dn.v[i] : b;
    b = (ymv && Up.v[i] != L) ? Up.v[i] : b;
    b = (xpv && xp != L) ? xp : b;
    b = (xmv && xm != L) ? xm : b;
    const int nvalid = INNER ? 0 : (int)xmv + (int)xpv + (int)ymv + (int)ypv + (int)zmv + (int)zpv;
    const bool boundary = ok && (b != L);
    const bool border = !INNER && ok && !boundary && nvalid < nfull;
    const float v = V.v[i];
    // reference bin rule (util/image_stats.hxx:24-35) with float-exact thresholds
    int c = 0;
#pragma unroll
    for (int k = 0; k < BINS; ++k) c += (v >= fb[k]) ? 1 : 0;
    const bool inside = (v > lo_f) && (v < hi_f);
    const int bin = inside ? c : ((v <= lo_f) ? 0 : nbins - 1);
    const bool drop = inside && c >= nbins;
    const uint32_t hinc = drop ? 0u : (1u << ((bin & 3) * 8));
    const int hw = bin >> 2;
    const double dv = (double)v;
    const double dv2 = dv * dv;                       // exact: 24-bit x 24-bit significands
    const uint32_t ov = float_ord(v);

    // ---- region run ----
    {
      const uint32_t rkey = L + 1u;
      const bool fresh = ok && (rkey != rr.klo);
      const bool ev = fresh && rr.klo != 0;
      if (!(dbg & 1)) enqueue(ev, rr, true);
      const uint32_t xr = (uint32_t)(xrel0 + i);
      const uint32_t rel = ((uint32_t)zrel << 11) | ((uint32_t)yrel << kXB) | xr;
      const uint32_t xb = (kXM - xr) | (xr << 16);
      if (fresh) {
        rr.klo = rkey; rr.cnt = 0; rr.aux = 0; rr.sum = 0.0; rr.sq = 0.0; rr.omin = 0xFFFFFFFFu; rr.omax = 0;
        rr.h0 = rr.h1 = rr.h2 = rr.h3 = 0; rr.first = rel; rr.xbox = xb;
      }
      if (ok) {
        rr.cnt += 1; rr.aux += border ? 1u : 0u;
        rr.sum += dv; rr.sq += dv2;
        rr.omin = min(rr.omin, ov); rr.omax = max(rr.omax, ov);
        rr.h0 += (hw == 0) ? hinc : 0u; rr.h1 += (hw == 1) ? hinc : 0u;
        if (BINS > 8) { rr.h2 += (hw == 2) ? hinc : 0u; rr.h3 += (hw == 3) ? hinc : 0u; }
        rr.first = min(rr.first, rel); rr.last = rel;
        rr.xbox = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(ushort2_t, rr.xbox),
                                                                        __builtin_bit_cast(ushort2_t, xb)));
      }
    }
    // ---- directed pair run ----
    {
      const uint32_t klo = b + 1u, khi = L + 1u;
      const bool diff = boundary && (klo != pr.klo || khi != pr.khi);
      const bool ev = diff && pr.klo != 0;
      if (!(dbg & 2)) enqueue(ev, pr, false);
      if (diff) {
        pr.klo = klo; pr.khi = khi; pr.cnt = 0; pr.aux = 0; pr.sum = 0.0; pr.sq = 0.0; pr.omin = 0xFFFFFFFFu; pr.omax = 0;
        pr.h0 = pr.h1 = pr.h2 = pr.h3 = 0;
      }
      if (boundary) {
        pr.cnt += 1;
        pr.aux += ((v >= t0) ? 1u : 0u) | ((v >= t1) ? 0x100u : 0u) | ((v >= t2) ? 0x10000u : 0u) | ((v >= t3) ? 0x1000000u : 0u);
        pr.sum += dv; pr.sq += dv2;
        pr.omin = min(pr.omin, ov); pr.omax = max(pr.omax, ov);
        pr.h0 += (hw == 0) ? hinc : 0u; pr.h1 += (hw == 1) ? hinc : 0u;
        if (BINS > 8) { pr.h2 += (hw == 2) ? hinc : 0u; pr.h3 += (hw == 3) ? hinc : 0u; }
      }
    }
  };

  // software pipeline: the rows of plane z+1 (and the centre row of z+2) are requested before plane z is
  // processed -- with one workgroup per CU (8 waves) memory latency is not hidden by other waves alone
  auto halo = [&](int64_t z, uint32_t& left, uint32_t& right) __attribute__((always_inline)) {
    left = 0u; right = 0u;
    if ((lane % kLanesPerRow) == 0) left = (rowOk && x0 > 0 && z < z1) ? p.lab[z * sz + y * sy + x0 - 1] : 0u;
    if ((lane % kLanesPerRow) == kLanesPerRow - 1) right = (rowOk && x0 + kVX < nx && z < z1) ? p.lab[z * sz + y * sy + x0 + kVX] : 0u;
  };
  U4 Lp = loadLab(y, tile.z0 - 1, tile.z0 > 0);
  U4 Lc = loadLab(y, tile.z0, true);
  U4 Ln = loadLab(y, tile.z0 + 1, tile.z0 + 1 < nz);
  U4 Up = loadLab(y - 1, tile.z0, y > 0);
  U4 Dn = loadLab(y + 1, tile.z0, y + 1 < ny);
  F4 V = loadImg(y, tile.z0);
  uint32_t hl, hr;
  halo(tile.z0, hl, hr);
  // is this an inner tile?  (full tile, not masked, 3D, one voxel away from every face of the -- global -- volume)
  const bool inner_tile = !MASK && VEC && is3d && dbg == 0 && tile.x0 > 0 && tile.x0 + kTileX < nx && tile.y0 > 0 && tile.y0 + kTileY < ny &&
                          tile.z0 + gz0 > 0 && z1 + gz0 < gnz && tile.z0 > 0 && z1 < nz;
  auto march = [&](auto INNER_T) __attribute__((always_inline)) {
  for (int64_t z = tile.z0; z < z1; ++z) {
    const int zrel = (int)(z - tile.z0);
    // requests for the next plane
    const bool more = z + 1 < z1;
    U4 Ln2 = loadLab(y, z + 2, more && z + 2 < nz);
    U4 Up2 = loadLab(y - 1, z + 1, more && y > 0);
    U4 Dn2 = loadLab(y + 1, z + 1, more && y + 1 < ny);
    F4 V2 = V;
    if (more) V2 = loadImg(y, z + 1);
    uint32_t hl2, hr2;
    halo(more ? z + 1 : z1, hl2, hr2);
    uint32_t left = __shfl_up(Lc.v[3], 1, kLanesPerRow);
    uint32_t right = __shfl_down(Lc.v[0], 1, kLanesPerRow);
    if ((lane % kLanesPerRow) == 0) left = hl;
    if ((lane % kLanesPerRow) == kLanesPerRow - 1) right = hr;
    const bool zmv = is3d && (z + gz0) > 0, zpv = is3d && (z + gz0) + 1 < gnz;
    const bool ymv = y > 0, ypv = y + 1 < ny;
    U4 Cc = Lc;
    if (MASK && sepCentre) Cc = loadLabFrom(p.lab_c, y, z, true);
    // a run lasts at most kTZ planes x 4 voxels = 128 voxels, so the 8-bit packed counters cannot overflow.
    // Serpentine x order: a lane that straddles a wall changes key once per plane instead of twice.
    if ((zrel & 1) == 0) {
      voxel(INNER_T, std::integral_constant<int, 0>{}, Lp, Lc, Cc, Ln, Up, Dn, V, left, right, zrel, zmv, zpv, ymv, ypv);
      voxel(INNER_T, std::integral_constant<int, 1>{}, Lp, Lc, Cc, Ln, Up, Dn, V, left, right, zrel, zmv, zpv, ymv, ypv);
      voxel(INNER_T, std::integral_constant<int, 2>{}, Lp, Lc, Cc, Ln, Up, Dn, V, left, right, zrel, zmv, zpv, ymv, ypv);
      voxel(INNER_T, std::integral_constant<int, 3>{}, Lp, Lc, Cc, Ln, Up, Dn, V, left, right, zrel, zmv, zpv, ymv, ypv);
    } else {
      voxel(INNER_T, std::integral_constant<int, 3>{}, Lp, Lc, Cc, Ln, Up, Dn, V, left, right, zrel, zmv, zpv, ymv, ypv);
      voxel(INNER_T, std::integral_constant<int, 2>{}, Lp, Lc, Cc, Ln, Up, Dn, V, left, right, zrel, zmv, zpv, ymv, ypv);
      voxel(INNER_T, std::integral_constant<int, 1>{}, Lp, Lc, Cc, Ln, Up, Dn, V, left, right, zrel, zmv, zpv, ymv, ypv);
      voxel(INNER_T, std::integral_constant<int, 0>{}, Lp, Lc, Cc, Ln, Up, Dn, V, left, right, zrel, zmv, zpv, ymv, ypv);
    }
    Lp = Lc; Lc = Ln; Ln = Ln2; Up = Up2; Dn = Dn2; V = V2; hl = hl2; hr = hr2;
  }
  };
  if (inner_tile) march(std::true_type{}); else march(std::false_type{});
  {
    if (!(dbg & 1)) enqueue(rr.klo != 0, rr, true);
    if (!(dbg & 2)) enqueue(pr.klo != 0, pr, false);
    drain_ring<BINS>(s, ring, ringCount, lane);
  }
  __syncthreads();
  if (dbg & 8) return;

  // ---- fold the workgroup's LDS tables into the global tables ----
  constexpr int kRegSlots = Lds<BINS>::kRegSlots, kPairSlots = Lds<BINS>::kPairSlots;
  int* gslot = reinterpret_cast<int*>(&s.ring[0][0]);      // rings are idle now
  static_assert(sizeof(s.ring) >= (kRegSlots + kPairSlots) * sizeof(int), "gslot does not fit in the rings");
  for (int i = tid; i < kRegSlots + kPairSlots; i += kThreads) {
    int g = -1;
    if (i < kRegSlots) { unsigned long long k = s.rkey[i]; if (k) g = global_region_slot(p, (uint32_t)k); }
    else { unsigned long long k = s.pkey[i - kRegSlots]; if (k) g = global_pair_slot(p, k); }
    gslot[i] = g;
  }
  __syncthreads();
  constexpr int RW = Lds<BINS>::kRegWords, PW = Lds<BINS>::kPairWordsL;
  const uint32_t tx = (uint32_t)tile.x0, ty = (uint32_t)tile.y0, tz = (uint32_t)(tile.z0 + gz0);
  for (int it = tid; it < kRegSlots * RW; it += kThreads) {
    const int slot = it / RW, w = it % RW;
    const int g = gslot[slot];
    if (g < 0) continue;
    uint32_t* dst = &p.rrec[(size_t)g * kRegionWords];
    const uint32_t* src = &s.rrec[slot * RW];
    const uint32_t val = src[w];
    if (w == LR_SUM || w == LR_SQ) {
      double dd = *reinterpret_cast<const double*>(&src[w]);
      if (dd != 0.0) atomicAdd(reinterpret_cast<double*>(&dst[w == LR_SUM ? R_SUM : R_SQ]), dd);
    } else if (w == LR_SUM + 1 || w == LR_SQ + 1 || w == 15) {
    } else if (w == LR_CNT) { atomicAdd(&dst[R_CNT], val); }
    else if (w == LR_BORDER) { if (val) atomicAdd(&dst[R_BORDER], val); }
    else if (w == LR_BOX + 0) { atomicMax(&dst[R_LO + 0], 0x7fffffffu - (tx + (kXM - val))); }
    else if (w == LR_BOX + 1) { atomicMax(&dst[R_HI + 0], tx + val); }
    else if (w == LR_BOX + 2) { atomicMax(&dst[R_LO + 1], 0x7fffffffu - (ty + (kYM - val))); }
    else if (w == LR_BOX + 3) { atomicMax(&dst[R_HI + 1], ty + val); }
    else if (w == LR_BOX + 4) { atomicMax(&dst[R_LO + 2], 0x7fffffffu - (tz + (31u - val))); }
    else if (w == LR_BOX + 5) { atomicMax(&dst[R_HI + 2], tz + val); }
    else if (w == LR_MIN) { atomicMax(&dst[R_MIN], val); }
    else if (w == LR_MAX) { atomicMax(&dst[R_MAX], val); }
    else if (w == LR_FIRST) {
      const uint32_t first = 0xFFFFFu - val;
      const unsigned long long fidx = (unsigned long long)((tile.z0 + gz0 + (first >> 11)) * sz + (tile.y0 + ((first >> kXB) & kYM)) * sy +
                                                           (tile.x0 + (first & kXM)));
      atomicMax(reinterpret_cast<unsigned long long*>(&dst[R_FIRST]), ~fidx);
    } else if (w >= LR_HIST) { if (val) atomicAdd(&dst[R_HIST + (w - LR_HIST)], val); }
  }
  for (int it = tid; it < kPairSlots * PW; it += kThreads) {
    const int slot = it / PW, w = it % PW;
    const int g = gslot[kRegSlots + slot];
    if (g < 0) continue;
    uint32_t* dst = &p.prec[(size_t)g * kPairWords];
    const uint32_t* src = &s.prec[slot * PW];
    const uint32_t val = src[w];
    if (w == LP_SUM || w == LP_SQ) {
      double dd = *reinterpret_cast<const double*>(&src[w]);
      if (dd != 0.0) atomicAdd(reinterpret_cast<double*>(&dst[w == LP_SUM ? P_SUM : P_SQ]), dd);
    } else if (w == LP_SUM + 1 || w == LP_SQ + 1 || w == 3) {
    } else if (w == LP_CNT) { atomicAdd(&dst[P_CNT], val); }
    else if (w == LP_MIN) { atomicMax(&dst[P_MIN], val); }
    else if (w == LP_MAX) { atomicMax(&dst[P_MAX], val); }
    else if (w >= LP_THR && w < LP_THR + 4) { if (val) atomicAdd(&dst[P_THR + (w - LP_THR)], val); }
    else if (w >= LP_HIST) { if (val) atomicAdd(&dst[P_HIST + (w - LP_HIST)], val); }
  }
}

}  // namespace

template <int BINS, bool VEC>
static void launch_bv(const AccParams& p, uint32_t nb, hipStream_t stream) {
  if (p.masked) hipLaunchKernelGGL((rag_accumulate_kernel<BINS, VEC, true>), dim3(nb), dim3(kThreads), 0, stream, p);
  else hipLaunchKernelGGL((rag_accumulate_kernel<BINS, VEC, false>), dim3(nb), dim3(kThreads), 0, stream, p);
}

int launch_accumulate(const AccParams& p, hipStream_t stream) {
  const uint32_t nb = (uint32_t)p.nbx * p.nby * p.nbz;
  const bool vec = (p.nx % kTileX) == 0;
  if (p.hist.bins <= 8) { if (vec) launch_bv<8, true>(p, nb, stream); else launch_bv<8, false>(p, nb, stream); }
  else { if (vec) launch_bv<16, true>(p, nb, stream); else launch_bv<16, false>(p, nb, stream); }
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
