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
// MI355X mapping.  HBM-bound: 8 algorithmic bytes per voxel (4 label + 4 image).  A wave owns one
// x-row segment of 256 voxels (lane = 4 consecutive x = one 16-byte load) and marches kTZ planes in z,
// keeping the z-1 / z / z+1 label rows of its segment in registers; y+-1 rows are re-read (L1/L2 hits:
// the neighbouring waves of the same workgroup stream them).  Because supervoxels are spatially
// coherent, a lane sees long runs of one label / one directed pair: it reduces each run in registers
// (f64 sums, 8-bit packed histogram and threshold counters) and only flushes on a key change, into a
// per-workgroup LDS hash table.  After the march the workgroup folds its LDS tables into the global
// hash tables with one find-or-insert per distinct key and contiguous per-record atomics.
// All reductions are integer adds, unsigned max, or f64 adds, so results are exact (hence
// order-independent and bit-reproducible) whenever the image is a multiple of 2^-k (Q8 pb).
#include "hmt_internal.hpp"

namespace glia {

namespace {

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t hash64(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
  return (uint32_t)k;
}

struct __attribute__((aligned(16))) Lds {
  unsigned long long pkey[kLdsPairSlots];
  uint32_t rkey[kLdsRegionSlots];
  uint32_t rrec[kLdsRegionSlots * kRegionWords];
  uint32_t prec[kLdsPairSlots * kPairWordsLds];
  int gslot[kLdsRegionSlots + kLdsPairSlots];
};
static_assert(sizeof(Lds) * 2 <= 160 * 1024, "two workgroups per CU must fit in LDS");

constexpr int kLdsProbes = 24;
constexpr int kGlobalProbes = 512;

__device__ __forceinline__ int lds_region_slot(Lds& s, uint32_t key) {
  uint32_t h = hash32(key) & (kLdsRegionSlots - 1);
  for (int i = 0; i < kLdsProbes; ++i) {
    uint32_t cur = s.rkey[h];
    if (cur == key) return (int)h;
    if (cur == 0) {
      uint32_t old = atomicCAS(&s.rkey[h], 0u, key);
      if (old == 0 || old == key) return (int)h;
    }
    h = (h + 1) & (kLdsRegionSlots - 1);
  }
  return -1;
}
__device__ __forceinline__ int lds_pair_slot(Lds& s, unsigned long long key) {
  uint32_t h = hash64(key) & (kLdsPairSlots - 1);
  for (int i = 0; i < kLdsProbes; ++i) {
    unsigned long long cur = s.pkey[h];
    if (cur == key) return (int)h;
    if (cur == 0) {
      unsigned long long old = atomicCAS(&s.pkey[h], 0ull, key);
      if (old == 0 || old == key) return (int)h;
    }
    h = (h + 1) & (kLdsPairSlots - 1);
  }
  return -1;
}
__device__ __forceinline__ int global_region_slot(const AccParams& p, uint32_t key) {
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
__device__ __forceinline__ int global_pair_slot(const AccParams& p, unsigned long long key) {
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

// ---- per-lane run accumulators ---------------------------------------------------------------
template <int BINS>
struct Moments {
  uint32_t cnt;
  double sum, sq;
  float mn, mx;
  unsigned long long h0, h1;   // 8-bit packed histogram counters (bins 0-7, 8-15)
  __device__ __forceinline__ void reset() { cnt = 0; sum = 0.0; sq = 0.0; mn = __builtin_inff(); mx = -__builtin_inff(); h0 = 0; h1 = 0; }
  __device__ __forceinline__ void add(float v, int bin) {
    ++cnt;
    double dv = (double)v;
    sum += dv;
    sq = __builtin_fma(dv, dv, sq);    // v*v is exact in double, so this equals (double)v*v added once
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
    if (bin >= 0) {
      unsigned long long one = 1ull << ((bin & 7) * 8);
      if (BINS <= 8 || bin < 8) h0 += one; else h1 += one;
    }
  }
};

// reference bin rule (util/image_stats.hxx:24-35) evaluated with float-exact thresholds; -1 = dropped
template <int BINS>
__device__ __forceinline__ int hist_bin(const HistSpec& hs, float v) {
  int c = 0;
#pragma unroll
  for (int k = 0; k < BINS; ++k) c += (v >= hs.fb[k]) ? 1 : 0;
  bool inside = (v > hs.lo_f) && (v < hs.hi_f);
  int bin = inside ? c : ((v <= hs.lo_f) ? 0 : hs.bins - 1);
  return (inside && c >= hs.bins) ? -1 : bin;
}

template <int BINS>
__device__ __forceinline__ void flush_moments(uint32_t* rec, const Moments<BINS>& m, int oCnt, int oSum, int oSq,
                                              int oMin, int oMax, int oHist) {
  atomicAdd(&rec[oCnt], m.cnt);
  atomicAdd((double*)&rec[oSum], m.sum);
  atomicAdd((double*)&rec[oSq], m.sq);
  atomicMax(&rec[oMin], ~float_ord(m.mn));
  atomicMax(&rec[oMax], float_ord(m.mx));
#pragma unroll
  for (int k = 0; k < BINS; ++k) {
    uint32_t c = (uint32_t)(((k < 8) ? (m.h0 >> (k * 8)) : (m.h1 >> ((k - 8) * 8))) & 0xFF);
    if (c) atomicAdd(&rec[oHist + k], c);
  }
}

template <int BINS>
struct RegionRun {
  uint32_t key;   // label + 1, 0 = none
  Moments<BINS> m;
  uint32_t border;
  int xlo, xhi, zlo, zhi;
  unsigned long long first;
};
template <int BINS>
struct PairRun {
  unsigned long long key;  // 0 = none
  Moments<BINS> m;
  uint32_t thr;            // 4 x 8-bit packed threshold counters
};

template <int BINS>
__device__ __forceinline__ void write_region(uint32_t* rec, const RegionRun<BINS>& r, int y) {
  flush_moments<BINS>(rec, r.m, R_CNT, R_SUM, R_SQ, R_MIN, R_MAX, R_HIST);
  if (r.border) atomicAdd(&rec[R_BORDER], r.border);
  atomicMax(&rec[R_LO + 0], 0x7fffffffu - (uint32_t)r.xlo);
  atomicMax(&rec[R_LO + 1], 0x7fffffffu - (uint32_t)y);
  atomicMax(&rec[R_LO + 2], 0x7fffffffu - (uint32_t)r.zlo);
  atomicMax(&rec[R_HI + 0], (uint32_t)r.xhi + 1u);
  atomicMax(&rec[R_HI + 1], (uint32_t)y + 1u);
  atomicMax(&rec[R_HI + 2], (uint32_t)r.zhi + 1u);
  atomicMax((unsigned long long*)&rec[R_FIRST], ~r.first);
}
template <int BINS>
__device__ __forceinline__ void write_pair(uint32_t* rec, const PairRun<BINS>& r, int nthr) {
  flush_moments<BINS>(rec, r.m, P_CNT, P_SUM, P_SQ, P_MIN, P_MAX, P_HIST);
#pragma unroll
  for (int t = 0; t < GLIA_HMT_MAX_THRESH; ++t) {
    uint32_t c = (r.thr >> (8 * t)) & 0xFF;
    if (t < nthr && c) atomicAdd(&rec[P_THR + t], c);
  }
}

template <int BINS>
__device__ __forceinline__ void flush_region(Lds& s, const AccParams& p, const RegionRun<BINS>& r, int y) {
  if (p.debug & 16) atomicAdd(&p.flags[2], 1u);
  int slot = lds_region_slot(s, r.key);
  if (slot >= 0) { write_region<BINS>(&s.rrec[slot * kRegionWords], r, y); return; }
  if (p.debug & 16) atomicAdd(&p.flags[4], 1u);
  int g = global_region_slot(p, r.key);     // LDS table saturated: straight to HBM (slow, still exact)
  if (g >= 0) write_region<BINS>(&p.rrec[(size_t)g * kRegionWords], r, y);
}
template <int BINS>
__device__ __forceinline__ void flush_pair(Lds& s, const AccParams& p, const PairRun<BINS>& r) {
  if (p.debug & 16) atomicAdd(&p.flags[3], 1u);
  int slot = lds_pair_slot(s, r.key);
  if (slot >= 0) { write_pair<BINS>(&s.prec[slot * kPairWordsLds], r, p.nthr); return; }
  if (p.debug & 16) atomicAdd(&p.flags[5], 1u);
  int g = global_pair_slot(p, r.key);
  if (g >= 0) write_pair<BINS>(&p.prec[(size_t)g * kPairWords], r, p.nthr);
}

struct U4 { uint32_t v[4]; };
struct F4 { float v[4]; };

// VEC: every row segment of the volume is a whole, 16-byte aligned 256-voxel run (nx % 256 == 0):
// one dwordx4 load per lane and row.  Otherwise the guarded scalar form handles any nx.
template <int BINS, bool VEC>
__global__ __launch_bounds__(kThreads, 4) void rag_accumulate_kernel(const AccParams p) {
  __shared__ Lds s;
  const int tid = threadIdx.x;
  {
    uint32_t* w = reinterpret_cast<uint32_t*>(&s);
    for (int i = tid; i < (int)(sizeof(Lds) / 4); i += kThreads) w[i] = 0;
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
  const int64_t x0 = (int64_t)bx * kRowX + lane * kVX;
  const int64_t y = (int64_t)by * kRows + wave;
  const int64_t z0 = (int64_t)bz * kTZ;
  const int64_t z1 = (z0 + kTZ < nz) ? z0 + kTZ : nz;
  const bool rowOk = (y < ny) && (x0 < nx);
  const int64_t sy = nx, sz = nx * ny;
  const bool is3d = p.dim == 3;
  const int nfull = 2 * p.dim;

  auto loadLab = [&](int64_t yy, int64_t zz, bool ok) -> U4 {
    U4 r;
    r.v[0] = r.v[1] = r.v[2] = r.v[3] = 0;
    if (ok && rowOk) {
      const uint32_t* q = p.lab + zz * sz + yy * sy + x0;
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
  auto loadImg = [&](int64_t yy, int64_t zz) -> F4 {
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

  RegionRun<BINS> rr;
  rr.key = 0; rr.m.reset(); rr.border = 0; rr.xlo = rr.xhi = rr.zlo = rr.zhi = 0; rr.first = 0;
  PairRun<BINS> pr;
  pr.key = 0; pr.m.reset(); pr.thr = 0;

  U4 Lp = loadLab(y, z0 - 1, z0 > 0);
  U4 Lc = loadLab(y, z0, true);
  for (int64_t z = z0; z < z1; ++z) {
    U4 Ln = loadLab(y, z + 1, z + 1 < nz);
    U4 Up = loadLab(y - 1, z, y > 0);
    U4 Dn = loadLab(y + 1, z, y + 1 < ny);
    F4 V = loadImg(y, z);
    uint32_t left = __shfl_up(Lc.v[3], 1);
    uint32_t right = __shfl_down(Lc.v[0], 1);
    if (lane == 0) left = (rowOk && x0 > 0) ? p.lab[z * sz + y * sy + x0 - 1] : 0u;
    if (lane == 63) right = (rowOk && x0 + kVX < nx) ? p.lab[z * sz + y * sy + x0 + kVX] : 0u;
    const bool zmv = is3d && z > 0, zpv = is3d && z + 1 < nz;
    const bool ymv = y > 0, ypv = y + 1 < ny;
#pragma unroll
    for (int i = 0; i < kVX; ++i) {
      const int64_t x = x0 + i;
      if (!(rowOk && x < nx) || (p.debug & 4)) continue;
      const uint32_t L = Lc.v[i];
      const uint32_t xm = (i == 0) ? left : Lc.v[i > 0 ? i - 1 : 0];
      const uint32_t xp = (i == kVX - 1) ? right : Lc.v[i < kVX - 1 ? i + 1 : kVX - 1];
      const bool xmv = x > 0, xpv = x + 1 < nx;
      uint32_t b = L;
      if (zpv && Ln.v[i] != L) b = Ln.v[i];
      if (zmv && Lp.v[i] != L) b = Lp.v[i];
      if (ypv && Dn.v[i] != L) b = Dn.v[i];
      if (ymv && Up.v[i] != L) b = Up.v[i];
      if (xpv && xp != L) b = xp;
      if (xmv && xm != L) b = xm;
      const int nvalid = (int)xmv + (int)xpv + (int)ymv + (int)ypv + (int)zmv + (int)zpv;
      const bool boundary = b != L;
      const bool border = !boundary && nvalid < nfull;
      const float v = V.v[i];
      const int bin = hist_bin<BINS>(p.hist, v);

      const uint32_t rkey = L + 1u;
      if (rkey != rr.key) {
        if (rr.key && !(p.debug & 1)) flush_region<BINS>(s, p, rr, (int)y);
        rr.key = rkey; rr.m.reset(); rr.border = 0;
        rr.xlo = rr.xhi = (int)x; rr.zlo = (int)z;
        rr.first = (unsigned long long)(z * sz + y * sy + x);
      }
      rr.m.add(v, bin);
      rr.border += border ? 1u : 0u;
      rr.xlo = min(rr.xlo, (int)x); rr.xhi = max(rr.xhi, (int)x); rr.zhi = (int)z;

      if (boundary) {
        const unsigned long long pkey = ((unsigned long long)rkey << 32) | (unsigned long long)(b + 1u);
        if (pkey != pr.key) {
          if (pr.key && !(p.debug & 2)) flush_pair<BINS>(s, p, pr);
          pr.key = pkey; pr.m.reset(); pr.thr = 0;
        }
        pr.m.add(v, bin);
        uint32_t t = 0;
#pragma unroll
        for (int k = 0; k < GLIA_HMT_MAX_THRESH; ++k) t |= (k < p.nthr && v >= p.thr_f[k]) ? (1u << (8 * k)) : 0u;
        pr.thr += t;
      }
    }
    Lp = Lc;
    Lc = Ln;
  }
  if (rr.key) flush_region<BINS>(s, p, rr, (int)y);
  if (pr.key) flush_pair<BINS>(s, p, pr);
  __syncthreads();

  if (p.debug & 8) return;
  // ---- fold the workgroup's LDS tables into the global tables ----
  for (int i = tid; i < kLdsRegionSlots + kLdsPairSlots; i += kThreads) {
    int g = -1;
    if (i < kLdsRegionSlots) { uint32_t k = s.rkey[i]; if (k) g = global_region_slot(p, k); }
    else { unsigned long long k = s.pkey[i - kLdsRegionSlots]; if (k) g = global_pair_slot(p, k); }
    s.gslot[i] = g;
  }
  __syncthreads();
  for (int it = tid; it < kLdsRegionSlots * kRegionWords; it += kThreads) {
    const int slot = it / kRegionWords, w = it % kRegionWords;
    const int g = s.gslot[slot];
    if (g < 0) continue;
    uint32_t* dst = &p.rrec[(size_t)g * kRegionWords];
    const uint32_t* src = &s.rrec[slot * kRegionWords];
    if (w == R_SUM || w == R_SQ) {
      double d = *reinterpret_cast<const double*>(&src[w]);
      if (d != 0.0) atomicAdd(reinterpret_cast<double*>(&dst[w]), d);
    } else if (w == R_SUM + 1 || w == R_SQ + 1 || w == R_FIRST + 1) {
    } else if (w == R_FIRST) {
      atomicMax(reinterpret_cast<unsigned long long*>(&dst[w]), *reinterpret_cast<const unsigned long long*>(&src[w]));
    } else if ((w >= R_LO && w < R_SUM) || w == R_MIN || w == R_MAX) {
      if (src[w]) atomicMax(&dst[w], src[w]);
    } else {
      if (src[w]) atomicAdd(&dst[w], src[w]);
    }
  }
  for (int it = tid; it < kLdsPairSlots * kPairWordsLds; it += kThreads) {
    const int slot = it / kPairWordsLds, w = it % kPairWordsLds;
    const int g = s.gslot[kLdsRegionSlots + slot];
    if (g < 0) continue;
    uint32_t* dst = &p.prec[(size_t)g * kPairWords];
    const uint32_t* src = &s.prec[slot * kPairWordsLds];
    if (w == P_SUM || w == P_SQ) {
      double d = *reinterpret_cast<const double*>(&src[w]);
      if (d != 0.0) atomicAdd(reinterpret_cast<double*>(&dst[w]), d);
    } else if (w == P_SUM + 1 || w == P_SQ + 1 || w == 3) {
    } else if (w == P_MIN || w == P_MAX) {
      if (src[w]) atomicMax(&dst[w], src[w]);
    } else {
      if (src[w]) atomicAdd(&dst[w], src[w]);
    }
  }
}

}  // namespace

int launch_accumulate(const AccParams& p, hipStream_t stream) {
  const uint32_t nb = (uint32_t)p.nbx * p.nby * p.nbz;
  const bool vec = (p.nx % kRowX) == 0;
  if (p.hist.bins <= 8) {
    if (vec) hipLaunchKernelGGL((rag_accumulate_kernel<8, true>), dim3(nb), dim3(kThreads), 0, stream, p);
    else hipLaunchKernelGGL((rag_accumulate_kernel<8, false>), dim3(nb), dim3(kThreads), 0, stream, p);
  } else {
    if (vec) hipLaunchKernelGGL((rag_accumulate_kernel<16, true>), dim3(nb), dim3(kThreads), 0, stream, p);
    else hipLaunchKernelGGL((rag_accumulate_kernel<16, false>), dim3(nb), dim3(kThreads), 0, stream, p);
  }
  GLIA_HIP_TRY(hipGetLastError());
  return GLIA_HMT_OK;
}

}  // namespace glia
