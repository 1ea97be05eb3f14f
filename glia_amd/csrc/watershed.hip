// glia_amd/csrc/watershed.hip -- the step before the RAG: morphological watershed of a float image.
//
// Reference: glia::watershed (util/image_alg.hxx:9-21, gadget/main_watershed.cxx) = itk::MorphologicalWatershedImageFilter
// with SetLevel(level), MarkWatershedLineOff(), face connectivity.  ITK is not available here and its flooding resolves ties
// by the arrival order of a sequential hierarchical queue, which no parallel algorithm reproduces in general: PARITY WITH ITK IS
// UNPINNED BY CONSTRUCTION (tie order only: the pipeline below is the filter's).  What is implemented -- here and, independently, in oracle/hmt_oracle.cc (orc_watershed), bit for
// bit the same labels -- is the same morphological watershed with every tie decided by a rule that does not depend on any order:
//   1. g = h-minima transform of f: reconstruction by erosion of (float)(f + level) above f (ITK's HMinima step);
//   2. markers = regional minima of g (face-connected plateaus without a lower neighbour), numbered 1..n in raster order of
//      their first voxel (ITK's RegionalMinima + ConnectedComponent steps);
//   3. flooding ON THE ORIGINAL IMAGE f (MorphologicalWatershedFromMarkers keeps the filter's input; only the regional-minima
//      step sees the h-minima image): every voxel takes the label of the marker that reaches it at the lowest cost (L, d):
//      L = the highest f on the path (the flood level), d = steps walked since the level last rose (distance on the plateau);
//      equal costs: the smaller label.  Markers keep their label.  (Rounds 1-2 flooded on g: where g > f -- filled shallow
//      minima -- basins were then split by plateau distance on g instead of by f's relief.)
// Steps 1 and 3 are fixed points of local rules.  Rounds 1-2 computed them by whole-volume Jacobi sweeps (one launch moves
// information by one voxel: 240 sweeps at 512^3, ~300x the algorithmic bytes); round 3: a workgroup iterates a 16^3 tile (2D:
// 64^2) in LDS until it is stable before it writes it back, so a launch moves information by a tile, and the plateaus of step 2
// are components of a union-find forest (one uniting pass, one flattening pass) instead of a min-index propagation.
#include <rocprim/device/device_scan.hpp>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <utility>

#include "greedy_common.hpp"
#include "hmt_internal.hpp"
#include "skew.hpp"

namespace glia {
namespace {

struct WsGrid { long long nx, ny, nz, n; int dim; };

template <typename F>
__device__ __forceinline__ void ws_neighbours(const WsGrid& G, long long p, F f) {
  const long long x = p % G.nx, y = (p / G.nx) % G.ny, z = p / (G.nx * G.ny);
  if (x > 0) f(p - 1);
  if (x + 1 < G.nx) f(p + 1);
  if (y > 0) f(p - G.nx);
  if (y + 1 < G.ny) f(p + G.nx);
  if (G.dim == 3) {
    if (z > 0) f(p - G.nx * G.ny);
    if (z + 1 < G.nz) f(p + G.nx * G.ny);
  }
}

__global__ void ws_shift(const float* f, float* g, long long n, double level) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) g[p] = (float)((double)f[p] + level);
}
__global__ void ws_iota(uint32_t* c, long long n) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) c[p] = (uint32_t)p;
}

// ---- tiles --------------------------------------------------------------------------------------------------------------------
// Both fixed points below propagate one voxel per whole-volume sweep when every sweep is a launch (rounds 1-2: 240 sweeps at
// 512^3).  Here a workgroup loads a tile and its one-voxel halo into LDS, iterates the local rule THERE until the tile is stable,
// and writes the tile back: information crosses a tile per launch instead of a voxel, and the global rounds drop to about
// (longest flood path) / (tile edge).  3D tiles are 16^3 (+ halo 18^3), 2D tiles 64 x 64 (+ halo 66 x 66).
constexpr int kWsThreads = 512;
constexpr int kWsCells = 18 * 18 * 18;       // >= 66 * 66
struct WsTile {
  int tx, ty, tz;                            // interior extent
  int hx, hy, hz;                            // extent with halo
  long long x0, y0, z0;                      // volume coordinates of the interior's first voxel
};
__device__ __forceinline__ WsTile ws_tile(const WsGrid& G) {
  WsTile T;
  if (G.dim == 3) { T.tx = T.ty = T.tz = 16; } else { T.tx = T.ty = 64; T.tz = 1; }
  T.hx = T.tx + 2; T.hy = T.ty + 2; T.hz = G.dim == 3 ? T.tz + 2 : 1;
  const long long nbx = (G.nx + T.tx - 1) / T.tx, nby = (G.ny + T.ty - 1) / T.ty;
  const long long b = blockIdx.x;
  T.x0 = (b % nbx) * T.tx; T.y0 = ((b / nbx) % nby) * T.ty; T.z0 = (b / (nbx * nby)) * T.tz;
  return T;
}
// halo cell index -> volume index (or -1 outside the volume)
__device__ __forceinline__ long long ws_cell_voxel(const WsGrid& G, const WsTile& T, int c, int& cx, int& cy, int& cz) {
  cx = c % T.hx; cy = (c / T.hx) % T.hy; cz = c / (T.hx * T.hy);
  const long long x = T.x0 + cx - 1, y = T.y0 + cy - 1, z = G.dim == 3 ? T.z0 + cz - 1 : 0;
  if (x < 0 || y < 0 || z < 0 || x >= G.nx || y >= G.ny || z >= G.nz) return -1;
  return (z * G.ny + y) * G.nx + x;
}
__device__ __forceinline__ bool ws_interior(const WsGrid& G, const WsTile& T, int cx, int cy, int cz) {
  return cx >= 1 && cx <= T.tx && cy >= 1 && cy <= T.ty && (G.dim != 3 || (cz >= 1 && cz <= T.tz));
}

// Work list: a tile runs in a round only if it or one of its face neighbours changed in the round before (dirty[round parity]);
// a tile that changes marks itself and its six neighbours for the next round.  After the first rounds nearly every launch is a
// flag test.
__device__ __forceinline__ bool ws_tile_take(const uint8_t* dirty_now) { return dirty_now[blockIdx.x] != 0; }
__device__ __forceinline__ void ws_tile_mark(const WsGrid& G, const WsTile& T, uint8_t* dirty_next) {
  const long long nbx = (G.nx + T.tx - 1) / T.tx, nby = (G.ny + T.ty - 1) / T.ty, nbz = (G.nz + T.tz - 1) / T.tz;
  const long long b = blockIdx.x, bx = b % nbx, by = (b / nbx) % nby, bz = b / (nbx * nby);
  dirty_next[b] = 1;
  if (bx > 0) dirty_next[b - 1] = 1;
  if (bx + 1 < nbx) dirty_next[b + 1] = 1;
  if (by > 0) dirty_next[b - nbx] = 1;
  if (by + 1 < nby) dirty_next[b + nbx] = 1;
  if (bz > 0) dirty_next[b - nbx * nby] = 1;
  if (bz + 1 < nbz) dirty_next[b + nbx * nby] = 1;
}

// reconstruction by erosion, one tile: g <- max(f, min over the voxel and its neighbours of g) until the tile is stable.  The
// rule is a monotone decrease towards the greatest fixed point below the start, so any order of updates -- in place in LDS, halo
// values of neighbouring tiles from any earlier moment -- reaches the same limit.
__global__ __launch_bounds__(kWsThreads) void ws_hmin_tile(WsGrid G, const float* f, float* g, uint32_t* changed, uint8_t* dirty_now, uint8_t* dirty_next) {
  __shared__ float sg[kWsCells];
  __shared__ float sf[kWsCells];
  __shared__ int any;
  if (!ws_tile_take(dirty_now)) return;
  __syncthreads();
  if (threadIdx.x == 0) dirty_now[blockIdx.x] = 0;      // (this buffer is the round after next's "next")
  const WsTile T = ws_tile(G);
  const int ncell = T.hx * T.hy * T.hz;
  for (int c = threadIdx.x; c < ncell; c += kWsThreads) {
    int cx, cy, cz;
    const long long v = ws_cell_voxel(G, T, c, cx, cy, cz);
    sg[c] = v >= 0 ? g[v] : __builtin_inff();       // outside the volume: never the minimum
    sf[c] = v >= 0 ? f[v] : __builtin_inff();
  }
  __syncthreads();
  const int sy = T.hx, sz = T.hx * T.hy;
  bool mine = false;
  for (int it = 0; it < 96; ++it) {
    bool ch = false;
    for (int c = threadIdx.x; c < ncell; c += kWsThreads) {
      const int cx = c % T.hx, cy = (c / T.hx) % T.hy, cz = c / (T.hx * T.hy);
      if (!ws_interior(G, T, cx, cy, cz)) continue;
      float m = sg[c];
      const float g0 = m;
      m = fminf(m, sg[c - 1]); m = fminf(m, sg[c + 1]); m = fminf(m, sg[c - sy]); m = fminf(m, sg[c + sy]);
      if (G.dim == 3) { m = fminf(m, sg[c - sz]); m = fminf(m, sg[c + sz]); }
      m = fmaxf(m, sf[c]);
      if (m != g0) { sg[c] = m; ch = true; }
    }
    mine = mine || ch;
    if (!__syncthreads_or(ch ? 1 : 0)) break;
  }
  if (threadIdx.x == 0) any = 0;
  __syncthreads();
  if (mine) any = 1;
  for (int c = threadIdx.x; c < ncell; c += kWsThreads) {
    int cx, cy, cz;
    const long long v = ws_cell_voxel(G, T, c, cx, cy, cz);
    if (v >= 0 && ws_interior(G, T, cx, cy, cz)) g[v] = sg[c];
  }
  __syncthreads();
  if (threadIdx.x == 0 && any) { *changed = 1u; ws_tile_mark(G, T, dirty_next); }
}

// plateau components by union-find: comp(p) -> the smallest linear index of p's face-connected set of equal g.  One pass unites
// every voxel with its +x / +y / +z neighbour of equal value (the larger root is linked under the smaller with atomicMin), one
// pass flattens.
__device__ __forceinline__ uint32_t ws_root(const uint32_t* comp, uint32_t x) {
  for (;;) {
    const uint32_t p = __hip_atomic_load(&comp[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p == x) return x;
    x = p;
  }
}
__device__ __forceinline__ void ws_unite(uint32_t* comp, uint32_t a, uint32_t b) {
  for (;;) {
    a = ws_root(comp, a); b = ws_root(comp, b);
    if (a == b) return;
    if (a > b) { const uint32_t t = a; a = b; b = t; }
    const uint32_t old = atomicMin(&comp[b], a);       // b was a root: now it points at the smaller root a
    if (old == b) return;
    b = old;                                           // somebody linked b meanwhile: go on from there
  }
}
__global__ void ws_comp_unite(WsGrid G, const float* g, uint32_t* comp) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= G.n) return;
  const long long x = p % G.nx, y = (p / G.nx) % G.ny, z = p / (G.nx * G.ny);
  const float gp = g[p];
  if (x + 1 < G.nx && g[p + 1] == gp) ws_unite(comp, (uint32_t)p, (uint32_t)(p + 1));
  if (y + 1 < G.ny && g[p + G.nx] == gp) ws_unite(comp, (uint32_t)p, (uint32_t)(p + G.nx));
  if (G.dim == 3 && z + 1 < G.nz && g[p + G.nx * G.ny] == gp) ws_unite(comp, (uint32_t)p, (uint32_t)(p + G.nx * G.ny));
}
__global__ void ws_comp_flatten(WsGrid G, uint32_t* comp) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < G.n) comp[p] = ws_root(comp, (uint32_t)p);
}
// "this plateau has a lower neighbour": one BIT per voxel index, set at the plateau's root
__global__ void ws_lower_flag(WsGrid G, const float* g, const uint32_t* comp, uint32_t* haslower) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= G.n) return;
  const float gp = g[p];
  bool lower = false;
  ws_neighbours(G, p, [&](long long q) { lower = lower || g[q] < gp; });
  if (lower) { const uint32_t c = comp[p]; atomicOr(&haslower[c >> 5], 1u << (c & 31u)); }
}
__device__ __forceinline__ bool ws_bit(const uint32_t* bits, uint32_t i) { return (bits[i >> 5] >> (i & 31u)) & 1u; }
__global__ void ws_root_flag(WsGrid G, const uint32_t* comp, const uint32_t* haslower, uint32_t* root) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < G.n) root[p] = (comp[p] == (uint32_t)p && !ws_bit(haslower, (uint32_t)p)) ? 1u : 0u;
}
// flooding in two fixed points.  A voxel's final state is the lexicographic minimum over its neighbours q of
// (max(L_q, f_p), L unchanged ? d_q + 1 : 0, label_q), markers fixed.  Its first component alone obeys L_p = max(f_p, min_q L_q)
// -- the erosion rule again (markers start at f, the rest at +inf, and a marker keeps its level because its own value takes part in
// the minimum): ws_hmin_tile computes it in place.  With the levels final, only neighbours that reach p AT its level count:
// L_q == L_p continues the plateau (d_q + 1), L_q < L_p with f_p == L_p is a rise (distance 0); (d, label) is ONE 64-bit word and
// its rule a monotone decrease, so it is updated in place -- in LDS and in the volume -- in any order.
// markers: label = raster rank of the plateau's first voxel + 1, everything else 0 -- into the caller's label volume, which then
// says "fixed or free" for the whole flood (the plateau arrays are dead from here on: their storage becomes the flood's state)
__global__ void ws_marker_labels(WsGrid G, const uint32_t* comp, const uint32_t* haslower, const uint32_t* rank, uint32_t* lab) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= G.n) return;
  const uint32_t c = comp[p];
  lab[p] = ws_bit(haslower, c) ? 0u : rank[c] + 1u;
}
__global__ void ws_init_flood(WsGrid G, const float* f, const uint32_t* lab, float* L, unsigned long long* st) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= G.n) return;
  const uint32_t m = lab[p];
  L[p] = m ? f[p] : __builtin_inff();
  st[p] = m ? (unsigned long long)m : ~0ull;       // marker: distance 0, its label; else unreached
}
__global__ __launch_bounds__(kWsThreads) void ws_label_tile(WsGrid G, const float* f, const float* L, const uint32_t* marker,
                                                            unsigned long long* st, uint32_t* changed, uint8_t* dirty_now, uint8_t* dirty_next) {
  if (!ws_tile_take(dirty_now)) return;
  __shared__ float sL[kWsCells];
  __shared__ unsigned long long ss[kWsCells];
  __shared__ int any;
  __syncthreads();
  if (threadIdx.x == 0) dirty_now[blockIdx.x] = 0;
  const WsTile T = ws_tile(G);
  const int ncell = T.hx * T.hy * T.hz;
  constexpr int kPer = (kWsCells + kWsThreads - 1) / kWsThreads;
  // per own cell: 0 = fixed, 1 = free, 3 = free and a rise may enter it (f_p == L_p)
  uint32_t kind[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int c = threadIdx.x + k * kWsThreads;
    kind[k] = 0;
    if (c >= ncell) continue;
    int cx, cy, cz;
    const long long v = ws_cell_voxel(G, T, c, cx, cy, cz);
    sL[c] = v >= 0 ? L[v] : __builtin_inff();
    ss[c] = v >= 0 ? st[v] : ~0ull;          // outside the volume: unreached for ever
    if (v >= 0 && ws_interior(G, T, cx, cy, cz) && marker[v] == 0u) kind[k] = (f[v] == L[v]) ? 3u : 1u;
  }
  __syncthreads();
  const int sy = T.hx, sz = T.hx * T.hy;
  bool mine = false;
  for (int it = 0; it < 128; ++it) {
    bool ch = false;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int c = threadIdx.x + k * kWsThreads;
      if (!kind[k]) continue;
      const float Lp = sL[c];
      unsigned long long best = ~0ull;
      auto relax = [&](int q) {
        const unsigned long long sq = ss[q];
        if (sq == ~0ull) return;
        const float Lq = sL[q];
        unsigned long long cand = ~0ull;
        if (Lq == Lp) cand = sq + (1ull << 32);                        // the plateau goes on: one step further
        else if (Lq < Lp && kind[k] == 3u) cand = sq & 0xFFFFFFFFull;    // the level rises here: distance 0
        best = cand < best ? cand : best;
      };
      relax(c - 1); relax(c + 1); relax(c - sy); relax(c + sy);
      if (G.dim == 3) { relax(c - sz); relax(c + sz); }
      if (best != ss[c]) { ss[c] = best; ch = true; }
    }
    mine = mine || ch;
    if (!__syncthreads_or(ch ? 1 : 0)) break;
  }
  if (threadIdx.x == 0) any = 0;
  __syncthreads();
  if (mine) any = 1;
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int c = threadIdx.x + k * kWsThreads;
    if (!kind[k]) continue;
    int cx, cy, cz;
    const long long v = ws_cell_voxel(G, T, c, cx, cy, cz);
    st[v] = ss[c];
  }
  __syncthreads();
  if (threadIdx.x == 0 && any) { *changed = 1u; ws_tile_mark(G, T, dirty_next); }
}
__global__ void ws_labels_out(const unsigned long long* st, long long n, uint32_t* out) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) out[p] = (uint32_t)st[p];
}

}  // namespace

int watershed_labels(int dim, const int64_t dims[3], const float* d_img, double level, uint32_t* d_out, uint32_t* n_labels, int* sweeps,
                     hipStream_t stream) {
  WsGrid G;
  G.dim = dim; G.nx = dims[0]; G.ny = dims[1]; G.nz = dim == 3 ? dims[2] : 1; G.n = G.nx * G.ny * G.nz;
  const long long n = G.n;
  if (n <= 0 || n >= (1ll << 32)) { set_error("watershed: between 1 and 2^32 - 1 voxels"); return GLIA_HMT_ERR_ARG; }
  const unsigned blocks = (unsigned)((n + 255) / 256);
  const long long tx = dim == 3 ? 16 : 64, tz = dim == 3 ? 16 : 1;
  const unsigned tiles = (unsigned)(((G.nx + tx - 1) / tx) * ((G.ny + tx - 1) / tx) * ((G.nz + tz - 1) / tz));
  // Scratch: 12.1 bytes per voxel, in two blocks that change roles (round 4; 28 bytes in round 3 -- at 1024^3 that was 28 GB,
  // more than the block cache parks, and the call's time was the driver's allocator: 0.44 s or 1.11 s).
  //   A (4 n bytes)  the h-minima image g            -> the flood levels L          (g is dead once the markers are numbered)
  //   B (8 n bytes)  plateau roots | raster ranks    -> the flood's (distance, label) words
  //   bits (n / 8)   "plateau has a lower neighbour"
  // The caller's label volume carries the markers' labels through the flood (0 = free voxel) and receives the result.
  float* A = nullptr;
  unsigned long long* B = nullptr;
  uint32_t *haslower = nullptr, *changed = nullptr;
  char* tmp = nullptr;
  int total_sweeps = 0;
  // the scratch blocks come from the process-wide block cache (greedy_common.hpp): hipMalloc / hipFree of gigabyte blocks cost
  // more than the kernels (measured: 160 ms of a 215 ms call at 512^3)
  DeviceBuffers buf;
  auto fail = [&](int rc) { return rc; };
#define WS_TRY(e) do { hipError_t _e = (e); if (_e != hipSuccess) { set_error(std::string("watershed: ") + hipGetErrorString(_e)); return fail(GLIA_HMT_ERR_HIP); } } while (0)
#define WS_GET(ptr, count) do { int _rc = buf.get(&(ptr), (size_t)(count), false, stream); if (_rc) return _rc; } while (0)
  const bool trace = option("GLIA_HMT_TRACE");
  const auto tr0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    (void)hipStreamSynchronize(stream);
    fprintf(stderr, "[trace] watershed: %s at %.2f ms (%d launches so far)\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr0).count(), total_sweeps);
  };
  WS_GET(A, n); WS_GET(changed, 4);
  float* const g0 = A;
  uint8_t* dirty[2] = {nullptr, nullptr};
  WS_GET(dirty[0], tiles); WS_GET(dirty[1], tiles);
  WS_TRY(hipMemsetAsync(dirty[0], 1, tiles, stream)); WS_TRY(hipMemsetAsync(dirty[1], 0, tiles, stream));
  // 1. h-minima transform: tile rounds (in place: the rule is monotone) until no tile changes
  hipLaunchKernelGGL(ws_shift, dim3(blocks), dim3(256), 0, stream, d_img, g0, n, level);
  for (;;) {
    uint32_t h = 0;
    WS_TRY(hipMemsetAsync(changed, 0, 4, stream));
    for (int rep = 0; rep < 2; ++rep) {        // two rounds per host round trip
      hipLaunchKernelGGL(ws_hmin_tile, dim3(tiles), dim3(kWsThreads), 0, stream, G, d_img, g0, changed, dirty[rep], dirty[rep ^ 1]);
      ++total_sweeps;
    }
    WS_TRY(hipMemcpyAsync(&h, changed, 4, hipMemcpyDeviceToHost, stream));
    WS_TRY(hipStreamSynchronize(stream));
    if (!h) break;
  }
  lap("h-minima done");
  const float* g = g0;
  // 2. plateaus, regional minima, raster-order numbering
  WS_GET(B, n + 1); WS_GET(haslower, (n + 31) / 32);
  uint32_t* const comp = reinterpret_cast<uint32_t*>(B);
  uint32_t* const rank = comp + n;                    // [n]
  uint32_t* const root = d_out;                       // (the output volume is free until the markers are written)
  hipLaunchKernelGGL(ws_iota, dim3(blocks), dim3(256), 0, stream, comp, n);
  hipLaunchKernelGGL(ws_comp_unite, dim3(blocks), dim3(256), 0, stream, G, g, comp);
  hipLaunchKernelGGL(ws_comp_flatten, dim3(blocks), dim3(256), 0, stream, G, comp);
  total_sweeps += 2;
  WS_TRY(hipMemsetAsync(haslower, 0, 4 * (size_t)((n + 31) / 32), stream));
  hipLaunchKernelGGL(ws_lower_flag, dim3(blocks), dim3(256), 0, stream, G, g, comp, haslower);
  hipLaunchKernelGGL(ws_root_flag, dim3(blocks), dim3(256), 0, stream, G, comp, haslower, root);
  {
    size_t bytes = 0;
    WS_TRY(rocprim::exclusive_scan(nullptr, bytes, root, rank, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
    WS_GET(tmp, bytes ? bytes : 16);
    WS_TRY(rocprim::exclusive_scan(tmp, bytes, root, rank, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
  }
  uint32_t last[2] = {0, 0};                          // markers = rank[n-1] + root[n-1]
  WS_TRY(hipMemcpyAsync(&last[0], rank + (n - 1), 4, hipMemcpyDeviceToHost, stream));
  WS_TRY(hipMemcpyAsync(&last[1], root + (n - 1), 4, hipMemcpyDeviceToHost, stream));
  hipLaunchKernelGGL(ws_marker_labels, dim3(blocks), dim3(256), 0, stream, G, comp, haslower, rank, d_out);
  lap("plateaus + markers done");
  WS_TRY(hipStreamSynchronize(stream));
  const uint32_t nlab = last[0] + last[1];
  // 3. flooding: levels (the erosion kernel again, in place), then (distance, label) in place
  float* const L0 = A;
  unsigned long long* const stw = B;
  hipLaunchKernelGGL(ws_init_flood, dim3(blocks), dim3(256), 0, stream, G, d_img, d_out, L0, stw);
  WS_TRY(hipMemsetAsync(dirty[0], 1, tiles, stream)); WS_TRY(hipMemsetAsync(dirty[1], 0, tiles, stream));
  for (;;) {
    uint32_t h = 0;
    WS_TRY(hipMemsetAsync(changed, 0, 16, stream));
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(ws_hmin_tile, dim3(tiles), dim3(kWsThreads), 0, stream, G, d_img, L0, changed, dirty[rep], dirty[rep ^ 1]);
      ++total_sweeps;
    }
    WS_TRY(hipMemcpyAsync(&h, changed, 4, hipMemcpyDeviceToHost, stream));
    WS_TRY(hipStreamSynchronize(stream));
    if (!h) break;
  }
  lap("flood levels done");
  WS_TRY(hipMemsetAsync(dirty[0], 1, tiles, stream)); WS_TRY(hipMemsetAsync(dirty[1], 0, tiles, stream));
  for (;;) {
    uint32_t h = 0;
    WS_TRY(hipMemsetAsync(changed, 0, 16, stream));
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(ws_label_tile, dim3(tiles), dim3(kWsThreads), 0, stream, G, d_img, L0, d_out, stw, changed, dirty[rep], dirty[rep ^ 1]);
      ++total_sweeps;
    }
    WS_TRY(hipMemcpyAsync(&h, changed, 4, hipMemcpyDeviceToHost, stream));
    WS_TRY(hipStreamSynchronize(stream));
    if (!h) break;
  }
  hipLaunchKernelGGL(ws_labels_out, dim3(blocks), dim3(256), 0, stream, stw, n, d_out);
  WS_TRY(hipGetLastError());
  lap("flooding done");
  if (n_labels) *n_labels = nlab;
  if (sweeps) *sweeps = total_sweeps;
  WS_TRY(hipStreamSynchronize(stream));      // the blocks go back to the cache when buf dies: nothing may still use them
#undef WS_GET
#undef WS_TRY
  return fail(GLIA_HMT_OK);
}

}  // namespace glia
