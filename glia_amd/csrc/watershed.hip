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
// Each step is a fixed point of a local rule, computed by whole-volume sweeps until nothing changes (HBM-streaming passes:
// 4..28 B per voxel and sweep); the sweeps of step 3 are double-buffered (a voxel's state is three words).
#include <rocprim/device/device_scan.hpp>
#include <utility>

#include "hmt_internal.hpp"

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
__global__ void ws_iota(unsigned long long* c, long long n) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) c[p] = (unsigned long long)p;
}
// one sweep of the reconstruction by erosion: g <- max(f, min over the voxel and its neighbours of g)
__global__ void ws_hmin_sweep(WsGrid G, const float* f, const float* gin, float* gout, uint32_t* changed) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= G.n) return;
  float m = gin[p];
  ws_neighbours(G, p, [&](long long q) { const float v = gin[q]; m = v < m ? v : m; });
  const float fp = f[p];
  m = m > fp ? m : fp;
  gout[p] = m;
  if (m != gin[p]) *changed = 1u;
}
// plateau components: comp(p) -> the smallest linear index of p's face-connected set of equal g.  In place: the rule is a
// monotone minimum, so any interleaving reaches the same fixed point; one pointer jump per sweep shortens the chains.
__global__ void ws_comp_sweep(WsGrid G, const float* g, unsigned long long* comp, uint32_t* changed) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= G.n) return;
  const float gp = g[p];
  unsigned long long c = comp[p];
  const unsigned long long c0 = c;
  ws_neighbours(G, p, [&](long long q) { if (g[q] == gp) { const unsigned long long cq = comp[q]; c = cq < c ? cq : c; } });
  const unsigned long long cc = comp[c];
  c = cc < c ? cc : c;
  if (c < c0) { atomicMin(&comp[p], c); *changed = 1u; }
}
__global__ void ws_lower_flag(WsGrid G, const float* g, const unsigned long long* comp, uint32_t* haslower) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= G.n) return;
  const float gp = g[p];
  bool lower = false;
  ws_neighbours(G, p, [&](long long q) { lower = lower || g[q] < gp; });
  if (lower) haslower[comp[p]] = 1u;
}
__global__ void ws_root_flag(WsGrid G, const unsigned long long* comp, const uint32_t* haslower, uint32_t* root) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < G.n) root[p] = (comp[p] == (unsigned long long)p && !haslower[p]) ? 1u : 0u;
}
__global__ void ws_init_flood(WsGrid G, const float* f, const unsigned long long* comp, const uint32_t* haslower, const uint32_t* rank, float* L,
                              uint32_t* d, uint32_t* lab, uint32_t* marker) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= G.n) return;
  const unsigned long long c = comp[p];
  const bool m = !haslower[c];
  marker[p] = m ? 1u : 0u;
  L[p] = m ? f[p] : __builtin_inff();
  d[p] = m ? 0u : 0xFFFFFFFFu;
  lab[p] = m ? rank[c] + 1u : 0u;
}
// one sweep of the flooding: a voxel's state (L, d, label) = the lexicographic minimum over its labelled neighbours q of
// (max(L_q, f_p), L unchanged ? d_q + 1 : 0, label_q); markers are fixed sources
__global__ void ws_flood_sweep(WsGrid G, const float* f, const uint32_t* marker, const float* Li, const uint32_t* di, const uint32_t* li,
                               float* Lo, uint32_t* dout, uint32_t* lo, uint32_t* changed) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= G.n) return;
  float bL = Li[p]; uint32_t bd = di[p], bl = li[p];
  if (!marker[p]) {
    // recomputed from the neighbours alone, never from the voxel's own previous state: a neighbour whose cost improves may pass on
    // a LARGER label at an unchanged cost for this voxel (the cost does not always rise strictly with the predecessor's), and a state
    // kept from before would then rest on nothing.  Costs settle first (their rule is monotone), then labels along the acyclic
    // relation "best predecessor".
    bL = __builtin_inff(); bd = 0xFFFFFFFFu; bl = 0u;
    const float gp = f[p];
    ws_neighbours(G, p, [&](long long q) {
      const uint32_t lq = li[q];
      if (lq == 0u) return;
      const float Lq = Li[q];
      const float Lc = Lq > gp ? Lq : gp;
      const uint32_t dc = Lc == Lq ? di[q] + 1u : 0u;
      const bool better = bl == 0u || Lc < bL || (Lc == bL && (dc < bd || (dc == bd && lq < bl)));
      if (better) { bL = Lc; bd = dc; bl = lq; }
    });
    if (bl != li[p] || bd != di[p] || bL != Li[p]) *changed = 1u;
  }
  Lo[p] = bL; dout[p] = bd; lo[p] = bl;
}

}  // namespace

int watershed_labels(int dim, const int64_t dims[3], const float* d_img, double level, uint32_t* d_out, uint32_t* n_labels, int* sweeps,
                     hipStream_t stream) {
  WsGrid G;
  G.dim = dim; G.nx = dims[0]; G.ny = dims[1]; G.nz = dim == 3 ? dims[2] : 1; G.n = G.nx * G.ny * G.nz;
  const long long n = G.n;
  if (n <= 0 || n >= (1ll << 32)) { set_error("watershed: between 1 and 2^32 - 1 voxels"); return GLIA_HMT_ERR_ARG; }
  const unsigned blocks = (unsigned)((n + 255) / 256);
  float *g0 = nullptr, *g1 = nullptr, *L0 = nullptr, *L1 = nullptr;
  uint32_t *d0 = nullptr, *d1 = nullptr, *l1 = nullptr, *haslower = nullptr, *root = nullptr, *rank = nullptr, *marker = nullptr, *changed = nullptr;
  unsigned long long* comp = nullptr;
  void* tmp = nullptr;
  int total_sweeps = 0;
  auto fail = [&](int rc) {
    for (void* p : {(void*)g0, (void*)g1, (void*)L0, (void*)L1, (void*)d0, (void*)d1, (void*)l1, (void*)haslower, (void*)root, (void*)rank, (void*)marker,
                    (void*)changed, (void*)comp, tmp}) if (p) (void)hipFree(p);
    return rc;
  };
#define WS_TRY(e) do { hipError_t _e = (e); if (_e != hipSuccess) { set_error(std::string("watershed: ") + hipGetErrorString(_e)); return fail(GLIA_HMT_ERR_HIP); } } while (0)
  WS_TRY(hipMalloc(&g0, 4 * n)); WS_TRY(hipMalloc(&g1, 4 * n)); WS_TRY(hipMalloc(&changed, 4));
  // 1. h-minima transform
  hipLaunchKernelGGL(ws_shift, dim3(blocks), dim3(256), 0, stream, d_img, g0, n, level);
  for (;;) {
    uint32_t h = 0;
    WS_TRY(hipMemsetAsync(changed, 0, 4, stream));
    for (int rep = 0; rep < 8; ++rep) {        // a few sweeps per host round trip
      hipLaunchKernelGGL(ws_hmin_sweep, dim3(blocks), dim3(256), 0, stream, G, d_img, g0, g1, changed);
      std::swap(g0, g1);
      ++total_sweeps;
    }
    WS_TRY(hipMemcpyAsync(&h, changed, 4, hipMemcpyDeviceToHost, stream));
    WS_TRY(hipStreamSynchronize(stream));
    if (!h) break;
  }
  (void)hipFree(g1); g1 = nullptr;
  const float* g = g0;
  // 2. plateaus, regional minima, raster-order numbering
  WS_TRY(hipMalloc(&comp, 8 * n)); WS_TRY(hipMalloc(&haslower, 4 * n)); WS_TRY(hipMalloc(&root, 4 * (n + 1))); WS_TRY(hipMalloc(&rank, 4 * (n + 1)));
  hipLaunchKernelGGL(ws_iota, dim3(blocks), dim3(256), 0, stream, comp, n);
  for (;;) {
    uint32_t h = 0;
    WS_TRY(hipMemsetAsync(changed, 0, 4, stream));
    for (int rep = 0; rep < 8; ++rep) { hipLaunchKernelGGL(ws_comp_sweep, dim3(blocks), dim3(256), 0, stream, G, g, comp, changed); ++total_sweeps; }
    WS_TRY(hipMemcpyAsync(&h, changed, 4, hipMemcpyDeviceToHost, stream));
    WS_TRY(hipStreamSynchronize(stream));
    if (!h) break;
  }
  WS_TRY(hipMemsetAsync(haslower, 0, 4 * n, stream));
  hipLaunchKernelGGL(ws_lower_flag, dim3(blocks), dim3(256), 0, stream, G, g, comp, haslower);
  hipLaunchKernelGGL(ws_root_flag, dim3(blocks), dim3(256), 0, stream, G, comp, haslower, root);
  WS_TRY(hipMemsetAsync(root + n, 0, 4, stream));
  {
    size_t bytes = 0;
    WS_TRY(rocprim::exclusive_scan(nullptr, bytes, root, rank, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), stream));
    WS_TRY(hipMalloc(&tmp, bytes ? bytes : 16));
    WS_TRY(rocprim::exclusive_scan(tmp, bytes, root, rank, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), stream));
  }
  uint32_t nlab = 0;
  WS_TRY(hipMemcpyAsync(&nlab, rank + n, 4, hipMemcpyDeviceToHost, stream));
  // 3. flooding (double-buffered; the output volume is one of the two label planes)
  WS_TRY(hipMalloc(&L0, 4 * n)); WS_TRY(hipMalloc(&L1, 4 * n)); WS_TRY(hipMalloc(&d0, 4 * n)); WS_TRY(hipMalloc(&d1, 4 * n));
  WS_TRY(hipMalloc(&l1, 4 * n)); WS_TRY(hipMalloc(&marker, 4 * n));
  uint32_t* l0 = d_out;
  hipLaunchKernelGGL(ws_init_flood, dim3(blocks), dim3(256), 0, stream, G, d_img, comp, haslower, rank, L0, d0, l0, marker);
  for (;;) {
    uint32_t h = 0;
    WS_TRY(hipMemsetAsync(changed, 0, 4, stream));
    for (int rep = 0; rep < 4; ++rep) {        // an even number of sweeps: the current state ends in (L0, d0, d_out)
      hipLaunchKernelGGL(ws_flood_sweep, dim3(blocks), dim3(256), 0, stream, G, d_img, marker, L0, d0, l0, L1, d1, l1, changed);
      hipLaunchKernelGGL(ws_flood_sweep, dim3(blocks), dim3(256), 0, stream, G, d_img, marker, L1, d1, l1, L0, d0, l0, changed);
      total_sweeps += 2;
    }
    WS_TRY(hipMemcpyAsync(&h, changed, 4, hipMemcpyDeviceToHost, stream));
    WS_TRY(hipStreamSynchronize(stream));
    if (!h) break;
  }
  WS_TRY(hipGetLastError());
  if (n_labels) *n_labels = nlab;
  if (sweeps) *sweeps = total_sweeps;
#undef WS_TRY
  return fail(GLIA_HMT_OK);
}

}  // namespace glia
