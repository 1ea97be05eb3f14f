// glia_amd/csrc/synth.hip -- synthetic supervoxels + boundary probability on the device
// (SURVEY.md 8d).  Bit-identical to oracle/hmt_oracle.cc:orc_synth (integer arithmetic, one final
// double->float conversion); used by bench.py and the full-size GPU tests.  Not part of the hot path.
#include "hmt_internal.hpp"

namespace glia {
namespace {

__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

struct CellGrid {
  int D; int S; uint64_t seed, salt; int64_t nc[3];
  __device__ void seedPos(int64_t cx, int64_t cy, int64_t cz, int64_t p[3]) const {
    uint64_t lin = (uint64_t)(cx + nc[0] * (cy + nc[1] * cz));
    uint64_t h = splitmix64(seed ^ salt ^ splitmix64(lin));
    p[0] = cx * S + (int64_t)((h & 0xFFFF) % (uint64_t)S);
    p[1] = cy * S + (int64_t)(((h >> 16) & 0xFFFF) % (uint64_t)S);
    p[2] = (D == 3) ? cz * S + (int64_t)(((h >> 32) & 0xFFFF) % (uint64_t)S) : 0;
  }
  __device__ uint32_t cellOf(int64_t x, int64_t y, int64_t z) const {
    int64_t cx = x / S, cy = y / S, cz = (D == 3) ? z / S : 0;
    int64_t best = INT64_MAX; uint32_t bestId = 0;
    for (int64_t dz = (D == 3 ? -1 : 0); dz <= (D == 3 ? 1 : 0); ++dz)
      for (int64_t dy = -1; dy <= 1; ++dy)
        for (int64_t dx = -1; dx <= 1; ++dx) {
          int64_t ex = cx + dx, ey = cy + dy, ez = cz + dz;
          if (ex < 0 || ey < 0 || ez < 0 || ex >= nc[0] || ey >= nc[1] || ez >= nc[2]) continue;
          int64_t q[3];
          seedPos(ex, ey, ez, q);
          int64_t d = (q[0] - x) * (q[0] - x) + (q[1] - y) * (q[1] - y) + (q[2] - z) * (q[2] - z);
          uint32_t id = (uint32_t)(ex + nc[0] * (ey + nc[1] * ez));
          if (d < best || (d == best && id < bestId)) { best = d; bestId = id; }
        }
    return bestId;
  }
};

__global__ void synth_cells(CellGrid sv, CellGrid tr, int64_t nx, int64_t ny, int64_t nz, uint32_t* labels,
                            uint32_t* truth) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nx * ny * nz) return;
  int64_t x = i % nx, y = (i / nx) % ny, z = i / (nx * ny);
  labels[i] = 1u + sv.cellOf(x, y, z);
  truth[i] = tr.cellOf(x, y, z);
}

__global__ void synth_pb(int dim, int64_t nx, int64_t ny, int64_t nz, uint64_t seed, int variant,
                         const uint32_t* labels, const uint32_t* truth, float* pb) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nx * ny * nz) return;
  int64_t c[3] = {i % nx, (i / nx) % ny, i / (nx * ny)};
  const int64_t st[3] = {1, nx, nx * ny};
  const int64_t nn[3] = {nx, ny, nz};
  int otherTruth = 0, otherSv = 0;
  for (int d = 0; d < dim; ++d) {
    if (c[d] > 0) { int64_t j = i - st[d]; otherTruth |= truth[j] != truth[i]; otherSv |= labels[j] != labels[i]; }
    if (c[d] + 1 < nn[d]) { int64_t j = i + st[d]; otherTruth |= truth[j] != truth[i]; otherSv |= labels[j] != labels[i]; }
  }
  uint64_t r = splitmix64((uint64_t)i ^ seed);
  int q = (int)(r % 77) + 154 * otherTruth + 38 * otherSv;
  if (q > 255) q = 255;
  double v = q / 256.0;
  if (variant == 1) {
    uint64_t r2 = splitmix64(r ^ 0xF32ull);
    v += (double)(r2 >> 40) * (1.0 / 16777216.0) * (1.0 / 256.0);
  }
  pb[i] = (float)v;
}

}  // namespace

int launch_synth(int dim, const int64_t dims[3], int S, int G, uint64_t seed, int variant, uint32_t* d_labels,
                 uint32_t* d_truth_tmp, float* d_pb, hipStream_t stream) {
  CellGrid sv, tr;
  int64_t n[3] = {dims[0], dims[1], dim == 3 ? dims[2] : 1};
  sv.D = tr.D = dim; sv.S = S; tr.S = G; sv.seed = tr.seed = seed;
  sv.salt = 0x5350ull; tr.salt = 0x54525554ull;
  for (int i = 0; i < 3; ++i) { sv.nc[i] = (n[i] + S - 1) / S; tr.nc[i] = (n[i] + G - 1) / G; }
  if (dim == 2) { sv.nc[2] = tr.nc[2] = 1; }
  int64_t N = n[0] * n[1] * n[2];
  dim3 grid((unsigned)((N + 255) / 256)), block(256);
  hipLaunchKernelGGL(synth_cells, grid, block, 0, stream, sv, tr, n[0], n[1], n[2], d_labels, d_truth_tmp);
  hipLaunchKernelGGL(synth_pb, grid, block, 0, stream, dim, n[0], n[1], n[2], seed, variant, d_labels, d_truth_tmp, d_pb);
  GLIA_HIP_TRY(hipGetLastError());
  return GLIA_HMT_OK;
}

}  // namespace glia
