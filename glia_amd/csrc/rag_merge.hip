// glia_amd/csrc/rag_merge.hip -- C1: combine the partial region adjacency structures of several z-slabs.
// Every leaf / directed-pair statistic is a commutative monoid (adds, unsigned max, f64 adds), so partial
// records with the same key are reduced; parts are combined in part order, which makes the f64 sums
// reproducible (and exact for Q8 images).  There is no reference counterpart: GLIA is single-node
// (SURVEY.md 8e); the algebra is that of rag_accumulate.hip's table fold.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "greedy_common.hpp"

namespace glia {
namespace {

__global__ void make_region_keys(const uint32_t* lab, unsigned long long* keys, uint32_t* idx, uint32_t n, uint32_t base, uint32_t part) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // sort key: label, then part (radix sort is stable, so equal labels keep part order anyway)
  keys[base + i] = ((unsigned long long)lab[i] << 8) | part;
  idx[base + i] = base + i;
}
__global__ void make_pair_keys(const uint32_t* a, const uint32_t* b, unsigned long long* keys, uint32_t* idx, uint32_t n, uint32_t base) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[base + i] = ((unsigned long long)a[i] << 32) | b[i];
  idx[base + i] = base + i;
}
__global__ void head_flags(const unsigned long long* keys, uint32_t* flag, uint32_t n, int shift) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flag[i] = (i == 0 || (keys[i] >> shift) != (keys[i - 1] >> shift)) ? 1u : 0u;
}

// one thread per output word of every segment head
template <bool REGION>
__global__ void combine_records(const unsigned long long* keys, const uint32_t* idx, const uint32_t* flag, const uint32_t* oidx,
                                const uint32_t* src /*concatenated records*/, uint32_t n, int shift, uint32_t* dst,
                                uint32_t* out_a, uint32_t* out_b) {
  constexpr int W = REGION ? kRegionWords : kPairWords;
  unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t i = (uint32_t)(t / W);
  const int w = (int)(t % W);
  if (i >= n || !flag[i]) return;
  const uint32_t o = oidx[i];
  const unsigned long long k = keys[i] >> shift;
  if (w == 0) {
    if (REGION) out_a[o] = (uint32_t)k;
    else { out_a[o] = (uint32_t)(k >> 32); out_b[o] = (uint32_t)k; }
  }
  bool isF64 = false, isF64hi = false, isMax = false, isU64max = false, isU64hi = false;
  if (REGION) {
    isF64 = (w == R_SUM || w == R_SQ); isF64hi = (w == R_SUM + 1 || w == R_SQ + 1);
    isMax = (w >= R_LO && w < R_SUM) || w == R_MIN || w == R_MAX;
    isU64max = (w == R_FIRST); isU64hi = (w == R_FIRST + 1);
  } else {
    isF64 = (w == P_SUM || w == P_SQ); isF64hi = (w == P_SUM + 1 || w == P_SQ + 1);
    isMax = (w == P_MIN || w == P_MAX);
  }
  if (isF64hi || isU64hi) return;
  uint32_t acc = 0;
  double dacc = 0.0;
  unsigned long long uacc = 0;
  for (uint32_t j = i; j < n && (keys[j] >> shift) == k; ++j) {
    const uint32_t* r = &src[(size_t)idx[j] * W];
    if (isF64) { double d; memcpy(&d, &r[w], 8); dacc += d; }
    else if (isU64max) { unsigned long long u; memcpy(&u, &r[w], 8); uacc = u > uacc ? u : uacc; }
    else if (isMax) acc = r[w] > acc ? r[w] : acc;
    else acc += r[w];
  }
  uint32_t* d = &dst[(size_t)o * W];
  if (isF64) memcpy(&d[w], &dacc, 8);
  else if (isU64max) memcpy(&d[w], &uacc, 8);
  else d[w] = acc;
}

// ---- value runs (boundary-voxel values per directed pair) ----
__global__ void run_counts(const unsigned long long* src_off, const uint32_t* order, uint32_t n, unsigned long long* cnt) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j > n) return;
  cnt[j] = j < n ? src_off[order[j] + 1] - src_off[order[j]] : 0ull;
}
__global__ void run_gather(const unsigned long long* src_off, const float* src_vals, const uint32_t* order, uint32_t n, const unsigned long long* new_off,
                           unsigned long long nV, float* out) {
  const unsigned long long v = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nV) return;
  uint32_t lo = 0, hi = n;                      // the last run that starts at or before v (empty runs share a start: take the last)
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (new_off[mid + 1] <= v) lo = mid + 1; else hi = mid; }
  out[v] = src_vals[src_off[order[lo]] + (v - new_off[lo])];
}
}  // namespace
int gather_value_runs(const unsigned long long* src_off, const float* src_vals, const uint32_t* order, uint32_t n, unsigned long long** out_off,
                      float** out_vals, unsigned long long* nV, hipStream_t stream) {
  GLIA_HIP_TRY(hipMalloc(out_off, sizeof(unsigned long long) * ((size_t)n + 1)));
  *out_vals = nullptr; *nV = 0;
  unsigned long long total = 0;
  if (n) {
    DeviceBuffers buf;
    int rc;
    unsigned long long* cnt;
    if ((rc = buf.get(&cnt, (size_t)n + 1, false, stream))) return rc;
    hipLaunchKernelGGL(run_counts, dim3((n + 256) / 256), dim3(256), 0, stream, src_off, order, n, cnt);
    size_t tmp = 0;
    GLIA_HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, cnt, *out_off, 0ull, (size_t)n + 1, rocprim::plus<unsigned long long>(), stream));
    char* d_tmp;
    if ((rc = buf.get(&d_tmp, tmp ? tmp : 16, false, stream))) return rc;
    GLIA_HIP_TRY(rocprim::exclusive_scan((void*)d_tmp, tmp, cnt, *out_off, 0ull, (size_t)n + 1, rocprim::plus<unsigned long long>(), stream));
    GLIA_HIP_TRY(hipMemcpyAsync(&total, *out_off + n, sizeof(total), hipMemcpyDeviceToHost, stream));
    GLIA_HIP_TRY(hipStreamSynchronize(stream));
  } else GLIA_HIP_TRY(hipMemsetAsync(*out_off, 0, sizeof(unsigned long long), stream));
  GLIA_HIP_TRY(hipMalloc(out_vals, sizeof(float) * (size_t)(total ? total : 1)));
  if (total) {
    hipLaunchKernelGGL(run_gather, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src_off, src_vals, order, n, *out_off, total, *out_vals);
    GLIA_HIP_TRY(hipGetLastError());
    GLIA_HIP_TRY(hipStreamSynchronize(stream));
  }
  *nV = total;
  return GLIA_HMT_OK;
}
namespace {
__global__ void pair_counts_u64(const uint32_t* prec, uint32_t P, unsigned long long* cnt) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > P) return;
  cnt[i] = i < P ? (unsigned long long)prec[(size_t)i * kPairWords + P_CNT] : 0ull;
}
// the neighbour rule of rag_accumulate.hip / type/neighbor.hxx:109-126 for the voxels of the owned planes: value -> its pair's run
__global__ void pair_values_scatter(VolumeRef vol, const uint32_t* pa, const uint32_t* pb, long long P, const unsigned long long* off, uint32_t* cursor,
                                    float* out) {
  const long long plane = vol.nx * vol.ny;
  const long long n_owned = (vol.ze - vol.zb) * plane;
  const long long q0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q0 >= n_owned) return;
  const long long p = q0 + vol.zb * plane;
  const long long x = p % vol.nx, y = (p / vol.nx) % vol.ny, z = p / plane;
  const uint32_t t = vol.lab[p];
  uint32_t nb = t;
  const long long sy = vol.nx, sz = plane;
  const uint32_t* L = vol.lab_nb;
  do {
    uint32_t q;
    if (x > 0 && (q = L[p - 1]) != t) { nb = q; break; }
    if (x + 1 < vol.nx && (q = L[p + 1]) != t) { nb = q; break; }
    if (y > 0 && (q = L[p - sy]) != t) { nb = q; break; }
    if (y + 1 < vol.ny && (q = L[p + sy]) != t) { nb = q; break; }
    if (z > 0 && (q = L[p - sz]) != t) { nb = q; break; }
    if (z + 1 < vol.nz && (q = L[p + sz]) != t) { nb = q; break; }
  } while (false);
  if (nb == t) return;
  const long long i = find_pair(pa, pb, P, t, nb);
  if (i < 0) return;
  out[off[i] + atomicAdd(&cursor[i], 1u)] = vol.pb[p];
}
}  // namespace
int collect_pair_values(RagArrays* rag, const VolumeRef& vol, hipStream_t stream) {
  const uint32_t P = (uint32_t)rag->P;
  if (vol.dim != 3 || !vol.lab || !vol.pb || vol.ze < 0) { set_error("collect_pair_values: needs the slab's label and image planes"); return GLIA_HMT_ERR_ARG; }
  DeviceBuffers buf;
  int rc;
  unsigned long long* cnt; uint32_t* cursor;
  if ((rc = buf.get(&cnt, (size_t)P + 1, false, stream))) return rc;
  if ((rc = buf.get(&cursor, (size_t)P + 1, true, stream))) return rc;
  GLIA_HIP_TRY(hipMalloc(&rag->d_pv_off, sizeof(unsigned long long) * ((size_t)P + 1)));
  hipLaunchKernelGGL(pair_counts_u64, dim3((P + 256) / 256), dim3(256), 0, stream, rag->d_prec, P, cnt);
  size_t tmp = 0;
  GLIA_HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, cnt, rag->d_pv_off, 0ull, (size_t)P + 1, rocprim::plus<unsigned long long>(), stream));
  char* d_tmp;
  if ((rc = buf.get(&d_tmp, tmp ? tmp : 16, false, stream))) return rc;
  GLIA_HIP_TRY(rocprim::exclusive_scan((void*)d_tmp, tmp, cnt, rag->d_pv_off, 0ull, (size_t)P + 1, rocprim::plus<unsigned long long>(), stream));
  unsigned long long total = 0;
  GLIA_HIP_TRY(hipMemcpyAsync(&total, rag->d_pv_off + P, sizeof(total), hipMemcpyDeviceToHost, stream));
  GLIA_HIP_TRY(hipStreamSynchronize(stream));
  GLIA_HIP_TRY(hipMalloc(&rag->d_pv, sizeof(float) * (size_t)(total ? total : 1)));
  rag->nV = total;
  const long long n_owned = (vol.ze - vol.zb) * vol.nx * vol.ny;
  if (total && n_owned > 0) {
    hipLaunchKernelGGL(pair_values_scatter, dim3((unsigned)((n_owned + 255) / 256)), dim3(256), 0, stream, vol, rag->d_pa, rag->d_pb, (long long)P, rag->d_pv_off,
                       cursor, rag->d_pv);
    GLIA_HIP_TRY(hipGetLastError());
    GLIA_HIP_TRY(hipStreamSynchronize(stream));
  }
  return GLIA_HMT_OK;
}
namespace {
__global__ void concat_offsets(const unsigned long long* off, uint32_t m, unsigned long long vbase, unsigned long long* cat_off, uint32_t base) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) cat_off[base + i] = off[i] + vbase;
}
__global__ void merged_offsets(const uint32_t* flag, const uint32_t* oidx, const unsigned long long* gscan, uint32_t n, uint32_t nout, unsigned long long nV,
                               unsigned long long* out_off) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n && flag[j]) out_off[oidx[j]] = gscan[j];
  if (j == 0) out_off[nout] = nV;
}

template <bool REGION>
int merge_kind(const RagArrays* parts, int n_parts, RagArrays* out, hipStream_t stream) {
  constexpr int W = REGION ? kRegionWords : kPairWords;
  uint64_t total = 0;
  for (int p = 0; p < n_parts; ++p) total += (uint64_t)(REGION ? parts[p].R : parts[p].P);
  if (total >= (1ull << 32)) { set_error("rag_merge: too many records"); return GLIA_HMT_ERR_ARG; }
  const uint32_t n = (uint32_t)total;
  DeviceBuffers buf;
  int rc;
  unsigned long long *k0, *k1; uint32_t *i0, *i1, *flag, *oidx, *src;
  if ((rc = buf.get(&k0, n, false, stream))) return rc;
  if ((rc = buf.get(&k1, n, false, stream))) return rc;
  if ((rc = buf.get(&i0, n, false, stream))) return rc;
  if ((rc = buf.get(&i1, n, false, stream))) return rc;
  if ((rc = buf.get(&flag, (size_t)n + 1, true, stream))) return rc;
  if ((rc = buf.get(&oidx, (size_t)n + 1, false, stream))) return rc;
  if ((rc = buf.get(&src, (size_t)n * W, false, stream))) return rc;
  const int K = parts[0].K > 0 ? parts[0].K : 1;
  uint32_t base = 0;
  for (int p = 0; p < n_parts; ++p) {
    const uint32_t m = (uint32_t)(REGION ? parts[p].R : parts[p].P);
    if (!m) continue;
    if (REGION) hipLaunchKernelGGL(make_region_keys, dim3((m + 255) / 256), dim3(256), 0, stream, parts[p].d_rlabel, k0, i0, m, base, (uint32_t)p);
    else hipLaunchKernelGGL(make_pair_keys, dim3((m + 255) / 256), dim3(256), 0, stream, parts[p].d_pa, parts[p].d_pb, k0, i0, m, base);
    base += m;
  }
  const int shift = REGION ? 8 : 0;
  uint32_t nout = 0;
  if (n) {
    size_t tmp = 0;
    GLIA_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp, k0, k1, i0, i1, (size_t)n, 0, 64, stream));
    char* d_tmp;
    if ((rc = buf.get(&d_tmp, tmp ? tmp : 16, false, stream))) return rc;
    GLIA_HIP_TRY(rocprim::radix_sort_pairs((void*)d_tmp, tmp, k0, k1, i0, i1, (size_t)n, 0, 64, stream));
    hipLaunchKernelGGL(head_flags, dim3((n + 255) / 256), dim3(256), 0, stream, k1, flag, n, shift);
    size_t tmp2 = 0;
    GLIA_HIP_TRY(rocprim::exclusive_scan(nullptr, tmp2, flag, oidx, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), stream));
    char* d_tmp2;
    if ((rc = buf.get(&d_tmp2, tmp2 ? tmp2 : 16, false, stream))) return rc;
    GLIA_HIP_TRY(rocprim::exclusive_scan((void*)d_tmp2, tmp2, flag, oidx, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), stream));
    GLIA_HIP_TRY(hipMemcpyAsync(&nout, oidx + n, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    GLIA_HIP_TRY(hipStreamSynchronize(stream));
  }
  uint32_t *oa, *ob = nullptr;
  GLIA_HIP_TRY(hipMalloc(&oa, sizeof(uint32_t) * (size_t)(nout ? nout : 1)));
  if (!REGION) GLIA_HIP_TRY(hipMalloc(&ob, sizeof(uint32_t) * (size_t)(nout ? nout : 1)));
  // the keys are shared by all image channels; every channel's records are reduced in the same sorted order
  for (int ch = 0; ch < K; ++ch) {
    uint32_t* dst;
    GLIA_HIP_TRY(hipMalloc(&dst, sizeof(uint32_t) * W * (size_t)(nout ? nout : 1)));
    uint32_t b2 = 0;
    for (int p = 0; p < n_parts; ++p) {
      const uint32_t m = (uint32_t)(REGION ? parts[p].R : parts[p].P);
      if (!m) continue;
      const uint32_t* rec = REGION ? (ch == 0 ? parts[p].d_rrec : parts[p].c_rrec[ch]) : (ch == 0 ? parts[p].d_prec : parts[p].c_prec[ch]);
      GLIA_HIP_TRY(hipMemcpyAsync(src + (size_t)b2 * W, rec, sizeof(uint32_t) * (size_t)m * W, hipMemcpyDeviceToDevice, stream));
      b2 += m;
    }
    if (n) {
      const unsigned long long threads = (unsigned long long)n * W;
      hipLaunchKernelGGL(combine_records<REGION>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, k1, i1, flag, oidx, src, n,
                         shift, dst, oa, ob);
      GLIA_HIP_TRY(hipGetLastError());
      GLIA_HIP_TRY(hipStreamSynchronize(stream));
    }
    if (REGION) { out->c_rrec[ch] = dst; if (ch == 0) out->d_rrec = dst; }
    else { out->c_prec[ch] = dst; if (ch == 0) out->d_prec = dst; }
  }
  if (REGION) { out->R = nout; out->d_rlabel = oa; }
  else { out->P = nout; out->d_pa = oa; out->d_pb = ob; }
  if (!REGION) {
    // value runs: when every part with pairs carries them, the merged pair's run is the concatenation of its sources' runs, i.e.
    // the source runs laid out in the sorted order (a merged pair's offset = the offset of its first source)
    bool all = n > 0;
    for (int p = 0; p < n_parts; ++p) if (parts[p].P && !parts[p].d_pv_off) all = false;
    if (all) {
      unsigned long long vtot = 0;
      for (int p = 0; p < n_parts; ++p) vtot += parts[p].P ? parts[p].nV : 0;
      unsigned long long* cat_off; float* cat_vals;
      if ((rc = buf.get(&cat_off, (size_t)n + 1, false, stream))) return rc;
      if ((rc = buf.get(&cat_vals, (size_t)(vtot ? vtot : 1), false, stream))) return rc;
      uint32_t b3 = 0; unsigned long long vb = 0;
      for (int p = 0; p < n_parts; ++p) {
        const uint32_t m = (uint32_t)parts[p].P;
        if (!m) continue;
        hipLaunchKernelGGL(concat_offsets, dim3((m + 255) / 256), dim3(256), 0, stream, parts[p].d_pv_off, m, vb, cat_off, b3);
        if (parts[p].nV) GLIA_HIP_TRY(hipMemcpyAsync(cat_vals + vb, parts[p].d_pv, sizeof(float) * (size_t)parts[p].nV, hipMemcpyDeviceToDevice, stream));
        b3 += m; vb += parts[p].nV;
      }
      GLIA_HIP_TRY(hipMemcpyAsync(cat_off + n, &vtot, sizeof(vtot), hipMemcpyHostToDevice, stream));
      GLIA_HIP_TRY(hipStreamSynchronize(stream));      // (vtot lives on this stack)
      unsigned long long* gscan; float* vals; unsigned long long nV = 0;
      if ((rc = gather_value_runs(cat_off, cat_vals, i1, n, &gscan, &vals, &nV, stream))) return rc;
      unsigned long long* ooff;
      GLIA_HIP_TRY(hipMalloc(&ooff, sizeof(unsigned long long) * ((size_t)nout + 1)));
      hipLaunchKernelGGL(merged_offsets, dim3((n + 255) / 256), dim3(256), 0, stream, flag, oidx, gscan, n, nout, nV, ooff);
      GLIA_HIP_TRY(hipGetLastError());
      GLIA_HIP_TRY(hipStreamSynchronize(stream));
      (void)hipFree(gscan);
      out->d_pv_off = ooff; out->d_pv = vals; out->nV = nV;
    }
  }
  out->K = K;
  for (int ch = 0; ch < K; ++ch) out->c_bins[ch] = parts[0].c_bins[ch];
  return GLIA_HMT_OK;
}

}  // namespace

namespace {
__global__ void gather_plane_labels(const uint32_t* lab, long long plane_voxels, const long long* plane_z, int n_planes, uint32_t* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= plane_voxels * n_planes) return;
  const int pl = (int)(i / plane_voxels);
  out[i] = lab[plane_z[pl] * plane_voxels + (i - (long long)pl * plane_voxels)];
}
__device__ __forceinline__ bool in_sorted(const uint32_t* a, uint32_t n, uint32_t key) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
  return lo < n && a[lo] == key;
}
__global__ void cut_flags_kernel(const uint32_t* cutlab, uint32_t ncut, const uint32_t* rlabel, uint32_t R, const uint32_t* pa, const uint32_t* pb,
                                 uint32_t P, uint8_t* rflag, uint8_t* pflag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < R) rflag[i] = in_sorted(cutlab, ncut, rlabel[i]) ? 1 : 0;
  if (i < P) pflag[i] = (in_sorted(cutlab, ncut, pa[i]) || in_sorted(cutlab, ncut, pb[i])) ? 1 : 0;
}
}  // namespace

// Which records of a slab's partial map can have a counterpart in another slab: those whose label (either label of a
// pair) occurs on a plane next to a cut -- the slab's first / last owned plane or the halo plane beyond it.  (A connected
// region that crosses a cut has voxels there; records this test misses are still combined by the final merge, the flags
// only decide which records take the keyed owner exchange.)
int rag_cut_flags(const RagArrays& rag, const uint32_t* d_lab, int64_t nx, int64_t ny, int64_t nzl, int64_t zb, int64_t ze,
                  uint8_t* d_rflag, uint8_t* d_pflag, hipStream_t stream) {
  long long planes[4]; int np = 0;
  if (zb > 0) { planes[np++] = zb - 1; planes[np++] = zb; }
  if (ze < nzl) { planes[np++] = ze - 1; planes[np++] = ze; }
  const uint32_t R = (uint32_t)rag.R, P = (uint32_t)rag.P;
  if (np == 0) {
    if (R) GLIA_HIP_TRY(hipMemsetAsync(d_rflag, 0, R, stream));
    if (P) GLIA_HIP_TRY(hipMemsetAsync(d_pflag, 0, P, stream));
    return GLIA_HMT_OK;
  }
  const long long pv = (long long)nx * ny, n = pv * np;
  DeviceBuffers buf;
  int rc;
  long long* d_planes; uint32_t *l0, *l1;
  if ((rc = buf.get(&d_planes, 4, false, stream))) return rc;
  if ((rc = buf.get(&l0, (size_t)n, false, stream))) return rc;
  if ((rc = buf.get(&l1, (size_t)n, false, stream))) return rc;
  GLIA_HIP_TRY(hipMemcpyAsync(d_planes, planes, sizeof(long long) * np, hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL(gather_plane_labels, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_lab, pv, d_planes, np, l0);
  size_t tmp = 0;
  GLIA_HIP_TRY(rocprim::radix_sort_keys(nullptr, tmp, l0, l1, (size_t)n, 0, 32, stream));
  char* d_tmp;
  if ((rc = buf.get(&d_tmp, tmp ? tmp : 16, false, stream))) return rc;
  GLIA_HIP_TRY(rocprim::radix_sort_keys((void*)d_tmp, tmp, l0, l1, (size_t)n, 0, 32, stream));
  const uint32_t m = R > P ? R : P;
  if (m) hipLaunchKernelGGL(cut_flags_kernel, dim3((m + 255) / 256), dim3(256), 0, stream, l1, (uint32_t)n, rag.d_rlabel, R, rag.d_pa, rag.d_pb, P, d_rflag, d_pflag);
  GLIA_HIP_TRY(hipGetLastError());
  GLIA_HIP_TRY(hipStreamSynchronize(stream));
  return GLIA_HMT_OK;
}

int merge_rag_arrays(const RagArrays* parts, int n_parts, RagArrays* out, hipStream_t stream) {
  int rc = merge_kind<true>(parts, n_parts, out, stream);
  if (rc) return rc;
  return merge_kind<false>(parts, n_parts, out, stream);
}

}  // namespace glia
