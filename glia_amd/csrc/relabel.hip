// glia_amd/csrc/relabel.hip -- the label-volume rewrites either side of the merge path:
//   * transformKeys (util/struct_merge.hxx:188-210): merge order -> leaf key -> final key map (host, path-compressed)
//   * transformImage (util/image.hxx:227-242): every unmasked voxel whose label has a mapping is rewritten, in place
//   * relabelImage (util/image.hxx:992-1001) = itk::RelabelComponentImageFilter: consecutive labels by decreasing size
// The rewrite kernels are streaming passes: 4 B read (+4 B mask) + 4 B written per voxel, the label map is a dense
// lookup table that stays in L2 / Infinity Cache (labels < 2^28) or a sorted key array searched per voxel.
#include <algorithm>
#include <cstring>
#include <unordered_map>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>

#include "hmt_internal.hpp"

namespace glia {

namespace {

constexpr uint32_t kNoMap = 0xFFFFFFFFu;
constexpr uint32_t kDenseLimit = 1u << 28;

__device__ __forceinline__ uint32_t map_dense(const uint32_t* lut, uint32_t lut_n, uint32_t v, int fill) {
  const uint32_t m = v < lut_n ? lut[v] : kNoMap;
  return m != kNoMap ? m : (fill ? 0u /* BG_VAL */ : v);
}
__device__ __forceinline__ uint32_t map_sorted(const uint32_t* src, const uint32_t* dst, uint32_t n, uint32_t v, int fill) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (src[mid] < v) lo = mid + 1; else hi = mid; }
  return (lo < n && src[lo] == v) ? dst[lo] : (fill ? 0u : v);
}

// one thread = 4 consecutive voxels (dwordx4 in, dwordx4 out)
template <bool DENSE>
__global__ void transform_kernel(uint32_t* lab, long long n, const uint32_t* a, const uint32_t* b, uint32_t m,
                                 const uint32_t* mask, int fill) {
  const long long i4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= n) return;
  auto f = [&](uint32_t v) { return DENSE ? map_dense(a, m, v, fill) : map_sorted(a, b, m, v, fill); };
  if (i4 + 4 <= n) {
    uint4 v = *reinterpret_cast<const uint4*>(lab + i4);
    if (mask) {
      const uint4 k = *reinterpret_cast<const uint4*>(mask + i4);
      v.x = k.x != 0u ? f(v.x) : v.x; v.y = k.y != 0u ? f(v.y) : v.y;       // MASK_OUT_VAL = 0 (glia_image.hxx:28)
      v.z = k.z != 0u ? f(v.z) : v.z; v.w = k.w != 0u ? f(v.w) : v.w;
    } else { v.x = f(v.x); v.y = f(v.y); v.z = f(v.z); v.w = f(v.w); }
    *reinterpret_cast<uint4*>(lab + i4) = v;
  } else {
    for (long long i = i4; i < n; ++i) if (!mask || mask[i] != 0u) lab[i] = f(lab[i]);
  }
}

__global__ void max_label_kernel(const uint32_t* lab, long long n, uint32_t* out) {
  uint32_t m = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) m = max(m, lab[i]);
  for (int off = 32; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// label sizes: runs of equal labels along x are counted in registers, one atomic per run
__global__ void count_labels_kernel(const uint32_t* lab, long long n, unsigned long long* cnt) {
  const long long i0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i0 >= n) return;
  const long long i1 = i0 + 8 < n ? i0 + 8 : n;
  uint32_t cur = lab[i0];
  unsigned long long run = 1;
  for (long long i = i0 + 1; i < i1; ++i) {
    const uint32_t v = lab[i];
    if (v == cur) ++run; else { atomicAdd(&cnt[cur], run); cur = v; run = 1; }
  }
  atomicAdd(&cnt[cur], run);
}

// genBoundaryConfidenceImage (hmt/tree_segment.hxx:145-171): every voxel of a directed boundary receives the value of its
// leaf pair; all other voxels 0.  The voxel's pair is re-derived with the neighbour rule of the accumulation pass.
__global__ void paint_pairs_kernel(VolumeRef vol, const uint32_t* pa, const uint32_t* pb, long long P, const float* val, float* out) {
  const long long N = vol.nx * vol.ny * vol.nz;
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  float r = 0.0f;
  const uint32_t t = vol.lab[p];
  if (t != kMaskedLabel) {
    const long long x = p % vol.nx, y = (p / vol.nx) % vol.ny, z = p / (vol.nx * vol.ny);
    const long long sy = vol.nx, sz = vol.nx * vol.ny;
    const uint32_t* L = vol.lab_nb;
    uint32_t nb = t, q;
    do {
      if (x > 0 && (q = L[p - 1]) != t && q != kMaskedLabel) { nb = q; break; }
      if (x + 1 < vol.nx && (q = L[p + 1]) != t && q != kMaskedLabel) { nb = q; break; }
      if (y > 0 && (q = L[p - sy]) != t && q != kMaskedLabel) { nb = q; break; }
      if (y + 1 < vol.ny && (q = L[p + sy]) != t && q != kMaskedLabel) { nb = q; break; }
      if (vol.dim == 3) {
        if (z > 0 && (q = L[p - sz]) != t && q != kMaskedLabel) { nb = q; break; }
        if (z + 1 < vol.nz && (q = L[p + sz]) != t && q != kMaskedLabel) { nb = q; break; }
      }
    } while (false);
    if (nb != t) {
      long long lo = 0, hi = P;
      while (lo < hi) { const long long mid = (lo + hi) >> 1; if (pa[mid] < t || (pa[mid] == t && pb[mid] < nb)) lo = mid + 1; else hi = mid; }
      if (lo < P && pa[lo] == t && pb[lo] == nb && val[lo] > 0.0f) r = val[lo];
    }
  }
  out[p] = r;
}

}  // namespace

int paint_pair_values(const VolumeRef& vol, const uint32_t* d_pa, const uint32_t* d_pb, int64_t P, const float* h_val, float* d_out,
                      hipStream_t stream) {
  float* d_val = nullptr;
  GLIA_HIP_TRY(hipMalloc(&d_val, sizeof(float) * (size_t)(P ? P : 1)));
  GLIA_HIP_TRY(hipMemcpyAsync(d_val, h_val, sizeof(float) * (size_t)P, hipMemcpyHostToDevice, stream));
  const long long N = vol.nx * vol.ny * vol.nz;
  hipLaunchKernelGGL(paint_pairs_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, vol, d_pa, d_pb, (long long)P, d_val, d_out);
  GLIA_HIP_TRY(hipGetLastError());
  GLIA_HIP_TRY(hipStreamSynchronize(stream));
  (void)hipFree(d_val);
  return GLIA_HMT_OK;
}

// transformKeys: for every merged key that is not itself created by a merge, the key it ends up in
int transform_keys(const uint32_t* order, int64_t n, std::vector<uint32_t>* src, std::vector<uint32_t>* dst) {
  std::unordered_map<uint32_t, uint32_t> omap, isnew;
  omap.reserve((size_t)n * 2);
  for (int64_t i = 0; i < n; ++i) { omap[order[3 * i]] = order[3 * i + 2]; omap[order[3 * i + 1]] = order[3 * i + 2]; isnew[order[3 * i + 2]] = 1; }
  std::vector<uint32_t> path;
  std::unordered_map<uint32_t, uint32_t> root;          // memo: key -> final key
  root.reserve(omap.size());
  auto find = [&](uint32_t k) {
    path.clear();
    uint32_t d = k;
    while (true) {
      auto r = root.find(d);
      if (r != root.end()) { d = r->second; break; }
      auto o = omap.find(d);
      if (o == omap.end()) break;
      path.push_back(d);
      d = o->second;
      if (path.size() > omap.size() + 1) return kNoMap;      // a cycle: not a merge order
    }
    for (uint32_t p : path) root[p] = d;
    return d;
  };
  src->clear(); dst->clear();
  for (auto const& op : omap) {
    if (isnew.count(op.first)) continue;
    const uint32_t d = find(op.first);
    if (d == kNoMap) { set_error("transform_keys: the merge list contains a cycle"); return GLIA_HMT_ERR_ARG; }
    src->push_back(op.first); dst->push_back(d);
  }
  std::vector<size_t> idx(src->size());
  for (size_t i = 0; i < idx.size(); ++i) idx[i] = i;
  std::sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return (*src)[x] < (*src)[y]; });
  std::vector<uint32_t> s2(idx.size()), d2(idx.size());
  for (size_t i = 0; i < idx.size(); ++i) { s2[i] = (*src)[idx[i]]; d2[i] = (*dst)[idx[i]]; }
  src->swap(s2); dst->swap(d2);
  return GLIA_HMT_OK;
}

int transform_image(uint32_t* d_lab, int64_t n, const uint32_t* h_src, const uint32_t* h_dst, int64_t m, const uint32_t* d_mask,
                    int fill_missing, hipStream_t stream, double* ms) {
  if (n == 0) return GLIA_HMT_OK;
  if ((reinterpret_cast<uintptr_t>(d_lab) & 15) || (d_mask && (reinterpret_cast<uintptr_t>(d_mask) & 15))) {
    set_error("transform_image: volumes must be 16-byte aligned");
    return GLIA_HMT_ERR_ARG;
  }
  // sorted copy of the map (later entries of a duplicated key win, like unordered_map::operator[] assignments)
  std::vector<std::pair<uint32_t, uint32_t>> kv((size_t)m);
  for (int64_t i = 0; i < m; ++i) kv[i] = {h_src[i], h_dst[i]};
  std::stable_sort(kv.begin(), kv.end(), [](auto const& x, auto const& y) { return x.first < y.first; });
  std::vector<uint32_t> s, d;
  for (size_t i = 0; i < kv.size(); ++i) {
    if (!s.empty() && s.back() == kv[i].first) d.back() = kv[i].second;
    else { s.push_back(kv[i].first); d.push_back(kv[i].second); }
  }
  const uint32_t maxsrc = s.empty() ? 0u : s.back();
  const bool dense = maxsrc < kDenseLimit;
  uint32_t *da = nullptr, *db = nullptr;
  uint32_t mm = 0;
  if (dense) {
    std::vector<uint32_t> lut((size_t)maxsrc + 1, kNoMap);
    for (size_t i = 0; i < s.size(); ++i) lut[s[i]] = d[i];
    mm = maxsrc + 1;
    GLIA_HIP_TRY(hipMalloc(&da, sizeof(uint32_t) * lut.size()));
    GLIA_HIP_TRY(hipMemcpyAsync(da, lut.data(), sizeof(uint32_t) * lut.size(), hipMemcpyHostToDevice, stream));
    GLIA_HIP_TRY(hipStreamSynchronize(stream));
  } else {
    mm = (uint32_t)s.size();
    GLIA_HIP_TRY(hipMalloc(&da, sizeof(uint32_t) * s.size()));
    GLIA_HIP_TRY(hipMalloc(&db, sizeof(uint32_t) * s.size()));
    GLIA_HIP_TRY(hipMemcpyAsync(da, s.data(), sizeof(uint32_t) * s.size(), hipMemcpyHostToDevice, stream));
    GLIA_HIP_TRY(hipMemcpyAsync(db, d.data(), sizeof(uint32_t) * s.size(), hipMemcpyHostToDevice, stream));
    GLIA_HIP_TRY(hipStreamSynchronize(stream));
  }
  hipEvent_t e0, e1;
  GLIA_HIP_TRY(hipEventCreate(&e0)); GLIA_HIP_TRY(hipEventCreate(&e1));
  // The first launch of a kernel of this translation unit loads its code object (this file's holds rocPRIM's sorts since the sparse
  // relabelling path of round 3: several milliseconds) -- on the host, between the two events.  Asking for the kernel's attributes loads it
  // here, so that the reported time is the kernel's (round 3's closing set reported 12.0 ms for a 4 ms kernel; tools/transform_bench.py).
  {
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, dense ? reinterpret_cast<const void*>(&transform_kernel<true>) : reinterpret_cast<const void*>(&transform_kernel<false>));
  }
  const long long threads = (n + 3) / 4;
  const unsigned blocks = (unsigned)((threads + 255) / 256);
  GLIA_HIP_TRY(hipEventRecord(e0, stream));
  if (dense) hipLaunchKernelGGL(transform_kernel<true>, dim3(blocks), dim3(256), 0, stream, d_lab, (long long)n, da, db, mm, d_mask, fill_missing);
  else hipLaunchKernelGGL(transform_kernel<false>, dim3(blocks), dim3(256), 0, stream, d_lab, (long long)n, da, db, mm, d_mask, fill_missing);
  GLIA_HIP_TRY(hipGetLastError());
  GLIA_HIP_TRY(hipEventRecord(e1, stream));
  GLIA_HIP_TRY(hipEventSynchronize(e1));
  float t = 0;
  (void)hipEventElapsedTime(&t, e0, e1);
  if (ms) *ms = t;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(da);
  if (db) (void)hipFree(db);
  return GLIA_HMT_OK;
}

int relabel_image(uint32_t* d_lab, int64_t n, int64_t min_size, uint32_t* n_labels, hipStream_t stream) {
  *n_labels = 0;
  if (n == 0) return GLIA_HMT_OK;
  uint32_t* d_max;
  GLIA_HIP_TRY(hipMalloc(&d_max, sizeof(uint32_t)));
  GLIA_HIP_TRY(hipMemsetAsync(d_max, 0, sizeof(uint32_t), stream));
  hipLaunchKernelGGL(max_label_kernel, dim3(2048), dim3(256), 0, stream, d_lab, (long long)n, d_max);
  uint32_t maxl = 0;
  GLIA_HIP_TRY(hipMemcpyAsync(&maxl, d_max, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  GLIA_HIP_TRY(hipStreamSynchronize(stream));
  (void)hipFree(d_max);
  std::vector<uint32_t> labs;                   // labels present (not 0), and their voxel counts
  std::vector<unsigned long long> cntOf;
  if (maxl >= kDenseLimit) {
    // sparse labels: no count array over the label range -- sort a copy of the labels, run lengths of the sorted copy.  rocPRIM's
    // run-length interface takes a 32-bit size and returns 32-bit counts: larger volumes are refused, not truncated
    if (n > 0xFFFFFFFFll) { set_error("relabel_image: labels >= 2^28 in a volume of more than 2^32 - 1 voxels are not supported"); return GLIA_HMT_ERR_UNSUPPORTED; }
    uint32_t *d_a = nullptr, *d_b = nullptr, *d_u = nullptr, *d_c = nullptr, *d_nr = nullptr;
    void* d_tmp = nullptr;
    auto freeAll = [&]() { for (void* q : {(void*)d_a, (void*)d_b, (void*)d_u, (void*)d_c, (void*)d_nr, d_tmp}) if (q) (void)hipFree(q); };
    auto fail = [&](hipError_t e) { freeAll(); set_error(std::string("relabel_image: ") + hipGetErrorString(e)); return GLIA_HMT_ERR_HIP; };
    hipError_t e;
    if ((e = hipMalloc(&d_a, sizeof(uint32_t) * (size_t)n)) != hipSuccess || (e = hipMalloc(&d_b, sizeof(uint32_t) * (size_t)n)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(d_a, d_lab, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice, stream)) != hipSuccess) return fail(e);
    size_t tmp_bytes = 0;
    if ((e = rocprim::radix_sort_keys(nullptr, tmp_bytes, d_a, d_b, (size_t)n, 0, 32, stream)) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16)) != hipSuccess) return fail(e);
    if ((e = rocprim::radix_sort_keys(d_tmp, tmp_bytes, d_a, d_b, (size_t)n, 0, 32, stream)) != hipSuccess) return fail(e);
    (void)hipFree(d_tmp); d_tmp = nullptr;
    // run lengths: unique labels into d_a (reused), counts into d_c; a run has at most n <= 2^32 - 1... counts are 32-bit in
    // rocPRIM's interface: volumes beyond 2^32 voxels of ONE label are out of reach of this path
    if ((e = hipMalloc(&d_c, sizeof(uint32_t) * (size_t)n)) != hipSuccess || (e = hipMalloc(&d_nr, sizeof(uint32_t))) != hipSuccess) return fail(e);
    d_u = d_a; d_a = nullptr;
    tmp_bytes = 0;
    if ((e = rocprim::run_length_encode(nullptr, tmp_bytes, d_b, (unsigned int)n, d_u, d_c, d_nr, stream)) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16)) != hipSuccess) return fail(e);
    if ((e = rocprim::run_length_encode(d_tmp, tmp_bytes, d_b, (unsigned int)n, d_u, d_c, d_nr, stream)) != hipSuccess) return fail(e);
    uint32_t nr = 0;
    if ((e = hipMemcpyAsync(&nr, d_nr, sizeof(uint32_t), hipMemcpyDeviceToHost, stream)) != hipSuccess || (e = hipStreamSynchronize(stream)) != hipSuccess) return fail(e);
    std::vector<uint32_t> u(nr), c32(nr);
    if (nr) {
      if ((e = hipMemcpy(u.data(), d_u, sizeof(uint32_t) * nr, hipMemcpyDeviceToHost)) != hipSuccess) return fail(e);
      if ((e = hipMemcpy(c32.data(), d_c, sizeof(uint32_t) * nr, hipMemcpyDeviceToHost)) != hipSuccess) return fail(e);
    }
    freeAll();
    for (uint32_t i = 0; i < nr; ++i) if (u[i] != 0u) { labs.push_back(u[i]); cntOf.push_back(c32[i]); }
  } else {
  unsigned long long* d_cnt;
  GLIA_HIP_TRY(hipMalloc(&d_cnt, sizeof(unsigned long long) * ((size_t)maxl + 1)));
  GLIA_HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long) * ((size_t)maxl + 1), stream));
  const long long threads = (n + 7) / 8;
  hipLaunchKernelGGL(count_labels_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, d_lab, (long long)n, d_cnt);
  std::vector<unsigned long long> cnt((size_t)maxl + 1);
  GLIA_HIP_TRY(hipMemcpyAsync(cnt.data(), d_cnt, sizeof(unsigned long long) * cnt.size(), hipMemcpyDeviceToHost, stream));
  GLIA_HIP_TRY(hipStreamSynchronize(stream));
  (void)hipFree(d_cnt);
  for (uint32_t l = 1; l <= maxl; ++l) if (cnt[l]) { labs.push_back(l); cntOf.push_back(cnt[l]); }
  }
  // RelabelComponentImageFilter: objects (label != 0) sorted by size, largest first, ties by the smaller original
  // label; objects below the minimum size become background
  std::vector<uint32_t> idx(labs.size());
  for (size_t i = 0; i < idx.size(); ++i) idx[i] = (uint32_t)i;
  std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return cntOf[a] != cntOf[b] ? cntOf[a] > cntOf[b] : labs[a] < labs[b]; });
  std::vector<uint32_t> src, dst;
  uint32_t next = 1;
  for (uint32_t i : idx) {
    src.push_back(labs[i]);
    dst.push_back((min_size > 0 && cntOf[i] < (unsigned long long)min_size) ? 0u : next++);
  }
  *n_labels = next - 1;
  return transform_image(d_lab, n, src.data(), dst.data(), (int64_t)src.size(), nullptr, 0, stream, nullptr);
}

}  // namespace glia
