// glia_amd/csrc/rag_compact.hip -- K4a: turn the sparse accumulation hash tables into the dense RAG:
// regions ascending by label, directed pairs ascending by (a,b) -- the lexicographic order
// TBoundaryTable's std::map imposes on its keys (type/boundary_table.hxx:19,109-113).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "hmt_internal.hpp"
#include "greedy_common.hpp"

namespace glia {
namespace {

// Used slots of a hash table -> dense (key, slot) list, consuming the keys.  A workgroup sweeps kGatherIters * 256
// slots twice: first it counts (one LDS atomic per wave), reserves its range with ONE global atomic, then it writes.
// (One global atomic per wave -- half a million on a 2^25-slot table -- serialised the whole kernel: 5.9 ms.)
constexpr int kGatherIters = 32;
template <typename K>
__global__ __launch_bounds__(256) void gather_used(K* keys, uint32_t cap, K* out_keys, uint32_t* out_slots, uint32_t* counter) {
  __shared__ uint32_t s_count, s_base;
  const int lane = threadIdx.x & 63;
  const uint32_t first = blockIdx.x * (uint32_t)(kGatherIters * 256) + threadIdx.x;
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  for (int it = 0; it < kGatherIters; ++it) {
    const uint32_t i = first + (uint32_t)it * 256u;
    const bool used = i < cap && keys[i] != 0;
    const unsigned long long m = __ballot(used);
    if (m && lane == 0) atomicAdd(&s_count, (uint32_t)__popcll(m));
  }
  __syncthreads();
  if (threadIdx.x == 0) { s_base = s_count ? atomicAdd(counter, s_count) : 0u; s_count = 0; }
  __syncthreads();
  const uint32_t base = s_base;
  for (int it = 0; it < kGatherIters; ++it) {
    const uint32_t i = first + (uint32_t)it * 256u;
    const K k = i < cap ? keys[i] : (K)0;
    const bool used = k != 0;
    const unsigned long long m = __ballot(used);
    if (m == 0) continue;
    uint32_t wbase = 0;
    if (lane == 0) wbase = atomicAdd(&s_count, (uint32_t)__popcll(m));
    wbase = __shfl(wbase, 0);
    if (used) {
      keys[i] = 0;   // consume: the table is clean again for the next build
      const uint32_t pos = base + wbase + (uint32_t)__popcll(m & ((1ull << lane) - 1));
      out_keys[pos] = k;
      out_slots[pos] = i;
    }
  }
}

__global__ void gather_records(uint32_t* src, const uint32_t* slots, uint32_t* dst, int64_t n, int words) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = n * words;
  if (i >= total) return;
  int64_t rec = i / words;
  int w = (int)(i % words);
  uint32_t* q = &src[(size_t)slots[rec] * words + w];
  dst[i] = *q;
  *q = 0;
}

__global__ void decode_region_keys(const uint32_t* keys, uint32_t* labels, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) labels[i] = keys[i] - 1u;
}
__global__ void decode_pair_keys(const unsigned long long* keys, uint32_t* a, uint32_t* b, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    a[i] = (uint32_t)(keys[i] >> 32) - 1u;
    b[i] = (uint32_t)(keys[i] & 0xFFFFFFFFull) - 1u;
  }
}

// (scratch comes from the process-wide block cache, greedy_common.hpp: a hipMalloc / hipFree pair per temporary cost the
// build several milliseconds and a device-wide synchronisation each)
template <typename K>
int sort_used(DeviceBuffers& buf, K* d_keys, uint32_t cap, K** d_sorted_keys, uint32_t** d_sorted_slots, int64_t* n_out,
              hipStream_t stream) {
  int rc;
  uint32_t* d_counter = nullptr;
  if ((rc = buf.get(&d_counter, 1, true, stream))) return rc;
  K* d_k = nullptr;
  uint32_t* d_s = nullptr;
  // first pass only counts (outputs sized afterwards would need two passes; cap-sized scratch is fine here)
  if ((rc = buf.get(&d_k, (size_t)cap, false, stream))) return rc;
  if ((rc = buf.get(&d_s, (size_t)cap, false, stream))) return rc;
  hipLaunchKernelGGL(gather_used<K>, dim3((cap + kGatherIters * 256 - 1) / (kGatherIters * 256)), dim3(256), 0, stream, d_keys, cap, d_k, d_s,
                     d_counter);
  uint32_t n = 0;
  GLIA_HIP_TRY(hipMemcpyAsync(&n, d_counter, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  GLIA_HIP_TRY(hipStreamSynchronize(stream));
  K* d_k2 = nullptr;
  uint32_t* d_s2 = nullptr;
  if ((rc = buf.get(&d_k2, (size_t)(n ? n : 1), false, stream))) return rc;
  if ((rc = buf.get(&d_s2, (size_t)(n ? n : 1), false, stream))) return rc;
  if (n) {
    size_t tmp_bytes = 0;
    GLIA_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_k, d_k2, d_s, d_s2, (size_t)n, 0, sizeof(K) * 8, stream));
    char* d_tmp = nullptr;
    if ((rc = buf.get(&d_tmp, tmp_bytes ? tmp_bytes : 16, false, stream))) return rc;
    GLIA_HIP_TRY(rocprim::radix_sort_pairs((void*)d_tmp, tmp_bytes, d_k, d_k2, d_s, d_s2, (size_t)n, 0, sizeof(K) * 8, stream));
  }
  *d_sorted_keys = d_k2;
  *d_sorted_slots = d_s2;
  *n_out = n;
  return GLIA_HMT_OK;
}

}  // namespace

int compact_tables(const AccParams& p, uint32_t rcap, uint32_t pcap, RagArrays* out, hipStream_t stream) {
  uint32_t* d_rk = nullptr; uint32_t* d_rs = nullptr;
  unsigned long long* d_pk = nullptr; uint32_t* d_ps = nullptr;
  DeviceBuffers buf;                               // (returns its blocks to the cache when this function ends: after the last sync)
  int rc = sort_used<uint32_t>(buf, p.rkeys, rcap, &d_rk, &d_rs, &out->R, stream);
  if (rc) return rc;
  rc = sort_used<unsigned long long>(buf, p.pkeys, pcap, &d_pk, &d_ps, &out->P, stream);
  if (rc) return rc;
  const int64_t R = out->R, P = out->P;
  GLIA_HIP_TRY(hipMalloc(&out->d_rlabel, sizeof(uint32_t) * (size_t)(R ? R : 1)));
  GLIA_HIP_TRY(hipMalloc(&out->d_rrec, sizeof(uint32_t) * kRegionWords * (size_t)(R ? R : 1)));
  GLIA_HIP_TRY(hipMalloc(&out->d_pa, sizeof(uint32_t) * (size_t)(P ? P : 1)));
  GLIA_HIP_TRY(hipMalloc(&out->d_pb, sizeof(uint32_t) * (size_t)(P ? P : 1)));
  GLIA_HIP_TRY(hipMalloc(&out->d_prec, sizeof(uint32_t) * kPairWords * (size_t)(P ? P : 1)));
  if (R) {
    hipLaunchKernelGGL(decode_region_keys, dim3((R + 255) / 256), dim3(256), 0, stream, d_rk, out->d_rlabel, R);
    int64_t tot = R * kRegionWords;
    hipLaunchKernelGGL(gather_records, dim3((tot + 255) / 256), dim3(256), 0, stream, p.rrec, d_rs, out->d_rrec, R, kRegionWords);
  }
  if (P) {
    hipLaunchKernelGGL(decode_pair_keys, dim3((P + 255) / 256), dim3(256), 0, stream, d_pk, out->d_pa, out->d_pb, P);
    int64_t tot = P * kPairWords;
    hipLaunchKernelGGL(gather_records, dim3((tot + 255) / 256), dim3(256), 0, stream, p.prec, d_ps, out->d_prec, P, kPairWords);
  }
  GLIA_HIP_TRY(hipGetLastError());
  GLIA_HIP_TRY(hipStreamSynchronize(stream));
  return GLIA_HMT_OK;
}

}  // namespace glia
