// tools/micro/lat.hip -- ground truth for the single-workgroup greedy loop: cost of a barrier, a dependent global
// load (pointer chase over a footprint beyond L2), a dependent LDS load, a wave reduction, an s_memtime tick.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

__global__ void k_barrier(int n, unsigned long long* out) {
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) __syncthreads();
  if (threadIdx.x == 0) out[0] = __builtin_readcyclecounter() - t0;
}
__global__ void k_chase(const unsigned* next, int n, unsigned* sink, unsigned long long* out) {
  unsigned long long t0 = __builtin_readcyclecounter();
  unsigned p = threadIdx.x;
  for (int i = 0; i < n; ++i) p = next[p];
  sink[threadIdx.x] = p;
  if (threadIdx.x == 0) out[0] = __builtin_readcyclecounter() - t0;
}
__global__ void k_chase_barrier(const unsigned* next, int n, unsigned* sink, unsigned long long* out) {
  unsigned long long t0 = __builtin_readcyclecounter();
  unsigned p = threadIdx.x;
  for (int i = 0; i < n; ++i) { if (threadIdx.x < 64) p = next[p]; __syncthreads(); }
  sink[threadIdx.x] = p;
  if (threadIdx.x == 0) out[0] = __builtin_readcyclecounter() - t0;
}
__global__ void k_store_barrier(unsigned* buf, const unsigned* next, int n, unsigned long long* out) {
  unsigned long long t0 = __builtin_readcyclecounter();
  unsigned p = threadIdx.x;
  for (int i = 0; i < n; ++i) { p = (p * 1664525u + 1013904223u); buf[p & 0xFFFFFFu] = p; __syncthreads(); }
  if (threadIdx.x == 0) out[0] = __builtin_readcyclecounter() - t0;
}
__global__ void k_shfl(int n, double* sink, unsigned long long* out) {
  unsigned long long t0 = __builtin_readcyclecounter();
  double v = threadIdx.x;
  for (int i = 0; i < n; ++i) { for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off)) + 1.0; }
  sink[threadIdx.x] = v;
  if (threadIdx.x == 0) out[0] = __builtin_readcyclecounter() - t0;
}
int main() {
  const size_t N = 1u << 26;   // 256 MiB of indices
  std::vector<unsigned> h(N);
  std::iota(h.begin(), h.end(), 0u);
  std::mt19937 rng(1);
  // random cyclic permutation in blocks to keep setup fast
  for (size_t i = N - 1; i > 0; --i) { size_t j = rng() % i; std::swap(h[i], h[j]); }
  unsigned *d_next, *d_sink, *d_buf; unsigned long long* d_out; double* d_ds;
  hipMalloc(&d_next, N * 4); hipMalloc(&d_sink, 4096); hipMalloc(&d_out, 64); hipMalloc(&d_buf, (1u << 24) * 4); hipMalloc(&d_ds, 8192);
  hipMemcpy(d_next, h.data(), N * 4, hipMemcpyHostToDevice);
  auto run = [&](const char* name, auto launch, int n) {
    launch(); hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    launch(); hipDeviceSynchronize();
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    unsigned long long c = 0; hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost);
    printf("%-28s n=%d  %.3f us/iter  %.0f ticks/iter (tick = %.3f ns)\n", name, n, us / n, (double)c / n, us * 1000.0 / (double)c);
  };
  for (int threads : {64, 512}) {
    printf("-- %d threads --\n", threads);
    run("barrier", [&] { hipLaunchKernelGGL(k_barrier, dim3(1), dim3(threads), 0, 0, 100000, d_out); }, 100000);
    run("dependent global load", [&] { hipLaunchKernelGGL(k_chase, dim3(1), dim3(threads), 0, 0, d_next, 20000, d_sink, d_out); }, 20000);
    run("wave0 load + barrier", [&] { hipLaunchKernelGGL(k_chase_barrier, dim3(1), dim3(threads), 0, 0, d_next, 20000, d_sink, d_out); }, 20000);
    run("random store + barrier", [&] { hipLaunchKernelGGL(k_store_barrier, dim3(1), dim3(threads), 0, 0, d_buf, d_next, 20000, d_out); }, 20000);
    run("6-step f64 shfl reduction", [&] { hipLaunchKernelGGL(k_shfl, dim3(1), dim3(threads), 0, 0, 20000, d_ds, d_out); }, 20000);
  }
  return 0;
}
