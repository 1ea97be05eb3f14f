// glia_amd/csrc/skew.hpp -- the wave-skew build (make skew / make skew2; never shipped).
//
// Round 3 lost a week of GPU time to one shape of bug: an LDS word that every wave reads behind a barrier and ONE thread
// rewrites before the next barrier; a wave that leaves the barrier a few hundred cycles late reads the new value
// (greedy_window_kernel's edge counter).  Such a bug needs adverse timing to show.  This build makes the adverse
// timing the rule: behind EVERY workgroup barrier of the loop kernels -- full_barrier / lds_barrier / __syncthreads and its
// _or / _count forms -- every wave but wave 0 (GLIA_HMT_SKEW=1: the writer is early, the readers late) or wave 0 alone
// (GLIA_HMT_SKEW=2: the writer late) sleeps for eight thousand cycles.  A kernel whose barriers separate every such
// read from its rewrite gives the same bytes under both; the GPU tests are run once against each library
// (tools/skew_tests.sh, record in profiles/).
#pragma once
#include <hip/hip_runtime.h>

#ifdef GLIA_HMT_SKEW
#ifndef GLIA_HMT_SKEW_TICKS
#define GLIA_HMT_SKEW_TICKS 127         // s_sleep argument: x 64 cycles
#endif
#ifndef GLIA_HMT_SKEW_SLEEPS
#define GLIA_HMT_SKEW_SLEEPS 1          // x s_sleep 127 (8128 cycles each: ~20 x the barrier-to-barrier distance of the loops)
#endif
// (bisecting a failure of the skew build: only the barriers on source lines LO .. HI are skewed)
#ifndef GLIA_HMT_SKEW_LO
#define GLIA_HMT_SKEW_LO 0
#endif
#ifndef GLIA_HMT_SKEW_HI
#define GLIA_HMT_SKEW_HI (1 << 30)
#endif
namespace glia {
__device__ __forceinline__ void skew_delay(const int line) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool late = ((GLIA_HMT_SKEW == 1) ? (wave != 0) : (wave == 0)) && line >= GLIA_HMT_SKEW_LO && line <= GLIA_HMT_SKEW_HI;
  if (late) {
#pragma unroll 1
    for (int i = 0; i < GLIA_HMT_SKEW_SLEEPS; ++i) __builtin_amdgcn_s_sleep(GLIA_HMT_SKEW_TICKS);
  }
}
__device__ __forceinline__ void skew_syncthreads(const int line) { __syncthreads(); skew_delay(line); }
__device__ __forceinline__ int skew_syncthreads_or(int p, const int line) { const int r = __syncthreads_or(p); skew_delay(line); return r; }
__device__ __forceinline__ int skew_syncthreads_count(int p, const int line) { const int r = __syncthreads_count(p); skew_delay(line); return r; }
}  // namespace glia
#define __syncthreads() ::glia::skew_syncthreads(__LINE__)
#define __syncthreads_or(p) ::glia::skew_syncthreads_or(p, __LINE__)
#define __syncthreads_count(p) ::glia::skew_syncthreads_count(p, __LINE__)
#define GLIA_SKEW_DELAY(line) ::glia::skew_delay(line)
#else
#define GLIA_SKEW_DELAY(line) do {} while (0)
#endif
