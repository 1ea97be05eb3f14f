// tools/micro/xcd_pingpong.hip -- how fast can two workgroups talk, on the same XCD and across XCDs, with which accesses?
// Workgroup 0 and workgroup `partner` play ping-pong on two flags in device memory (N rounds); everything in between exits.
// hipcc --offload-arch=gfx950 -O3 xcd_pingpong.hip -o xcd_pingpong && ./xcd_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE> __device__ __forceinline__ uint32_t ld(uint32_t* p) {
  if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (MODE == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (MODE == 2) return *(volatile uint32_t*)p;
  if (MODE == 3) return __hip_atomic_fetch_or(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return __hip_atomic_fetch_or(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int MODE> __device__ __forceinline__ void st(uint32_t* p, uint32_t v) {
  if (MODE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if (MODE == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else if (MODE == 2) *(volatile uint32_t*)p = v;
  else if (MODE == 3) (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE>
__global__ void pingpong(uint32_t* flags, unsigned long long* out, uint32_t partner, uint32_t rounds) {
  const uint32_t b = blockIdx.x;
  if (b != 0 && b != partner) return;
  if (threadIdx.x != 0) return;
  uint32_t xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  uint32_t* A = flags;          // main -> partner
  uint32_t* B = flags + 64;     // partner -> main (another cache line)
  const unsigned long long limit = 1ull << 22;
  unsigned long long t0 = __builtin_readcyclecounter();
  uint32_t fails = 0;
  if (b == 0) {
    for (uint32_t i = 1; i <= rounds; ++i) {
      st<MODE>(A, i);
      unsigned long long spins = 0;
      while (ld<MODE>(B) != i) { if (++spins > limit) { fails = i; break; } }
      if (fails) break;
    }
    // let a stuck partner go (agent scope always gets there)
    __hip_atomic_store(A, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out[0] = __builtin_readcyclecounter() - t0; out[1] = fails; out[2] = xcc & 0xF;
  } else {
    for (uint32_t i = 1; i <= rounds; ++i) {
      unsigned long long spins = 0;
      uint32_t v;
      while ((v = ld<MODE>(A)) != i) {
        if (v == 0xFFFFFFFFu) { fails = i; break; }
        // (the escape check uses another kind of load: only now and then, so that it cannot help the mode under test)
        if ((++spins & 0xFFFFull) == 0 && __hip_atomic_load(A, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0xFFFFFFFFu) { fails = i; break; }
        if (spins > limit) { fails = i; break; }
      }
      if (fails) break;
      st<MODE>(B, i);
    }
    out[3] = fails; out[4] = xcc & 0xF;
  }
}

template <int MODE> int run(const char* name, uint32_t partner) {
  uint32_t* flags; unsigned long long* out;
  CHECK(hipMalloc(&flags, 4096)); CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(flags, 0, 4096)); CHECK(hipMemset(out, 0, 64));
  const uint32_t rounds = 2000;
  hipLaunchKernelGGL(pingpong<MODE>, dim3(partner + 1), dim3(64), 0, 0, flags, out, partner, rounds);
  CHECK(hipDeviceSynchronize());
  unsigned long long h[8];
  CHECK(hipMemcpy(h, out, 64, hipMemcpyDeviceToHost));
  printf("%-34s partner %3u  xcc %llu/%llu  %s  %8.0f cycles per round trip\n", name, partner, h[2], h[4],
         (h[1] || h[3]) ? "STUCK (not visible)" : "ok", (double)h[0] / rounds);
  (void)hipFree(flags); (void)hipFree(out);
  return 0;
}

int main() {
  for (uint32_t partner : {8u, 16u, 1u, 2u, 3u, 4u, 5u, 6u, 7u}) {
    run<0>("agent-scope atomic load/store", partner);
    run<4>("agent-scope atomic RMW", partner);
    run<3>("workgroup-scope atomic RMW", partner);
    run<1>("workgroup-scope atomic load/store", partner);
    run<2>("volatile load/store", partner);
  }
  return 0;
}
