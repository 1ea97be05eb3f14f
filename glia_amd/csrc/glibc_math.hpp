// glia_amd/csrc/glibc_math.hpp -- the host libm's double-precision log2() and log(), restated operation for operation
// for host AND device, so that entropy features (std::log2, util/stats.hxx:145-152) and --logs features (std::log via
// slog, glia_base.hxx:80-81) carry the very bits the reference computes on the CPU.
//
// glibc >= 2.28 implements both with the table-driven algorithm of ARM's optimized-routines
// (sysdeps/ieee754/dbl-64/e_log2.c, e_log.c): x = 2^k z, z in [OFF, 2 OFF), c = centre of z's sub-interval,
// r = z/c - 1, result = k + log(c) + polynomial(r), every step in a fixed order with error-compensated sums.  The
// result is within ~0.55 ulp but NOT correctly rounded, and the build with FMA differs in the last bit from the one
// without -- so the operation order below is that of the machine code of glibc 2.35 / x86-64 (this image's libm,
// read with objdump; tools/gen_glibc_tables.py extracts the coefficient tables from the same binary):
//   log2:  one build only (not an IFUNC in 2.35): the non-FMA path with the {chi, clo} table       -> log2_sse2
//   log :  IFUNC __log_finite: __log_sse2 / __log_avx (same operations)                            -> log_sse2
//                              __log_fma (gcc contracted the polynomial, transcribed from the code) -> log_fma
// glia_hmt_ctx_create probes the host's std::log2 / std::log on a fixed vector and selects the variant that
// reproduces it bit for bit (api.cpp: probe_host_libm); with no match the device falls back to its own libm and the
// context reports the feature as unpinned.  Everything here must be compiled with -ffp-contract=off: fused operations
// are written as __builtin_fma where (and only where) the machine code has them.
#pragma once
#include <stdint.h>

namespace glia {
namespace glibc {

#ifndef __HIP_DEVICE_COMPILE__
#define GLIBC_TABLE_QUALIFIER static const
#else
#define GLIBC_TABLE_QUALIFIER static __device__ const
#endif
#include "glibc_tables.inc"
#undef GLIBC_TABLE_QUALIFIER

// On the device the restatements are real functions (one copy each): inlined at every call site of a large kernel they cost
// more in code size and registers than the calls do (greedy_bc.hip: -8 % loop time with ~40 inlined copies).
#ifdef __HIP_DEVICE_COMPILE__
#define GLIBC_FN __host__ __device__ __noinline__
#else
#define GLIBC_FN __host__ __device__ inline
#endif
__host__ __device__ __forceinline__ double as_f64(uint64_t u) { return __builtin_bit_cast(double, u); }
__host__ __device__ __forceinline__ uint64_t as_u64(double d) { return __builtin_bit_cast(uint64_t, d); }

constexpr uint64_t kOff = 0x3fe6000000000000ull;

// x > 0 finite only matters to the callers (p = c/n in (0, 1]; slog guards x > 0); other inputs get the IEEE answers.
__host__ __device__ inline double special(double x) {
  if (x == 0.0) return -__builtin_inf();
  if (x != x || x < 0.0) return __builtin_nan("");
  return x;   // +inf
}

// ---- log2, glibc 2.35 x86-64 (function at libm.so.6 + 0x2f6b0) --------------------------------------------------
// (the tables are parameters so that a kernel can keep a copy in LDS: H = kLog2Head, T = kLog2Tab, T2 = kLog2Tab2)
GLIBC_FN double log2_sse2_tab(double x, const uint64_t* H, const uint64_t* T, const uint64_t* T2) {
  uint64_t ix = as_u64(x);
  const uint32_t top = (uint32_t)(ix >> 48);
  const uint64_t LO = 0x3feea4af00000000ull /* 1 - 0x1.5b51p-5 */, HI = 0x3ff0b55900000000ull /* 1 + 0x1.6ab2p-5 */;
  const double hi_c = as_f64(H[0]), lo_c = as_f64(H[1]);
  if (ix - LO < HI - LO) {
    if (ix == 0x3ff0000000000000ull) return 0.0;
    const double* B = nullptr; (void)B;
    const double r = x - 1.0;
    const double rhi = as_f64(as_u64(r) & 0xffffffff00000000ull);
    const double rlo = r - rhi;
    const double hi = rhi * hi_c;
    double lo = rlo * hi_c + r * lo_c;
    const double r2 = r * r, r4 = r2 * r2;
#define B_(i) as_f64(H[8 + (i)])
    const double p = r2 * (B_(0) + r * B_(1));
    double y = hi + p;
    lo = ((hi - y) + p) + lo;
    const double q = ((B_(2) + r * B_(3)) + r2 * (B_(4) + r * B_(5))) + r4 * ((B_(6) + r * B_(7)) + r2 * (B_(8) + r * B_(9)));
#undef B_
    lo = r4 * q + lo;
    y = y + lo;
    return y;
  }
  if (top - 0x0010u >= 0x7ff0u - 0x0010u) {
    if (ix * 2 == 0 || ix == 0x7ff0000000000000ull || (top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) return special(x);
    ix = as_u64(x * 0x1p52) - (52ull << 52);    // subnormal: normalise
  }
  const uint64_t tmp = ix - kOff;
  const int i = (int)((tmp >> (52 - 6)) & 63);
  const int k = (int)((int64_t)tmp >> 52);
  const uint64_t iz = ix - (tmp & (0xfffull << 52));
  const double invc = as_f64(T[2 * i]), logc = as_f64(T[2 * i + 1]);
  const double chi = as_f64(T2[2 * i]), clo = as_f64(T2[2 * i + 1]);
  const double z = as_f64(iz), kd = (double)k;
  const double r = ((z - chi) - clo) * invc;
  const double rhi = as_f64(as_u64(r) & 0xffffffff00000000ull);
  const double rlo = r - rhi;
  const double t1 = rhi * hi_c;
  const double t2 = rlo * hi_c + r * lo_c;
  const double t3 = kd + logc;
  const double hi = t3 + t1;
  const double lo = ((t3 - hi) + t1) + t2;
  const double r2 = r * r, r4 = r2 * r2;
#define A_(i) as_f64(H[2 + (i)])
  const double p = ((A_(0) + r * A_(1)) + r2 * (A_(2) + r * A_(3))) + r4 * (A_(4) + r * A_(5));
#undef A_
  return (lo + r2 * p) + hi;
}
__host__ __device__ inline double log2_sse2(double x) { return log2_sse2_tab(x, kLog2Head, kLog2Tab, kLog2Tab2); }
constexpr int kLog2TabWords = 18 + 128 + 128;       // H | T | T2 in one array (LDS copy)

// ---- log, glibc 2.35 x86-64: common range reduction ---------------------------------------------------------------
struct LogArgs { double z, kd, invc, logc, chi, clo; };
__host__ __device__ inline bool log_reduce(double x, uint64_t ix, LogArgs& a, double& special_out) {
  const uint32_t top = (uint32_t)(ix >> 48);
  if (top - 0x0010u >= 0x7ff0u - 0x0010u) {
    if (ix * 2 == 0 || ix == 0x7ff0000000000000ull || (top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) { special_out = special(x); return false; }
    ix = as_u64(x * 0x1p52) - (52ull << 52);
  }
  const uint64_t tmp = ix - kOff;
  const int i = (int)((tmp >> (52 - 7)) & 127);
  const int k = (int)((int64_t)tmp >> 52);
  a.z = as_f64(ix - (tmp & (0xfffull << 52)));
  a.kd = (double)k;
  a.invc = as_f64(kLogTab[2 * i]); a.logc = as_f64(kLogTab[2 * i + 1]);
  a.chi = as_f64(kLogTab2[2 * i]); a.clo = as_f64(kLogTab2[2 * i + 1]);
  return true;
}
#define LA_(i) as_f64(kLogHead[2 + (i)])
#define LB_(i) as_f64(kLogHead[7 + (i)])
constexpr uint64_t kLogLo = 0x3fee000000000000ull /* 1 - 0x1p-4 */, kLogHi = 0x3ff1090000000000ull /* 1 + 0x1.09p-4 */;

// __log_sse2 (libm.so.6 + 0x29200; __log_avx is the same sequence VEX-encoded)
GLIBC_FN double log_sse2(double x) {
  const uint64_t ix = as_u64(x);
  if (ix - kLogLo < kLogHi - kLogLo) {
    if (ix == 0x3ff0000000000000ull) return 0.0;
    const double r = x - 1.0;
    const double r2 = r * r, r3 = r * r2;
    double y = ((LB_(7) + r * LB_(8)) + r2 * LB_(9)) + r3 * LB_(10);
    y = ((LB_(4) + r * LB_(5)) + r2 * LB_(6)) + r3 * y;
    y = ((LB_(1) + r * LB_(2)) + r2 * LB_(3)) + r3 * y;
    y = r3 * y;
    double w = r * 0x1p27;
    const double rhi = (r + w) - w;
    const double rlo = r - rhi;
    w = (rhi * rhi) * LB_(0);
    const double hi = r + w;
    double lo = (r - hi) + w;
    lo = ((LB_(0) * rlo) * (rhi + r)) + lo;
    y = lo + y;
    return y + hi;
  }
  LogArgs a; double sp;
  if (!log_reduce(x, ix, a, sp)) return sp;
  const double ln2hi = as_f64(kLogHead[0]), ln2lo = as_f64(kLogHead[1]);
  const double r = ((a.z - a.chi) - a.clo) * a.invc;
  const double w = a.kd * ln2hi + a.logc;
  const double hi = w + r;
  const double lo = ((w - hi) + r) + a.kd * ln2lo;
  const double r2 = r * r;
  const double q = (LA_(1) + r * LA_(2)) + r2 * (LA_(3) + r * LA_(4));
  return ((lo + r2 * LA_(0)) + (r * r2) * q) + hi;
}

// __log_fma (libm.so.6 + 0x76660): the __FP_FAST_FMA source path, with the contractions gcc made
GLIBC_FN double log_fma(double x) {
  const uint64_t ix = as_u64(x);
  if (ix - kLogLo < kLogHi - kLogLo) {
    if (ix == 0x3ff0000000000000ull) return 0.0;
    const double r = x - 1.0;
    const double r2 = r * r, r3 = r * r2;
    const double p1 = __builtin_fma(r2, LB_(3), __builtin_fma(r, LB_(2), LB_(1)));
    const double p2 = __builtin_fma(r2, LB_(6), __builtin_fma(r, LB_(5), LB_(4)));
    double p3 = __builtin_fma(r2, LB_(9), __builtin_fma(r, LB_(8), LB_(7)));
    p3 = __builtin_fma(r3, LB_(10), p3);
    double P = __builtin_fma(p3, r3, p2);
    P = __builtin_fma(P, r3, p1);
    const double rw = __builtin_fma(r, 0x1p27, r);
    const double rhi = __builtin_fma(-0x1p27, r, rw);
    const double rlo = r - rhi;
    const double rh2 = rhi * rhi;
    const double hi = __builtin_fma(rh2, LB_(0), r);
    double lo = __builtin_fma(rh2, LB_(0), r - hi);
    lo = __builtin_fma(LB_(0) * rlo, r + rhi, lo);
    const double y = __builtin_fma(P, r3, lo);
    return hi + y;
  }
  LogArgs a; double sp;
  if (!log_reduce(x, ix, a, sp)) return sp;
  const double ln2hi = as_f64(kLogHead[0]), ln2lo = as_f64(kLogHead[1]);
  const double r = __builtin_fma(a.z, a.invc, -1.0);
  const double w = __builtin_fma(a.kd, ln2hi, a.logc);
  const double q0 = __builtin_fma(r, LA_(2), LA_(1));
  const double hi = r + w;
  const double r2 = r * r;
  double lo = (w - hi) + r;
  lo = __builtin_fma(a.kd, ln2lo, lo);
  const double r3 = r * r2;
  const double q1 = __builtin_fma(r, LA_(4), LA_(3));
  const double t = __builtin_fma(r2, LA_(0), lo);
  const double q = __builtin_fma(q1, r2, q0);
  return __builtin_fma(r3, q, t) + hi;
}
#undef LA_
#undef LB_

// ---- pow(x, y) for x > 0 normal, 2^-65 < |y log x| < 1024 (e_pow.c; the callers: std::pow(perim, 1.5), type/feat.hxx:78-79) -----
// log_inline gives log(x) = hi + lo in double-double from the {invc, logc, logctail} table, pow multiplies by y, exp_inline
// evaluates exp(ehi + elo) = 2^(k/128) * (1 + tail + polynomial(r)).  The IFUNC __pow_finite picks __pow_fma
// (libm.so.6 + 0x768b0: the __FP_FAST_FMA source paths plus the contractions gcc made, transcribed from the code) or
// __pow_sse2 (the non-FMA source paths: Dekker splits, no contraction possible).
constexpr uint64_t kPowOff = 0x3fe6955500000000ull;
#define PA_(i) as_f64(kPowLogHead[2 + (i)])
#define EC_(i) as_f64(kExpHead[4 + (i)])     /* C2..C5 */
__host__ __device__ inline double pow_exp_fma_tail(double ehi, double elo, bool fma) {
  const double invln2n = as_f64(kExpHead[0]), shift = as_f64(kExpHead[1]), nhi = as_f64(kExpHead[2]), nlo = as_f64(kExpHead[3]);
  double kd = fma ? __builtin_fma(ehi, invln2n, shift) : invln2n * ehi + shift;
  const uint64_t ki = as_u64(kd);
  kd = kd - shift;
  double r = fma ? __builtin_fma(kd, nlo, __builtin_fma(kd, nhi, ehi)) : (ehi + kd * nhi) + kd * nlo;
  r = r + elo;
  const uint64_t idx = 2 * (ki & 127);
  const double tail = as_f64(kExpTab[idx]);
  const double scale = as_f64(kExpTab[idx + 1] + (ki << 45));
  const double r2 = r * r;
  if (fma) {
    const double a = __builtin_fma(r, EC_(1), EC_(0));          // C2 + r C3
    const double b = __builtin_fma(r, EC_(3), EC_(2));          // C4 + r C5
    const double t = __builtin_fma(a, r2, tail + r);
    const double tmp = __builtin_fma(b, r2 * r2, t);
    return __builtin_fma(tmp, scale, scale);
  }
  const double tmp = ((tail + r) + r2 * (EC_(0) + r * EC_(1))) + (r2 * r2) * (EC_(2) + r * EC_(3));
  return scale + scale * tmp;
}
// domain check shared by both variants: everything else (zero, negative, subnormal, inf/nan, over/underflow) is not a perimeter
__host__ __device__ inline bool pow_in_domain(double x, double y) {
  const uint64_t ix = as_u64(x), iy = as_u64(y);
  const uint32_t tx = (uint32_t)(ix >> 52), ty = (uint32_t)(iy >> 52) & 0x7ff;
  return tx - 1u < 0x7feu && ty - 0x3beu < 0x43eu - 0x3beu;
}
GLIBC_FN double pow_fma(double x, double y) {
  const uint64_t ix = as_u64(x);
  const uint64_t tmp = ix - kPowOff;
  const int i = (int)((tmp >> (52 - 7)) & 127);
  const double kd = (double)(int)((int64_t)tmp >> 52);
  const double z = as_f64(ix - (tmp & (0xfffull << 52)));
  const double invc = as_f64(kPowLogTab[4 * i]), logc = as_f64(kPowLogTab[4 * i + 2]), logctail = as_f64(kPowLogTab[4 * i + 3]);
  const double ln2hi = as_f64(kPowLogHead[0]), ln2lo = as_f64(kPowLogHead[1]);
  const double r = __builtin_fma(z, invc, -1.0);
  const double t1 = __builtin_fma(kd, ln2hi, logc);
  const double t2 = t1 + r;
  const double lo1 = __builtin_fma(kd, ln2lo, logctail);
  const double lo2 = (t1 - t2) + r;
  const double ar = PA_(0) * r, ar2 = r * ar, ar3 = r * ar2;
  const double hi = t2 + ar2;
  const double lo3 = __builtin_fma(ar, r, -ar2);
  const double lo4 = (t2 - hi) + ar2;
  const double q = __builtin_fma(ar2, __builtin_fma(__builtin_fma(r, PA_(6), PA_(5)), ar2, __builtin_fma(r, PA_(4), PA_(3))),
                                 __builtin_fma(r, PA_(2), PA_(1)));
  const double lo = __builtin_fma(ar3, q, ((lo1 + lo2) + lo3) + lo4);
  const double lh = hi + lo;
  const double ll = (hi - lh) + lo;
  const double ehi = y * lh;
  const double elo = __builtin_fma(y, ll, __builtin_fma(y, lh, -ehi));
  if (((as_u64(ehi) >> 52) & 0x7ff) - 0x3c9u >= 0x3fu) return ehi == 0.0 || ((as_u64(ehi) >> 52) & 0x7ff) < 0x3c9u ? 1.0 : __builtin_nan("");
  return pow_exp_fma_tail(ehi, elo, true);
}
GLIBC_FN double pow_sse2(double x, double y) {
  const uint64_t ix = as_u64(x), iy = as_u64(y);
  const uint64_t tmp = ix - kPowOff;
  const int i = (int)((tmp >> (52 - 7)) & 127);
  const double kd = (double)(int)((int64_t)tmp >> 52);
  const uint64_t iz = ix - (tmp & (0xfffull << 52));
  const double z = as_f64(iz);
  const double invc = as_f64(kPowLogTab[4 * i]), logc = as_f64(kPowLogTab[4 * i + 2]), logctail = as_f64(kPowLogTab[4 * i + 3]);
  const double ln2hi = as_f64(kPowLogHead[0]), ln2lo = as_f64(kPowLogHead[1]);
  const double zhi = as_f64((iz + (1ull << 31)) & (~0ull << 32));
  const double zlo = z - zhi;
  const double rhi = zhi * invc - 1.0;
  const double rlo = zlo * invc;
  const double r = rhi + rlo;
  const double t1 = kd * ln2hi + logc;
  const double t2 = t1 + r;
  const double lo1 = kd * ln2lo + logctail;
  const double lo2 = (t1 - t2) + r;
  const double ar = PA_(0) * r, ar2 = r * ar, ar3 = r * ar2;
  const double arhi = PA_(0) * rhi, arhi2 = rhi * arhi;
  const double hi = t2 + arhi2;
  const double lo3 = rlo * (ar + arhi);
  const double lo4 = (t2 - hi) + arhi2;
  const double p = ar3 * ((PA_(1) + r * PA_(2)) + ar2 * ((PA_(3) + r * PA_(4)) + ar2 * (PA_(5) + r * PA_(6))));
  const double lo = (((lo1 + lo2) + lo3) + lo4) + p;
  const double lh = hi + lo;
  const double ll = (hi - lh) + lo;
  const double yhi = as_f64(iy & (~0ull << 27)), ylo = y - yhi;
  const double lhi = as_f64(as_u64(lh) & (~0ull << 27));
  const double llo = (lh - lhi) + ll;
  const double ehi = yhi * lhi;
  const double elo = ylo * lhi + y * llo;
  if (((as_u64(ehi) >> 52) & 0x7ff) - 0x3c9u >= 0x3fu) return ehi == 0.0 || ((as_u64(ehi) >> 52) & 0x7ff) < 0x3c9u ? 1.0 : __builtin_nan("");
  return pow_exp_fma_tail(ehi, elo, false);
}
#undef PA_
#undef EC_

}  // namespace glibc

// Variant selection (glia_hmt_ctx_create -> kernels): which restatement reproduces the host's libm.
enum : int { kLibmDevice = 0 /* unpinned: device libm */, kLibmSse2 = 1, kLibmFma = 2 };
struct LibmSel { int log2_variant, log_variant, pow_variant; };

}  // namespace glia
