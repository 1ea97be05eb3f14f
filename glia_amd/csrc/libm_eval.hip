// glia_amd/csrc/libm_eval.hip -- evaluates the device restatements of the host libm's logarithms over an array
// (glia_hmt_libm_eval: parity tests compare the bits with std::log2 / std::log on the host; see glibc_math.hpp).
#include "bc_features.hpp"

namespace glia {

__global__ void libm_eval_kernel(int function, int variant, const double* __restrict__ in, double* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = function == 0 ? feat::host_log2(in[i], variant) : function == 1 ? feat::host_log(in[i], variant) : feat::pow_perim(in[i], 3, variant);
}

int launch_libm_eval(int function, int variant, const double* d_in, double* d_out, int64_t n, hipStream_t stream) {
  if (n == 0) return GLIA_HMT_OK;
  libm_eval_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(function, variant, d_in, d_out, (long long)n);
  GLIA_HIP_TRY(hipGetLastError());
  return GLIA_HMT_OK;
}

}  // namespace glia
