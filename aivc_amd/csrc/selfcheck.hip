// selfcheck.hip -- device-side check of the lean square root / division of the fused GDN epilogues (common.h) against
// the compiler's full IEEE sequences.  A diagnostic entry point: nothing on the coded path calls it.
#include "common.h"

namespace aivc {

// every float in [GDN_SAFE_LO, GDN_SAFE_HI] (120 binades x 2^23 values): sqrt_rn_safe(s) == __builtin_sqrtf(s)
__global__ __launch_bounds__(256) void selfcheck_sqrt_kernel(uint32_t first, uint32_t last, unsigned long long *bad) {
  unsigned long long n = 0;
  for (uint64_t b = (uint64_t)first + (uint64_t)blockIdx.x * 256 + threadIdx.x; b <= last; b += (uint64_t)gridDim.x * 256) {
    const float s = __uint_as_float((uint32_t)b);
    n += __float_as_uint(sqrt_rn_safe(s)) != __float_as_uint(__builtin_sqrtf(s));
  }
  if (n) atomicAdd(bad, n);
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// random (numerator, denominator) pairs: numerators of either sign with |v| in [2^-60, 2^60], denominators the square
// roots' range [2^-30, 2^30]; every fourth pair shares the numerator's mantissa neighbourhood with the denominator's
// (quotients near 1 and near powers of two, where the last correction step decides the rounding)
__global__ __launch_bounds__(256) void selfcheck_div_kernel(uint64_t n_pairs, uint32_t seed, unsigned long long *bad) {
  unsigned long long n = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_pairs; i += (uint64_t)gridDim.x * 256) {
    const uint64_t h = mix64(i + ((uint64_t)seed << 32) + 0x9E3779B97F4A7C15ull), h2 = mix64(h);
    const uint32_t ev = 127u - 60u + (uint32_t)((h >> 23) % 120u), ed = 127u - 30u + (uint32_t)((h2 >> 23) % 60u);
    uint32_t mv = (uint32_t)h & 0x7FFFFFu;
    const uint32_t md = (uint32_t)h2 & 0x7FFFFFu;
    if ((i & 3) == 3) mv = (md + (uint32_t)((h >> 50) & 7u) - 3u) & 0x7FFFFFu;
    const float v = __uint_as_float(((uint32_t)(h >> 63) << 31) | (ev << 23) | mv);
    const float d = __uint_as_float((ed << 23) | md);
    n += __float_as_uint(div_rn_safe(v, d)) != __float_as_uint(v / d);
  }
  if (n) atomicAdd(bad, n);
}

}  // namespace aivc

AIVC_EXPORT int aivc_selfcheck_gdn_math(uint64_t n_div_pairs, uint32_t seed, uint64_t *mismatch, aivc_stream_t stream) {
  if (!mismatch) return AIVC_ERR_ARG;
  hipStream_t s = aivc::to_stream(stream);
  if (hipMemsetAsync(mismatch, 0, 2 * sizeof(uint64_t), s) != hipSuccess) return aivc::check_launch("selfcheck memset");
  const uint32_t first = __builtin_bit_cast(uint32_t, aivc::GDN_SAFE_LO), last = __builtin_bit_cast(uint32_t, aivc::GDN_SAFE_HI);
  hipLaunchKernelGGL(aivc::selfcheck_sqrt_kernel, dim3(4096), dim3(256), 0, s, first, last, reinterpret_cast<unsigned long long *>(mismatch));
  if (n_div_pairs)
    hipLaunchKernelGGL(aivc::selfcheck_div_kernel, dim3(4096), dim3(256), 0, s, n_div_pairs, seed, reinterpret_cast<unsigned long long *>(mismatch) + 1);
  return aivc::check_launch("selfcheck_gdn_math");
}
