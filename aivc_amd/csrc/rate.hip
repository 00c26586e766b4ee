// rate.hip -- rate ESTIMATION (logging only, never on the coded path): the reference prints an estimated rate next to
// the real one (src/real_life/bitstream.py:307-329, src/real_life/encode.py:153-170) from the probabilities its entropy
// models give the coded symbols (src/layers/entropy_coding/entropy_coder.py:18-30, pdf_estimator.py:27-65,185-202).
// Elementwise kernels + one deterministic fp64 sum (fixed lanes, fixed tree: the same bits on the CPU twin).
#include "common.h"

namespace aivc {

// lane j of AIVC_RATE_LANES adds its elements i = j, j + L, j + 2L, ... in that order
template <typename F>
__global__ void __launch_bounds__(256) rate_lanes_kernel(size_t n, double *lanes, F term) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  double acc = 0.0;
  for (size_t i = j; i < n; i += AIVC_RATE_LANES) acc = acc + term(i);
  lanes[j] = acc;
}
// lanes[j] += lanes[j + s] for s = L/2, L/4, ..., 1
__global__ void __launch_bounds__(1024) rate_tree_kernel(double *lanes, double *sum) {
  for (int s = AIVC_RATE_LANES / 2; s >= 1; s >>= 1) {
    for (int j = threadIdx.x; j < s; j += 1024) lanes[j] = lanes[j] + lanes[j + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *sum = lanes[0];
}

struct BoundsTerm {
  const uint32_t *b;
  __device__ double operator()(size_t i) const { return aivc_rate_of_bounds(b[i]); }
};
struct ProbTerm {
  const float *p;
  float lo, hi;
  float *rate;
  __device__ double operator()(size_t i) const {
    const float r = aivc_rate_of_prob(p[i], lo, hi);
    if (rate) rate[i] = r;
    return (double)r;
  }
};

__global__ void __launch_bounds__(256) laplace_prob_kernel(const float *y, const float *mu, const float *sigma, size_t n, float *prob) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float m = mu ? mu[i] : 0.0f;
  prob[i] = aivc_laplace_bin_prob(y[i], m, sigma[i]);
}

__global__ void __launch_bounds__(256) table_prob_kernel(const float *x, const float *cdf, size_t n, size_t hw, int c, float *prob) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int ch = (int)((i / hw) % (size_t)c);
  prob[i] = aivc_table_bin_prob(x[i], cdf + (size_t)ch * AIVC_LP);
}

}  // namespace aivc

using namespace aivc;

AIVC_EXPORT int aivc_bounds_rate(const uint32_t *bounds, size_t n, double *lanes, double *sum, aivc_stream_t stream) {
  if (!lanes || !sum || (n && !bounds)) return AIVC_ERR_ARG;
  hipLaunchKernelGGL(rate_lanes_kernel<BoundsTerm>, dim3(AIVC_RATE_LANES / 256), dim3(256), 0, to_stream(stream), n, lanes, BoundsTerm{bounds});
  hipLaunchKernelGGL(rate_tree_kernel, dim3(1), dim3(1024), 0, to_stream(stream), lanes, sum);
  return check_launch("bounds_rate");
}

AIVC_EXPORT int aivc_rate_bits(const float *prob, size_t n, float p_min, float p_max, float *rate, double *lanes, double *sum,
                               aivc_stream_t stream) {
  if (!lanes || !sum || (n && !prob) || !(p_min > 0.0f) || !(p_max >= p_min)) return AIVC_ERR_ARG;
  hipLaunchKernelGGL(rate_lanes_kernel<ProbTerm>, dim3(AIVC_RATE_LANES / 256), dim3(256), 0, to_stream(stream), n, lanes,
                     ProbTerm{prob, p_min, p_max, rate});
  hipLaunchKernelGGL(rate_tree_kernel, dim3(1), dim3(1024), 0, to_stream(stream), lanes, sum);
  return check_launch("rate_bits");
}

AIVC_EXPORT int aivc_laplace_prob(const float *y, const float *mu, const float *sigma, size_t n, float *prob, aivc_stream_t stream) {
  if (n == 0) return AIVC_OK;
  if (!y || !sigma || !prob) return AIVC_ERR_ARG;
  hipLaunchKernelGGL(laplace_prob_kernel, dim3(cdiv(n, 256)), dim3(256), 0, to_stream(stream), y, mu, sigma, n, prob);
  return check_launch("laplace_prob");
}

AIVC_EXPORT int aivc_table_prob(const float *x, const float *cdf_f32, size_t n, size_t hw, int32_t c, float *prob, aivc_stream_t stream) {
  if (n == 0) return AIVC_OK;
  if (!x || !cdf_f32 || !prob || c <= 0 || hw == 0) return AIVC_ERR_ARG;
  hipLaunchKernelGGL(table_prob_kernel, dim3(cdiv(n, 256)), dim3(256), 0, to_stream(stream), x, cdf_f32, n, hw, c, prob);
  return check_launch("table_prob");
}
