// conv_thin.hip -- transposed convolution with a THIN output (c_out of 3 or 6: the last layer of
// the synthesis transforms, 64 -> 3/6 channels at full resolution).
//
// On the matrix cores N would be padded from 3 to 32 (9 % useful work).  The fp32 VALU has the same
// peak as the fp32 MFMA on gfx950, so this layer runs as a register-tiled VALU kernel instead:
//   - a workgroup owns an 8 x 16 tile of INPUT pixels and produces the 16 x 32 output pixels of all
//     four output-parity classes, so the input patch (tile + 1-pixel halo, all channels) is staged in
//     LDS exactly once per workgroup;
//   - each thread owns one input position and keeps 4 x c_out accumulators (one per parity class);
//   - weights are wave-uniform: they arrive through the scalar cache as SGPR operands of v_fma_f32;
//   - per output value the accumulation is the same (ky, kx) ascending / channel ascending fmaf chain
//     as every other implementation (taps outside the image contribute fmaf(0, w, acc) = acc).
#include "common.h"

namespace aivc {

// taps of output-parity class (pc) along one axis, in ascending kernel index: count and the t-th
template <int KS>
__device__ __host__ constexpr int class_ntaps(int pc) {
  constexpr int TPAD = (KS + 1) / 2 - 1;
  int n = 0;
  for (int k = 0; k < KS; ++k) n += (((pc + TPAD - k) & 1) == 0);
  return n;
}
template <int KS>
__device__ __host__ constexpr int class_tap(int pc, int t) {
  constexpr int TPAD = (KS + 1) / 2 - 1;
  int n = 0;
  for (int k = 0; k < KS; ++k)
    if (((pc + TPAD - k) & 1) == 0) {
      if (n == t) return k;
      ++n;
    }
  return -1;
}

template <int KS, int CO, int TH>
__global__ __launch_bounds__(TH * 16) void thin_tconv_kernel(aivc_conv_params p) {
  constexpr int TW = 16;
  constexpr int NT = TH * TW;
  constexpr int TPAD = (KS + 1) / 2 - 1;
  constexpr int MAXT = ((KS + 1) / 2) * ((KS + 1) / 2);  // taps of the largest class
  extern __shared__ __attribute__((aligned(16))) float patch[];  // [(TH+2)*(TW+2)][Cin + 4]
  const int Cin = p.c_in, H = p.h_in, W = p.w_in;
  const int stride_px = Cin + 4;
  const int tiles_x = (W + TW - 1) / TW;
  const int tx0 = (blockIdx.x % tiles_x) * TW, ty0 = (blockIdx.x / tiles_x) * TH;
  const int n = blockIdx.y;
  constexpr int PW = TW + 2, PH = TH + 2;
  const float *xn = p.x + (size_t)n * H * W * Cin;

  // ---- stage the input patch (zero outside the image) -----------------------------------------
  const int quads = Cin / 4;
  for (int i = threadIdx.x; i < PH * PW * quads; i += NT) {
    const int q = i % quads, pp = i / quads;
    const int iy = ty0 - 1 + pp / PW, ix = tx0 - 1 + pp % PW;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const float4 *>(xn + ((size_t)iy * W + ix) * Cin + q * 4);
    *reinterpret_cast<float4 *>(patch + pp * stride_px + q * 4) = v;
  }
  __syncthreads();

  const int lx = threadIdx.x % TW, ly = threadIdx.x / TW;
  const int qx = tx0 + lx, qy = ty0 + ly;
  float acc[4][CO];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[c][o] = 0.0f;
  const float *center = patch + ((ly + 1) * PW + (lx + 1)) * stride_px;

  // The 4 parity classes are independent fmaf chains: they advance together (step t = t-th tap of
  // each class, then all channels), which gives the VALU 4 x CO chains to interleave.
  // (CO = 6: two classes at a time, 12 chains -- 24 would spill the SGPR file with weights)
  constexpr int PAR = CO <= 3 ? 4 : 2;
#pragma unroll
  for (int c0 = 0; c0 < 4; c0 += PAR)
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    for (int ci = 0; ci < Cin; ci += 4) {
#pragma unroll
      for (int c = c0; c < c0 + PAR; ++c) {
        const int pyc = c >> 1, pxc = c & 1;
        constexpr int dummy = 0;
        (void)dummy;
        const int nx = class_ntaps<KS>(pxc);
        if (t < class_ntaps<KS>(pyc) * nx) {
          const int ky = class_tap<KS>(pyc, t / nx), kx = class_tap<KS>(pxc, t % nx);
          const int dy = (pyc + TPAD - ky) >> 1, dx = (pxc + TPAD - kx) >> 1;
          const float4 xv = *reinterpret_cast<const float4 *>(center + (dy * PW + dx) * stride_px + ci);
          const float *wt = p.w + (size_t)(ky * KS + kx) * Cin + ci;
#pragma unroll
          for (int o = 0; o < CO; ++o) {
            const float *wo = wt + (size_t)o * KS * KS * Cin;
            float a = acc[c][o];
            a = __builtin_fmaf(xv.x, wo[0], a);
            a = __builtin_fmaf(xv.y, wo[1], a);
            a = __builtin_fmaf(xv.z, wo[2], a);
            a = __builtin_fmaf(xv.w, wo[3], a);
            acc[c][o] = a;
          }
        }
      }
    }
  }
  if (qx >= W || qy >= H) return;
  Epilogue ep{p.bias, p.mul, p.res, p.x, p.y, p.act1, p.act2, AIVC_MODE_TCONV};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const size_t opix = ((size_t)n * p.h_out + (2 * qy + (c >> 1))) * p.w_out + (2 * qx + (c & 1));
#pragma unroll
    for (int o = 0; o < CO; ++o) ep.store(opix, o, CO, acc[c][o]);
  }
}

bool conv2d_thin_supported(const aivc_conv_params &p) {
  if (p.mode != AIVC_MODE_TCONV || p.gdn) return false;
  if (p.c_out != 3 && p.c_out != 6) return false;
  if (p.ksize != 3 && p.ksize != 5) return false;
  return p.c_in % 4 == 0 && p.c_in >= 16 && p.c_in <= 128;
}

template <int KS, int CO>
static int launch_thin(const aivc_conv_params &p, hipStream_t s) {
  constexpr int TH = 8;  // 8 x 16 input pixels per workgroup: <= 49 KB of LDS at 64 channels (3 groups per CU)
  const size_t lds = (size_t)(TH + 2) * 18 * (p.c_in + 4) * sizeof(float);
  const int tiles = ((p.w_in + 15) / 16) * ((p.h_in + TH - 1) / TH);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(thin_tconv_kernel<KS, CO, TH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((thin_tconv_kernel<KS, CO, TH>), dim3(tiles, p.n), dim3(TH * 16), lds, s, p);
  return check_launch("thin_tconv");
}

int conv2d_thin(const aivc_conv_params &p, hipStream_t s) {
  if (!conv2d_thin_supported(p)) return AIVC_ERR_UNSUPPORTED;
  if (p.ksize == 5) return p.c_out == 3 ? launch_thin<5, 3>(p, s) : launch_thin<5, 6>(p, s);
  return p.c_out == 3 ? launch_thin<3, 3>(p, s) : launch_thin<3, 6>(p, s);
}

}  // namespace aivc
