// conv_direct.hip -- scalar (one thread per output value) implementation of the conv family.
// It is the always-available path: every shape the ABI accepts runs here, with exactly the
// fmaf-chain arithmetic (AIVC_K_ORDER) of include/aivc_hip.h.  The MFMA kernels (conv_mfma.hip) must agree with
// it bit for bit; thin layers (c_out of 3 or 6) stay on this path.
#include "common.h"

namespace aivc {

template <int MODE>
__global__ __launch_bounds__(256) void conv_direct_kernel(aivc_conv_params p) {
  const int Co = p.c_out;
  const size_t total = (size_t)p.n * p.h_out * p.w_out * Co;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int co = (int)(gid % Co);
  const size_t opix = gid / Co;
  const int ox = (int)(opix % p.w_out);
  const int oy = (int)((opix / p.w_out) % p.h_out);
  const int n = (int)(opix / ((size_t)p.w_out * p.h_out));
  const int k = p.ksize, H = p.h_in, W = p.w_in, Ci = p.c_in;
  const float *wrow = p.w + (size_t)co * k * k * Ci;
  const float *xn = p.x + (size_t)n * H * W * Ci;
  float acc = 0.0f;
  // reduction index kk = t * Ci + ci over the tap list of this output (all k*k taps; transposed conv: the taps
  // of the pixel's parity class), walked in groups of 8 in AIVC_K_ORDER -- the arithmetic contract of
  // include/aivc_hip.h.  Out-of-image taps of the transposed conv are exact no-ops and skipped.
  int ky0 = 0, kx0 = 0, nkx = k, ntap = k * k, tstep = 1, tpad = 0;
  if (MODE == AIVC_MODE_TCONV) {
    tpad = (k + 1) / 2 - 1;
    ky0 = (oy + tpad) & 1;
    kx0 = (ox + tpad) & 1;
    nkx = (k - kx0 + 1) / 2;
    ntap = ((k - ky0 + 1) / 2) * nkx;
    tstep = 2;
  }
  const int K = ntap * Ci;
  for (int g = 0; g < K; g += 8) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kk = g + AIVC_K_ORDER(i);
      if (kk >= K) continue;
      const int t = kk / Ci, ci = kk - t * Ci;
      const int ky = ky0 + tstep * (t / nkx), kx = kx0 + tstep * (t % nkx);
      int iy, ix;
      if (MODE == AIVC_MODE_TCONV) {
        iy = (oy + tpad - ky) >> 1;  // even by construction of the class
        ix = (ox + tpad - kx) >> 1;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      } else {
        iy = oy * p.stride + ky - p.pad;
        ix = ox * p.stride + kx - p.pad;
        iy = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy);
        ix = ix < 0 ? 0 : (ix > W - 1 ? W - 1 : ix);
      }
      float a = xn[((size_t)iy * W + ix) * Ci + ci];
      if (MODE == AIVC_MODE_GDN || MODE == AIVC_MODE_IGDN) a = a * a;
      acc = __builtin_fmaf(a, wrow[(size_t)(ky * k + kx) * Ci + ci], acc);
    }
  }
  Epilogue ep{p.bias, p.mul, p.res, p.x, p.y, p.act1, p.act2, MODE};
  ep.store(opix, co, Co, acc);
}

int conv2d_direct(const aivc_conv_params &p, hipStream_t s) {
  const size_t total = (size_t)p.n * p.h_out * p.w_out * p.c_out;
  const unsigned grid = cdiv(total, 256);
  switch (p.mode) {
    case AIVC_MODE_CONV: hipLaunchKernelGGL(conv_direct_kernel<AIVC_MODE_CONV>, dim3(grid), dim3(256), 0, s, p); break;
    case AIVC_MODE_TCONV: hipLaunchKernelGGL(conv_direct_kernel<AIVC_MODE_TCONV>, dim3(grid), dim3(256), 0, s, p); break;
    case AIVC_MODE_GDN: hipLaunchKernelGGL(conv_direct_kernel<AIVC_MODE_GDN>, dim3(grid), dim3(256), 0, s, p); break;
    case AIVC_MODE_IGDN: hipLaunchKernelGGL(conv_direct_kernel<AIVC_MODE_IGDN>, dim3(grid), dim3(256), 0, s, p); break;
    default: return AIVC_ERR_UNSUPPORTED;
  }
  return check_launch("conv_direct");
}

}  // namespace aivc
