// conv_direct.hip -- scalar (one thread per output value) implementation of the conv family.
// It is the always-available path: every shape the ABI accepts runs here, with exactly the
// fmaf-chain arithmetic of include/aivc_hip.h.  The MFMA kernels (conv_mfma.hip) must agree with
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
  if (MODE == AIVC_MODE_TCONV) {
    const int tpad = (k + 1) / 2 - 1;
    for (int ky = 0; ky < k; ++ky) {
      const int ty = oy + tpad - ky;
      if (ty < 0 || (ty & 1) || (ty >> 1) >= H) continue;
      for (int kx = 0; kx < k; ++kx) {
        const int tx = ox + tpad - kx;
        if (tx < 0 || (tx & 1) || (tx >> 1) >= W) continue;
        const float *xp = xn + ((size_t)(ty >> 1) * W + (tx >> 1)) * Ci;
        const float *wp = wrow + (size_t)(ky * k + kx) * Ci;
        for (int ci = 0; ci < Ci; ++ci) acc = __builtin_fmaf(xp[ci], wp[ci], acc);
      }
    }
  } else {
    for (int ky = 0; ky < k; ++ky) {
      int iy = oy * p.stride + ky - p.pad;
      iy = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy);
      for (int kx = 0; kx < k; ++kx) {
        int ix = ox * p.stride + kx - p.pad;
        ix = ix < 0 ? 0 : (ix > W - 1 ? W - 1 : ix);
        const float *xp = xn + ((size_t)iy * W + ix) * Ci;
        const float *wp = wrow + (size_t)(ky * k + kx) * Ci;
        if (MODE == AIVC_MODE_GDN || MODE == AIVC_MODE_IGDN) {
          for (int ci = 0; ci < Ci; ++ci) {
            const float a = xp[ci];
            acc = __builtin_fmaf(a * a, wp[ci], acc);
          }
        } else {
          for (int ci = 0; ci < Ci; ++ci) acc = __builtin_fmaf(xp[ci], wp[ci], acc);
        }
      }
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
