// metrics.hip -- quality metrics on the device, fp64 (gfx950 runs fp64 VALU at the fp32 rate):
//   aivc_sq_err      sum of squared differences of two planes
//   aivc_ssim_means  mean SSIM / mean contrast-structure of one scale ("valid" Gaussian window)
//   aivc_pool2x2     the 2x2 mean between MS-SSIM scales (two edge rules for odd sizes)
// The window is separable (the reference's 2-D windows are outer products of a normalised 1-D Gaussian);
// every reduction is a fixed-order tree -- no atomics -- so results are reproducible run to run.
#include "common.h"

namespace aivc {

constexpr int MT = 16;   // output tile edge
constexpr int MAXW = 11; // largest window

struct SsimArgs {
  const double *a, *b;
  int h, w, ws, tiles_x, tiles_y;
  double win[MAXW];
  double c1, c2;
  double *partial;  // [n][tiles][2]
};

__device__ __forceinline__ double block_sum_256(double v, double *red) {
  const int t = threadIdx.x;
  red[t] = v;
  __syncthreads();
#pragma unroll
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) red[t] = red[t] + red[t + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void ssim_tile_kernel(SsimArgs g) {
  constexpr int TI = MT + MAXW - 1;  // input tile edge (26)
  __shared__ double ta[TI][TI + 1], tb[TI][TI + 1];
  __shared__ double hx[5][TI][MT];
  __shared__ double red[256];
  const int tile = blockIdx.x, n = blockIdx.y;
  const int ty0 = (tile / g.tiles_x) * MT, tx0 = (tile % g.tiles_x) * MT;
  const double *a = g.a + (size_t)n * g.h * g.w, *b = g.b + (size_t)n * g.h * g.w;
  const int ws = g.ws;
  for (int i = threadIdx.x; i < TI * TI; i += 256) {
    const int r = i / TI, c = i % TI;
    const int y = ty0 + r, x = tx0 + c;
    const bool ok = y < g.h && x < g.w;
    ta[r][c] = ok ? a[(size_t)y * g.w + x] : 0.0;
    tb[r][c] = ok ? b[(size_t)y * g.w + x] : 0.0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TI * MT; i += 256) {
    const int r = i / MT, c = i % MT;
    double s1 = 0, s2 = 0, s11 = 0, s22 = 0, s12 = 0;
    for (int k = 0; k < ws; ++k) {
      const double wv = g.win[k], av = ta[r][c + k], bv = tb[r][c + k];
      s1 += wv * av;
      s2 += wv * bv;
      s11 += wv * (av * av);
      s22 += wv * (bv * bv);
      s12 += wv * (av * bv);
    }
    hx[0][r][c] = s1;
    hx[1][r][c] = s2;
    hx[2][r][c] = s11;
    hx[3][r][c] = s22;
    hx[4][r][c] = s12;
  }
  __syncthreads();
  const int ly = threadIdx.x / MT, lx = threadIdx.x % MT;
  const int oy = ty0 + ly, ox = tx0 + lx;
  double ssim = 0.0, cs = 0.0;
  if (oy < g.h - ws + 1 && ox < g.w - ws + 1) {
    double mu1 = 0, mu2 = 0, e11 = 0, e22 = 0, e12 = 0;
    for (int k = 0; k < ws; ++k) {
      const double wv = g.win[k];
      mu1 += wv * hx[0][ly + k][lx];
      mu2 += wv * hx[1][ly + k][lx];
      e11 += wv * hx[2][ly + k][lx];
      e22 += wv * hx[3][ly + k][lx];
      e12 += wv * hx[4][ly + k][lx];
    }
    const double mu11 = mu1 * mu1, mu22 = mu2 * mu2, mu12 = mu1 * mu2;
    const double v1 = 2.0 * (e12 - mu12) + g.c2;
    const double v2 = (e11 - mu11) + (e22 - mu22) + g.c2;
    cs = v1 / v2;
    ssim = ((2.0 * mu12 + g.c1) * v1) / ((mu11 + mu22 + g.c1) * v2);
  }
  const double bs = block_sum_256(ssim, red);
  const double bc = block_sum_256(cs, red);
  if (threadIdx.x == 0) {
    double *p = g.partial + ((size_t)n * gridDim.x + tile) * 2;
    p[0] = bs;
    p[1] = bc;
  }
}

// out[n][q] = (sum over `count` partials of component q) * scale; one workgroup per n
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double *partial, int count, int comps, double scale,
                                                            double *out) {
  __shared__ double red[256];
  const int n = blockIdx.x;
  for (int q = 0; q < comps; ++q) {
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += 256) s += partial[((size_t)n * count + i) * comps + q];
    const double t = block_sum_256(s, red);
    if (threadIdx.x == 0) out[n * comps + q] = t * scale;
  }
}

__global__ __launch_bounds__(256) void pool2x2_kernel(const double *in, int n, int h, int w, int edge, double *out) {
  const int h2 = (h + 1) / 2, w2 = (w + 1) / 2;
  const size_t total = (size_t)n * h2 * w2;
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int x = (int)(gid % w2), y = (int)((gid / w2) % h2);
  const size_t img = gid / ((size_t)w2 * h2);
  const double *p = in + img * (size_t)h * w;
  // one sample past the end: edge = 0 mirrors without repeating the border (ReflectionPad2d), 1 repeats it
  auto idx = [edge](int i, int len) { return i < len ? i : (edge ? len - 1 : (len >= 2 ? len - 2 : 0)); };
  const int y0 = 2 * y, y1 = idx(2 * y + 1, h), x0 = 2 * x, x1 = idx(2 * x + 1, w);
  out[gid] = (p[(size_t)y0 * w + x0] + p[(size_t)y0 * w + x1] + p[(size_t)y1 * w + x0] + p[(size_t)y1 * w + x1]) * 0.25;
}

__global__ __launch_bounds__(256) void sq_err_kernel(const double *a, const double *b, size_t count, double *partial) {
  __shared__ double red[256];
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
    const double d = a[i] - b[i];
    s += d * d;
  }
  const double t = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

}  // namespace aivc

using namespace aivc;

AIVC_EXPORT size_t aivc_metrics_workspace(int32_t n, int32_t h, int32_t w) {
  const size_t tiles = (size_t)cdiv((size_t)w, MT) * cdiv((size_t)h, MT);
  const size_t ssim = (size_t)(n > 0 ? n : 1) * tiles * 2;
  return (ssim > 1024 ? ssim : 1024) * sizeof(double);
}

AIVC_EXPORT int aivc_ssim_means(const double *a, const double *b, int32_t n, int32_t h, int32_t w, const double *win,
                                int32_t ws, double c1, double c2, double *workspace, double *out,
                                aivc_stream_t stream) {
  if (!a || !b || !win || !workspace || !out || n <= 0 || ws < 1 || ws > MAXW || h < ws || w < ws) return AIVC_ERR_ARG;
  SsimArgs g;
  g.a = a;
  g.b = b;
  g.h = h;
  g.w = w;
  g.ws = ws;
  const int oh = h - ws + 1, ow = w - ws + 1;
  g.tiles_x = (ow + MT - 1) / MT;
  g.tiles_y = (oh + MT - 1) / MT;
  for (int k = 0; k < MAXW; ++k) g.win[k] = k < ws ? win[k] : 0.0;  // host pointer: copied into the launch
  g.c1 = c1;
  g.c2 = c2;
  g.partial = workspace;
  const int tiles = g.tiles_x * g.tiles_y;
  hipLaunchKernelGGL(ssim_tile_kernel, dim3(tiles, n), dim3(256), 0, to_stream(stream), g);
  int rc = check_launch("ssim_tile");
  if (rc != AIVC_OK) return rc;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(n), dim3(256), 0, to_stream(stream), workspace, tiles, 2,
                     1.0 / ((double)oh * (double)ow), out);
  return check_launch("ssim_reduce");
}

AIVC_EXPORT int aivc_pool2x2(const double *in, int32_t n, int32_t h, int32_t w, int32_t edge, double *out,
                             aivc_stream_t stream) {
  if (!in || !out || n <= 0 || h <= 0 || w <= 0 || edge < 0 || edge > 1) return AIVC_ERR_ARG;
  const size_t total = (size_t)n * ((h + 1) / 2) * ((w + 1) / 2);
  hipLaunchKernelGGL(pool2x2_kernel, dim3(cdiv(total, 256)), dim3(256), 0, to_stream(stream), in, n, h, w, edge, out);
  return check_launch("pool2x2");
}

AIVC_EXPORT int aivc_sq_err(const double *a, const double *b, size_t count, double *workspace, double *out,
                            aivc_stream_t stream) {
  if (!a || !b || !workspace || !out) return AIVC_ERR_ARG;
  const int blocks = count == 0 ? 1 : (int)((count + 255) / 256 < 1024 ? (count + 255) / 256 : 1024);
  hipLaunchKernelGGL(sq_err_kernel, dim3(blocks), dim3(256), 0, to_stream(stream), a, b, count, workspace);
  int rc = check_launch("sq_err");
  if (rc != AIVC_OK) return rc;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, to_stream(stream), workspace, blocks, 1, 1.0, out);
  return check_launch("sq_err_reduce");
}
