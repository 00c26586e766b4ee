// pixel_ops.hip -- HBM-bound frame / latent kernels: 4:2:0 <-> 4:4:4, motion-compensation warp +
// blend, reconstruction + 8-bit cast, hyper-prior parameter split, gains, (de)quantisation.
// One thread per pixel (channels are innermost, so a pixel's channels sit in one or two
// cache lines); grids are sized >> 256 CUs for every frame size the codec handles.
#include "common.h"

namespace aivc {

// ---------------------------------------------------------------- 4:2:0 -> 4:4:4 (InputLayer)
// One workgroup converts a run of 1024 pixels of one row, 4 pixels per thread spaced 256 apart (so every load and
// every 16-byte store of a wave is contiguous across its lanes); no integer division (row and image come from
// the grid), and for 8-bit planes the level k / 255.0f -- which must be the correctly rounded quotient the
// reference's to_tensor produces -- from a 256-entry table built once per workgroup in LDS instead of three IEEE
// divisions per pixel.
template <typename T>
__global__ __launch_bounds__(256) void yuv420_to_444_kernel(const T *__restrict__ y, const T *__restrict__ u,
                                                            const T *__restrict__ v, int n, int h, int w,
                                                            float *__restrict__ out, int c_store, int c_off,
                                                            int zero_pad) {
  __shared__ float lut[256];
  if (sizeof(T) == 1) {
    lut[threadIdx.x] = (float)threadIdx.x / 255.0f;
    __syncthreads();
  }
  const int r = blockIdx.y, b = blockIdx.z;
  const int hc = (h + 1) / 2, wc = (w + 1) / 2;
  const T *yr = y + ((size_t)b * h + r) * w;
  const T *ur = u + ((size_t)b * hc + r / 2) * wc;
  const T *vr = v + ((size_t)b * hc + r / 2) * wc;
  float *orow = out + ((size_t)b * h + r) * w * c_store + c_off;
  const bool vec = zero_pad && ((c_store & 3) == 0) && ((c_off & 3) == 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int x = blockIdx.x * 1024 + i * 256 + (int)threadIdx.x;
    if (x < w) {
      float fy, fu, fv;
      if (sizeof(T) == 1) {
        fy = lut[(uint8_t)yr[x]];
        fu = lut[(uint8_t)ur[x >> 1]];
        fv = lut[(uint8_t)vr[x >> 1]];
      } else {
        fy = (float)yr[x];
        fu = (float)ur[x >> 1];
        fv = (float)vr[x >> 1];
      }
      float *o = orow + (size_t)x * c_store;
      if (vec) {
        *reinterpret_cast<float4 *>(o) = make_float4(fy, fu, fv, 0.0f);
      } else {
        o[0] = fy;
        o[1] = fu;
        o[2] = fv;
        if (zero_pad) o[3] = 0.0f;
      }
    }
  }
}

// ---------------------------------------------------------------- padded multi-image conv input
struct PackArgs {
  aivc_image_src src[AIVC_MAX_IMAGES];
  int n_img, n, h, w;
  float *out;
};
// A lane owns one 16-byte chunk = (pixel, image): consecutive lanes write consecutive chunks of the row, so each
// store instruction covers 1024 contiguous bytes whatever the number of images (a per-image pass stores 16 of
// every 32 / 48 bytes).  Row and batch index come from the grid; k / 255.0f from a table in LDS.
__global__ __launch_bounds__(256) void pack_images_kernel(PackArgs a) {
  __shared__ float lut[256];
  lut[threadIdx.x] = (float)threadIdx.x / 255.0f;
  __syncthreads();
  const int r = blockIdx.y, b = blockIdx.z;
  const int w = a.w, h = a.h, hc = (h + 1) / 2, wc = (w + 1) / 2, ni = a.n_img;
  const int chunks = w * ni;
  float4 *orow = reinterpret_cast<float4 *>(a.out + ((size_t)b * h + r) * (size_t)w * 4 * ni);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = blockIdx.x * 1024 + i * 256 + (int)threadIdx.x;
    if (c >= chunks) break;
    const int x = ni == 1 ? c : (ni == 2 ? c >> 1 : (int)(((unsigned)c * 43691u) >> 17));  // c / 3 for c < 98304
    const int img = c - x * ni;
    const aivc_image_src &s = a.src[img];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s.y) {
      v.x = lut[s.y[((size_t)b * h + r) * w + x]];
      v.y = lut[s.u[((size_t)b * hc + r / 2) * wc + (x >> 1)]];
      v.z = lut[s.v[((size_t)b * hc + r / 2) * wc + (x >> 1)]];
    } else if (s.f) {
      const float *f = s.f + (((size_t)b * h + r) * w + x) * s.f_channels;
      v.x = f[0];
      v.y = f[1];
      v.z = f[2];
    }
    orow[c] = v;
  }
}

// ---------------------------------------------------------------- reconstruction tail
__device__ __forceinline__ float cast8(float x, uint8_t *byte) {
  float c = x < 0.0f ? 0.0f : x;
  c = c > 1.0f ? 1.0f : c;
  const float r = __builtin_rintf(255.0f * c);
  *byte = (uint8_t)r;
  return r / 255.0f;
}

struct RecArgs {
  const float *x, *skip;
  int n, hx, wx, cx, cs, h, w;
  float *y, *u, *v;
  uint8_t *y8, *u8, *v8;
};

__device__ __forceinline__ float xhat(const RecArgs &a, int b, int r, int c, int ch) {
  float val = a.x[(((size_t)b * a.hx + r) * a.wx + c) * a.cx + ch];
  if (a.skip) val = val + a.skip[(((size_t)b * a.h + r) * a.w + c) * a.cs + ch];
  return val;
}

// one thread per chroma sample: writes U, V and the (up to) 2x2 luma samples it covers
// VEC: x has 3 channels and an even row length, skip (if any) 4, the frame has even sides: the two pixels of a row
// of the block are 24 contiguous bytes of x (three 8-byte loads) and two 16-byte loads of skip instead of 12 scalar
// loads; same arithmetic per value.
template <bool VEC>
__global__ __launch_bounds__(256) void frame_to_yuv420_kernel(RecArgs a) {
  const int hc = (a.h + 1) / 2, wc = (a.w + 1) / 2;
  const int hf = a.h / 2, wf = a.w / 2;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)a.n * hc * wc;
  if (gid >= total) return;
  const int c = (int)(gid % wc), r = (int)((gid / wc) % hc), b = (int)(gid / ((size_t)wc * hc));
  if constexpr (VEC) {
    float px[2][2][3];  // [row][column][channel]
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const float2 *xp = reinterpret_cast<const float2 *>(a.x + (((size_t)b * a.hx + 2 * r + dy) * a.wx + 2 * c) * 3);
      const float2 f0 = xp[0], f1 = xp[1], f2 = xp[2];
      px[dy][0][0] = f0.x; px[dy][0][1] = f0.y; px[dy][0][2] = f1.x;
      px[dy][1][0] = f1.y; px[dy][1][1] = f2.x; px[dy][1][2] = f2.y;
      if (a.skip) {
        const float4 *sp = reinterpret_cast<const float4 *>(a.skip + (((size_t)b * a.h + 2 * r + dy) * a.w + 2 * c) * 4);
        const float4 s0 = sp[0], s1 = sp[1];
        px[dy][0][0] = px[dy][0][0] + s0.x; px[dy][0][1] = px[dy][0][1] + s0.y; px[dy][0][2] = px[dy][0][2] + s0.z;
        px[dy][1][0] = px[dy][1][0] + s1.x; px[dy][1][1] = px[dy][1][1] + s1.y; px[dy][1][2] = px[dy][1][2] + s1.z;
      }
    }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      uint8_t b0, b1;
      const float l0 = cast8(px[dy][0][0], &b0), l1 = cast8(px[dy][1][0], &b1);
      const size_t o = ((size_t)b * a.h + 2 * r + dy) * a.w + 2 * c;
      if (a.y) {
        a.y[o] = l0;
        a.y[o + 1] = l1;
      }
      if (a.y8) *reinterpret_cast<uint16_t *>(a.y8 + o) = (uint16_t)b0 | ((uint16_t)b1 << 8);
    }
#pragma unroll
    for (int ch = 1; ch <= 2; ++ch) {
      const float top = 0.5f * px[0][0][ch] + 0.5f * px[0][1][ch];
      const float bot = 0.5f * px[1][0][ch] + 0.5f * px[1][1][ch];
      const float val = 0.5f * top + 0.5f * bot;
      uint8_t byte;
      const float lv = cast8(val, &byte);
      float *dst = ch == 1 ? a.u : a.v;
      uint8_t *dst8 = ch == 1 ? a.u8 : a.v8;
      if (dst) dst[gid] = lv;
      if (dst8) dst8[gid] = byte;
    }
    return;
  }
  for (int dy = 0; dy < 2; ++dy)
    for (int dx = 0; dx < 2; ++dx) {
      const int yy = 2 * r + dy, xx = 2 * c + dx;
      if (yy < a.h && xx < a.w) {
        uint8_t byte;
        const float lv = cast8(xhat(a, b, yy, xx, 0), &byte);
        const size_t o = ((size_t)b * a.h + yy) * a.w + xx;
        if (a.y) a.y[o] = lv;
        if (a.y8) a.y8[o] = byte;
      }
    }
  const int rr = min(r, max(hf - 1, 0)), cc = min(c, max(wf - 1, 0));
  for (int ch = 1; ch <= 2; ++ch) {
    float val = 0.0f;
    if (hf > 0 && wf > 0) {
      const float p00 = xhat(a, b, 2 * rr, 2 * cc, ch), p01 = xhat(a, b, 2 * rr, 2 * cc + 1, ch);
      const float p10 = xhat(a, b, 2 * rr + 1, 2 * cc, ch), p11 = xhat(a, b, 2 * rr + 1, 2 * cc + 1, ch);
      const float top = 0.5f * p00 + 0.5f * p01;
      const float bot = 0.5f * p10 + 0.5f * p11;
      val = 0.5f * top + 0.5f * bot;
    }
    uint8_t byte;
    const float lv = cast8(val, &byte);
    float *dst = ch == 1 ? a.u : a.v;
    uint8_t *dst8 = ch == 1 ? a.u8 : a.v8;
    if (dst) dst[gid] = lv;
    if (dst8) dst8[gid] = byte;
  }
}

__global__ __launch_bounds__(256) void downsample2x_kernel(const float *__restrict__ x, int n, int h, int w, int c,
                                                           int ch0, int nch, float *__restrict__ out) {
  const int hf = h / 2, wf = w / 2;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)n * nch * hf * wf) return;
  const int q = (int)(gid % wf), r = (int)((gid / wf) % hf);
  const int j = (int)((gid / ((size_t)wf * hf)) % nch), b = (int)(gid / ((size_t)wf * hf * nch));
  const float *p00 = x + (((size_t)b * h + 2 * r) * w + 2 * q) * c + ch0 + j;
  const float top = 0.5f * p00[0] + 0.5f * p00[c];
  const float bot = 0.5f * p00[(size_t)w * c] + 0.5f * p00[(size_t)w * c + c];
  out[gid] = 0.5f * top + 0.5f * bot;
}

// ---------------------------------------------------------------- warp (grid_sample bilinear/border)
__device__ __forceinline__ float warp_coord(float pos, int size) {
  const int d = size - 1 > 1 ? size - 1 : 1;
  const float g = 2.0f * pos / (float)d - 1.0f;
  float ix = ((g + 1.0f) / 2.0f) * (float)(size - 1);
  ix = ix < 0.0f ? 0.0f : ix;
  ix = ix > (float)(size - 1) ? (float)(size - 1) : ix;
  return ix;
}

struct WarpTap {
  int o00, o01, o10, o11;  // pixel offsets (or -1 when out of bounds)
  float nw, ne, sw, se;
};
__device__ __forceinline__ WarpTap warp_taps(int h, int w, float fx, float fy, int row, int col) {
  const float ix = warp_coord((float)col + fx, w);
  const float iy = warp_coord((float)row + fy, h);
  const float x0 = __builtin_floorf(ix), y0 = __builtin_floorf(iy);
  const float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
  WarpTap t;
  t.nw = (x1 - ix) * (y1 - iy);
  t.ne = (ix - x0) * (y1 - iy);
  t.sw = (x1 - ix) * (iy - y0);
  t.se = (ix - x0) * (iy - y0);
  const int xi0 = (int)x0, yi0 = (int)y0, xi1 = xi0 + 1, yi1 = yi0 + 1;
  const bool x0ok = xi0 >= 0 && xi0 < w, x1ok = xi1 >= 0 && xi1 < w;
  const bool y0ok = yi0 >= 0 && yi0 < h, y1ok = yi1 >= 0 && yi1 < h;
  t.o00 = (y0ok && x0ok) ? yi0 * w + xi0 : -1;
  t.o01 = (y0ok && x1ok) ? yi0 * w + xi1 : -1;
  t.o10 = (y1ok && x0ok) ? yi1 * w + xi0 : -1;
  t.o11 = (y1ok && x1ok) ? yi1 * w + xi1 : -1;
  return t;
}
__device__ __forceinline__ float warp_apply(const float *img, int c, int ch, const WarpTap &t) {
  float res = 0.0f;
  if (t.o00 >= 0) res = res + img[(size_t)t.o00 * c + ch] * t.nw;
  if (t.o01 >= 0) res = res + img[(size_t)t.o01 * c + ch] * t.ne;
  if (t.o10 >= 0) res = res + img[(size_t)t.o10 * c + ch] * t.sw;
  if (t.o11 >= 0) res = res + img[(size_t)t.o11 * c + ch] * t.se;
  return res;
}

__global__ __launch_bounds__(256) void warp_kernel(const float *__restrict__ x, const float *__restrict__ flow,
                                                   int n, int h, int w, int c, float *__restrict__ out) {
  const size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= (size_t)n * h * w) return;
  const int q = (int)(pix % w), r = (int)((pix / w) % h), b = (int)(pix / ((size_t)w * h));
  const float *img = x + (size_t)b * h * w * c;
  const WarpTap t = warp_taps(h, w, flow[pix * 2], flow[pix * 2 + 1], r, q);
  for (int ch = 0; ch < c; ++ch) out[pix * c + ch] = warp_apply(img, c, ch, t);
}

struct BlendArgs {
  const float *mof, *prev, *next;
  int hm, wm, cm, cr, n, h, w, frame_type, co;
  float *pred, *skip, *x_warp, *alpha_out, *beta_out;
  int row0, rows;  // the launch covers frame rows [row0, row0 + rows); mof and the outputs are indexed by the LOCAL row
};
__global__ __launch_bounds__(256) void warp_blend_kernel(BlendArgs a) {
  const size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= (size_t)a.n * a.rows * a.w) return;
  const int q = (int)(pix % a.w), r = (int)((pix / a.w) % a.rows), b = (int)(pix / ((size_t)a.w * a.rows));
  const int gr = a.row0 + r;  // row of the frame
  const float *m = a.mof + (((size_t)b * a.hm + r) * a.wm + q) * a.cm;
  float alpha = m[0] + 0.5f;
  alpha = alpha < 0.0f ? 0.0f : (alpha > 1.0f ? 1.0f : alpha);
  float beta = m[1] + 0.5f;
  beta = beta < 0.0f ? 0.0f : (beta > 1.0f ? 1.0f : beta);
  float vpx = m[2], vpy = m[3], vnx = m[4], vny = m[5];
  if (a.frame_type == 1) {
    beta = 1.0f;
    vnx = 0.0f;
    vny = 0.0f;
  }
  if (a.alpha_out) a.alpha_out[pix] = alpha;
  if (a.beta_out) a.beta_out[pix] = beta;
  const float *pimg = a.prev + (size_t)b * a.h * a.w * a.cr;
  const float *nimg = a.next + (size_t)b * a.h * a.w * a.cr;
  const WarpTap tp = warp_taps(a.h, a.w, vpx, vpy, gr, q);
  const WarpTap tn = warp_taps(a.h, a.w, vnx, vny, gr, q);
  for (int ch = 0; ch < a.co; ++ch) {
    float xw = 0.0f, pr = 0.0f, sk = 0.0f;
    if (ch < 3) {
      const float wp = warp_apply(pimg, a.cr, ch, tp);
      const float wn = warp_apply(nimg, a.cr, ch, tn);
      const float t1 = beta * wp;
      const float t2 = (1.0f - beta) * wn;
      xw = t1 + t2;
      pr = xw * alpha;
      sk = (1.0f - alpha) * xw;
    }
    if (a.x_warp) a.x_warp[pix * a.co + ch] = xw;
    if (a.pred) a.pred[pix * a.co + ch] = pr;
    if (a.skip) a.skip[pix * a.co + ch] = sk;
  }
}

// The codec's layout (references and outputs stored as 4 channels, 6 flow / mask channels): every tap is ONE 16-byte
// gather instead of three 4-byte ones, the outputs are 16-byte stores.  Same arithmetic per channel, in the same
// order (0 + nw * a, + ne * b, + sw * c, + se * d; out-of-frame taps are skipped, not added as zeros).
__device__ __forceinline__ void warp_apply4(const float *img, const WarpTap &t, float &r0, float &r1, float &r2) {
  r0 = r1 = r2 = 0.0f;
  if (t.o00 >= 0) {
    const float4 v = *reinterpret_cast<const float4 *>(img + (size_t)t.o00 * 4);
    r0 = r0 + v.x * t.nw; r1 = r1 + v.y * t.nw; r2 = r2 + v.z * t.nw;
  }
  if (t.o01 >= 0) {
    const float4 v = *reinterpret_cast<const float4 *>(img + (size_t)t.o01 * 4);
    r0 = r0 + v.x * t.ne; r1 = r1 + v.y * t.ne; r2 = r2 + v.z * t.ne;
  }
  if (t.o10 >= 0) {
    const float4 v = *reinterpret_cast<const float4 *>(img + (size_t)t.o10 * 4);
    r0 = r0 + v.x * t.sw; r1 = r1 + v.y * t.sw; r2 = r2 + v.z * t.sw;
  }
  if (t.o11 >= 0) {
    const float4 v = *reinterpret_cast<const float4 *>(img + (size_t)t.o11 * 4);
    r0 = r0 + v.x * t.se; r1 = r1 + v.y * t.se; r2 = r2 + v.z * t.se;
  }
}
__global__ __launch_bounds__(256) void warp_blend4_kernel(BlendArgs a) {
  const size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= (size_t)a.n * a.rows * a.w) return;
  const int q = (int)(pix % a.w), r = (int)((pix / a.w) % a.rows), b = (int)(pix / ((size_t)a.w * a.rows));
  const int gr = a.row0 + r;  // row of the frame
  const float2 *m = reinterpret_cast<const float2 *>(a.mof + (((size_t)b * a.hm + r) * a.wm + q) * a.cm);
  const float2 m01 = m[0], m23 = m[1], m45 = m[2];
  float alpha = m01.x + 0.5f;
  alpha = alpha < 0.0f ? 0.0f : (alpha > 1.0f ? 1.0f : alpha);
  float beta = m01.y + 0.5f;
  beta = beta < 0.0f ? 0.0f : (beta > 1.0f ? 1.0f : beta);
  float vpx = m23.x, vpy = m23.y, vnx = m45.x, vny = m45.y;
  if (a.frame_type == 1) {
    beta = 1.0f;
    vnx = 0.0f;
    vny = 0.0f;
  }
  if (a.alpha_out) a.alpha_out[pix] = alpha;
  if (a.beta_out) a.beta_out[pix] = beta;
  const float *pimg = a.prev + (size_t)b * a.h * a.w * 4;
  const float *nimg = a.next + (size_t)b * a.h * a.w * 4;
  const WarpTap tp = warp_taps(a.h, a.w, vpx, vpy, gr, q);
  const WarpTap tn = warp_taps(a.h, a.w, vnx, vny, gr, q);
  float wp[3], wn[3];
  warp_apply4(pimg, tp, wp[0], wp[1], wp[2]);
  warp_apply4(nimg, tn, wn[0], wn[1], wn[2]);
  float xw[3], pr[3], sk[3];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float t1 = beta * wp[ch];
    const float t2 = (1.0f - beta) * wn[ch];
    xw[ch] = t1 + t2;
    pr[ch] = xw[ch] * alpha;
    sk[ch] = (1.0f - alpha) * xw[ch];
  }
  if (a.x_warp) reinterpret_cast<float4 *>(a.x_warp)[pix] = make_float4(xw[0], xw[1], xw[2], 0.0f);
  if (a.pred) reinterpret_cast<float4 *>(a.pred)[pix] = make_float4(pr[0], pr[1], pr[2], 0.0f);
  if (a.skip) reinterpret_cast<float4 *>(a.skip)[pix] = make_float4(sk[0], sk[1], sk[2], 0.0f);
}

// ---------------------------------------------------------------- latent ops
__global__ __launch_bounds__(256) void hyper_params_kernel(const float *__restrict__ hs, int n, int hh, int wh,
                                                           int c, int h, int w, float *__restrict__ mu,
                                                           float *__restrict__ sigma) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)n * h * w * c;
  if (gid >= total) return;
  const int ch = (int)(gid % c);
  const size_t pix = gid / c;
  const int q = (int)(pix % w), r = (int)((pix / w) % h), b = (int)(pix / ((size_t)w * h));
  const float *src = hs + (((size_t)b * hh + r) * wh + q) * 2 * c;
  mu[gid] = src[ch];
  float lv = src[c + ch];
  lv = lv < -18.4207f ? -18.4207f : lv;
  lv = lv > 10.0f ? 10.0f : lv;
  sigma[gid] = aivc_expf_det(0.5f * lv);
}

__global__ __launch_bounds__(256) void channel_gain_kernel(const float *__restrict__ in, const float *__restrict__ gain,
                                                           size_t total, int c, float *__restrict__ out) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  out[gid] = gain ? in[gid] * __builtin_fabsf(gain[gid % c]) : in[gid];
}

__global__ __launch_bounds__(256) void gain_interp_kernel(const float *__restrict__ g_r, const float *__restrict__ g_t,
                                                          int c, float l, float *__restrict__ out) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch < c) out[ch] = aivc_gain_interp_one(g_r[ch], g_t[ch], l);
}

__global__ __launch_bounds__(256) void quantize_center_kernel(const float *__restrict__ y, const float *__restrict__ mu,
                                                              const float *__restrict__ gain, size_t total, int c,
                                                              int16_t *__restrict__ q, float *__restrict__ y_hat) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const float m = mu ? mu[gid] : 0.0f;
  float r = __builtin_rintf(mu ? y[gid] - m : y[gid]);
  r = r < -256.0f ? -256.0f : (r > 256.0f ? 256.0f : r);  // the alphabet of the coder: symbols 0 .. 512
  if (q) q[gid] = (int16_t)r;
  if (y_hat) {
    float v = mu ? r + m : r;
    if (gain) v = v * __builtin_fabsf(gain[gid % c]);
    y_hat[gid] = v;
  }
}

__global__ __launch_bounds__(256) void dequantize_kernel(const int16_t *__restrict__ q, const float *__restrict__ mu,
                                                         const float *__restrict__ gain, size_t total, int c,
                                                         float *__restrict__ y_hat) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  float v = mu ? (float)q[gid] + mu[gid] : (float)q[gid];
  if (gain) v = v * __builtin_fabsf(gain[gid % c]);
  y_hat[gid] = v;
}

__global__ __launch_bounds__(256) void pad_channels_kernel(const float *__restrict__ in, size_t npix, int c_in,
                                                           float *__restrict__ out, int c_out) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= npix * c_out) return;
  const int c = (int)(gid % c_out);
  const size_t p = gid / c_out;
  out[gid] = c < c_in ? in[p * c_in + c] : 0.0f;
}

__global__ __launch_bounds__(256) void gdn_reparam_kernel(const float *__restrict__ beta, const float *__restrict__ gamma,
                                                          int c, float beta_bound, float gamma_bound, float pedestal,
                                                          float *__restrict__ beta_eff, float *__restrict__ gamma_eff) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < c) {
    const float b = beta[gid] > beta_bound ? beta[gid] : beta_bound;
    beta_eff[gid] = b * b - pedestal;
  }
  if (gid < c * c) {
    const float g = gamma[gid] > gamma_bound ? gamma[gid] : gamma_bound;
    gamma_eff[gid] = g * g - pedestal;
  }
}

}  // namespace aivc

using namespace aivc;

AIVC_EXPORT int aivc_gdn_reparam(const float *beta, const float *gamma, int32_t c, float beta_bound,
                                 float gamma_bound, float pedestal, float *beta_eff, float *gamma_eff,
                                 aivc_stream_t stream) {
  if (!beta || !gamma || !beta_eff || !gamma_eff || c <= 0) return AIVC_ERR_ARG;
  hipLaunchKernelGGL(gdn_reparam_kernel, dim3(cdiv((size_t)c * c, 256)), dim3(256), 0, to_stream(stream), beta,
                     gamma, c, beta_bound, gamma_bound, pedestal, beta_eff, gamma_eff);
  return check_launch("gdn_reparam");
}

AIVC_EXPORT int aivc_pad_channels(const float *in, size_t npix, int32_t c_in, float *out, int32_t c_out,
                                  aivc_stream_t stream) {
  if (!in || !out || c_out < c_in || c_in <= 0) return AIVC_ERR_ARG;
  if (npix == 0) return AIVC_OK;
  hipLaunchKernelGGL(pad_channels_kernel, dim3(cdiv(npix * c_out, 256)), dim3(256), 0, to_stream(stream), in, npix,
                     c_in, out, c_out);
  return check_launch("pad_channels");
}

AIVC_EXPORT int aivc_yuv420_to_444(const float *y, const float *u, const float *v, int32_t n, int32_t h, int32_t w,
                                   float *out, int32_t c_store, int32_t c_off, int32_t zero_pad,
                                   aivc_stream_t stream) {
  if (!y || !u || !v || !out || n <= 0 || h <= 0 || w <= 0) return AIVC_ERR_ARG;
  if (c_off < 0 || c_off + 3 + (zero_pad ? 1 : 0) > c_store) return AIVC_ERR_ARG;
  if (h > 65535 || n > 65535) return AIVC_ERR_UNSUPPORTED;  // grid y / z limits
  hipLaunchKernelGGL(yuv420_to_444_kernel<float>, dim3(cdiv((size_t)w, 1024), h, n), dim3(256), 0, to_stream(stream),
                     y, u, v, n, h, w, out, c_store, c_off, zero_pad);
  return check_launch("yuv420_to_444");
}

AIVC_EXPORT int aivc_yuv420u8_to_444(const uint8_t *y, const uint8_t *u, const uint8_t *v, int32_t n, int32_t h,
                                     int32_t w, float *out, int32_t c_store, int32_t c_off, int32_t zero_pad,
                                     aivc_stream_t stream) {
  if (!y || !u || !v || !out || n <= 0 || h <= 0 || w <= 0) return AIVC_ERR_ARG;
  if (c_off < 0 || c_off + 3 + (zero_pad ? 1 : 0) > c_store) return AIVC_ERR_ARG;
  if (h > 65535 || n > 65535) return AIVC_ERR_UNSUPPORTED;  // grid y / z limits
  // (four consecutive pixels per thread -- one 4-byte load of Y, 64 contiguous bytes out per lane -- measured 10 %
  // slower at 64 frames and 70 % slower at 4 than this mapping, whose stores are contiguous ACROSS the lanes)
  hipLaunchKernelGGL(yuv420_to_444_kernel<uint8_t>, dim3(cdiv((size_t)w, 1024), h, n), dim3(256), 0,
                     to_stream(stream), y, u, v, n, h, w, out, c_store, c_off, zero_pad);
  return check_launch("yuv420u8_to_444");
}

AIVC_EXPORT int aivc_pack_images(const aivc_image_src *src, int32_t n_img, int32_t n, int32_t h, int32_t w, float *out,
                                 aivc_stream_t stream) {
  if (!src || !out || n_img < 1 || n_img > AIVC_MAX_IMAGES || n <= 0 || h <= 0 || w <= 0) return AIVC_ERR_ARG;
  if ((long)w * n_img >= 98304 || h > 65535 || n > 65535) return AIVC_ERR_UNSUPPORTED;
  PackArgs a;
  for (int i = 0; i < AIVC_MAX_IMAGES; ++i) a.src[i] = aivc_image_src{nullptr, nullptr, nullptr, nullptr, 0, 0};
  for (int i = 0; i < n_img; ++i) {
    a.src[i] = src[i];
    if (src[i].y && (!src[i].u || !src[i].v)) return AIVC_ERR_ARG;
    if (!src[i].y && src[i].f && src[i].f_channels < 3) return AIVC_ERR_ARG;
  }
  a.n_img = n_img;
  a.n = n;
  a.h = h;
  a.w = w;
  a.out = out;
  hipLaunchKernelGGL(pack_images_kernel, dim3(cdiv((size_t)w * n_img, 1024), h, n), dim3(256), 0, to_stream(stream), a);
  return check_launch("pack_images");
}

AIVC_EXPORT int aivc_frame_to_yuv420(const float *x, int32_t n, int32_t hx, int32_t wx, int32_t cx, const float *skip,
                                     int32_t cs, int32_t h, int32_t w, float *y, float *u, float *v, uint8_t *y8,
                                     uint8_t *u8, uint8_t *v8, aivc_stream_t stream) {
  if (!x || n <= 0 || h <= 0 || w <= 0 || hx < h || wx < w || cx < 3) return AIVC_ERR_ARG;
  if (skip && cs < 3) return AIVC_ERR_ARG;
  RecArgs a{x, skip, n, hx, wx, cx, cs, h, w, y, u, v, y8, u8, v8};
  const size_t total = (size_t)n * ((h + 1) / 2) * ((w + 1) / 2);
  // vector path: 3-channel x with an even row length (8-byte aligned pixel pairs), 4-channel skip, even frame sides
  const bool vec = cx == 3 && (wx & 1) == 0 && (h & 1) == 0 && (w & 1) == 0 && (!skip || cs == 4) &&
                   ((uintptr_t)x & 7) == 0 && (!skip || ((uintptr_t)skip & 15) == 0) && (!y8 || ((uintptr_t)y8 & 1) == 0);
  if (vec) hipLaunchKernelGGL(frame_to_yuv420_kernel<true>, dim3(cdiv(total, 256)), dim3(256), 0, to_stream(stream), a);
  else hipLaunchKernelGGL(frame_to_yuv420_kernel<false>, dim3(cdiv(total, 256)), dim3(256), 0, to_stream(stream), a);
  return check_launch("frame_to_yuv420");
}

AIVC_EXPORT int aivc_downsample2x(const float *x, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ch0,
                                  int32_t nch, float *out, aivc_stream_t stream) {
  if (!x || !out || n <= 0 || h <= 0 || w <= 0 || ch0 < 0 || nch <= 0 || ch0 + nch > c) return AIVC_ERR_ARG;
  const size_t total = (size_t)n * nch * (h / 2) * (w / 2);
  if (total == 0) return AIVC_OK;
  hipLaunchKernelGGL(downsample2x_kernel, dim3(cdiv(total, 256)), dim3(256), 0, to_stream(stream), x, n, h, w, c, ch0,
                     nch, out);
  return check_launch("downsample2x");
}

AIVC_EXPORT int aivc_warp(const float *x, const float *flow, int32_t n, int32_t h, int32_t w, int32_t c, float *out,
                          aivc_stream_t stream) {
  if (!x || !flow || !out || n <= 0 || h <= 0 || w <= 0 || c <= 0) return AIVC_ERR_ARG;
  hipLaunchKernelGGL(warp_kernel, dim3(cdiv((size_t)n * h * w, 256)), dim3(256), 0, to_stream(stream), x, flow, n, h,
                     w, c, out);
  return check_launch("warp");
}

AIVC_EXPORT int aivc_warp_blend_rows(const float *mof, int32_t hm, int32_t wm, int32_t cm, const float *prev,
                                     const float *next, int32_t cr, int32_t n, int32_t h, int32_t w, int32_t row0,
                                     int32_t rows, int32_t frame_type, float *pred, float *skip, float *x_warp,
                                     int32_t co, float *alpha_out, float *beta_out, aivc_stream_t stream) {
  if (!mof || !prev || !next || n <= 0 || h <= 0 || w <= 0) return AIVC_ERR_ARG;
  if (row0 < 0 || rows < 0 || row0 + rows > h) return AIVC_ERR_ARG;
  if (hm < rows || wm < w || cm < 6 || cr < 3 || co < 3) return AIVC_ERR_ARG;
  if (rows == 0) return AIVC_OK;
  BlendArgs a{mof, prev, next, hm, wm, cm, cr, n, h, w, frame_type, co, pred, skip, x_warp, alpha_out, beta_out, row0, rows};
  auto al16 = [](const void *p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  if (cr == 4 && co == 4 && cm % 2 == 0 && (reinterpret_cast<uintptr_t>(mof) & 7u) == 0 && al16(prev) && al16(next) &&
      al16(pred) && al16(skip) && al16(x_warp)) {
    hipLaunchKernelGGL(warp_blend4_kernel, dim3(cdiv((size_t)n * rows * w, 256)), dim3(256), 0, to_stream(stream), a);
    return check_launch("warp_blend");
  }
  hipLaunchKernelGGL(warp_blend_kernel, dim3(cdiv((size_t)n * rows * w, 256)), dim3(256), 0, to_stream(stream), a);
  return check_launch("warp_blend");
}

AIVC_EXPORT int aivc_warp_blend(const float *mof, int32_t hm, int32_t wm, int32_t cm, const float *prev,
                                const float *next, int32_t cr, int32_t n, int32_t h, int32_t w, int32_t frame_type,
                                float *pred, float *skip, float *x_warp, int32_t co, float *alpha_out,
                                float *beta_out, aivc_stream_t stream) {
  if (hm < h) return AIVC_ERR_ARG;
  return aivc_warp_blend_rows(mof, hm, wm, cm, prev, next, cr, n, h, w, 0, h, frame_type, pred, skip, x_warp, co,
                              alpha_out, beta_out, stream);
}

AIVC_EXPORT int aivc_hyper_params(const float *hs, int32_t n, int32_t hh, int32_t wh, int32_t c, int32_t h, int32_t w,
                                  float *mu, float *sigma, aivc_stream_t stream) {
  if (!hs || !mu || !sigma || n <= 0 || c <= 0 || h <= 0 || w <= 0 || hh < h || wh < w) return AIVC_ERR_ARG;
  hipLaunchKernelGGL(hyper_params_kernel, dim3(cdiv((size_t)n * h * w * c, 256)), dim3(256), 0, to_stream(stream), hs,
                     n, hh, wh, c, h, w, mu, sigma);
  return check_launch("hyper_params");
}

AIVC_EXPORT int aivc_channel_gain(const float *in, const float *gain, size_t npix, int32_t c, float *out,
                                  aivc_stream_t stream) {
  if (!in || !out || c <= 0) return AIVC_ERR_ARG;
  if (npix == 0) return AIVC_OK;
  hipLaunchKernelGGL(channel_gain_kernel, dim3(cdiv(npix * c, 256)), dim3(256), 0, to_stream(stream), in, gain,
                     npix * c, c, out);
  return check_launch("channel_gain");
}

AIVC_EXPORT int aivc_gain_interp(const float *g_r, const float *g_t, int32_t c, float l, float *out,
                                 aivc_stream_t stream) {
  if (!g_r || !g_t || !out || c <= 0) return AIVC_ERR_ARG;
  hipLaunchKernelGGL(gain_interp_kernel, dim3(cdiv(c, 256)), dim3(256), 0, to_stream(stream), g_r, g_t, c, l, out);
  return check_launch("gain_interp");
}

AIVC_EXPORT int aivc_quantize_center(const float *y, const float *mu, const float *gain_dec, size_t npix, int32_t c,
                                     int16_t *q, float *y_hat, aivc_stream_t stream) {
  if (!y || c <= 0 || (!q && !y_hat)) return AIVC_ERR_ARG;
  if (npix == 0) return AIVC_OK;
  hipLaunchKernelGGL(quantize_center_kernel, dim3(cdiv(npix * c, 256)), dim3(256), 0, to_stream(stream), y, mu,
                     gain_dec, npix * c, c, q, y_hat);
  return check_launch("quantize_center");
}

AIVC_EXPORT int aivc_dequantize(const int16_t *q, const float *mu, const float *gain_dec, size_t npix, int32_t c,
                                float *y_hat, aivc_stream_t stream) {
  if (!q || !y_hat || c <= 0) return AIVC_ERR_ARG;
  if (npix == 0) return AIVC_OK;
  hipLaunchKernelGGL(dequantize_kernel, dim3(cdiv(npix * c, 256)), dim3(256), 0, to_stream(stream), q, mu, gain_dec,
                     npix * c, c, y_hat);
  return check_launch("dequantize");
}
