// entropy.hip -- hyperprior entropy model + torchac-compatible range coder for gfx950.
//
// CDF build (massively parallel, fp64 transcendentals from aivc_detmath.h):
//   balle_cdf_table      one thread per (channel, k)          once per model load
//   laplace_cdf_rows     one thread per 8 CDF points (16 B)   decoder: a 1040-byte uint16 row per
//                        symbol position, so the serial decoder only does lookups
//   laplace_bounds       one thread per symbol                encoder: the 2 CDF values it needs
// Range coder (format-mandated: ONE serial stream per latent section):
//   one 64-lane wavefront per stream, streams of a batch run concurrently on different CUs.
//   All coder state is wave-uniform, so it lives in SGPRs / the scalar ALU; the vector lanes are
//   used for what is parallel inside one symbol step:
//     encode: lanes fetch 64 packed (c_lo,c_hi) pairs at once (coalesced), v_readlane feeds the
//             scalar update; E1/E2/E3 renormalisation loops are collapsed into clz-based closed
//             forms (s_flbit), bits are packed MSB-first through a 64-bit shifter.
//     decode: each lane owns 8 consecutive CDF entries of the symbol's row (one 16-byte load,
//             prefetched two groups ahead because the row address never depends on coder state);
//             "largest m with cdf[m] <= count" is a ballot + popcount instead of torchac's
//             10-step binary search.
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace aivc {

// ------------------------------------------------------------------ factorised prior (z) table
__device__ float balle_cdf_point(const float *P, float t) {
  const float *h0 = P, *h1 = P + 3, *h2 = P + 12, *h3 = P + 21;
  const float *b0 = P + 24, *b1 = P + 27, *b2 = P + 30, *b3 = P + 33;
  const float *a0 = P + 34, *a1 = P + 37, *a2 = P + 40;
  float v[3], nv[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float x = t * aivc_softplusf_det(h0[r]);
    x = x + b0[r];
    v[r] = x + aivc_tanhf_det(a0[r]) * aivc_tanhf_det(x);
  }
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    const float *hh = l == 0 ? h1 : h2;
    const float *bb = l == 0 ? b1 : b2;
    const float *aa = l == 0 ? a1 : a2;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float acc = v[0] * aivc_softplusf_det(hh[0 * 3 + r]);
      acc = __builtin_fmaf(v[1], aivc_softplusf_det(hh[1 * 3 + r]), acc);
      acc = __builtin_fmaf(v[2], aivc_softplusf_det(hh[2 * 3 + r]), acc);
      const float x = acc + bb[r];
      nv[r] = x + aivc_tanhf_det(aa[r]) * aivc_tanhf_det(x);
    }
    v[0] = nv[0];
    v[1] = nv[1];
    v[2] = nv[2];
  }
  float acc = v[0] * aivc_softplusf_det(h3[0]);
  acc = __builtin_fmaf(v[1], aivc_softplusf_det(h3[1]), acc);
  acc = __builtin_fmaf(v[2], aivc_softplusf_det(h3[2]), acc);
  return aivc_sigmoidf_det(acc + b3[0]);
}

__global__ __launch_bounds__(256) void balle_cdf_table_kernel(const float *__restrict__ params, int c,
                                                              uint16_t *__restrict__ table,
                                                              float *__restrict__ cdf_f32) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= c * AIVC_CDF_ROW) return;
  const int ch = gid / AIVC_CDF_ROW, k = gid % AIVC_CDF_ROW;
  if (k >= AIVC_LP) {
    table[gid] = 0;
    return;
  }
  const float cdf = balle_cdf_point(params + (size_t)ch * AIVC_BALLE_PARAMS, (float)k - 256.5f);
  if (cdf_f32) cdf_f32[(size_t)ch * AIVC_LP + k] = cdf;
  table[gid] = aivc_cdf_quant(cdf, k);
}

// ------------------------------------------------------------------ non-zero feature maps
__global__ __launch_bounds__(256) void nonzero_maps_kernel(const int16_t *__restrict__ q, size_t npix, int c,
                                                           uint8_t *__restrict__ flags) {
  // grid.y = image of a batch: q is [n][npix][c], flags [n][c]
  q += (size_t)blockIdx.y * npix * c;
  flags += (size_t)blockIdx.y * c;
  // every writer stores the same value (1): deterministic without atomics
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix * c; i += (size_t)gridDim.x * blockDim.x)
    if (q[i] != 0) flags[i % c] = 1;
}

// ------------------------------------------------------------------ Laplace CDF rows / bounds
__global__ __launch_bounds__(256) void laplace_cdf_rows_kernel(const float *__restrict__ sigma, size_t npix, int c,
                                                               aivc_map_list maps, uint16_t *__restrict__ rows) {
  constexpr int CHUNKS = AIVC_CDF_ROW / 8;  // 65 x 16 B per row
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)maps.n_maps * npix * CHUNKS;
  if (gid >= total) return;
  const int chunk = (int)(gid % CHUNKS);
  const size_t pos = gid / CHUNKS;
  const int m = (int)(pos / npix);
  const size_t pix = pos % npix;
  const float s = sigma[pix * c + maps.idx[m]];
  uint32_t w[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k0 = chunk * 8 + 2 * j, k1 = k0 + 1;
    const uint32_t lo = k0 < AIVC_LP ? aivc_laplace_cdf_u16(k0, s) : 0;
    const uint32_t hi = k1 < AIVC_LP ? aivc_laplace_cdf_u16(k1, s) : 0;
    w[j] = lo | (hi << 16);
  }
  *reinterpret_cast<uint4 *>(rows + pos * AIVC_CDF_ROW + chunk * 8) = make_uint4(w[0], w[1], w[2], w[3]);
}

// The decoder's fast path only ever looks at the 64 entries DEC_WIN0 .. DEC_WIN0 + 63 of a row (symbols -32 .. +31);
// this kernel produces just those (128 B per coded position instead of 1040) plus sigma of the position, from which
// the decoder's slow path rebuilds the rest of a row on demand with the same function.
constexpr int CDF_WIN0 = 224, CDF_WIN = 64;
// A wavefront builds ONE 8-entry chunk (16 bytes) of the windows of 64 consecutive positions, so the saturation test
// is wave-uniform: for the small sigmas of P / B latents the outer chunks of a window lie where the function returns
// expm1 = -1 exactly (|t| / b > 17.5: cdf 0 or 1, entry k or 65023 + k) for every position of the wavefront, and
// the fp64 evaluation is skipped for the whole wavefront (one diverging lane used to keep all 64 on the long path).
__global__ __launch_bounds__(256) void laplace_cdf_windows_kernel(const float *__restrict__ sigma, size_t npix, int c,
                                                                  aivc_map_list maps, uint16_t *__restrict__ win,
                                                                  float *__restrict__ sigma_pos) {
  constexpr int CHUNKS = CDF_WIN / 8;  // 8 x 16 B per position
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)maps.n_maps * npix;
  const size_t wv = gid >> 6;
  const int lane = (int)(gid & 63), chunk = (int)(wv % CHUNKS);
  const size_t pos = (wv / CHUNKS) * 64 + (size_t)lane;
  const bool valid = pos < total;
  float s = 1.0f;
  if (valid) {
    const int m = (int)(pos / npix);
    const size_t pix = pos % npix;
    s = sigma[pix * c + maps.idx[m]];
    if (chunk == 0) sigma_pos[pos] = s;
  }
  const int k0 = CDF_WIN0 + chunk * 8;
  uint32_t w[4];
  // the entry of this chunk nearest to the centre decides: |t| / b is monotonic in |t| (correctly rounded division)
  bool sat = false;
  if (chunk != CHUNKS / 2) {  // (the chunk around t = 0 holds both signs)
    const float tmin = chunk < CHUNKS / 2 ? 256.5f - (float)(k0 + 7) : (float)k0 - 256.5f;
    const float b = s / 1.41421354f;  // as in aivc_laplace_cdf
    sat = tmin / b > 17.5f;           // aivc_expm1f_det(-a) returns -1.0f for -a < -17.5f
  }
  if (__all(sat || !valid)) {
    const uint32_t base = chunk < CHUNKS / 2 ? 0u : 65023u;  // rint(0 * 65023), rint(1 * 65023)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t k = (uint32_t)(k0 + 2 * j);
      w[j] = ((base + k) & 0xFFFFu) | (((base + k + 1u) & 0xFFFFu) << 16);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + 2 * j;
      w[j] = (uint32_t)aivc_laplace_cdf_u16(k, s) | ((uint32_t)aivc_laplace_cdf_u16(k + 1, s) << 16);
    }
  }
  if (valid) *reinterpret_cast<uint4 *>(win + pos * CDF_WIN + chunk * 8) = make_uint4(w[0], w[1], w[2], w[3]);
}

__global__ __launch_bounds__(256) void laplace_bounds_kernel(const float *__restrict__ sigma,
                                                             const int16_t *__restrict__ q, size_t npix, int c,
                                                             aivc_map_list maps, uint32_t *__restrict__ bounds) {
  const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= (size_t)maps.n_maps * npix) return;
  const int ch = maps.idx[pos / npix];
  const size_t pix = pos % npix;
  const float s = sigma[pix * c + ch];
  // symbols outside the alphabet [0, 512] never reach this kernel through the codec (quantize_center clamps,
  // the path API raises like torchac's check_input_bounds); the clamp keeps a misuse memory-safe.
  // Symbol 512 (torchac's max_symbol): upper bound 2^16, packed as 0 (include/aivc_hip.h)
  const int sym = min(max((int)q[pix * c + ch] + AIVC_AC_MAX_VAL, 0), AIVC_MAX_SYMBOL);
  const uint32_t lo = aivc_laplace_cdf_u16(sym, s), hi = sym == AIVC_MAX_SYMBOL ? 0u : aivc_laplace_cdf_u16(sym + 1, s);
  bounds[pos] = lo | (hi << 16);
}

__global__ __launch_bounds__(256) void table_bounds_kernel(const uint16_t *__restrict__ table,
                                                           const int16_t *__restrict__ q, size_t npix, int c,
                                                           uint32_t *__restrict__ bounds) {
  const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= (size_t)c * npix) return;
  const int ch = (int)(pos / npix);
  const size_t pix = pos % npix;
  const int sym = min(max((int)q[pix * c + ch] + AIVC_AC_MAX_VAL, 0), AIVC_MAX_SYMBOL);
  const uint16_t *row = table + (size_t)ch * AIVC_CDF_ROW;
  bounds[pos] = (uint32_t)row[sym] | (sym == AIVC_MAX_SYMBOL ? 0u : (uint32_t)row[sym + 1] << 16);
}

__global__ __launch_bounds__(256) void table_bounds_batch_kernel(const uint16_t *__restrict__ table, const int16_t *__restrict__ q,
                                                                 size_t npix, int c, uint32_t *__restrict__ bounds) {
  const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= (size_t)c * npix) return;
  const int ch = (int)(pos / npix);
  const size_t pix = pos % npix;
  const int sym = min(max((int)q[((size_t)blockIdx.y * npix + pix) * c + ch] + AIVC_AC_MAX_VAL, 0), AIVC_MAX_SYMBOL);
  const uint16_t *row = table + (size_t)ch * AIVC_CDF_ROW;
  bounds[(size_t)blockIdx.y * c * npix + pos] = (uint32_t)row[sym] | (sym == AIVC_MAX_SYMBOL ? 0u : (uint32_t)row[sym + 1] << 16);
}

struct InvMap {
  int16_t slot[AIVC_MAX_MAPS];  // channel -> position in the coded list, or -1
};
// A thread writes 8 consecutive channels (16 bytes) of one pixel; consecutive lanes take consecutive pixels, so every
// read of a coded map is 128 contiguous bytes per wavefront (one thread per (pixel, channel) with the channel
// fastest read 64 different maps per load: 36 us for a 1080p latent).
__global__ __launch_bounds__(256) void scatter_symbols_kernel(const uint16_t *__restrict__ sym, size_t npix, int c,
                                                              InvMap inv, int16_t *__restrict__ q, int vec) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int groups = (c + 7) / 8;
  if (gid >= npix * groups) return;
  const size_t pix = gid % npix;
  const int ch0 = (int)(gid / npix) * 8;
  int16_t v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ch = ch0 + e;
    const int m = ch < c ? inv.slot[ch] : -1;
    v[e] = m < 0 ? (int16_t)0 : (int16_t)((int)sym[(size_t)m * npix + pix] - AIVC_AC_MAX_VAL);
  }
  int16_t *dst = q + pix * c + ch0;
  if (vec) {  // c % 8 == 0 and q 16-byte aligned
    uint4 o;
    o.x = (uint32_t)(uint16_t)v[0] | ((uint32_t)(uint16_t)v[1] << 16);
    o.y = (uint32_t)(uint16_t)v[2] | ((uint32_t)(uint16_t)v[3] << 16);
    o.z = (uint32_t)(uint16_t)v[4] | ((uint32_t)(uint16_t)v[5] << 16);
    o.w = (uint32_t)(uint16_t)v[6] | ((uint32_t)(uint16_t)v[7] << 16);
    *reinterpret_cast<uint4 *>(dst) = o;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (ch0 + e < c) dst[e] = v[e];
  }
}

// ---- frame-batch variants: blockIdx.y = frame, its map list and stream offset read from a device table ------------
__global__ __launch_bounds__(256) void laplace_cdf_windows_batch_kernel(const float *__restrict__ sigma, size_t npix, int c,
                                                                        const aivc_frame_maps *__restrict__ frames,
                                                                        uint16_t *__restrict__ win, float *__restrict__ sigma_pos) {
  constexpr int CHUNKS = CDF_WIN / 8;
  const aivc_frame_maps &fm = frames[blockIdx.y];
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)fm.n_maps * npix;
  const size_t wv = gid >> 6;
  if ((wv / CHUNKS) * 64 >= total) return;  // wave-uniform
  const int lane = (int)(gid & 63), chunk = (int)(wv % CHUNKS);
  const size_t pos = (wv / CHUNKS) * 64 + (size_t)lane;
  const bool valid = pos < total;
  const float *sg = sigma + (size_t)blockIdx.y * npix * c;
  float s = 1.0f;
  if (valid) {
    const int m = (int)(pos / npix);
    const size_t pix = pos % npix;
    s = sg[pix * c + fm.idx[m]];
    if (chunk == 0) sigma_pos[fm.pos_off + pos] = s;
  }
  const int k0 = CDF_WIN0 + chunk * 8;
  uint32_t w[4];
  bool sat = false;
  if (chunk != CHUNKS / 2) {
    const float tmin = chunk < CHUNKS / 2 ? 256.5f - (float)(k0 + 7) : (float)k0 - 256.5f;
    const float b = s / 1.41421354f;
    sat = tmin / b > 17.5f;
  }
  if (__all(sat || !valid)) {
    const uint32_t base = chunk < CHUNKS / 2 ? 0u : 65023u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t k = (uint32_t)(k0 + 2 * j);
      w[j] = ((base + k) & 0xFFFFu) | (((base + k + 1u) & 0xFFFFu) << 16);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + 2 * j;
      w[j] = (uint32_t)aivc_laplace_cdf_u16(k, s) | ((uint32_t)aivc_laplace_cdf_u16(k + 1, s) << 16);
    }
  }
  if (valid) *reinterpret_cast<uint4 *>(win + (fm.pos_off + pos) * CDF_WIN + chunk * 8) = make_uint4(w[0], w[1], w[2], w[3]);
}

__global__ __launch_bounds__(256) void laplace_bounds_batch_kernel(const float *__restrict__ sigma, const int16_t *__restrict__ q,
                                                                   size_t npix, int c, const aivc_frame_maps *__restrict__ frames,
                                                                   uint32_t *__restrict__ bounds) {
  const aivc_frame_maps &fm = frames[blockIdx.y];
  const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= (size_t)fm.n_maps * npix) return;
  const int ch = fm.idx[pos / npix];
  const size_t pix = pos % npix;
  const size_t o = ((size_t)blockIdx.y * npix + pix) * c + ch;
  const float s = sigma[o];
  const int sym = min(max((int)q[o] + AIVC_AC_MAX_VAL, 0), AIVC_MAX_SYMBOL);
  const uint32_t lo = aivc_laplace_cdf_u16(sym, s), hi = sym == AIVC_MAX_SYMBOL ? 0u : aivc_laplace_cdf_u16(sym + 1, s);
  bounds[fm.pos_off + pos] = lo | (hi << 16);
}

__global__ __launch_bounds__(256) void scatter_symbols_batch_kernel(const uint16_t *__restrict__ sym, size_t npix, int c,
                                                                    const aivc_frame_maps *__restrict__ frames,
                                                                    int16_t *__restrict__ q, int vec) {
  __shared__ int16_t slot[AIVC_MAX_MAPS];  // channel -> position in the frame's coded list, or -1
  const aivc_frame_maps &fm = frames[blockIdx.y];
  for (int i = threadIdx.x; i < AIVC_MAX_MAPS; i += blockDim.x) slot[i] = -1;
  __syncthreads();
  for (int i = threadIdx.x; i < fm.n_maps; i += blockDim.x) slot[fm.idx[i]] = (int16_t)i;
  __syncthreads();
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int groups = (c + 7) / 8;
  if (gid >= npix * groups) return;
  const size_t pix = gid % npix;
  const int ch0 = (int)(gid / npix) * 8;
  const uint16_t *fs = sym + fm.pos_off;
  int16_t v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ch = ch0 + e;
    const int m = ch < c ? slot[ch] : -1;
    v[e] = m < 0 ? (int16_t)0 : (int16_t)((int)fs[(size_t)m * npix + pix] - AIVC_AC_MAX_VAL);
  }
  int16_t *dst = q + ((size_t)blockIdx.y * npix + pix) * c + ch0;
  if (vec) {
    uint4 o;
    o.x = (uint32_t)(uint16_t)v[0] | ((uint32_t)(uint16_t)v[1] << 16);
    o.y = (uint32_t)(uint16_t)v[2] | ((uint32_t)(uint16_t)v[3] << 16);
    o.z = (uint32_t)(uint16_t)v[4] | ((uint32_t)(uint16_t)v[5] << 16);
    o.w = (uint32_t)(uint16_t)v[6] | ((uint32_t)(uint16_t)v[7] << 16);
    *reinterpret_cast<uint4 *>(dst) = o;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (ch0 + e < c) dst[e] = v[e];
  }
}

// ------------------------------------------------------------------ range encoder
__device__ __forceinline__ uint32_t rl(uint32_t v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// t = (span * e) >> 16 with span = hl + 1 (up to 2^32): e * hl + e < 2^48, exact in ONE v_mad_u64_u32 (written out:
// the compiler turns the C expression into a 33-bit (hl + 1) * e, five instructions)
__device__ __forceinline__ uint32_t scaled(uint32_t e, uint32_t hl) {
  uint64_t r, carry;
  const uint64_t e64 = e;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=&v"(r), "=s"(carry) : "v"(e), "s"(hl), "v"(e64));
  uint32_t t = (uint32_t)(r >> 16);
  asm("" : "+v"(t));  // the shift stays on the vector unit (one v_alignbit; moved behind a readlane it costs a second readlane)
  return t;
}

// MSB-first bit packer.  All state is wave-uniform (SGPRs); a 32-bit word is stored (every lane issues the same
// store: one transaction) only when it is complete -- at the ~1 bit per symbol of these latents that is one store
// per ~30 symbols instead of one per call -- behind a wave-uniform branch.  finish() writes the partial last word.
struct BitSink {
  uint32_t *out;       // 4-byte aligned
  uint32_t cap_words;  // capacity in 32-bit words (>= 1)
  uint32_t n_words;
  uint64_t acc;        // the low `nbits` bits are pending output
  uint32_t nbits;      // < 32 between calls
  uint32_t overflow;
  __device__ __forceinline__ void put(uint32_t bits, uint32_t nb) {  // nb in [0, 32]
    acc = (acc << nb) | (uint64_t)bits;
    nbits += nb;
    if (nbits >= 32) {
      nbits -= 32;
      const uint32_t w = (uint32_t)(acc >> nbits);
      if (n_words < cap_words) out[n_words] = __builtin_bswap32(w);
      else overflow = 1;
      n_words++;
    }
  }
  __device__ __forceinline__ void put_run(uint32_t bit, uint32_t count) {
    while (count > 0) {
      const uint32_t r = count > 32 ? 32 : count;
      const uint32_t ones = r == 32 ? 0xFFFFFFFFu : ((1u << r) - 1u);
      put(bit ? ones : 0u, r);
      count -= r;
    }
  }
  __device__ __forceinline__ void finish() {  // bits left over after the last complete word: MSB aligned, zero padded
    if (nbits > 0) {
      const uint32_t w = (uint32_t)(acc << (32 - nbits));
      if (n_words < cap_words) out[n_words] = __builtin_bswap32(w);
      else overflow = 1;
    }
  }
};

__global__ __launch_bounds__(64) void range_encode_kernel(const uint32_t *__restrict__ bounds, aivc_rc_batch batch,
                                                          uint8_t *__restrict__ out, uint32_t *__restrict__ out_len) {
  const aivc_rc_stream st = batch.s[blockIdx.x];
  const int lane = threadIdx.x;
  // latency-critical serial wave: win the issue arbitration against co-resident conv waves
  __builtin_amdgcn_s_setprio(3);
  BitSink sink{reinterpret_cast<uint32_t *>(out + st.out_off), st.out_cap / 4, 0, 0, 0, 0};
  uint32_t low = 0, high = 0xFFFFFFFFu, pending = 0;
  const uint32_t *src = bounds + st.in_off;
  // 64 symbols per block, lane j holds the bounds of symbol base + j.  Per symbol both interval ends
  // t = (span * c) >> 16 are one v_mad_u64_u32 each over the whole block (lane j's result is the one read back):
  // 6 instructions instead of the ~18 of two 33 x 16 bit products on the scalar unit.  The next block's bounds are
  // in flight while this one is coded.
  uint32_t nxt = lane < (int)st.n_sym ? src[lane] : 0u;
#pragma unroll 1
  for (uint32_t base = 0; base < st.n_sym; base += 64) {
    const uint32_t mine = nxt;
    nxt = (base + 64u + lane < st.n_sym) ? src[base + 64u + lane] : 0u;
    const uint32_t vlo = mine & 0xFFFFu, vhi = (mine >> 16) ? (mine >> 16) : 0x10000u;  // 0 = 2^16: symbol 512
    const uint32_t cnt = min(64u, st.n_sym - base);
#pragma unroll 1
    for (uint32_t j = 0; j < cnt; ++j) {
      const uint32_t hl = high - low;  // span - 1
      const uint32_t t_lo = rl(scaled(vlo, hl), (int)j), t_hi = rl(scaled(vhi, hl), (int)j);
      high = low + t_hi - 1u;
      low = low + t_lo;
      // E1 / E2: the n leading bits on which low and high agree are final (low < high: n <= 31); nothing to do
      // for the many symbols that settle no bit
      const uint32_t x = low ^ high;
      if ((int32_t)x >= 0) {
        const uint32_t n = (uint32_t)__builtin_clz(x | 1u);
        const uint32_t top = (uint32_t)(((uint64_t)low << n) >> 32);  // the n leading bits of low
        uint32_t bits = top, nb = n;
        if (pending != 0) {  // the first settled bit releases the pending straddle bits (its complement)
          const uint32_t b0 = low >> 31;
          sink.put(b0, 1);
          sink.put_run(b0 ^ 1u, pending);
          pending = 0;
          nb = n - 1u;
          bits = top & ((1u << nb) - 1u);
        }
        sink.put(bits, nb);
        low <<= n;
        high = (high << n) | ~(0xFFFFFFFFu << n);
      }
      asm volatile("" : "+s"(high));  // keeps the test below on the scalar unit
      // E3: now low = 0..., high = 1...; every further position with (low,high) = (1,0) straddles
      const uint32_t yy = low & ~high;
      if (yy & 0x40000000u) {
        const uint32_t m = (uint32_t)__builtin_clz(~(yy << 1));  // leading ones (1..31)
        pending += m;
        low = (low << m) & 0x7FFFFFFFu;
        high = (high << m) | 0x80000000u | ((1u << m) - 1u);
      }
    }
  }
  pending += 1;
  const uint32_t fb = low < 0x40000000u ? 0u : 1u;
  sink.put(fb, 1);
  sink.put_run(fb ^ 1u, pending);
  sink.finish();
  // count the bytes of the partial last word
  const uint32_t total = sink.n_words * 4u + (sink.nbits + 7u) / 8u;
  if (total > st.out_cap) sink.overflow = 1;
  if (lane == 0) out_len[blockIdx.x] = sink.overflow ? 0xFFFFFFFFu : total;
}

// ------------------------------------------------------------------ range encoder, one stream per LANE (round 5)
// What a lone wavefront pays (tools/lat_probe.hip, profiles/r05_lat_probe.txt): 4 cycles per instruction of any kind,
// dependent or not; +10 for a branch not taken, +24 taken, +25 when it tests a VALU compare; +16..20 for every
// VALU -> SGPR -> SALU crossing; 18 for v_mad_u64_u32 + shift in a chain.  The wave-per-stream encoder above pays two
// crossings and two to four branches per symbol: 0.09 us per symbol when nothing settles, 0.16-0.18 at 2-6 bit per
// symbol -- and a launch of 64 streams occupies 64 wavefronts on 64 CUs for as long as its longest stream, next to the
// convolutions.  Here a stream is a LANE: all state is per-lane VGPR data, every step is straight-line VALU code
// (selects, no branch per symbol, so the cost does not depend on the bit rate), no cross-lane traffic, and the serial
// chain is split over TWO wavefronts of one workgroup that run on different SIMDs:
//   wave A  the interval arithmetic, the chain that is serial by format (30 instructions per symbol):
//             t = (span * c) >> 16 by one v_mad_u64_u32 per end;
//             E1 / E2: n = clz(low ^ high) leading bits are final;  E3: m = leading ones of (low & ~high) << (n + 1)
//             straddle steps (span > 2^30 before a symbol and a symbol's probability >= 2^-16 bound n + m <= 18, so the
//             two renormalisations are ONE shift by n + m);
//             per symbol it leaves {low before the shift, n, pending count released by the settled bit} in LDS;
//   wave B  the bit packer, one chunk of 32 symbols behind A: [b0][pending x ~b0][n - 1 more bits of low] is ONE push of
//             n + pending <= 32 bits into a 64-bit shifter per lane; the candidate word is stored on every symbol
//             (overwritten until it is complete: no branch, no flush test), a chunk in which some lane's run of straddle
//             bits exceeded one push is redone from its records by the generic loop.
//   wave C  the loader: the packed bounds of chunk c + 1 (per-lane streams: scattered dwords) into LDS, loads issued two
//             chunks ahead.
// One s_barrier per chunk.  64 streams cost three wavefronts; a symbol costs wave A ~0.055 us whatever the bit rate.
// Symbols past a lane's end are coded as the null symbol (bounds 0 / 2^16): t_lo = 0, t_hi = span -- the state does not
// move and nothing is emitted.
constexpr int ENC_CH = 32;  // symbols per chunk (LDS: 2 buffers x 32 x 64 lanes x 8 B = 32 KB)

struct LanePacker {
  uint32_t *dst;       // this lane's output (4-byte aligned)
  uint32_t last_word;  // cap_words - 1
  uint32_t widx;       // complete words produced
  uint64_t acc;        // low `nbits` bits pending
  uint32_t nbits;      // < 32 between pushes
  // straight-line push: the candidate word (the top 32 of the nbits accumulated bits) is stored at dst[widx] on EVERY call
  // -- garbage while the word is incomplete, overwritten until it is; widx advances when 32 bits are there
  __device__ __forceinline__ void push(uint32_t val, uint32_t len) {  // len in [0, 32], val < 2^len
    acc = (acc << len) | (uint64_t)val;
    nbits += len;
    dst[min(widx, last_word)] = __builtin_bswap32((uint32_t)(acc >> ((nbits - 32u) & 63u)));
    widx += nbits >> 5;
    nbits &= 31u;
  }
  __device__ __forceinline__ void put_run(uint32_t bit, uint32_t count) {
    while (count > 0) {
      const uint32_t r = count > 32 ? 32 : count;
      push(bit ? (r == 32 ? 0xFFFFFFFFu : ((1u << r) - 1u)) : 0u, r);
      count -= r;
    }
  }
  // E1 / E2 output of one symbol in general (any run length): b0, the pending run, the other n - 1 bits
  __device__ __forceinline__ void settle_long(uint32_t low, uint32_t n, uint32_t P) {
    if (n != 0) {
      const uint32_t b0 = low >> 31;
      push(b0, 1);
      put_run(b0 ^ 1u, P);
      push(((low << 1) >> 1) >> (32u - n), n - 1u);
    }
  }
};

__global__ __launch_bounds__(192) void range_encode_lanes_kernel(const uint32_t *__restrict__ bounds, aivc_rc_batch batch,
                                                                 uint8_t *__restrict__ out, uint32_t *__restrict__ out_len) {
  __shared__ uint32_t inb[2][ENC_CH][64];  // packed bounds of a chunk, staged by wave C
  __shared__ uint2 rec[2][ENC_CH][64];     // {low after the interval update, n | pending released << 5}, from wave A
  __shared__ uint2 fin[64];                // {low, pending} after the last symbol
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool live = lane < batch.n_streams;
  const aivc_rc_stream st = batch.s[live ? lane : 0];
  const uint32_t n_sym = live ? st.n_sym : 0u;
  uint32_t n_max = n_sym;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) n_max = max(n_max, (uint32_t)__shfl_xor((int)n_max, off));
  n_max = __builtin_amdgcn_readfirstlane(n_max);
  const uint32_t n_chunks = (n_max + ENC_CH - 1) / ENC_CH;
  const uint32_t T = n_chunks + 2;  // pipeline steps: at step t wave C stages chunk t, A codes chunk t - 1, B packs chunk t - 2
  __builtin_amdgcn_s_setprio(3);
  if (wave == 2) {
    // ---------------------------------------------------------------- wave C: the bounds, two chunks ahead of wave A
    // (per-lane streams: 64 scattered dword loads per symbol index.  In wave A, eight symbols ahead, they were what the
    // chain waited for: under the convolutions' memory traffic a load takes 2-3 us and the encoder ran at 0.2 us per
    // symbol next to 0.09 alone -- profiles/r05_kernel_stats_4k_high_rate_before_loader.csv)
    const uint32_t *src = bounds + st.in_off;
    uint32_t r0[ENC_CH], r1[ENC_CH];
    auto fetch = [&](uint32_t (&r)[ENC_CH], uint32_t c) {
#pragma unroll
      for (int k = 0; k < ENC_CH; ++k) {
        const uint32_t i = c * ENC_CH + (uint32_t)k;
        r[k] = i < n_sym ? src[i] : 0u;  // past the end: the null symbol
      }
    };
    auto stage = [&](const uint32_t (&r)[ENC_CH], uint32_t c) {
#pragma unroll
      for (int k = 0; k < ENC_CH; ++k) inb[c & 1][k][lane] = r[k];
    };
    fetch(r0, 0);
    fetch(r1, 1);
#pragma unroll 1
    for (uint32_t t = 0; t < T; t += 2) {
      if (t < n_chunks) {
        stage(r0, t);
        fetch(r0, t + 2);
      }
      __syncthreads();
      if (t + 1 < T) {
        if (t + 1 < n_chunks) {
          stage(r1, t + 1);
          fetch(r1, t + 3);
        }
        __syncthreads();
      }
    }
  } else if (wave == 0) {
    // ---------------------------------------------------------------- wave A: intervals and renormalisation
    uint32_t low = 0, high = 0xFFFFFFFFu, pending = 0;
#pragma unroll 1
    for (uint32_t t = 0; t < T; ++t) {
      if (t >= 1 && t <= n_chunks) {
        const uint32_t c = t - 1;
        const uint32_t *ib = &inb[c & 1][0][lane];
        uint2 *buf = &rec[c & 1][0][lane];
#pragma unroll 8
        for (int k = 0; k < ENC_CH; ++k) {
          const uint32_t w = ib[k * 64];
          const uint32_t c_lo = w & 0xFFFFu;
          uint32_t c_hi = w >> 16;
          c_hi = c_hi ? c_hi : 0x10000u;  // 0 = 2^16: symbol 512, and the null symbol
          const uint32_t hl = high - low;  // span - 1
          const uint32_t t_lo = (uint32_t)(((uint64_t)hl * c_lo + c_lo) >> 16);
          const uint32_t t_hi = (uint32_t)(((uint64_t)hl * c_hi + c_hi) >> 16);
          high = low + t_hi - 1u;
          low = low + t_lo;
          const uint32_t n = (uint32_t)__builtin_clz(low ^ high);  // low < high: never 0; n <= 18
          const uint32_t P = n ? pending : 0u;
          pending = n ? 0u : pending;
          buf[k * 64] = make_uint2(low, n | (P << 5));
          // E3 on the values E1 / E2 would leave: positions with (low, high) = (1, 0) below the settled bits straddle
          const uint32_t yy = (low & ~high) << (n + 1u);
          const uint32_t m = (uint32_t)__builtin_clz(~yy);  // leading ones of yy: yy is never all ones (n + m <= 18)
          const uint32_t sh = n + m;
          pending += m;
          low = (low << sh) & 0x7FFFFFFFu;
          high = (high << sh) | ((1u << sh) - 1u) | 0x80000000u;
        }
      }
      if (t == n_chunks) fin[lane] = make_uint2(low, pending);
      __syncthreads();
    }
  } else {
    // ---------------------------------------------------------------- wave B: bit packing
    LanePacker pk{reinterpret_cast<uint32_t *>(out + st.out_off), st.out_cap / 4 - 1u, 0, 0, 0};
#pragma unroll 1
    for (uint32_t t = 0; t < T; ++t) {
      if (t >= 2 && live) {  // (a lane without a stream owns no output; the wave still takes every barrier)
        const uint32_t c = t - 2;
        const uint2 *buf = &rec[c & 1][0][lane];
        const LanePacker at_start = pk;
        uint32_t longest = 0;
#pragma unroll 8
        for (int k = 0; k < ENC_CH; ++k) {
          const uint2 r = buf[k * 64];
          const uint32_t low = r.x, n = r.y & 31u, P = r.y >> 5;
          const uint32_t len = n + P;
          longest = max(longest, len);
          // [b0][P x ~b0][the other n - 1 bits] as one value: head = b0 ? 1 << P : (1 << P) - 1, then bits 30 .. 32 - n of low
          const uint32_t n1 = n ? n - 1u : 0u;
          const uint32_t head = ((1u << P) - 1u) + (low >> 31);
          const uint32_t rest = ((low << 1) >> 1) >> (31u - n1);
          uint32_t val = (head << n1) | rest;
          asm volatile("" : "+v"(val));  // computed for every symbol: as "n ? ... : 0" the compiler branches around it (+25 cycles)
          pk.push(n ? val : 0u, len);
        }
        if (__builtin_expect(__ballot(longest > 32u) != 0ull, 0)) {  // (uniform; practically never)
          pk = at_start;
#pragma unroll 1
          for (int k = 0; k < ENC_CH; ++k) {
            const uint2 r = buf[k * 64];
            pk.settle_long(r.x, r.y & 31u, r.y >> 5);
          }
        }
      }
      __syncthreads();
    }
    const uint2 f = fin[lane];  // wave A's final state (written at step n_chunks, a barrier ago at least)
    if (live) {
      const uint32_t fb = f.x < 0x40000000u ? 0u : 1u;
      pk.push(fb, 1);
      pk.put_run(fb ^ 1u, f.y + 1u);
      if (pk.nbits > 0) pk.dst[min(pk.widx, pk.last_word)] = __builtin_bswap32((uint32_t)(pk.acc << (32 - pk.nbits)));
      const uint32_t total = pk.widx * 4u + (pk.nbits + 7u) / 8u;
      out_len[lane] = total > st.out_cap ? 0xFFFFFFFFu : total;
    }
  }
}

// ------------------------------------------------------------------ range decoder
// Per symbol torchac computes count = ((value - low + 1) * 2^16 - 1) / span, binary-searches the largest m with
// cdf[m] <= count, then narrows  high = low - 1 + (span * cdf[m + 1] >> 16),  low = low + (span * cdf[m] >> 16).
// Here every lane holds one CDF entry e and computes t = (span * e) >> 16 -- the very quantity the update needs.
// cdf[m] <= count  <=>  e * span <= (d + 1) * 2^16 - 1  <=>  t <= d  with d = value - low, so the search is one
// 32-bit compare + ballot + popcount, and the two t the update needs are already in registers (v_readlane): no
// division, no 64-bit compare, no second multiplication.  The bit window keeps `value` as its upper half, so
// renormalisation is a 64-bit shift.  All coder state is wave-uniform (SGPRs).  Instruction count is what
// matters: the chain is serial by format, one dependent instruction after the other.
struct BitWin {
  const uint32_t *in;   // 4-byte aligned
  uint32_t n_words;     // words that may be read (beyond: zeros)
  uint32_t tail_bytes;  // payload bytes in the last word (0 = all four)
  uint32_t wi;          // next word index
  uint32_t chunk;       // per-lane: word (wi & ~63) + lane
  uint64_t win;         // upcoming bits, MSB aligned; value = upper 32 bits
  uint32_t avail;       // valid bits in win, kept in [33, 64]
  int lane;
  __device__ __forceinline__ void load_chunk() {
    const uint32_t idx = (wi & ~63u) + (uint32_t)lane;
    uint32_t w = idx < n_words ? in[idx] : 0u;
    // bytes past the payload end read as zero bits (torchac: "missing bits are 0")
    if (idx + 1 == n_words && tail_bytes) w &= (1u << (8 * tail_bytes)) - 1u;
    chunk = w;
  }
  __device__ __forceinline__ uint32_t next_word() {
    const uint32_t w = __builtin_bswap32(rl(chunk, (int)(wi & 63u)));
    wi++;
    if ((wi & 63u) == 0) load_chunk();
    return w;
  }
  __device__ __forceinline__ void init() {
    wi = 0;
    load_chunk();
    win = (uint64_t)next_word() << 32;
    win |= (uint64_t)next_word();
    avail = 64;
  }
  __device__ __forceinline__ uint32_t value() const { return (uint32_t)(win >> 32); }
  __device__ __forceinline__ void consume(uint32_t n) {  // n in [0, 31]
    win <<= n;
    avail -= n;
    if (__builtin_expect(avail <= 32, 0)) {  // (laid out as the taken branch: it fires once per 32 bits)
      win |= (uint64_t)next_word() << (32 - avail);
      avail += 32;
    }
  }
};

constexpr int DEC_HALF = 32;   // symbols per half of the window ring: the prefetching wave fills one half while the decoding wave reads the other
constexpr int DEC_SLOTS = 2 * DEC_HALF + 1;  // (+ 1: the decoding wave reads one slot ahead, past the end of the second half at a block's last symbol)
constexpr int DEC_WIN0 = 224;  // fast-path window: CDF entries 224..287 (symbol values -32..+31), one per lane
static_assert(DEC_WIN0 == CDF_WIN0, "window position");

// LDS-DMA of one uint16 per lane (128 contiguous bytes of the row -> one dword per lane in LDS, zero-extended): no
// register, no VGPR round trip.  Address = uniform row base + per-lane 32-bit byte offset; M0 (the LDS destination base)
// is written in the statement that uses it (gfx9 LDS instructions do not need it).  Round 5: the DMAs are issued by a
// second wavefront of the workgroup (decode_prefetcher) -- in the decoding wave they were 7 of its ~45 instructions per
// symbol (a lone wavefront issues one instruction per 4 cycles, whatever the kind: tools/lat_probe.hip).
__device__ __forceinline__ void window_dma(const uint16_t *base, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_ushort %0, %1" : : "v"(voff), "s"(base), "s"(lds_dst) : "memory", "m0");
}

// One stream, one decoding wavefront (+ the prefetching one).  PLANE: a CDF row serves `plane` consecutive symbols
// (factorised prior of z), else one row per symbol.  Written for what a lone wavefront pays (tools/lat_probe.hip: 4 cycles
// per instruction of any kind, +24 per taken branch, +16..20 per VALU -> SALU crossing): per symbol a window read, two
// subtractions, one mad + shift, compare + popcount, two readlanes, the interval update, ONE renormalisation test and,
// when it fires, one merged shift; the loop is unrolled by hand, the window's answer is taken unconditionally and
// replaced on the rare symbol outside the window.
// WINDOWED: `rows` holds only the 64-entry window of every position (laplace_cdf_windows_kernel); the slow path evaluates
// the entries it needs itself from the position's Laplace scale, which the prefetching wave has put beside the windows
// (`sigma_ring`, LDS).  At high rate a quarter of an I frame's symbols take that path and it is a serial chain of
// ~100 fp64 / fp32 instructions: aivc_laplace_cdf_u16_scale is the row function laid out without divergent branches.
template <bool PLANE, bool WINDOWED = false>
__device__ __forceinline__ uint32_t decode_stream(const uint8_t *__restrict__ bytes, const uint16_t *__restrict__ rows,
                                              const aivc_rc_stream &st, uint16_t *__restrict__ sym, uint32_t *ring,
                                              const int lane, const float *sigma_ring = nullptr) {
  static_assert(!(PLANE && WINDOWED), "windows are per-position Laplace rows");
  constexpr int ROWLEN = WINDOWED ? CDF_WIN : AIVC_CDF_ROW;  // uint16 entries per stored row
  constexpr int WOFF = WINDOWED ? 0 : DEC_WIN0;              // position of the window inside a stored row
  BitWin bw;
  bw.in = reinterpret_cast<const uint32_t *>(bytes + st.in_off);
  bw.n_words = (st.in_len + 3u) / 4u;
  bw.tail_bytes = st.in_len & 3u;
  bw.lane = lane;
  bw.init();
  uint32_t low = 0, high = 0xFFFFFFFFu;
  const uint32_t n_sym = st.n_sym, plane = st.plane;
  const uint16_t *base = rows + st.row_off * ROWLEN;
  auto row_of = [&](uint32_t i) -> const uint16_t * { return base + (uint64_t)(PLANE ? i / plane : i) * ROWLEN; };
  uint32_t ew_next = 0;

  uint32_t mysym = 0;
#pragma unroll 1
  for (uint32_t first = 0; first < n_sym; first += (uint32_t)DEC_HALF) {
    const uint32_t cnt = min((uint32_t)DEC_HALF, n_sym - first);
    __syncthreads();  // the prefetching wave has filled this block's half of the ring (and starts on the other one)
    const uint32_t *hw = ring + ((first / DEC_HALF) & 1u) * (DEC_HALF * 64) + lane;
    ew_next = hw[0];
    // the second half of a symbol step: record, interval update, renormalisation.  A lambda called from the fast AND
    // the slow path: as code after their join the compiler keeps a "came from the fast path" flag and tests it on
    // every symbol (a second branch per symbol)
    auto finish = [&](const uint32_t j, const uint32_t m, const uint32_t t_lo, const uint32_t t_hi) __attribute__((always_inline)) {
      mysym = (uint32_t)lane == ((first + j) & 63u) ? m : mysym;
      // interval update (also after the last symbol: the state is dead then, reads past the payload are zeros)
      high = low + t_hi - 1u;
      low = low + t_lo;
      // Renormalisation, ONE test and one shift (round 5: a taken branch costs a lone wavefront 24 cycles, an
      // instruction 4 -- tools/lat_probe.hip): E1 / E2 shift out the n = clz(low ^ high) leading bits on which low and
      // high agree, E3 the m3 positions below them where (low, high) = (1, 0) straddle the middle.  The span exceeds 2^30
      // before a symbol and a CDF step is at least 2^-16 of it, so n + m3 <= 18: both are one shift of the interval and
      // one of the bit window (the window's top bit flips once if any straddle step was taken, as after m3 single steps).
      const uint32_t x = low ^ high;
      const uint32_t yy = (low & ~high) << 1;
      if ((yy | ~x) & 0x80000000u) {  // a leading bit agrees, or the position below the top straddles
        const uint32_t n = (uint32_t)__builtin_clz(x);  // x != 0 (low < high)
        const uint32_t m3 = (uint32_t)__builtin_clz(~(yy << n));  // leading ones of yy << n: never all ones
        const uint32_t sh = n + m3;
        low = (low << sh) & 0x7FFFFFFFu;
        high = (high << sh) | ((1u << sh) - 1u) | 0x80000000u;
        bw.consume(sh);
        bw.win ^= (uint64_t)(m3 != 0u) << 63;
      }
    };
    // one symbol; the loop over a block is unrolled by hand (hipcc does not unroll this body, and the loop's own
    // taken branch costs a lone wavefront 24 cycles per symbol)
    auto step = [&](const uint32_t j) __attribute__((always_inline)) {
      const uint32_t ew = ew_next;
      // window of the next symbol: read now, used in the next step, so the LDS latency hides behind this symbol's
      // arithmetic (at a block's last symbol the slot belongs to the other half: read, never used)
      ew_next = hw[(j + 1u) * 64u];
      const uint32_t hl = high - low;  // span - 1
      const uint32_t d = bw.value() - low;
      uint32_t m, t_lo, t_hi;
      // fast path: entries are strictly increasing, so the lanes with t <= d form a prefix
      const uint32_t tw = scaled(ew, hl);
      const uint32_t cw = (uint32_t)__builtin_popcountll(__ballot(tw <= d));
      // The window's answer is taken unconditionally (v_readlane uses the low 6 bits of its lane operand: any cw is a
      // valid read) and REPLACED on the rare symbol outside it: an if without else is one not-taken branch per symbol,
      // the if / else form made the compiler keep and test a "came from the fast path" flag as well
      m = (uint32_t)DEC_WIN0 - 1u + cw;
      t_lo = rl(tw, (int)cw - 1);
      t_hi = rl(tw, (int)cw);
      if (__builtin_expect(cw - 1u >= 63u, 0)) {
        // symbol outside [-32, 30]: fetch and search the whole row (8 entries per lane); rare
        uint32_t i = first + j;
        asm volatile("" : "+s"(i));  // rare path: no running row offset kept in the loop for it
        if constexpr (WINDOWED) {
          // The symbol lies on a KNOWN side of the window (cw = 0: below entry 224, cw = 64: at or above entry 287) and
          // most often just beyond it: the wavefront evaluates the 64 entries adjacent on that side from the position's
          // sigma (one CDF point per lane, the same function that built the window) and searches them, then the next
          // 64 if it is not there.  (Round 4 rebuilt the whole row in two dependent evaluation passes -- every 8th entry,
          // then the octet: twice the fp64 work on the common near miss; I-frame streams at high rate take this path on
          // a quarter of their symbols.)  m = (entries 0 .. 511 with t <= d) - 1, as the full-row search defines it.
          // (the Laplace scale b = sigma / sqrt(2) of the position: the division is done by the prefetching wave)
          const float sg = sigma_ring[((first / DEC_HALF) & 1u) * DEC_HALF + j];
          if (cw == 0u) {
            uint32_t above = rl(tw, 0);  // t of the entry right above the block being searched
            int base = DEC_WIN0 - 64;
#pragma unroll 1
            for (;;) {
              const uint32_t tf = scaled((uint32_t)aivc_laplace_cdf_u16_scale(base + lane, sg), hl);
              uint32_t c = (uint32_t)__builtin_popcountll(__ballot(tf <= d));
              // (the block at base 0 ends the search whatever it holds: a corrupt stream can ask for less than entry 0
              // when sigma is large enough for entry 0 to be non-zero; the full-row search answers symbol 0 then)
              if (base == 0) c = max(c, 1u);
              if (c > 0u) {
                m = (uint32_t)base + c - 1u;
                t_lo = rl(tf, (int)c - 1);
                t_hi = c < 64u ? rl(tf, (int)c) : above;
                break;
              }
              above = rl(tf, 0);
              base = base >= 64 ? base - 64 : 0;
            }
          } else {
            uint32_t below = rl(tw, 63);  // t of the entry right below the block being searched
            int base = DEC_WIN0 + 64;
#pragma unroll 1
            for (;;) {
              const int k = base + lane;  // entries up to 512 exist (512 = lower bound of symbol 512); 0 .. 511 count
              const uint32_t tf = scaled((uint32_t)aivc_laplace_cdf_u16_scale(min(k, AIVC_MAX_SYMBOL), sg), hl);
              const uint32_t c = (uint32_t)__builtin_popcountll(__ballot(k < AIVC_MAX_SYMBOL && tf <= d));
              if (c < 64u) {  // (the block holding entry 511 has c <= 32)
                m = (uint32_t)base + c - 1u;
                t_lo = c > 0u ? rl(tf, (int)c - 1) : below;
                t_hi = rl(tf, (int)c);  // entry base + c <= 512
                break;
              }
              below = rl(tf, 63);
              base += 64;
            }
          }
        } else {
          uint32_t t[9];
          const uint16_t *row = row_of(i);
          const uint4 e = *reinterpret_cast<const uint4 *>(row + lane * 8);
          const uint32_t nx = row[lane * 8 + 8];
          t[0] = scaled(e.x & 0xFFFFu, hl); t[1] = scaled(e.x >> 16, hl);
          t[2] = scaled(e.y & 0xFFFFu, hl); t[3] = scaled(e.y >> 16, hl);
          t[4] = scaled(e.z & 0xFFFFu, hl); t[5] = scaled(e.z >> 16, hl);
          t[6] = scaled(e.w & 0xFFFFu, hl); t[7] = scaled(e.w >> 16, hl);
          t[8] = scaled(nx, hl);
          uint32_t total = 0;
#pragma unroll
          for (int k = 0; k < 8; ++k) total += (uint32_t)__builtin_popcountll(__ballot(t[k] <= d));
          m = total > 0 ? total - 1 : 0;
          const int L = (int)(m >> 3);
          const uint32_t idx = m & 7u;
          uint32_t s_lo = t[0], s_hi = t[1];
#pragma unroll
          for (int k = 1; k < 8; ++k) {
            s_lo = idx == (uint32_t)k ? t[k] : s_lo;
            s_hi = idx == (uint32_t)k ? t[k + 1] : s_hi;
          }
          t_lo = rl(s_lo, L);
          t_hi = rl(s_hi, L);
        }
        if (m == 511u && t_hi <= d) {  // symbol 512 (value +256, torchac's max_symbol): its upper bound is 2^16,
          m = 512u;                    // i.e. t = span: high stays
          t_lo = t_hi;
          t_hi = hl + 1u;
        }
      }
      finish(j, m, t_lo, t_hi);
    };
    uint32_t j = 0;
#pragma unroll 1
    for (; j + 4u <= cnt; j += 4u) {
      step(j);
      step(j + 1u);
      step(j + 2u);
      step(j + 3u);
    }
#pragma unroll 1
    for (; j < cnt; ++j) step(j);
    // lane l holds symbol l of the 64-symbol group this block belongs to: stored when the group (or the stream) ends
    const uint32_t g0 = first & ~63u, done = first + cnt;
    if (((done & 63u) == 0u || done == n_sym) && (uint32_t)lane < done - g0) sym[st.out_off + g0 + lane] = (uint16_t)mysym;
  }
  // bits shifted in by renormalisation over the whole stream (the window was primed with 64 at word index 2)
  return bw.wi * 32u - bw.avail;
}

// The second wavefront of a decoding workgroup: LDS-DMA of the 64-entry windows, DEC_HALF symbols per barrier.
template <bool PLANE, bool WINDOWED = false>
__device__ __forceinline__ void decode_prefetcher(const uint16_t *__restrict__ rows, const aivc_rc_stream &st, uint32_t *ring,
                                                  const int lane, const float *__restrict__ sigma_pos = nullptr,
                                                  float *sigma_ring = nullptr) {
  constexpr int ROWLEN = WINDOWED ? CDF_WIN : AIVC_CDF_ROW;
  constexpr int WOFF = WINDOWED ? 0 : DEC_WIN0;
  const uint32_t ring_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)ring;
  const uint32_t n_sym = st.n_sym, plane = st.plane;
  const uint32_t n_rows = PLANE ? (n_sym + plane - 1u) / plane : n_sym;
  const uint16_t *base = rows + st.row_off * ROWLEN;
  // per-lane byte offset of its window entry in the row of the symbol being fetched (host side: a stream's rows span
  // < 4 GiB); parked on the last row past the end of the stream (add + min, no counter)
  uint32_t voff = (uint32_t)(WOFF + lane) * 2u;
  const uint32_t vlast = voff + (n_rows - 1u) * (uint32_t)(ROWLEN * 2);
  uint32_t in_plane = 0;
#pragma unroll 1
  for (uint32_t first = 0; first < n_sym; first += (uint32_t)DEC_HALF) {
    const uint32_t half = ring_base + ((first / DEC_HALF) & 1u) * (DEC_HALF * 256);
    float sg = 1.f;
    if constexpr (WINDOWED)  // the block's sigmas, one per lane, for the decoding wave's slow path
      sg = sigma_pos[st.row_off + min(first + (uint32_t)(lane & (DEC_HALF - 1)), n_sym - 1u)];
#pragma unroll 8
    for (uint32_t k = 0; k < (uint32_t)DEC_HALF; ++k) {
      window_dma(base, voff, half + k * 256u);
      uint32_t step = (uint32_t)(ROWLEN * 2);
      if (PLANE) {
        ++in_plane;
        const bool wrap = in_plane == plane;
        step = wrap ? step : 0u;
        in_plane = wrap ? 0u : in_plane;
      }
      voff = min(voff + step, vlast);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the half is complete (and no DMA outlives the workgroup's LDS)
    if constexpr (WINDOWED)
      if (lane < DEC_HALF) sigma_ring[((first / DEC_HALF) & 1u) * DEC_HALF + lane] = aivc_laplace_scale(sg);
    __syncthreads();
  }
}

__global__ __launch_bounds__(128) void range_decode_kernel(const uint8_t *__restrict__ bytes,
                                                           const uint16_t *__restrict__ rows, aivc_rc_batch batch,
                                                           uint16_t *__restrict__ sym, uint32_t *__restrict__ consumed) {
  __shared__ uint32_t ring[DEC_SLOTS * 64];  // the DMA writes one dword per lane (a ushort load lands zero-extended at lane * 4)
  const aivc_rc_stream st = batch.s[blockIdx.x];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (st.n_sym == 0) {
    if (consumed && threadIdx.x == 0) consumed[blockIdx.x] = 0;
    return;
  }
  __builtin_amdgcn_s_setprio(3);  // latency-critical serial wave (see range_encode_kernel)
  if (wave == 1) {
    if (st.plane) decode_prefetcher<true>(rows, st, ring, lane);
    else decode_prefetcher<false>(rows, st, ring, lane);
    return;
  }
  uint32_t bits;
  if (st.plane) bits = decode_stream<true>(bytes, rows, st, sym, ring, lane);
  else bits = decode_stream<false>(bytes, rows, st, sym, ring, lane);
  if (consumed && lane == 0) consumed[blockIdx.x] = bits;
}

__global__ __launch_bounds__(128) void range_decode_windows_kernel(const uint8_t *__restrict__ bytes,
                                                                   const uint16_t *__restrict__ win,
                                                                   const float *__restrict__ sigma_pos, aivc_rc_batch batch,
                                                                   uint16_t *__restrict__ sym, uint32_t *__restrict__ consumed) {
  __shared__ uint32_t ring[DEC_SLOTS * 64];
  __shared__ float sigma_ring[2 * DEC_HALF];
  const aivc_rc_stream st = batch.s[blockIdx.x];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (st.n_sym == 0) {
    if (consumed && threadIdx.x == 0) consumed[blockIdx.x] = 0;
    return;
  }
  __builtin_amdgcn_s_setprio(3);  // (the convolutions' priority instead measured the same: 74.3 vs 74.0 fps at high rate)
  if (wave == 1) {
    decode_prefetcher<false, true>(win, st, ring, lane, sigma_pos, sigma_ring);
    return;
  }
  const uint32_t bits = decode_stream<false, true>(bytes, win, st, sym, ring, lane, sigma_ring);
  if (consumed && lane == 0) consumed[blockIdx.x] = bits;
}

}  // namespace aivc

using namespace aivc;

AIVC_EXPORT int aivc_balle_cdf_table(const float *params, int32_t c, uint16_t *table, float *cdf_f32,
                                     aivc_stream_t stream) {
  if (!params || !table || c <= 0) return AIVC_ERR_ARG;
  hipLaunchKernelGGL(balle_cdf_table_kernel, dim3(cdiv((size_t)c * AIVC_CDF_ROW, 256)), dim3(256), 0,
                     to_stream(stream), params, c, table, cdf_f32);
  return check_launch("balle_cdf_table");
}

AIVC_EXPORT int aivc_nonzero_maps_batch(const int16_t *q, int32_t n, size_t npix, int32_t c, uint8_t *flags,
                                        aivc_stream_t stream) {
  if (!q || !flags || c <= 0 || c > AIVC_MAX_MAPS || n <= 0 || n > 65535) return AIVC_ERR_ARG;
  if (hipMemsetAsync(flags, 0, (size_t)n * c, to_stream(stream)) != hipSuccess) return check_launch("nonzero_maps memset");
  if (npix == 0) return AIVC_OK;
  unsigned grid = cdiv(npix * c, 256 * 8);
  grid = grid > 2048 ? 2048 : (grid == 0 ? 1 : grid);
  hipLaunchKernelGGL(nonzero_maps_kernel, dim3(grid, (unsigned)n), dim3(256), 0, to_stream(stream), q, npix, c, flags);
  return check_launch("nonzero_maps");
}

AIVC_EXPORT int aivc_nonzero_maps(const int16_t *q, size_t npix, int32_t c, uint8_t *flags, aivc_stream_t stream) {
  return aivc_nonzero_maps_batch(q, 1, npix, c, flags, stream);
}

static int check_maps(const aivc_map_list *maps, int c) {
  if (!maps || maps->n_maps < 0 || maps->n_maps > c || c > AIVC_MAX_MAPS) return AIVC_ERR_ARG;
  for (int i = 0; i < maps->n_maps; ++i)
    if (maps->idx[i] >= c) return AIVC_ERR_ARG;
  return AIVC_OK;
}

AIVC_EXPORT int aivc_laplace_cdf_rows(const float *sigma, size_t npix, int32_t c, const aivc_map_list *maps,
                                      uint16_t *rows, aivc_stream_t stream) {
  if (!sigma || !rows || c <= 0) return AIVC_ERR_ARG;
  if (int rc = check_maps(maps, c)) return rc;
  const size_t total = (size_t)maps->n_maps * npix * (AIVC_CDF_ROW / 8);
  if (total == 0) return AIVC_OK;
  hipLaunchKernelGGL(laplace_cdf_rows_kernel, dim3(cdiv(total, 256)), dim3(256), 0, to_stream(stream), sigma, npix,
                     c, *maps, rows);
  return check_launch("laplace_cdf_rows");
}

AIVC_EXPORT int aivc_laplace_bounds(const float *sigma, const int16_t *q, size_t npix, int32_t c,
                                    const aivc_map_list *maps, uint32_t *bounds, aivc_stream_t stream) {
  if (!sigma || !q || !bounds || c <= 0) return AIVC_ERR_ARG;
  if (int rc = check_maps(maps, c)) return rc;
  const size_t total = (size_t)maps->n_maps * npix;
  if (total == 0) return AIVC_OK;
  hipLaunchKernelGGL(laplace_bounds_kernel, dim3(cdiv(total, 256)), dim3(256), 0, to_stream(stream), sigma, q, npix,
                     c, *maps, bounds);
  return check_launch("laplace_bounds");
}

AIVC_EXPORT int aivc_table_bounds(const uint16_t *table, const int16_t *q, size_t npix, int32_t c, uint32_t *bounds,
                                  aivc_stream_t stream) {
  if (!table || !q || !bounds || c <= 0) return AIVC_ERR_ARG;
  if (npix == 0) return AIVC_OK;
  hipLaunchKernelGGL(table_bounds_kernel, dim3(cdiv((size_t)c * npix, 256)), dim3(256), 0, to_stream(stream), table,
                     q, npix, c, bounds);
  return check_launch("table_bounds");
}

AIVC_EXPORT int aivc_scatter_symbols(const uint16_t *sym, size_t npix, int32_t c, const aivc_map_list *maps,
                                     int16_t *q, aivc_stream_t stream) {
  if (!q || c <= 0) return AIVC_ERR_ARG;
  if (int rc = check_maps(maps, c)) return rc;
  if (maps->n_maps > 0 && !sym) return AIVC_ERR_ARG;
  if (npix == 0) return AIVC_OK;
  InvMap inv;
  for (int i = 0; i < AIVC_MAX_MAPS; ++i) inv.slot[i] = -1;
  for (int i = 0; i < maps->n_maps; ++i) inv.slot[maps->idx[i]] = (int16_t)i;
  const int vec = (c & 7) == 0 && ((uintptr_t)q & 15) == 0;
  hipLaunchKernelGGL(scatter_symbols_kernel, dim3(cdiv(npix * ((c + 7) / 8), 256)), dim3(256), 0, to_stream(stream), sym, npix,
                     c, inv, q, vec);
  return check_launch("scatter_symbols");
}

AIVC_EXPORT int aivc_laplace_cdf_windows_batch(const float *sigma, int32_t n, size_t npix, int32_t c,
                                               const aivc_frame_maps *frames, int32_t max_maps, uint16_t *win,
                                               float *sigma_pos, aivc_stream_t stream) {
  if (!sigma || !frames || !win || !sigma_pos || c <= 0 || c > AIVC_MAX_MAPS || n <= 0 || n > 65535 || max_maps < 0 || max_maps > c)
    return AIVC_ERR_ARG;
  const size_t n_pos = (size_t)max_maps * npix;
  if (n_pos == 0) return AIVC_OK;
  const size_t total = ((n_pos + 63) / 64) * (CDF_WIN / 8) * 64;
  hipLaunchKernelGGL(laplace_cdf_windows_batch_kernel, dim3(cdiv(total, 256), (unsigned)n), dim3(256), 0, to_stream(stream),
                     sigma, npix, c, frames, win, sigma_pos);
  return check_launch("laplace_cdf_windows_batch");
}

AIVC_EXPORT int aivc_laplace_bounds_batch(const float *sigma, const int16_t *q, int32_t n, size_t npix, int32_t c,
                                          const aivc_frame_maps *frames, int32_t max_maps, uint32_t *bounds,
                                          aivc_stream_t stream) {
  if (!sigma || !q || !frames || !bounds || c <= 0 || c > AIVC_MAX_MAPS || n <= 0 || n > 65535 || max_maps < 0 || max_maps > c)
    return AIVC_ERR_ARG;
  const size_t total = (size_t)max_maps * npix;
  if (total == 0) return AIVC_OK;
  hipLaunchKernelGGL(laplace_bounds_batch_kernel, dim3(cdiv(total, 256), (unsigned)n), dim3(256), 0, to_stream(stream), sigma, q,
                     npix, c, frames, bounds);
  return check_launch("laplace_bounds_batch");
}

AIVC_EXPORT int aivc_table_bounds_batch(const uint16_t *table, const int16_t *q, int32_t n, size_t npix, int32_t c,
                                        uint32_t *bounds, aivc_stream_t stream) {
  if (!table || !q || !bounds || c <= 0 || n <= 0 || n > 65535) return AIVC_ERR_ARG;
  if (npix == 0) return AIVC_OK;
  hipLaunchKernelGGL(table_bounds_batch_kernel, dim3(cdiv((size_t)c * npix, 256), (unsigned)n), dim3(256), 0, to_stream(stream),
                     table, q, npix, c, bounds);
  return check_launch("table_bounds_batch");
}

AIVC_EXPORT int aivc_scatter_symbols_batch(const uint16_t *sym, int32_t n, size_t npix, int32_t c,
                                           const aivc_frame_maps *frames, int16_t *q, aivc_stream_t stream) {
  if (!q || !frames || !sym || c <= 0 || c > AIVC_MAX_MAPS || n <= 0 || n > 65535) return AIVC_ERR_ARG;
  if (npix == 0) return AIVC_OK;
  const int vec = (c & 7) == 0 && ((uintptr_t)q & 15) == 0;
  hipLaunchKernelGGL(scatter_symbols_batch_kernel, dim3(cdiv(npix * ((c + 7) / 8), 256), (unsigned)n), dim3(256), 0,
                     to_stream(stream), sym, npix, c, frames, q, vec);
  return check_launch("scatter_symbols_batch");
}

static int check_batch(const aivc_rc_batch *b) {
  if (!b || b->n_streams < 0 || b->n_streams > AIVC_RC_MAX_STREAMS) return AIVC_ERR_ARG;
  return AIVC_OK;
}

AIVC_EXPORT int aivc_range_encode(const uint32_t *bounds, const aivc_rc_batch *batch, uint8_t *out,
                                  uint32_t *out_len, aivc_stream_t stream) {
  if (!out || !out_len) return AIVC_ERR_ARG;
  if (int rc = check_batch(batch)) return rc;
  if (batch->n_streams == 0) return AIVC_OK;
  for (int i = 0; i < batch->n_streams; ++i) {
    if (batch->s[i].out_off % 4) return AIVC_ERR_ARG;
    // whole 32-bit words only (both encoders store words), and at least one: the lane packer's last-word clamp is
    // out_cap / 4 - 1
    if (batch->s[i].out_cap < 4 || batch->s[i].out_cap % 4) return AIVC_ERR_ARG;
    if (batch->s[i].n_sym && !bounds) return AIVC_ERR_ARG;
  }
  // one stream per lane (one wavefront for the whole batch) unless AIVC_RC_ENCODE=wave asks for the wave-per-stream kernel
  const char *mode = getenv("AIVC_RC_ENCODE");  // (read per call: the tests run both kernels in one process)
  if (mode && !strcmp(mode, "wave"))
    hipLaunchKernelGGL(range_encode_kernel, dim3(batch->n_streams), dim3(64), 0, to_stream(stream), bounds, *batch, out, out_len);
  else {
    // A long launch asks for LDS it does not use, so that no convolution workgroup (16-100 KB of ring each) fits beside it
    // on its CU: a coder wave that shares its SIMD with matrix-pipe waves runs 3.5-4x slower (every one of its dependent
    // instructions waits for the issue port: tools/bench_rangecoder.py LOAD=1, 0.07 -> 0.29 us per symbol), and at high
    // rate the last level's streams are the encoder's tail.  One CU of 256 for the duration of the launch; short launches
    // (hidden under the transforms anyway) take what is free.
    static const int own_cu = getenv("AIVC_RC_OWN_CU") ? atoi(getenv("AIVC_RC_OWN_CU")) : 1;  // tuning aid: 0 = never
    uint32_t longest = 0;
    for (int i = 0; i < batch->n_streams; ++i) longest = batch->s[i].n_sym > longest ? batch->s[i].n_sym : longest;
    size_t pad_lds = 0;
    if (own_cu && longest >= 65536u) {
      static LdsOptIn opt_in;
      pad_lds = 100 * 1024;  // + 48.5 KB of its own: 11 KB of the CU's 160 left
      if (!opt_in.raise(reinterpret_cast<const void *>(range_encode_lanes_kernel), pad_lds)) pad_lds = 0;
    }
    hipLaunchKernelGGL(range_encode_lanes_kernel, dim3(1), dim3(192), pad_lds, to_stream(stream), bounds, *batch, out, out_len);
  }
  return check_launch("range_encode");
}

AIVC_EXPORT int aivc_laplace_cdf_windows(const float *sigma, size_t npix, int32_t c, const aivc_map_list *maps,
                                         uint16_t *win, float *sigma_pos, aivc_stream_t stream) {
  if (!sigma || !win || !sigma_pos || c <= 0) return AIVC_ERR_ARG;
  if (int rc = check_maps(maps, c)) return rc;
  const size_t n_pos = (size_t)maps->n_maps * npix;
  if (n_pos == 0) return AIVC_OK;
  const size_t total = ((n_pos + 63) / 64) * (CDF_WIN / 8) * 64;  // a wavefront per (64 positions, chunk)
  hipLaunchKernelGGL(laplace_cdf_windows_kernel, dim3(cdiv(total, 256)), dim3(256), 0, to_stream(stream), sigma, npix, c,
                     *maps, win, sigma_pos);
  return check_launch("laplace_cdf_windows");
}

AIVC_EXPORT int aivc_range_decode_windows(const uint8_t *bytes, const uint16_t *win, const float *sigma_pos,
                                          const aivc_rc_batch *batch, uint16_t *sym, uint32_t *consumed_bits,
                                          aivc_stream_t stream) {
  if (!bytes || !win || !sigma_pos || !sym) return AIVC_ERR_ARG;
  if (int rc = check_batch(batch)) return rc;
  if (batch->n_streams == 0) return AIVC_OK;
  for (int i = 0; i < batch->n_streams; ++i) {
    if (batch->s[i].in_off % 4 || batch->s[i].plane != 0) return AIVC_ERR_ARG;
    if (((uint64_t)batch->s[i].n_sym + 1) * (uint64_t)(CDF_WIN * 2) >= 0x100000000ull) return AIVC_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(range_decode_windows_kernel, dim3(batch->n_streams), dim3(128), 0, to_stream(stream), bytes, win,
                     sigma_pos, *batch, sym, consumed_bits);
  return check_launch("range_decode_windows");
}

AIVC_EXPORT int aivc_range_decode(const uint8_t *bytes, const uint16_t *rows, const aivc_rc_batch *batch,
                                  uint16_t *sym, uint32_t *consumed_bits, aivc_stream_t stream) {
  if (!bytes || !rows || !sym) return AIVC_ERR_ARG;
  if (int rc = check_batch(batch)) return rc;
  if (batch->n_streams == 0) return AIVC_OK;
  for (int i = 0; i < batch->n_streams; ++i) {
    if (batch->s[i].in_off % 4) return AIVC_ERR_ARG;
    // the window prefetcher addresses a stream's rows with 32-bit byte offsets
    const uint64_t n_rows = batch->s[i].plane ? (batch->s[i].n_sym + batch->s[i].plane - 1) / batch->s[i].plane : batch->s[i].n_sym;
    if ((n_rows + 1) * (uint64_t)(AIVC_CDF_ROW * 2) >= 0x100000000ull) return AIVC_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(range_decode_kernel, dim3(batch->n_streams), dim3(128), 0, to_stream(stream), bytes, rows,
                     *batch, sym, consumed_bits);
  return check_launch("range_decode");
}
