// conv_images.hip -- the first layer of the analysis transforms: 5x5 stride-2 replicate-padded convolution to 64
// channels (+ bias, + fused GDN) over the concatenation of up to three 3-channel images, read straight from their
// sources (8-bit 4:2:0 planes or float NHWC).  Bit identical to aivc_pack_images followed by aivc_conv2d on the
// packed [n][h][w][4 * n_img] tensor (AIVC_CONV_SPARSE4), whose 48-byte-per-pixel round trip through HBM it removes.
//
// Why a kernel of its own: with 4 / 8 / 12 stored input channels a 32-wide K-tile of the generic implicit GEMM
// straddles several kernel taps, so its loader decodes (tap, channel) per 16-byte load and the layer ran at half
// the rate of the others.  Here the reduction is short enough (K = 100 / 200 / 300) to keep ALL of it on chip:
//   - a workgroup owns 4 x 32 output pixels; the 11 x 67 input patch is converted once (k / 255.0f from a table,
//     chroma by nearest neighbour, replicate padding = clamped coordinates) into LDS, split by column parity and
//     image so that the 32 pixels of a wavefront read 512 contiguous bytes for every (tap, image);
//   - the whole weight matrix [64][K] sits in LDS beside it (row stride chosen conflict-free for ds_read_b128);
//   - a workgroup walks 8 consecutive tiles: weights, gamma fragments (registers), bias / beta are fetched once, the
//     raw samples of the next tile are in flight during the matrix work of the current one;
//   - per tile 13 / 25 / 38 octets of v_mfma_f32_32x32x2_f32 with both operands read from LDS, no staging, no
//     barrier inside; the zero fourth channel of every image is never multiplied (an exact no-op);
//   - reduction order = the contract of the conv family (include/aivc_hip.h): kk = tap * c_in + 4 * image + c in
//     groups of 8, inside a group in AIVC_K_ORDER -- octet o is quads 2 o (lanes 0-31) and 2 o + 1 (lanes 32-63);
//   - the matrix product is taken TRANSPOSED (weights as the A operand, pixels as B: D[channel][pixel], the same fmaf
//     chain per output -- the two factors of a product commute): accumulator register 4 o + s of lane half h then holds
//     channel 8 o + s + 4 h of the lane's pixel, which is exactly the B operand the contract's K order asks of the
//     second GEMM at step (o, s);
//   - fused GDN as in conv_mfma.hip (second MFMA GEMM over the squares), but the squares go from the accumulators to
//     the matrix pipe in registers (no LDS round trip, no barrier) and gamma is the A operand, resident in registers;
//   - a lane owns 4 consecutive channels of its pixel per accumulator quad: outputs leave as 16-byte stores.
// roofline: fp32 MFMA (157.3 TFLOP/s); algorithmic FLOPs 2 * 25 * 3 n_img * 64 (+ 2 * 64 * 64 GDN) per output pixel.
// Measured (r02, 16 x 1080p): 2.1 / 2.8 / 5.2 ms for 1 / 2 / 3 images against 1.9 / 2.8 / 3.7 ms of the generic kernel
// on the packed tensor plus 0.2 / 0.4 / 0.6 ms to pack it: at par for one and two images (and 0.5 / 1 GB less HBM
// traffic and memory per call), slower for three (113 KB of LDS: one workgroup per CU) -- the codec sends those
// down the pack + conv path.  The layer is bounded by its epilogue (64 square roots and divisions per pixel against
// 75 / 150 / 225 multiply-adds per output), not by the matrix pipe (50 % busy).
#include <type_traits>

#include "common.h"

namespace aivc {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct ImgArgs {
  aivc_image_src src[AIVC_MAX_IMAGES];
  const float *w, *bias, *gdn_beta, *gdn_gamma;
  float *y;
  int n, h, w_in, ho, wo, gdn, act1;
  int tiles_x, tiles_y;
};

constexpr int IC_TH = 4, IC_TW = 32;    // output pixels per workgroup: one row of 32 per wavefront
constexpr int IC_PR = 2 * IC_TH + 3;    // patch rows
constexpr int IC_PC = 2 * IC_TW + 3;    // patch columns
constexpr int IC_HALF = IC_TW + 2;      // columns of one parity
constexpr int IC_PLANE = IC_HALF * 4;   // floats of one (row, parity, image) plane
constexpr int IC_CO = 64;

// weight row stride in floats: = 36 / 12 / 44 (mod 64), so the 16 lanes of a ds_read_b128 phase hit 16 distinct
// bank quads
template <int NIMG>
constexpr int ic_wstride() { return NIMG == 2 ? 204 : 100 * NIMG; }
template <int NIMG>
constexpr int ic_region_floats() { return IC_PR * 2 * NIMG * IC_PLANE; }  // the patch
template <int NIMG>
constexpr int ic_lds_floats() { return IC_CO * ic_wstride<NIMG>() + ic_region_floats<NIMG>(); }

constexpr int IC_TPW = 8;  // consecutive tiles per workgroup: weights and gamma are fetched once for all of them

template <int NIMG>
__global__ __launch_bounds__(256, 2) void conv_images_kernel(ImgArgs a) {
  constexpr int NQ = 25 * NIMG;        // quads (tap, image) of the reduction
  constexpr int NO = (NQ + 1) / 2;     // octets
  constexpr int WS = ic_wstride<NIMG>();
  constexpr int K = 100 * NIMG;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float lut[256];
  __shared__ __attribute__((aligned(16))) float chan[2][IC_CO];  // bias, beta (a lane needs them by channel quad: broadcast reads)
  float *Wl = smem;
  float *patch = smem + IC_CO * WS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 31, hh = lane >> 5;
  lut[tid] = (float)tid / 255.0f;
  if (tid < IC_CO) {
    chan[0][tid] = a.bias ? a.bias[tid] : 0.0f;
    chan[1][tid] = a.gdn ? a.gdn_beta[tid] : 0.0f;
  }
  const int H = a.h, W = a.w_in, hc = (H + 1) / 2, wc = (W + 1) / 2;
  const uint32_t ntiles = (uint32_t)a.n * (uint32_t)a.tiles_y * (uint32_t)a.tiles_x;
  const uint32_t t_first = blockIdx.x * (uint32_t)IC_TPW;
  const uint32_t t_end = min(t_first + (uint32_t)IC_TPW, ntiles);

  // ---- once per workgroup: weights -> LDS (float4 = one quad of one output channel), GDN operands -> registers
  {
    constexpr int WU = (IC_CO * NQ + 255) / 256;
    float4 wreg[WU];
#pragma unroll
    for (int i = 0; i < WU; ++i) {
      const int u = tid + 256 * i;
      const int uc = u < IC_CO * NQ ? u : 0;
      const int co = uc / NQ, q = uc - co * NQ;
      wreg[i] = *reinterpret_cast<const float4 *>(a.w + (size_t)co * K + 4 * q);
    }
#pragma unroll
    for (int i = 0; i < WU; ++i) {
      const int u = tid + 256 * i;
      if (u < IC_CO * NQ) {
        const int co = u / NQ, q = u - co * NQ;
        *reinterpret_cast<float4 *>(Wl + co * WS + 4 * q) = make_float4(wreg[i].x, wreg[i].y, wreg[i].z, wreg[i].w);
      }
    }
  }
  const bool gdn = a.gdn != 0;
  // A fragments of the GDN GEMM (gamma[i][k], i = 32 j + p, k = 32 c + 8 o + 4 hh ..): 16 float4 per lane, kept
  // for all tiles -- the second GEMM then needs no shared staging, no barrier and no global latency per tile
  float4 greg[2][4][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int o = 0; o < 4; ++o)
        greg[c][o][j] = gdn ? *reinterpret_cast<const float4 *>(a.gdn_gamma + (size_t)(32 * j + p) * IC_CO + 32 * c + 8 * o + 4 * hh)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // ---- patch staging units of this thread (unit = one (row, column, image) sample triple).  Their decomposition is
  // recomputed per tile from an opaque copy of the thread index: kept live across the matrix work (4 values per
  // unit) it pushed the kernel over the 256 registers of two waves per SIMD
  constexpr int PU = (IC_PR * IC_PC * NIMG + 255) / 256;
  auto unit_of = [&](int tt, int i, int &pr, int &pc, int &img, int &dst) {
    const int u = tt + 256 * i;
    const bool live = u < IC_PR * IC_PC * NIMG;
    const int uc = live ? u : 0;
    img = uc % NIMG;
    const int r2 = uc / NIMG;
    pc = r2 % IC_PC;
    pr = r2 / IC_PC;
    dst = live ? (((pr * 2 + (pc & 1)) * NIMG + img) * IC_HALF + (pc >> 1)) * 4 : -1;  // LDS float offset, < 0: none
  };
  uint32_t raw[PU][3];  // bytes (8-bit sources) or float bits of the NEXT tile's samples
  auto tile_of = [&](uint32_t t, int &b, int &oy0, int &ox0) {
    const uint32_t tx = t % (uint32_t)a.tiles_x;
    t /= (uint32_t)a.tiles_x;
    oy0 = (int)(t % (uint32_t)a.tiles_y) * IC_TH;
    b = (int)(t / (uint32_t)a.tiles_y);
    ox0 = (int)tx * IC_TW;
  };
  auto fetch = [&](uint32_t t) {  // every global load of the tile is issued before any is used
    int b, oy0, ox0;
    tile_of(t, b, oy0, ox0);
    int tt = tid;
    asm volatile("" : "+v"(tt));
#pragma unroll
    for (int i = 0; i < PU; ++i) {
      int pr, pc, img, dst;
      unit_of(tt, i, pr, pc, img, dst);
      const int iy = min(max(2 * oy0 - 2 + pr, 0), H - 1), ix = min(max(2 * ox0 - 2 + pc, 0), W - 1);
      const aivc_image_src &s = a.src[img];
      raw[i][0] = raw[i][1] = raw[i][2] = 0u;
      // 32-bit element offsets off the (uniform) plane pointers: scalar base + vector offset addressing, no 64-bit
      // address arithmetic per sample (conv_images_supported bounds the planes to 2^31 elements / 2^32 bytes)
      if (s.y) {
        const uint32_t oy = ((uint32_t)b * (uint32_t)H + (uint32_t)iy) * (uint32_t)W + (uint32_t)ix;
        const uint32_t oc = ((uint32_t)b * (uint32_t)hc + (uint32_t)(iy >> 1)) * (uint32_t)wc + (uint32_t)(ix >> 1);
        raw[i][0] = s.y[oy];
        raw[i][1] = s.u[oc];
        raw[i][2] = s.v[oc];
      } else if (s.f) {
        const uint32_t of = (((uint32_t)b * (uint32_t)H + (uint32_t)iy) * (uint32_t)W + (uint32_t)ix) * (uint32_t)s.f_channels;
        raw[i][0] = __float_as_uint(s.f[of]);
        raw[i][1] = __float_as_uint(s.f[of + 1u]);
        raw[i][2] = __float_as_uint(s.f[of + 2u]);
      }
    }
  };
  if (t_first < t_end) fetch(t_first);

  const float *abase = patch + (2 * wave * 2 * NIMG * IC_HALF + p) * 4;
  const float *bbase = Wl + p * WS;
  const float *cquad = &chan[0][0] + 4 * hh;  // this lane half's channel quads: channel 32 j + 8 q + 4 hh + s <-> register 4 q + s

  for (uint32_t t = t_first; t < t_end; ++t) {
    int b, oy0, ox0;
    tile_of(t, b, oy0, ox0);
    __syncthreads();  // lut / weights / bias / beta (first tile); everybody is done reading the previous tile's patch
    {
      int tt = tid;
      asm volatile("" : "+v"(tt));
#pragma unroll
      for (int i = 0; i < PU; ++i) {
        int pr, pc, img, dst;
        unit_of(tt, i, pr, pc, img, dst);
        if (dst >= 0) {
          const bool bytes = a.src[img].y != nullptr;
          float4 v;
          v.x = bytes ? lut[raw[i][0] & 255u] : __uint_as_float(raw[i][0]);
          v.y = bytes ? lut[raw[i][1] & 255u] : __uint_as_float(raw[i][1]);
          v.z = bytes ? lut[raw[i][2] & 255u] : __uint_as_float(raw[i][2]);
          v.w = 0.0f;
          *reinterpret_cast<float4 *>(patch + dst) = v;
        }
      }
    }
    __syncthreads();
    if (t + 1 < t_end) fetch(t + 1);  // in flight during this tile's matrix work

    // ---- main loop: operands straight from LDS --------------------------------------------------
    floatx16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      // compile-time offsets of quads 2 o (lanes 0-31) and 2 o + 1 (lanes 32-63)
      constexpr auto offa = [](int q) {
        const int tap = q / NIMG, img = q % NIMG, ky = tap / 5, kx = tap % 5;
        return (((ky * 2 + (kx & 1)) * NIMG + img) * IC_HALF + (kx >> 1)) * 4;
      };
      const bool odd_tail = 2 * o + 1 >= NQ;  // the last octet of an odd quad count has no second half
      const int a0 = offa(2 * o), a1 = odd_tail ? a0 : offa(2 * o + 1);
      const int b0 = 8 * o, b1 = odd_tail ? b0 : 8 * o + 4;
      const float *ap = abase + a0 + hh * (a1 - a0);
      const float *bp = bbase + b0 + hh * (b1 - b0);
      float4 af = *reinterpret_cast<const float4 *>(ap);
      float4 bf0 = *reinterpret_cast<const float4 *>(bp);
      float4 bf1 = *reinterpret_cast<const float4 *>(bp + 32 * WS);
      if (odd_tail && hh) {  // zero operands for the missing quad (what a zero-padded K tile multiplies)
        af = make_float4(0.f, 0.f, 0.f, 0.f);
        bf0 = af;
        bf1 = af;
      }
      // (weights are the A operand: D[channel][pixel])
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf0.x, af.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf1.x, af.x, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf0.y, af.y, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf1.y, af.y, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf0.z, af.z, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf1.z, af.z, acc[1], 0, 0, 0);
      // .w: the zero fourth channel of the image -- fmaf(0, w, acc) = acc, not issued
    }

    // ---- bias, fused (inverse) GDN ------------------------------------------------------------
    if (a.bias) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b4 = *reinterpret_cast<const float4 *>(cquad + 32 * j + 8 * q);
          acc[j][4 * q + 0] = acc[j][4 * q + 0] + b4.x;
          acc[j][4 * q + 1] = acc[j][4 * q + 1] + b4.y;
          acc[j][4 * q + 2] = acc[j][4 * q + 2] + b4.z;
          acc[j][4 * q + 3] = acc[j][4 * q + 3] + b4.w;
        }
    }
    floatx16 acc2[2];
    if (gdn) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[j][r] = 0.0f;
      // s[i][pixel] = sum over k = 32 c + 8 o + (s | 4 + s) of gamma[i][k] * x[k][pixel]^2, K order of the contract: the B
      // operand of step (c, o, s) is the square of accumulator register 4 o + s of block c, as it stands
#pragma unroll
      for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const float4 g0 = greg[c][o][0], g1 = greg[c][o][1];
#pragma unroll
          for (int st = 0; st < 4; ++st) {
            float xv = acc[c][4 * o + st];
            asm volatile("" : "+v"(xv));  // keeps the squares inside the loop (hoisted, they cost 32 registers)
            const float sq = xv * xv;
            const float ga = st == 0 ? g0.x : (st == 1 ? g0.y : (st == 2 ? g0.z : g0.w));
            const float gb = st == 0 ? g1.x : (st == 1 ? g1.y : (st == 2 ? g1.z : g1.w));
            acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, sq, acc2[0], 0, 0, 0);
            acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(gb, sq, acc2[1], 0, 0, 0);
          }
        }
      }
    }

    // GDN: beta joins the normalisation sums here, and the wavefront learns whether all of its operands are ordinary
    // numbers (common.h: the lean square root / division then replace the full IEEE sequences, same bits)
    bool lean = false;
    if (gdn) {
      float mx = 0.0f, mn = GDN_SAFE_HI;
      const bool inv = a.gdn == 2;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 be = *reinterpret_cast<const float4 *>(cquad + IC_CO + 32 * j + 8 * q);
          acc2[j][4 * q + 0] = acc2[j][4 * q + 0] + be.x;
          acc2[j][4 * q + 1] = acc2[j][4 * q + 1] + be.y;
          acc2[j][4 * q + 2] = acc2[j][4 * q + 2] + be.z;
          acc2[j][4 * q + 3] = acc2[j][4 * q + 3] + be.w;
        }
      if (inv) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; r += 2) gdn_range_pair(mx, mn, acc2[j][r], acc2[j][r + 1]);
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) gdn_range(mx, mn, acc[j][r], acc2[j][r]);
      }
      lean = gdn_range_ok(mx, mn);
    }
    const int oy = oy0 + wave;
    if (oy < a.ho) {
      // wave-uniform row base (scalar registers) + one 32-bit lane offset: the 32 store addresses of a lane are
      // immediates off it (as 64-bit per-element pointers they were what the kernel spilled)
      const uint64_t yb64 = (uint64_t)(uintptr_t)(a.y + (((size_t)b * a.ho + oy) * a.wo + ox0) * IC_CO);
      const uint32_t yb_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(yb64 >> 32));  // (the builtin returns int:
      const uint32_t yb_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)yb64);          //  no sign extension)
      // (a pointer rebuilt from integers has no address space: name it, or every store is a flat_store behind a 64-bit add)
      typedef float gfloat4 __attribute__((ext_vector_type(4)));
      typedef __attribute__((address_space(1))) gfloat4 ggfloat4;
      typedef __attribute__((address_space(1))) char gchar;
      gchar *yrow = reinterpret_cast<gchar *>((uintptr_t)(((uint64_t)yb_hi << 32) | (uint64_t)yb_lo));
      // the lane's pixel is p; its accumulator quad q of block j is channels 32 j + 8 q + 4 hh .. + 3: one 16-byte store
      const uint32_t lane_off = (uint32_t)(p * IC_CO + 4 * hh) * 4u;
      const int cols = a.wo - ox0;  // output columns of this tile that exist (>= 1)
      auto emit = [&](auto MODE, auto WHOLE, auto LEAN) {
        constexpr int MD = decltype(MODE)::value;  // 0 / 1 / 2: no GDN + none / leaky / relu, 3: GDN, 4: inverse GDN
        constexpr bool WH = decltype(WHOLE)::value;
        constexpr bool LN = decltype(LEAN)::value;
        if (!WH && p >= cols) return;  // (per lane: a pixel beyond the image edge)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float o4[4];
#pragma unroll
            for (int st = 0; st < 4; ++st) {
              const int r = 4 * q + st;
              float v = acc[j][r];
              if constexpr (MD >= 3) {
                if constexpr (LN) {
                  const float nrm = sqrt_rn_safe(acc2[j][r]);
                  v = MD == 4 ? v * nrm : div_rn_safe(v, nrm);
                } else {
                  const float nrm = __builtin_sqrtf(acc2[j][r]);
                  v = MD == 4 ? v * nrm : v / nrm;
                }
              }
              if constexpr (MD == 1) v = v > 0.0f ? v : v * 0.01f;
              if constexpr (MD == 2) v = v > 0.0f ? v : 0.0f;
              o4[st] = v;
              // keep the elements apart: interleaved sqrt / division sequences of many of them cost registers (the
              // gamma fragments already take 64).  The lean ones go in pairs: the second element's instructions fill
              // the wait state behind v_rsq / v_rcp (an s_nop otherwise)
              if (!LN || (st & 1)) __builtin_amdgcn_sched_barrier(0);
            }
            *reinterpret_cast<ggfloat4 *>(yrow + lane_off + (uint32_t)((32 * j + 8 * q) * 4)) = (gfloat4){o4[0], o4[1], o4[2], o4[3]};
          }
        }
      };
      using std::integral_constant;
      const int mode = gdn ? (a.gdn == 2 ? 4 : 3) : (a.act1 == AIVC_ACT_LEAKY ? 1 : (a.act1 == AIVC_ACT_RELU ? 2 : 0));
      using no = integral_constant<bool, false>;
      using yes = integral_constant<bool, true>;
      auto emit_m = [&](auto WHOLE) {
        switch (mode) {
          case 0: emit(integral_constant<int, 0>{}, WHOLE, no{}); break;
          case 1: emit(integral_constant<int, 1>{}, WHOLE, no{}); break;
          case 2: emit(integral_constant<int, 2>{}, WHOLE, no{}); break;
          case 3:
            if (lean) emit(integral_constant<int, 3>{}, WHOLE, yes{});
            else emit(integral_constant<int, 3>{}, WHOLE, no{});
            break;
          default:
            if (lean) emit(integral_constant<int, 4>{}, WHOLE, yes{});
            else emit(integral_constant<int, 4>{}, WHOLE, no{});
            break;
        }
      };
      if (cols >= IC_TW) emit_m(integral_constant<bool, true>{});
      else emit_m(integral_constant<bool, false>{});
    }
  }
}

template <int NIMG>
static int launch_images(const ImgArgs &a, hipStream_t s) {
  const size_t lds = (size_t)ic_lds_floats<NIMG>() * sizeof(float);
  static LdsOptIn opt_in;  // > 64 KB of dynamic LDS needs the opt-in, on every device
  if (lds > 64 * 1024 && !opt_in.raise(reinterpret_cast<const void *>(conv_images_kernel<NIMG>), lds)) return AIVC_ERR_LAUNCH;
  const unsigned ntiles = (unsigned)a.n * (unsigned)a.tiles_y * (unsigned)a.tiles_x;
  const unsigned grid = (ntiles + IC_TPW - 1) / IC_TPW;
  hipLaunchKernelGGL(conv_images_kernel<NIMG>, dim3(grid), dim3(256), lds, s, a);
  return check_launch("conv_images");
}

bool conv_images_supported(const aivc_image_src *src, int n_img, const aivc_conv_params &p) {
  if (!src || n_img < 1 || n_img > AIVC_MAX_IMAGES) return false;
  if (p.mode != AIVC_MODE_CONV || p.ksize != 5 || p.stride != 2 || p.pad != 2 || p.c_out != IC_CO || p.c_in != 4 * n_img) return false;
  if (p.mul || p.res || p.act2 != AIVC_ACT_NONE || p.tail_c_out) return false;
  if (p.gdn ? p.act1 != AIVC_ACT_NONE : p.act1 == AIVC_ACT_SIGMOID) return false;
  if ((uint64_t)p.n * ((p.h_out + IC_TH - 1) / IC_TH) * ((p.w_out + IC_TW - 1) / IC_TW) >= 0x7FFFFFFFull) return false;
  // 32-bit sample offsets in the kernel: float sources up to 2^32 bytes, 8-bit planes far below
  for (int i = 0; i < n_img; ++i) {
    const uint64_t elems = (uint64_t)p.n * p.h_in * p.w_in * (src[i].y ? 1 : (src[i].f ? src[i].f_channels : 0));
    if (elems >= (src[i].y ? 0x7FFFFFFFull : 0x3FFFFFFFull)) return false;
  }
  return true;
}

int conv_images(const aivc_image_src *src, int n_img, const aivc_conv_params &p, hipStream_t s) {
  ImgArgs a;
  for (int i = 0; i < AIVC_MAX_IMAGES; ++i) a.src[i] = aivc_image_src{nullptr, nullptr, nullptr, nullptr, 0, 0};
  for (int i = 0; i < n_img; ++i) a.src[i] = src[i];
  a.w = p.w;
  a.bias = p.bias;
  a.gdn_beta = p.gdn_beta;
  a.gdn_gamma = p.gdn_gamma;
  a.y = p.y;
  a.n = p.n;
  a.h = p.h_in;
  a.w_in = p.w_in;
  a.ho = p.h_out;
  a.wo = p.w_out;
  a.gdn = p.gdn;
  a.act1 = p.act1;
  a.tiles_x = (p.w_out + IC_TW - 1) / IC_TW;
  a.tiles_y = (p.h_out + IC_TH - 1) / IC_TH;
  switch (n_img) {
    case 1: return launch_images<1>(a, s);
    case 2: return launch_images<2>(a, s);
    default: return launch_images<3>(a, s);
  }
}

}  // namespace aivc
