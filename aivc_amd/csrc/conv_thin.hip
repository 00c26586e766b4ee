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
//   - per output value the accumulation is the same fmaf chain as every other implementation: the taps of the
//     class in (ky, kx) order, 8 channels at a time in AIVC_K_ORDER (include/aivc_hip.h); taps outside the
//     image contribute fmaf(0, w, acc) = acc.
#include <stdlib.h>
#include <type_traits>

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
    for (int ci = 0; ci < Cin; ci += 8) {  // a group of 8 channels, accumulated in AIVC_K_ORDER (0,4,1,5,2,6,3,7)
#pragma unroll
      for (int c = c0; c < c0 + PAR; ++c) {
        const int pyc = c >> 1, pxc = c & 1;
        const int nx = class_ntaps<KS>(pxc);
        if (t < class_ntaps<KS>(pyc) * nx) {
          const int ky = class_tap<KS>(pyc, t / nx), kx = class_tap<KS>(pxc, t % nx);
          const int dy = (pyc + TPAD - ky) >> 1, dx = (pxc + TPAD - kx) >> 1;
          const float4 x0 = *reinterpret_cast<const float4 *>(center + (dy * PW + dx) * stride_px + ci);
          const float4 x1 = *reinterpret_cast<const float4 *>(center + (dy * PW + dx) * stride_px + ci + 4);
          const float *wt = p.w + (size_t)(ky * KS + kx) * Cin + ci;
#pragma unroll
          for (int o = 0; o < CO; ++o) {
            const float *wo = wt + (size_t)o * KS * KS * Cin;
            float a = acc[c][o];
            a = __builtin_fmaf(x0.x, wo[0], a);
            a = __builtin_fmaf(x1.x, wo[4], a);
            a = __builtin_fmaf(x0.y, wo[1], a);
            a = __builtin_fmaf(x1.y, wo[5], a);
            a = __builtin_fmaf(x0.z, wo[2], a);
            a = __builtin_fmaf(x1.z, wo[6], a);
            a = __builtin_fmaf(x0.w, wo[3], a);
            a = __builtin_fmaf(x1.w, wo[7], a);
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

// ---- matrix-core version -------------------------------------------------------------------------
// The 4 parity classes x c_out outputs of one INPUT pixel form the N dimension of a small GEMM
// (N = 12 or 24, padded to 16 / 32), K runs over the 3x3 input neighbourhood x c_in, and the class/tap
// structure lives in the B operand: B[(dy, dx, ci)][class, o] = w[o][ky][kx][ci] with
// ky = pyc + TPAD - 2 dy, kx = pxc + TPAD - 2 dx when that is a kernel tap of the class, else 0.
// Walking the neighbourhood with dy, dx DESCENDING makes ky, kx ascend for every class at once, so each
// output sees exactly the contract's fmaf chain (taps of its class in (ky, kx) order, channels in groups of 8
// in AIVC_K_ORDER) with exact no-ops (fmaf(x, 0, acc)) in between.  v_mfma_f32_16x16x4_f32 accumulates its 4
// k-slots in order (tools/mfma_probe.hip checks it against an fmaf chain), lane group g = lane / 16 supplies
// slot g, so inside every group of 16 channels LDS position 4 g + e holds channel thin_chan(g, e): one
// ds_read_b128 feeds 4 consecutive MFMA steps.
// 52 % (c_out 3) / 39 % (c_out 6) of the MFMA work is useful, still 2-3x the VALU kernel above.
typedef float floatx4 __attribute__((ext_vector_type(4)));

// Channel (inside a group of 16) that k-slot g = lane / 16 multiplies at MFMA step e.  The contract order of
// include/aivc_hip.h inside 16 channels is 0 4 1 5 | 2 6 3 7 | 8 12 9 13 | 10 14 11 15 (two groups of 8 in
// AIVC_K_ORDER); v_mfma_f32_16x16x4_f32 accumulates its 4 slots in order, so step e takes the e-th quadruple
// and slot g its g-th entry.
__device__ __host__ constexpr int thin_chan(int g, int e) { return 8 * (e >> 1) + AIVC_K_ORDER(4 * (e & 1) + g); }

// Persistent workgroups (one per CU, 8 wavefronts): the B operand of a wavefront -- its 16 output columns
// over all K = ND^2 * c_in -- is gathered from the weight tensor ONCE into registers (144 VGPRs at
// c_in = 64), then the group walks over 4 x 32-pixel input tiles whose patches (tile + 1-pixel halo) are
// double-buffered in LDS: the global loads of the next patch are in flight during the MFMAs of the
// current one.  Wave roles: c_out 3 -> (row, half row), one 16-pixel group each; c_out 6 -> (row, N block),
// two 16-pixel groups each.
template <int KS, int CO, int G16>
__global__ __launch_bounds__(512) void thin_mfma_kernel(aivc_conv_params p, int tiles_x, int tiles_y, int ntiles) {
  constexpr int TH = 4, TW = 32, PW = TW + 2, PH = TH + 2;
  constexpr int NB = (4 * CO + 15) / 16;
  constexpr int NMG = NB;  // 16-pixel groups per wavefront
  constexpr int TPAD = (KS + 1) / 2 - 1;
  constexpr int LO = -((KS - 1 - TPAD) / 2), HI = (1 + TPAD) / 2;  // neighbourhood offsets carrying a tap
  constexpr int ND = HI - LO + 1;
  constexpr int NSTEP = ND * ND * G16;
  constexpr int Cin = 16 * G16, SPX = Cin + 4, PATCH = PH * PW * SPX;
  constexpr int UNITS = PH * PW * G16, UPT = (UNITS + 511) / 512;
  extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 patches [PH * PW][Cin + 4]
  const int H = p.h_in, W = p.w_in;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, g = lane >> 4;
  const int row = wave >> 1;
  const int nb = NB == 2 ? (wave & 1) : 0, mg0 = NB == 2 ? 0 : (wave & 1);

  // ---- this lane's output column and its B operand ------------------------------------------------
  const int nc = nb * 16 + col;
  const bool colok = nc < 4 * CO;
  const int cls = colok ? nc / CO : 0, o = colok ? nc % CO : 0;
  const int pyc = cls >> 1, pxc = cls & 1;
  float breg[NSTEP][4];
#pragma unroll
  for (int step = 0; step < NSTEP; ++step) {
    const int t = step / G16, j16 = step % G16;
    const int dy = HI - t / ND, dx = HI - t % ND;
    const int ky = pyc + TPAD - 2 * dy, kx = pxc + TPAD - 2 * dx;
    const bool ok = colok && ky >= 0 && ky < KS && kx >= 0 && kx < KS;
    const float *src = p.w + (size_t)((o * KS + (ok ? ky : 0)) * KS + (ok ? kx : 0)) * Cin + j16 * 16;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = src[thin_chan(g, e)];
      breg[step][e] = ok ? v : 0.0f;
    }
  }

  // ---- patch staging: unit = 16 channels of one patch pixel ------------------------------------------
  float4 sreg[UPT][4];
  auto tile_origin = [&](int tile, int &n, int &ty0, int &tx0) {
    const int per = tiles_x * tiles_y;
    n = tile / per;
    const int r = tile - n * per;
    ty0 = (r / tiles_x) * TH;
    tx0 = (r % tiles_x) * TW;
  };
  auto stage_load = [&](int tile) {
    int n, ty0, tx0;
    tile_origin(tile, n, ty0, tx0);
    const float *xn = p.x + (size_t)n * H * W * Cin;
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      const int i = tid + 512 * u;
      const int j16 = i % G16, pp = i / G16;
      const int iy = ty0 - 1 + pp / PW, ix = tx0 - 1 + pp % PW;
      const bool ok = i < UNITS && iy >= 0 && iy < H && ix >= 0 && ix < W;
      const float *src = xn + ((size_t)(ok ? iy : 0) * W + (ok ? ix : 0)) * Cin + j16 * 16;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float4 v = *reinterpret_cast<const float4 *>(src + 4 * e);
        sreg[u][e] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto stage_store = [&](float *buf) {
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      const int i = tid + 512 * u;
      if (i < UNITS) {
        float *dst = buf + (i / G16) * SPX + (i % G16) * 16;  // position 4 g + e <- channel thin_chan(g, e)
        *reinterpret_cast<float4 *>(dst) = make_float4(sreg[u][0].x, sreg[u][0].z, sreg[u][2].x, sreg[u][2].z);       // 0 2 8 10
        *reinterpret_cast<float4 *>(dst + 4) = make_float4(sreg[u][1].x, sreg[u][1].z, sreg[u][3].x, sreg[u][3].z);   // 4 6 12 14
        *reinterpret_cast<float4 *>(dst + 8) = make_float4(sreg[u][0].y, sreg[u][0].w, sreg[u][2].y, sreg[u][2].w);   // 1 3 9 11
        *reinterpret_cast<float4 *>(dst + 12) = make_float4(sreg[u][1].y, sreg[u][1].w, sreg[u][3].y, sreg[u][3].w);  // 5 7 13 15
      }
    }
  };

  const bool has_bias = p.bias != nullptr;
  const float bias_o = (has_bias && colok) ? p.bias[o] : 0.0f;
  const int act1 = p.act1, act2 = p.act2;
  // A fragments: corner (LO, LO) of the neighbourhood of this lane's pixel; steps add constant offsets
  const int a_off = ((row + 1 + LO) * PW + 1 + LO + mg0 * 16 + col) * SPX + 4 * g;

  // ---- epilogue of one tile: lane (col, g) holds rows 4 g + r of its 16-pixel groups ------------------------
  // (the variant without gate / residual operands has no load in it: with them in the same code the compiler drains
  // the memory counter -- the tile's own stores included -- after every element)
  auto emit_v = [&](auto EXTRA, int etile, const floatx4 (&eacc)[NMG]) {
    constexpr bool extra = decltype(EXTRA)::value;
    int n, ty0, tx0;
    tile_origin(etile, n, ty0, tx0);
    const int qy = ty0 + row;
    if (colok && qy < H) {
#pragma unroll
      for (int mg = 0; mg < NMG; ++mg)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qx = tx0 + (mg0 + mg) * 16 + 4 * g + r;
          if (qx < W) {
            const size_t off = (((size_t)n * p.h_out + (2 * qy + pyc)) * p.w_out + (2 * qx + pxc)) * CO + o;
            float v = eacc[mg][r];
            if (has_bias) v = v + bias_o;
            v = v > 0.0f ? v : (act1 == AIVC_ACT_LEAKY ? v * 0.01f : (act1 == AIVC_ACT_RELU ? 0.0f : v));
            if constexpr (extra) {
              if (p.mul) v = p.mul[off] * v;
              if (p.res) v = v + p.res[off];
            }
            v = v > 0.0f ? v : (act2 == AIVC_ACT_LEAKY ? v * 0.01f : (act2 == AIVC_ACT_RELU ? 0.0f : v));
            p.y[off] = v;
          }
        }
    }
  };
  const bool has_extra = p.mul != nullptr || p.res != nullptr;
  auto emit = [&](int etile, const floatx4 (&eacc)[NMG]) {
    if (has_extra) emit_v(std::true_type{}, etile, eacc);
    else emit_v(std::false_type{}, etile, eacc);
  };

  // Software pipeline over the tiles of this workgroup (round 3): the reduction of a tile is one dependent chain of
  // 16x16x4 MFMAs per 16-pixel group -- 32 cycles each with nothing else to issue -- so the epilogue of the PREVIOUS
  // tile (its accumulators are 4 registers per group) and the LDS stores of the NEXT tile's patch sit inside that
  // chain instead of between two chains, where all 8 wavefronts of the CU left the matrix pipe idle together.
  //   patch(next) -> registers (global loads in flight for a whole tile) -> LDS buffer cur ^ 1 at step STORE_AT of
  //   this tile's chain (its last readers passed the barrier at the end of the previous tile); one barrier per tile.
  // (EMIT_AT 6 and STORE_AT at 1/3 or 1/2 of the chain measure the same; the epilogue in the middle or at the end of
  // the chain -- EMIT_AT 12 / 20 of 36 -- runs TWICE as long)
  constexpr int EMIT_AT = NSTEP > 8 ? 2 : 0, STORE_AT = NSTEP > 8 ? NSTEP * 2 / 3 : NSTEP - 1;
  int tile = blockIdx.x, cur = 0;
  if (tile < ntiles) {
    stage_load(tile);
    stage_store(smem);
  }
  __syncthreads();
  if (tile + (int)gridDim.x < ntiles) stage_load(tile + gridDim.x);
  floatx4 pacc[NMG];
#pragma unroll
  for (int mg = 0; mg < NMG; ++mg) pacc[mg] = (floatx4){0.f, 0.f, 0.f, 0.f};
  int ptile = -1;
  for (; tile < ntiles; tile += gridDim.x) {
    const int next = tile + gridDim.x;
    const float *ap = smem + cur * PATCH + a_off;
    floatx4 acc[NMG];
#pragma unroll
    for (int mg = 0; mg < NMG; ++mg) acc[mg] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      const int t = step / G16, j16 = step % G16;
      const int dy = HI - t / ND, dx = HI - t % ND;
      float4 af[NMG];
#pragma unroll
      for (int mg = 0; mg < NMG; ++mg)
        af[mg] = *reinterpret_cast<const float4 *>(ap + ((dy - LO) * PW + (dx - LO) + mg * 16) * SPX + j16 * 16);
      if (step == EMIT_AT && ptile >= 0) emit(ptile, pacc);
      if (step == STORE_AT && next < ntiles) {
        stage_store(smem + (cur ^ 1) * PATCH);
        if (next + (int)gridDim.x < ntiles) stage_load(next + gridDim.x);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mg = 0; mg < NMG; ++mg) {
          const float av = e == 0 ? af[mg].x : (e == 1 ? af[mg].y : (e == 2 ? af[mg].z : af[mg].w));
          acc[mg] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, breg[step][e], acc[mg], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mg = 0; mg < NMG; ++mg) pacc[mg] = acc[mg];
    ptile = tile;
    __syncthreads();
    cur ^= 1;
  }
  if (ptile >= 0) emit(ptile, pacc);
}

static bool thin_mfma_ok(const aivc_conv_params &p) {
  if (p.c_in != 16 && p.c_in != 32 && p.c_in != 64) return false;  // instantiated widths (B operand in registers)
  return p.act1 != AIVC_ACT_SIGMOID && p.act2 != AIVC_ACT_SIGMOID;
}

template <int KS, int CO, int G16>
static int launch_thin_mfma_g(const aivc_conv_params &p, hipStream_t s) {
  const size_t lds = (size_t)2 * 6 * 34 * (16 * G16 + 4) * sizeof(float);
  const int tiles_x = (p.w_in + 31) / 32, tiles_y = (p.h_in + 3) / 4;
  const int ntiles = tiles_x * tiles_y * p.n;
  static LdsOptIn opt_in;  // per device
  static std::atomic<int> n_cu{0};
  if (!opt_in.raise(reinterpret_cast<const void *>(thin_mfma_kernel<KS, CO, G16>), 160 * 1024)) return check_launch("thin_mfma lds attribute");
  if (n_cu.load(std::memory_order_relaxed) == 0) {
    int dev = 0, cus = 0;
    n_cu = hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0 ? cus : 256;
  }
  const int cus = n_cu.load(std::memory_order_relaxed);
  // More workgroups than CUs (one fits per CU): a group that shares its CU with a range-coder wave of the entropy side
  // streams runs slower, and with ONE group per CU walking a fixed share of the tiles the slowest CU set the launch
  // time; with several rounds of groups the hardware dispatcher evens it out (the B operand is re-gathered per group:
  // 25 x 64 x c_out floats out of L2).  AIVC_THIN_GRID_MULT: tuning aid.
  static const int mult = getenv("AIVC_THIN_GRID_MULT") ? atoi(getenv("AIVC_THIN_GRID_MULT")) : 1;
  const int want = cus * (mult > 0 ? mult : 1);
  const int grid = ntiles < want ? ntiles : want;
  hipLaunchKernelGGL((thin_mfma_kernel<KS, CO, G16>), dim3(grid), dim3(512), lds, s, p, tiles_x, tiles_y, ntiles);
  return check_launch("thin_mfma");
}

template <int KS, int CO>
static int launch_thin_mfma(const aivc_conv_params &p, hipStream_t s) {
  switch (p.c_in) {
    case 16: return launch_thin_mfma_g<KS, CO, 1>(p, s);
    case 32: return launch_thin_mfma_g<KS, CO, 2>(p, s);
    default: return launch_thin_mfma_g<KS, CO, 4>(p, s);
  }
}

bool conv2d_thin_supported(const aivc_conv_params &p) {
  if (p.mode != AIVC_MODE_TCONV || p.gdn) return false;
  if (p.c_out != 3 && p.c_out != 6) return false;
  if (p.ksize != 3 && p.ksize != 5) return false;
  return p.c_in % 8 == 0 && p.c_in >= 16 && p.c_in <= 128;
}

template <int KS, int CO>
static int launch_thin(const aivc_conv_params &p, hipStream_t s) {
  constexpr int TH = 8;  // 8 x 16 input pixels per workgroup: <= 49 KB of LDS at 64 channels (3 groups per CU)
  const size_t lds = (size_t)(TH + 2) * 18 * (p.c_in + 4) * sizeof(float);
  const int tiles = ((p.w_in + 15) / 16) * ((p.h_in + TH - 1) / TH);
  static LdsOptIn opt_in;  // per device
  if (!opt_in.raise(reinterpret_cast<const void *>(thin_tconv_kernel<KS, CO, TH>), 160 * 1024)) return check_launch("thin_tconv lds attribute");
  hipLaunchKernelGGL((thin_tconv_kernel<KS, CO, TH>), dim3(tiles, p.n), dim3(TH * 16), lds, s, p);
  return check_launch("thin_tconv");
}

int conv2d_thin_variant(const aivc_conv_params &p) { return thin_mfma_ok(p) && !getenv("AIVC_THIN_VALU") ? 2 : 1; }

int conv2d_thin(const aivc_conv_params &p, hipStream_t s) {
  if (!conv2d_thin_supported(p)) return AIVC_ERR_UNSUPPORTED;
  if (thin_mfma_ok(p) && !getenv("AIVC_THIN_VALU")) {
    if (p.ksize == 5) return p.c_out == 3 ? launch_thin_mfma<5, 3>(p, s) : launch_thin_mfma<5, 6>(p, s);
    return p.c_out == 3 ? launch_thin_mfma<3, 3>(p, s) : launch_thin_mfma<3, 6>(p, s);
  }
  if (p.ksize == 5) return p.c_out == 3 ? launch_thin<5, 3>(p, s) : launch_thin<5, 6>(p, s);
  return p.c_out == 3 ? launch_thin<3, 3>(p, s) : launch_thin<3, 6>(p, s);
}

}  // namespace aivc
