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
//
// LDS image of a patch (round 5): FOUR PLANES, one per k-slot g = lane / 16, each [patch pixel][c_in / 16 chunks of 4
// floats + pad]: lane (col, g) reads the 16-byte chunk of pixel col in plane g.  ds_read_b128 is served in the lane
// groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS): each holds the 16 columns once,
// split over two values of g -- so the 16-byte slot of an address must depend on col only.  Pixel stride = an ODD
// number of slots (col -> slot is a bijection mod 16), plane stride = a multiple of 16 slots.  (The round-3 image
// [pixel][c_in + 4] with slot = col + g put (col 12, g 0) and (col 11, g 1) on one slot in every group: half of the
// LDS cycles were conflict cycles, profiles/r05_pmc_thin.json.)
//
// Per-tile bookkeeping is what the matrix pipe waits for here (144 MFMAs of 32 cycles per wave and tile against
// ~350 other instructions in round 3: three integer divisions per tile origin, per-unit staging addresses, 64-bit output
// offsets): tile origins advance incrementally (no division after the prologue), staging offsets and validity are
// per-thread constants + two adds, outputs go through one uniform 64-bit base per tile + a 32-bit per-lane offset
// with the 4 rows of a lane as immediate offsets.
struct ThinOrigin {
  int n, ry, rx;  // image, tile row, tile column
};

// LEAN: the layer as the codec launches it -- bias, one activation, no gate / residual, no second activation -- without the
// run-time flags of the general epilogue (the compiler turns them into a thicket of uniform branches around every store)
template <int KS, int CO, int G16, bool LEAN>
__global__ __launch_bounds__(512) void thin_mfma_kernel(aivc_conv_params p, int tiles_x, int tiles_y, int ntiles) {
  constexpr int TH = 4, TW = 32, PW = TW + 2, PH = TH + 2;
  constexpr int NB = (4 * CO + 15) / 16;
  constexpr int NMG = NB;  // 16-pixel groups per wavefront
  constexpr int TPAD = (KS + 1) / 2 - 1;
  constexpr int LO = -((KS - 1 - TPAD) / 2), HI = (1 + TPAD) / 2;  // neighbourhood offsets carrying a tap
  constexpr int ND = HI - LO + 1;
  constexpr int NSTEP = ND * ND * G16;
  constexpr int Cin = 16 * G16;
  constexpr int PS = 4 * (G16 | 1);                          // floats per patch pixel in a plane: an odd number of 16-byte slots
  constexpr int PLANE = (PH * PW * PS + 63) / 64 * 64;       // floats per plane: a multiple of 16 slots
  constexpr int PATCH = 4 * PLANE;
  constexpr int UNITS = PH * PW * G16, UPT = (UNITS + 511) / 512;
  extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 patches [4 planes][PH * PW][PS]
  const int H = p.h_in, W = p.w_in;
  const int tid = threadIdx.x, lane = tid & 63;
  // the wave id IS uniform, but anything derived from threadIdx is divergent to the compiler (64-bit per-lane output
  // addresses, exec-masked branches around uniform tests): say so
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, g = lane >> 4;
  const int row = wave >> 1;
  // c_out 6 (two N blocks): block nb holds the two classes with pxc = nb (12 of its 16 columns) -- the neighbours of the column
  // dx = LO carry no tap of the pxc = 1 classes, so the waves of block 1 skip those steps (6 of 9 neighbours instead of 9:
  // 15 instead of 18 MFMA steps per pixel group and channel quadruple over the pair); waves w and w + 4 share a SIMD: one of each
  const int nb = NB == 2 ? ((wave ^ (wave >> 2)) & 1) : 0, mg0 = NB == 2 ? 0 : (wave & 1);

  // ---- this lane's output column and its B operand ------------------------------------------------
  const bool colok = NB == 2 ? col < 2 * CO : col < 4 * CO;
  const int cls = !colok ? 0 : (NB == 2 ? 2 * (col / CO) + nb : col / CO), o = colok ? col % CO : 0;
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

  // ---- tile origins without divisions: the tiles of this group are blockIdx.x, + gridDim.x, ... ------------------
  const int per = tiles_x * tiles_y;
  const int G = (int)gridDim.x;
  const int g_n = G / per, g_ry = (G % per) / tiles_x, g_rx = (G % per) % tiles_x;  // (three divisions, once)
  auto origin_of = [&](int tile) {
    ThinOrigin t;
    t.n = tile / per;
    const int r = tile - t.n * per;
    t.ry = r / tiles_x;
    t.rx = r - t.ry * tiles_x;
    return t;
  };
  auto advance = [&](ThinOrigin t) {
    t.rx += g_rx;
    if (t.rx >= tiles_x) { t.rx -= tiles_x; t.ry += 1; }
    t.ry += g_ry;
    if (t.ry >= tiles_y) { t.ry -= tiles_y; t.n += 1; }
    t.n += g_n;
    return t;
  };

  // ---- patch staging: unit = 16 channels of one patch pixel; everything but the tile origin is a per-thread constant
  float4 sreg[UPT][4];
  int u_py[UPT], u_px[UPT], u_src[UPT], u_dst[UPT];
#pragma unroll
  for (int u = 0; u < UPT; ++u) {
    const int i = tid + 512 * u;
    const int j16 = i % G16, pp = i / G16;
    u_py[u] = i < UNITS ? pp / PW : -(1 << 20);  // (a unit beyond the patch is never valid)
    u_px[u] = pp % PW;
    u_src[u] = ((pp / PW) * W + pp % PW) * Cin + j16 * 16;
    u_dst[u] = pp * PS + j16 * 4;
  }
  // Loads go through a buffer descriptor over ONE image (base = image n, num_records = its bytes: a 64-frame batch of
  // 540x960x64 floats is 8.5 GB, beyond a 32-bit offset): a unit outside the image gets an offset beyond num_records
  // and the hardware returns zeros -- no select per loaded float (32 v_cndmask per tile before), no clamped address.
  const unsigned img_bytes = (unsigned)H * (unsigned)W * Cin * 4u;
  auto stage_load = [&](const ThinOrigin &t) {
    const int ty0 = t.ry * TH - 1, tx0 = t.rx * TW - 1;
    const float *img = p.x + (long)t.n * H * W * Cin;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(img), 0, (int)img_bytes, 0x00020000);
    const int tile_off = (ty0 * W + tx0) * Cin * 4;  // bytes; negative on the first tile row / column (valid units add up to >= 0)
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      const bool ok = (unsigned)(ty0 + u_py[u]) < (unsigned)H && (unsigned)(tx0 + u_px[u]) < (unsigned)W;
      const int voff = ok ? tile_off + u_src[u] * 4 : -64;  // 0xFFFFFFC0: out of range -> zeros
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // (whole-vector cast: __builtin_bit_cast of the single elements makes this compiler load ONE dword)
        typedef float f32x4 __attribute__((__vector_size__(16)));
        const f32x4 v = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 16 * e, 0, 0);
        sreg[u][e] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  };
  auto stage_store = [&](float *buf) {
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      if (tid + 512 * u < UNITS) {
        float *dst = buf + u_dst[u];  // plane g, position e <- channel thin_chan(g, e)
        *reinterpret_cast<float4 *>(dst) = make_float4(sreg[u][0].x, sreg[u][0].z, sreg[u][2].x, sreg[u][2].z);              // 0 2 8 10
        *reinterpret_cast<float4 *>(dst + PLANE) = make_float4(sreg[u][1].x, sreg[u][1].z, sreg[u][3].x, sreg[u][3].z);      // 4 6 12 14
        *reinterpret_cast<float4 *>(dst + 2 * PLANE) = make_float4(sreg[u][0].y, sreg[u][0].w, sreg[u][2].y, sreg[u][2].w);  // 1 3 9 11
        *reinterpret_cast<float4 *>(dst + 3 * PLANE) = make_float4(sreg[u][1].y, sreg[u][1].w, sreg[u][3].y, sreg[u][3].w);  // 5 7 13 15
      }
    }
  };

  const bool has_bias = p.bias != nullptr;
  const float bias_o = (has_bias && colok) ? p.bias[o] : 0.0f;
  const int act1 = p.act1, act2 = p.act2;
  // A fragments: corner (LO, LO) of the neighbourhood of this lane's pixel in plane g; steps add constant offsets
  const int a_off = g * PLANE + ((row + 1 + LO) * PW + 1 + LO + mg0 * 16 + col) * PS;

  // ---- epilogue of one tile: lane (col, g) holds rows 4 g + r of its 16-pixel groups ------------------------
  // (the variant without gate / residual operands has no load in it: with them in the same code the compiler drains
  // the memory counter -- the tile's own stores included -- after every element)
  const int w_out = p.w_out, h_out = p.h_out;
  // per-lane part of an output offset: class position inside the 2x2 block, output channel, first of the lane's 4 pixels
  const int lane_out = ((pyc * w_out + pxc) + 2 * (mg0 * 16 + 4 * g)) * CO + o;
  // activations as selects on uniform constants, not branches: x > 0 ? x : (x * slope) & keep, slope 0.01 (leaky) / 1 (none:
  // x * 1 = x exactly, -0 included), keep = 0 for relu (+0.0, as the scalar kernels give) else all ones
  const float slope1 = act1 == AIVC_ACT_LEAKY ? 0.01f : 1.0f, slope2 = act2 == AIVC_ACT_LEAKY ? 0.01f : 1.0f;
  const unsigned keep1 = act1 == AIVC_ACT_RELU ? 0u : 0xFFFFFFFFu, keep2 = act2 == AIVC_ACT_RELU ? 0u : 0xFFFFFFFFu;
  auto act = [](float v, float slope, unsigned keep) {
    const float ng = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v * slope) & keep);
    return v > 0.0f ? v : ng;
  };
  auto emit_v = [&](auto EXTRA, auto FULL, const ThinOrigin &t, const floatx4 (&eacc)[NMG]) {
    constexpr bool extra = decltype(EXTRA)::value, full = decltype(FULL)::value;
    const int qy = t.ry * TH + row, tx0 = t.rx * TW;
    if (qy < H) {  // (uniform)
      // uniform: output pixel (2 qy, 2 tx0) of image n; per lane a 32-bit offset, the lane's 4 pixels as immediates
      const long base = (((long)t.n * h_out + 2 * qy) * w_out + 2 * tx0) * CO;
      float *yb = p.y + base;
      const int qx0 = tx0 + mg0 * 16 + 4 * g;
#pragma unroll
      for (int mg = 0; mg < NMG; ++mg)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (colok && (full || qx0 + mg * 16 + r < W)) {
            const int off = lane_out + (2 * (mg * 16 + r)) * CO;
            float v = eacc[mg][r];
            if constexpr (LEAN) {
              v = act(v + bias_o, slope1, keep1);
            } else {
              const float vb = v + bias_o;
              v = has_bias ? vb : v;
              v = act(v, slope1, keep1);
              if constexpr (extra) {
                if (p.mul) v = p.mul[base + off] * v;
                if (p.res) v = v + p.res[base + off];
              }
              v = act(v, slope2, keep2);
            }
            yb[off] = v;
          }
        }
    }
  };
  const bool has_extra = p.mul != nullptr || p.res != nullptr;
  auto emit = [&](const ThinOrigin &t, const floatx4 (&eacc)[NMG]) {
    const bool full = t.rx * TW + TW <= W;  // (uniform) every pixel column of the tile is inside the image
    if constexpr (LEAN) {
      if (full) emit_v(std::false_type{}, std::true_type{}, t, eacc);
      else emit_v(std::false_type{}, std::false_type{}, t, eacc);
    } else if (has_extra) {
      if (full) emit_v(std::true_type{}, std::true_type{}, t, eacc);
      else emit_v(std::true_type{}, std::false_type{}, t, eacc);
    } else {
      if (full) emit_v(std::false_type{}, std::true_type{}, t, eacc);
      else emit_v(std::false_type{}, std::false_type{}, t, eacc);
    }
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
  // origins of the tile in the matrix pipe, of the one whose patch is in registers / on its way, and of the last one
  ThinOrigin t_cur = origin_of(tile < ntiles ? tile : 0), t_load = t_cur, t_prev = t_cur;
  if (tile < ntiles) {
    stage_load(t_cur);
    stage_store(smem);
  }
  __syncthreads();
  t_load = advance(t_cur);
  if (tile + G < ntiles) stage_load(t_load);
  floatx4 pacc[NMG];
#pragma unroll
  for (int mg = 0; mg < NMG; ++mg) pacc[mg] = (floatx4){0.f, 0.f, 0.f, 0.f};
  bool have_prev = false;
  for (; tile < ntiles; tile += G) {
    const int next = tile + G;
    const float *ap = smem + cur * PATCH + a_off;
    floatx4 acc[NMG];
#pragma unroll
    for (int mg = 0; mg < NMG; ++mg) acc[mg] = (floatx4){0.f, 0.f, 0.f, 0.f};
    const bool skip_lo = NB == 2 && nb == 1;  // (wave-uniform) this wave's columns have no tap at dx = LO
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      const int t = step / G16, j16 = step % G16;
      const int dy = HI - t / ND, dx = HI - t % ND;
      if (step == EMIT_AT && have_prev) emit(t_prev, pacc);
      if (step == STORE_AT && next < ntiles) {
        stage_store(smem + (cur ^ 1) * PATCH);  // the patch of `next` (origin t_load)
        if (next + G < ntiles) stage_load(advance(t_load));
      }
      // kx of the pxc = 1 classes at dx = LO lies beyond the kernel: exact no-ops, not issued (one uniform branch per neighbour column)
      if (NB == 2 && 1 + TPAD - 2 * dx >= KS && skip_lo) continue;
      float4 af[NMG];
#pragma unroll
      for (int mg = 0; mg < NMG; ++mg)
        af[mg] = *reinterpret_cast<const float4 *>(ap + ((dy - LO) * PW + (dx - LO) + mg * 16) * PS + j16 * 4);
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
    t_prev = t_cur;
    t_cur = t_load;
    t_load = advance(t_load);
    have_prev = true;
    __syncthreads();
    cur ^= 1;
  }
  if (have_prev) emit(t_prev, pacc);
}

static bool thin_mfma_ok(const aivc_conv_params &p) {
  if (p.c_in != 16 && p.c_in != 32 && p.c_in != 64) return false;  // instantiated widths (B operand in registers)
  // an image is addressed through a buffer descriptor with signed 32-bit byte offsets: larger ones go to the VALU kernel
  if ((uint64_t)p.h_in * (uint64_t)p.w_in * (uint64_t)p.c_in * 4u >= 0x7FFFFFC0ull) return false;
  return p.act1 != AIVC_ACT_SIGMOID && p.act2 != AIVC_ACT_SIGMOID;
}

template <int KS, int CO, int G16, bool LEAN>
static int launch_thin_mfma_l(const aivc_conv_params &p, hipStream_t s) {
  const size_t lds = (size_t)2 * 4 * ((6 * 34 * 4 * (G16 | 1) + 63) / 64 * 64) * sizeof(float);  // 2 patches x 4 planes
  const int tiles_x = (p.w_in + 31) / 32, tiles_y = (p.h_in + 3) / 4;
  const int ntiles = tiles_x * tiles_y * p.n;
  static LdsOptIn opt_in;  // per device
  static std::atomic<int> n_cu{0};
  if (!opt_in.raise(reinterpret_cast<const void *>(thin_mfma_kernel<KS, CO, G16, LEAN>), 160 * 1024)) return check_launch("thin_mfma lds attribute");
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
  int want = cus * (mult > 0 ? mult : 1);
  // AIVC_THIN_GRID_MAX (test aid): few persistent groups on a small input, so that the tile walk (several tiles per
  // group, origins advanced across rows and images) runs at sizes the CPU oracle checks in seconds
  if (const char *gm = getenv("AIVC_THIN_GRID_MAX")) want = atoi(gm) > 0 && atoi(gm) < want ? atoi(gm) : want;
  const int grid = ntiles < want ? ntiles : want;
  hipLaunchKernelGGL((thin_mfma_kernel<KS, CO, G16, LEAN>), dim3(grid), dim3(512), lds, s, p, tiles_x, tiles_y, ntiles);
  return check_launch("thin_mfma");
}

template <int KS, int CO, int G16>
static int launch_thin_mfma_g(const aivc_conv_params &p, hipStream_t s) {
  const bool lean = p.bias != nullptr && p.mul == nullptr && p.res == nullptr && p.act2 == AIVC_ACT_NONE;
  return lean ? launch_thin_mfma_l<KS, CO, G16, true>(p, s) : launch_thin_mfma_l<KS, CO, G16, false>(p, s);
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
