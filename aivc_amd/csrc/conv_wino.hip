// conv_wino.hip -- version 2 of the fp32 arithmetic contract (AIVC_PREC_FP32_WINO, include/aivc_hip.h): the stride-1 3x3
// convolutions with c_in % 32 == 0 and c_out % 64 == 0 as Winograd F(2x2, 3x3) on the gfx950 matrix cores.
//
//   why            the fp32 matrix pipe is the scarce unit of this part (157 TFLOP/s, 1/16 of the bf16 rate) and the codec's
//                  step is 0.75 of it end to end: what is left under the tap chain is ~10 %.  F(2x2, 3x3) issues 16
//                  multiplications per 2x2 output pixels and channel pair instead of 36.
//   GEMM view      M = output TILES (n, ty, tx) of 2x2 pixels, N = output channels, K = (position p = 0..15, ci): sixteen
//                  GEMMs of reduction length c_in that share one accumulator tile in turn -- after the last K-tile of a
//                  position the accumulators M_p are folded into the four output accumulators (a, b) with coefficients
//                  0 / +-1 and cleared.  Each M_p is the fixed-order fmaf chain of v_mfma_f32_32x32x2_f32 over ci in
//                  AIVC_K_ORDER; the fold adds the positions in ascending order: one fixed chain per output, the CPU oracle
//                  (oracle/aivc_oracle.c) walks the same one.
//   A operand      V_p = the input transform of the 4x4 patch, never in memory: the four input pixels a position combines
//                  (rows A[i] / B[i], columns A[j] / B[j] of the patch, replicate-clamped) go global -> LDS by LDS-DMA into
//                  four raw planes of [64 tiles][32 channels]; the workgroup turns them into the A tile in place of a
//                  ds_write pass of a register-staged loader: 4 ds_read_b128 + 12 v_fma + 1 ds_write_b128 per float4 of V.
//   B operand      U = G g G^T, transformed once per layer (aivc_winograd_weights, fp64, rounded once), [c_out][16][c_in]: a
//                  K-contiguous row per output channel like the OHWI weights of the tap kernels, fetched by the same
//                  LDS-DMA into a two-stage ring.
//   LDS image      rows of 128 bytes (32 channels), XOR-swizzled 16-byte slots exactly as in conv_mfma.hip (slot s of row R
//                  holds data chunk s ^ (R & 7) ^ ((R >> 3) & 3)); raw planes and A tile share the layout, so the transform
//                  is slot-wise.  32 KB raw + 8 KB A + 2 x 16 KB B = 72 KB: two workgroups per CU.
//   schedule       per K-tile: [DMA of this tile landed] barrier, transform raw -> A, barrier, issue the DMAs of the next
//                  tile (raw planes are free, the other B stage was read a tile ago), 32 MFMAs per wave.
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"

namespace aivc {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct WinoArgs {
  aivc_conv_params p;
  int M;                // output tiles: n * TH * TW
  int TH, TW;           // tiles per image column / row
  uint32_t tw_magic, th_magic, img_magic;  // floor(2^32 / TW), floor(2^32 / TH), floor(2^32 / (TH * TW)): quotients low by at most one
  int gy;               // c_out tiles
};

__device__ __forceinline__ void wino_glds16(const float *base, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds_dst) : "memory", "m0");
}

__device__ __forceinline__ uint32_t udiv_magic(uint32_t v, uint32_t magic, uint32_t d) {  // v / d for d >= 1 (magic = floor(2^32 / d); d = 1: magic saturates)
  if (d == 1u) return v;
  uint32_t q = __umulhi(v, magic);
  if (v - q * d >= d) ++q;
  return q;
}

// TN: 32-channel accumulator blocks per wave along N; the workgroup tile is 64 tiles x (64 TN) output channels, waves 2 x 2
template <int TN>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(WinoArgs a) {
  constexpr int BM = 64, BN = 64 * TN, ROWB = 128;
  constexpr int PLANE = BM * ROWB, RAW_B = 4 * PLANE, A_B = BM * ROWB, BSTAGE = BN * ROWB;
  constexpr int GB = BN / 32;  // B DMA instructions per wave and K-tile
  extern __shared__ __attribute__((aligned(16))) char wsmem[];
  char *raw = wsmem, *As = wsmem + RAW_B, *Bs = As + A_B;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)wsmem;

  const aivc_conv_params &p = a.p;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int H = p.h_in, W = p.w_in, Cin = p.c_in, Cout = p.c_out, M = a.M, TH = a.TH, TW = a.TW;

  // XCD-aware tile order (as conv_mfma.hip): one XCD works on a contiguous run of tiles
  uint32_t tile_id;
  {
    const uint32_t nwg = gridDim.x, bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int by = (int)(tile_id % (uint32_t)a.gy), bx = (int)(tile_id / (uint32_t)a.gy);
  const int m0 = bx * BM, n0 = by * BN;

  // every per-lane input offset is relative to the image of the tile's first row (32-bit byte offsets: a tile of 64 output
  // tiles spans few images; the host checks the span)
  const uint32_t img0 = udiv_magic((uint32_t)m0, a.img_magic, (uint32_t)(TH * TW));
  const float *xbase = p.x + (size_t)img0 * (size_t)H * W * Cin;

  const int l3 = lane >> 3;
  const uint32_t chunk_b = (uint32_t)(((lane & 7) ^ l3 ^ wave) << 4);  // this lane's data chunk (bytes) in a K row
  // the lane's two raw rows (rows 8 wave + l3 and + 32 of the tile)
  int g_y0[2], g_x0[2];
  uint32_t g_nb[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    int m = m0 + 8 * wave + 32 * g + l3;
    m = m < M ? m : M - 1;
    const uint32_t t = udiv_magic((uint32_t)m, a.tw_magic, (uint32_t)TW), tx = (uint32_t)m - t * (uint32_t)TW;
    const uint32_t n = udiv_magic(t, a.th_magic, (uint32_t)TH), ty = t - n * (uint32_t)TH;
    g_y0[g] = 2 * (int)ty - 1;
    g_x0[g] = 2 * (int)tx - 1;
    g_nb[g] = (n - img0) * (uint32_t)(H * W);
  }
  uint32_t g_avo[2][4], g_bvo[GB];
#pragma unroll
  for (int j = 0; j < GB; ++j) {
    const int co = n0 + 32 * j + 8 * wave + l3;  // (c_out % 64 == 0: every row exists)
    g_bvo[j] = (uint32_t)co * (uint32_t)(16 * Cin * 4) + chunk_b;
  }
  const uint32_t adst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)wave * 1024u);
  const uint32_t bdst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(RAW_B + A_B) + (uint32_t)wave * 1024u);

  // state of the NEXT tile to issue
  int nx_pos = 0, nx_c = 0;
  const float *urun = p.w_wino;
  auto set_position = [&](int pos) {  // per-lane byte offsets of the four input pixels position `pos` combines
    const int i = pos >> 2, j = pos & 3;
    const int ai = i == 0 ? 0 : (i == 2 ? 2 : 1), bi = i == 3 ? 3 : (i == 2 ? 1 : 2);
    const int aj = j == 0 ? 0 : (j == 2 ? 2 : 1), bj = j == 3 ? 3 : (j == 2 ? 1 : 2);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int ra = max(min(g_y0[g] + ai, H - 1), 0), rb = max(min(g_y0[g] + bi, H - 1), 0);
      const int ca = max(min(g_x0[g] + aj, W - 1), 0), cb = max(min(g_x0[g] + bj, W - 1), 0);
      const uint32_t c4 = (uint32_t)(Cin * 4);
      g_avo[g][0] = (g_nb[g] + (uint32_t)(ra * W + ca)) * c4 + chunk_b;
      g_avo[g][1] = (g_nb[g] + (uint32_t)(ra * W + cb)) * c4 + chunk_b;
      g_avo[g][2] = (g_nb[g] + (uint32_t)(rb * W + ca)) * c4 + chunk_b;
      g_avo[g][3] = (g_nb[g] + (uint32_t)(rb * W + cb)) * c4 + chunk_b;
    }
  };
  auto issue_tile = [&](int stage) {
    if (nx_c == 0) set_position(nx_pos);
    const float *ab = xbase + nx_c;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int g = 0; g < 2; ++g) wino_glds16(ab, g_avo[g][t], adst + (uint32_t)(t * PLANE + g * 4096));
    const uint32_t bd = bdst + (uint32_t)stage * BSTAGE;
#pragma unroll
    for (int j = 0; j < GB; ++j) wino_glds16(urun, g_bvo[j], bd + j * 4096);
    urun += 32;
    nx_c += 32;
    if (nx_c == Cin) {
      nx_c = 0;
      ++nx_pos;
    }
  };

  floatx16 acc[TN], yacc[4][TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[j][r] = 0.0f;
#pragma unroll
      for (int q = 0; q < 4; ++q) yacc[q][j][r] = 0.0f;
    }

  // fragment reads: lane reads row (lane & 31) of its 32-row blocks, data chunk 2 o + (lane >> 5)
  const int sw = (lane & 7) ^ ((lane >> 3) & 3);
  const char *a_rd = As + (wm * 32 + (lane & 31)) * ROWB;
  const char *b_rd = Bs + (wn * TN * 32 + (lane & 31)) * ROWB;
  int f_off[4];
#pragma unroll
  for (int o = 0; o < 4; ++o) f_off[o] = ((2 * o + (lane >> 5)) ^ sw) << 4;

  const int kc = Cin >> 5, nkt = 16 * kc;
  int cur_pos = 0, cur_c = 0;  // position of the tile being multiplied, K-tiles of it done
  issue_tile(0);
  for (int kt = 0; kt < nkt; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // this tile's raw planes and B stage are in LDS; everybody is done with the A tile
    {
      // V = (x_aa + s_j x_ab) + s_i (x_ba + s_j x_bb), slot-wise on the swizzled image
      const int i = cur_pos >> 2, j = cur_pos & 3;
      const float si = i == 1 ? 1.0f : -1.0f, sj = j == 1 ? 1.0f : -1.0f;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int off = (tid + 256 * u) * 16;
        const float4 xaa = *reinterpret_cast<const float4 *>(raw + off), xab = *reinterpret_cast<const float4 *>(raw + PLANE + off);
        const float4 xba = *reinterpret_cast<const float4 *>(raw + 2 * PLANE + off), xbb = *reinterpret_cast<const float4 *>(raw + 3 * PLANE + off);
        float4 v;
        v.x = __builtin_fmaf(si, __builtin_fmaf(sj, xbb.x, xba.x), __builtin_fmaf(sj, xab.x, xaa.x));
        v.y = __builtin_fmaf(si, __builtin_fmaf(sj, xbb.y, xba.y), __builtin_fmaf(sj, xab.y, xaa.y));
        v.z = __builtin_fmaf(si, __builtin_fmaf(sj, xbb.z, xba.z), __builtin_fmaf(sj, xab.z, xaa.z));
        v.w = __builtin_fmaf(si, __builtin_fmaf(sj, xbb.w, xba.w), __builtin_fmaf(sj, xab.w, xaa.w));
        *reinterpret_cast<float4 *>(As + off) = v;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // the A tile is complete, the raw planes are free
    if (kt + 1 < nkt) issue_tile((kt + 1) & 1);
    {
      const int sb = (kt & 1) * BSTAGE;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const float4 af = *reinterpret_cast<const float4 *>(a_rd + f_off[o]);
        float4 bf[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4 *>(b_rd + sb + j * 32 * ROWB + f_off[o]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float av = s == 0 ? af.x : (s == 1 ? af.y : (s == 2 ? af.z : af.w));
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const float bv = s == 0 ? bf[j].x : (s == 1 ? bf[j].y : (s == 2 ? bf[j].z : bf[j].w));
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
          }
        }
      }
    }
    if (++cur_c == kc) {
      // fold M_p into the four outputs of the tile: coefficient T[a][i] * T[b][j], T[0] = (1, 1, 1, 0), T[1] = (0, 1, -1, -1)
      const int i = cur_pos >> 2, j = cur_pos & 3;
      const int t0i = i < 3 ? 1 : 0, t1i = i == 0 ? 0 : (i == 1 ? 1 : -1);
      const int t0j = j < 3 ? 1 : 0, t1j = j == 0 ? 0 : (j == 1 ? 1 : -1);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cf = ((q >> 1) ? t1i : t0i) * ((q & 1) ? t1j : t0j);  // wave-uniform
        if (cf > 0) {
#pragma unroll
          for (int jj = 0; jj < TN; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) yacc[q][jj][r] = yacc[q][jj][r] + acc[jj][r];
        } else if (cf < 0) {
#pragma unroll
          for (int jj = 0; jj < TN; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) yacc[q][jj][r] = yacc[q][jj][r] - acc[jj][r];
        }
      }
#pragma unroll
      for (int jj = 0; jj < TN; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jj][r] = 0.0f;
      cur_c = 0;
      ++cur_pos;
    }
  }

  // ---- epilogue: the contract's order (Epilogue::finish, common.h) on the four outputs of every tile --------------------
  const float *__restrict__ g_mul = p.mul;
  const float *__restrict__ g_res = p.res;
  float *__restrict__ g_y = p.y;
  const int act1 = p.act1, act2 = p.act2;
  auto act_cheap = [](int act, float v) {  // NONE / LEAKY / RELU of act_apply() without branches
    const float neg = act == AIVC_ACT_LEAKY ? v * 0.01f : (act == AIVC_ACT_RELU ? 0.0f : v);
    return v > 0.0f ? v : neg;
  };
  float cb[TN];
  int cch[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    cch[j] = n0 + (wn * TN + j) * 32 + (lane & 31);
    cb[j] = p.bias ? p.bias[cch[j]] : 0.0f;
  }
  const bool has_bias = p.bias != nullptr;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (m >= M) continue;
    const uint32_t t = udiv_magic((uint32_t)m, a.tw_magic, (uint32_t)TW), tx = (uint32_t)m - t * (uint32_t)TW;
    const uint32_t n = udiv_magic(t, a.th_magic, (uint32_t)TH), ty = t - n * (uint32_t)TH;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int oy = 2 * (int)ty + (q >> 1), ox = 2 * (int)tx + (q & 1);
      if (oy >= H || ox >= W) continue;  // (odd sizes: the last tile row / column holds one pixel row / column)
      const size_t base = (((size_t)n * H + oy) * W + ox) * (size_t)Cout;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const size_t o = base + cch[j];
        float v = yacc[q][j][r];
        if (has_bias) v = v + cb[j];
        v = act_cheap(act1, v);
        if (g_mul) v = g_mul[o] * v;
        if (g_res) v = v + g_res[o];
        v = act_cheap(act2, v);
        g_y[o] = v;
      }
    }
  }
}

// U = G g G^T per (c_out, c_in), fp64 in the order of include/aivc_hip.h, rounded once
__device__ __forceinline__ void wino_g(double g0, double g1, double g2, double (&out)[4]) {
  out[0] = g0;
  out[1] = 0.5 * ((g0 + g1) + g2);
  out[2] = 0.5 * ((g0 - g1) + g2);
  out[3] = g2;
}
__global__ void __launch_bounds__(256) winograd_weights_kernel(const float *w, int c_out, int c_in, float *u) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)c_out * c_in) return;
  const int co = (int)(idx / c_in), ci = (int)(idx % c_in);
  double t[4][3], uu[4][4];
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    double col[4];
    wino_g((double)w[(((size_t)co * 3 + 0) * 3 + l) * c_in + ci], (double)w[(((size_t)co * 3 + 1) * 3 + l) * c_in + ci],
           (double)w[(((size_t)co * 3 + 2) * 3 + l) * c_in + ci], col);
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i][l] = col[i];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) wino_g(t[i][0], t[i][1], t[i][2], uu[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) u[((size_t)co * 16 + 4 * i + j) * c_in + ci] = (float)uu[i][j];
}

int winograd_weights(const float *w, int c_out, int c_in, float *u, hipStream_t s) {
  hipLaunchKernelGGL(winograd_weights_kernel, dim3(cdiv((size_t)c_out * c_in, 256)), dim3(256), 0, s, w, c_out, c_in, u);
  return check_launch("winograd_weights");
}

// what the kernel can address: 32-bit byte offsets relative to the image of a tile's first row
bool conv2d_wino_supported(const aivc_conv_params &p) {
  if (!aivc_winograd_covers(&p) || p.gdn) return false;
  const uint64_t th = (uint64_t)(p.h_in + 1) / 2, tw = (uint64_t)(p.w_in + 1) / 2;
  const uint64_t img_bytes = (uint64_t)p.h_in * p.w_in * p.c_in * 4u;
  const uint64_t span_imgs = 64u / (th * tw) + 2u;  // images a tile of 64 output tiles can touch
  if (img_bytes * span_imgs >= 0xFFFF0000ull) return false;
  if ((uint64_t)p.n * th * tw >= 0x7FFFFFFFull) return false;
  if ((uint64_t)p.c_out * 16u * p.c_in * 4u >= 0xFFFF0000ull) return false;
  return true;
}

int conv2d_wino_variant(const aivc_conv_params &p) { return p.c_out % 128 == 0 ? 305 : 301; }

template <int TN>
static int launch_wino(const aivc_conv_params &p, hipStream_t s) {
  WinoArgs a;
  a.p = p;
  a.TH = (p.h_in + 1) / 2;
  a.TW = (p.w_in + 1) / 2;
  a.M = p.n * a.TH * a.TW;
  a.tw_magic = (uint32_t)(0x100000000ull / (uint64_t)a.TW);
  a.th_magic = (uint32_t)(0x100000000ull / (uint64_t)a.TH);
  a.img_magic = (uint32_t)(0x100000000ull / ((uint64_t)a.TH * a.TW));
  a.gy = p.c_out / (64 * TN);
  const size_t lds = (size_t)4 * 64 * 128 + 64 * 128 + 2 * (64 * TN) * 128;
  static LdsOptIn opt_in;
  if (!opt_in.raise(reinterpret_cast<const void *>(conv_wino_kernel<TN>), lds)) return check_launch("conv_wino lds attribute");
  const unsigned grid = (unsigned)(((a.M + 63) / 64) * a.gy);
  hipLaunchKernelGGL(conv_wino_kernel<TN>, dim3(grid), dim3(256), lds, s, a);
  return check_launch("conv_wino");
}

int conv2d_wino(const aivc_conv_params &p, hipStream_t s) {
  if (!conv2d_wino_supported(p) || !p.w_wino) return AIVC_ERR_UNSUPPORTED;
  return p.c_out % 128 == 0 ? launch_wino<2>(p, s) : launch_wino<1>(p, s);
}

}  // namespace aivc
