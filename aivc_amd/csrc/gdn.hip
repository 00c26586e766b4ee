// gdn.hip -- stand-alone (inverse) GDN with gamma RESIDENT IN REGISTERS (round 6).
//
//   y[m][i] = x[m][i] / sqrt(beta[i] + sum_j gamma[i][j] x[m][j]^2)          (IGDN: x * sqrt(...))
//
//   why            a (I)GDN that cannot be fused into its convolution (version 2 of the contract: a Winograd workgroup owns
//                  64 of the 128 channels of its pixels; models whose widths are no tile width) was a GDN-mode launch of the
//                  generic implicit-GEMM kernel: gamma re-staged through LDS for every pixel tile (as many L2 -> LDS bytes as the
//                  activations themselves), the input read twice, two barriers per K-tile -- 3.9 ms per 64 frames of 272 x 480 x
//                  128 against 1.1 ms of HBM traffic and 1.7 ms of matrix work.
//   this kernel    persistent workgroups; wave w owns 32 output channels and keeps its slice of gamma as MFMA B operands in
//                  registers for the whole launch (C / 2 registers); the activations stream through a two-stage LDS ring
//                  (LDS-DMA, 64 pixels per stage, one barrier per tile); a wave reads its A fragments from the ring, squares
//                  them on the way to the matrix pipe and, in the epilogue, takes x in the accumulator layout from the same
//                  ring -- the input crosses the memory system once.
//   arithmetic     the contract's (include/aivc_hip.h): per output one v_mfma_f32_32x32x2_f32 chain from +0 over j in
//                  groups of 8 in AIVC_K_ORDER, squares rounded once, then + beta, sqrt, division (multiplication), residual:
//                  the bits of the generic kernel's GDN-mode launch and of the CPU oracle.
//   LDS image      row = one pixel (4 C bytes); the 16-byte chunk c of row R sits in slot c ^ swz(R), swz(R) = (R & 15) ^
//                  ((R & 4) << 1): conflict-free ds_read_b128 fragment reads (16-lane groups of the hardware) and
//                  conflict-free ds_read_b32 reads of the epilogue (rows R and R + 4 of the two lane halves land in
//                  different bank octets).  Applied on the SOURCE address of the DMA.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace aivc {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct GdnArgs {
  aivc_conv_params p;
  int M;       // pixels
  int ntiles;  // tiles of 64 pixels
};

constexpr int GDN_BM = 64;

__device__ __forceinline__ void gdn_glds16(const float *base, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds_dst) : "memory", "m0");
}

__device__ __forceinline__ int gdn_swz(int row) { return (row & 15) ^ ((row & 4) << 1); }

// C = 64: 2 channel blocks x 2 row blocks = 4 waves (one accumulator each); C = 128: 4 waves x 2 row blocks (C = 192: 6 waves x 2 --
// measured slower than the generic kernel, not dispatched)
template <int C>
struct GdnCfg {
  static constexpr int NB = C / 32;             // channel blocks of 32
  static constexpr int PW = C == 64 ? 2 : 1;    // waves along the pixel rows
  static constexpr int NW = NB * PW;            // waves per workgroup
  static constexpr int TM = 2 / PW;             // row blocks of 32 per wave
  static constexpr int ROWB = C * 4;            // bytes per LDS row
  static constexpr int STAGE = GDN_BM * ROWB;   // bytes per stage
  static constexpr int PIECES = STAGE / 1024;   // DMA instructions per stage
  static constexpr int PPW = (PIECES + NW - 1) / NW;  // ... per wave
  static constexpr int WGS = C == 192 ? 1 : 2;  // workgroups per CU (LDS: 2 stages each)
};

template <int C, bool INV, bool RES>
__global__ __launch_bounds__(64 * GdnCfg<C>::NW, GdnCfg<C>::WGS) void gdn_resident_kernel(GdnArgs a) {
  using Cfg = GdnCfg<C>;
  constexpr int NB = Cfg::NB, PW = Cfg::PW, NW = Cfg::NW, TM = Cfg::TM, ROWB = Cfg::ROWB, STAGE = Cfg::STAGE, PIECES = Cfg::PIECES, PPW = Cfg::PPW;
  constexpr int NOCT = C / 8;
  extern __shared__ __attribute__((aligned(16))) char gsmem[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)gsmem;
  const aivc_conv_params &p = a.p;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nb = wave % NB, pw = wave / NB;
  const int l31 = lane & 31, hh = lane >> 5;
  const int M = a.M;

  // ---- gamma slice: B operand of step t = 4 o + s is gamma[32 nb + l31][8 o + s + 4 hh] ------------------------------------
  float4 gb[NOCT];
  {
    const float *g = p.w + (size_t)(32 * nb + l31) * C + 4 * hh;
#pragma unroll
    for (int o = 0; o < NOCT; ++o) gb[o] = *reinterpret_cast<const float4 *>(g + 8 * o);
  }
  const float cbeta = p.bias[32 * nb + l31];

  // ---- loader plan: piece q of a stage = bytes 1024 q .. of the stage image; this lane's 16 bytes ---------------------------
  uint32_t d_voff[PPW];
#pragma unroll
  for (int k = 0; k < PPW; ++k) {
    const int q = wave + NW * k;
    const int b = q * 1024 + lane * 16;
    const int row = b / ROWB, chunk = ((b % ROWB) >> 4) ^ gdn_swz(row);
    d_voff[k] = (uint32_t)(row * ROWB + chunk * 16);
  }
  const uint32_t w_dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)wave * 1024u);
  auto issue = [&](int tile, int stage) {
    const int m0 = tile * GDN_BM;
    const float *src = p.x + (size_t)m0 * C;
    const uint32_t dst = w_dst + (uint32_t)(stage * STAGE);
    if (m0 + GDN_BM <= M) {
#pragma unroll
      for (int k = 0; k < PPW; ++k)
        if (PIECES % NW == 0 || wave + NW * k < PIECES) gdn_glds16(src, d_voff[k], dst + (uint32_t)(NW * k) * 1024u);
    } else {  // last, partial tile: rows beyond M fetch the last pixel again (computed, never stored)
      const int last = M - 1 - m0;
#pragma unroll
      for (int k = 0; k < PPW; ++k)
        if (PIECES % NW == 0 || wave + NW * k < PIECES)
          gdn_glds16(src, (uint32_t)min((int)(d_voff[k] / ROWB), last) * (uint32_t)ROWB + d_voff[k] % ROWB, dst + (uint32_t)(NW * k) * 1024u);
    }
  };

  // ---- fragment reads: row 32 (pw + i) + l31 (C = 64: the wave's row block is pw), chunk 2 o + hh ----------------------------
  const int frow = (PW == 2 ? 32 * pw : 0) + l31;
  int f_off[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) f_off[o] = frow * ROWB + (((2 * o + hh) ^ gdn_swz(frow)) << 4);  // + (o >> 3) * 256 + i * 32 * ROWB
  // ---- epilogue reads: x[row][32 nb + l31], row = rb + (r & 3) + 8 (r >> 2) + 4 hh --------------------------------------------
  // swz(row) = ((r & 3) | ((r >> 2) & 1) << 3) ^ (hh ? 12 : 0)  [row & 15 = (r & 3) + 8 ((r >> 2) & 1) + 4 hh, bit 3 ^= bit 2]
  const int e_chunk = (8 * nb + (l31 >> 2)) ^ (hh ? 12 : 0);
  const int e_lane = 4 * hh * ROWB + (l31 & 3) * 4;
  const int rb0 = PW == 2 ? 32 * pw : 0;

  const int G = gridDim.x;
  int tile = blockIdx.x;
  if (tile >= a.ntiles) return;
  issue(tile, 0);
  int stage = 0;
  for (; tile < a.ntiles; tile += G, stage ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the tile have landed (and its stores have left)
    __builtin_amdgcn_s_barrier();                     // ... everybody's; everybody is done with the other stage
    if (tile + G < a.ntiles) issue(tile + G, stage ^ 1);
    const char *sb = gsmem + stage * STAGE;

    floatx16 acc[TM];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    struct Frag { f32x2 lo, hi; };  // k = 8 o + 4 hh + {0, 1}, {2, 3}
    Frag cur[TM], nxt[TM];
    auto read_frag = [&](Frag (&f)[TM], int o) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float4 v = *reinterpret_cast<const float4 *>(sb + f_off[o & 7] + (o >> 3) * 256 + i * 32 * ROWB);
        f[i].lo = (f32x2){v.x, v.y};
        f[i].hi = (f32x2){v.z, v.w};
      }
    };
    read_frag(cur, 0);
#pragma unroll
    for (int o = 0; o < NOCT; ++o) {
      if (o + 1 < NOCT) read_frag(nxt, o + 1);  // in flight during this octet's MFMAs
#pragma unroll
      for (int i = 0; i < TM; ++i) {  // squares, rounded once (v_pk_mul_f32: two per instruction)
        cur[i].lo = cur[i].lo * cur[i].lo;
        cur[i].hi = cur[i].hi * cur[i].hi;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float bv = s == 0 ? gb[o].x : (s == 1 ? gb[o].y : (s == 2 ? gb[o].z : gb[o].w));
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float av = s == 0 ? cur[i].lo.x : (s == 1 ? cur[i].lo.y : (s == 2 ? cur[i].hi.x : cur[i].hi.y));
          if (o == 0 && s == 0) {
            const floatx16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, zero, 0, 0, 0);
          } else {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i) cur[i] = nxt[i];
    }

    // ---- epilogue ----------------------------------------------------------------------------------------------------------
    const int m0 = tile * GDN_BM;
    const int rows_left = M - m0;  // rows of this tile that exist
    typedef __attribute__((address_space(1))) char gchar;
    typedef __attribute__((address_space(1))) float gfloat;
    typedef __attribute__((address_space(1))) const float cgfloat;
    gchar *yb = (gchar *)(uintptr_t)(p.y + (size_t)m0 * C);
    const gchar *rsb = (const gchar *)(uintptr_t)(p.res + (size_t)m0 * C);
    const uint32_t lane_b = (uint32_t)((4 * hh) * ROWB + (32 * nb + l31) * 4);
    auto epilogue = [&](auto WHOLE) {
      constexpr bool whole = decltype(WHOLE)::value;
      constexpr int EG = RES ? 4 : 8;  // outputs per group
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += EG) {  // EG outputs at a time: sixteen interleaved sqrt / division sequences cost the second wave per SIMD
        const int rb = rb0 + 32 * i;
        float xv[EG], sv[EG], rv[EG];
        float mx = 1.0f, mn = 1.0f;
#pragma unroll
        for (int q = 0; q < EG; ++q) {
          const int r = r0 + q, rloc = (r & 3) + 8 * (r >> 2);
          const int cst = (r & 3) | (((r >> 2) & 1) << 3);
          xv[q] = *reinterpret_cast<const float *>(sb + (rb + rloc) * ROWB + e_lane + ((e_chunk ^ cst) << 4));
          if constexpr (RES) {
            const uint32_t off = lane_b + (uint32_t)((rb + rloc) * ROWB);
            rv[q] = (whole || rb + rloc + 4 * hh < rows_left) ? *reinterpret_cast<cgfloat *>(rsb + off) : 0.0f;
          }
        }
#pragma unroll
        for (int q = 0; q < EG; ++q) {
          sv[q] = acc[i][r0 + q] + cbeta;
          if constexpr (INV) gdn_range_pair(mx, mn, sv[q], sv[q]);
          else gdn_range(mx, mn, xv[q], sv[q]);
        }
        // x = +-0 in GDN mode falls out of the lean range (|x| >= 2^-60): such wavefronts take the compiler's sequences
        if (gdn_range_ok(mx, mn)) {
#pragma unroll
          for (int q = 0; q < EG; ++q) {
            const float nrm = sqrt_rn_safe(sv[q]);
            float v = INV ? xv[q] * nrm : div_rn_safe(xv[q], nrm);
            if constexpr (RES) v = v + rv[q];
            sv[q] = v;
          }
        } else {
#pragma unroll
          for (int q = 0; q < EG; ++q) {
            const float nrm = __builtin_sqrtf(sv[q]);
            float v = INV ? xv[q] * nrm : xv[q] / nrm;
            if constexpr (RES) v = v + rv[q];
            sv[q] = v;
            if (q & 1) __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
        for (int q = 0; q < EG; ++q) {
          const int r = r0 + q, rloc = (r & 3) + 8 * (r >> 2);
          const uint32_t off = lane_b + (uint32_t)((rb + rloc) * ROWB);
          if (whole || rb + rloc + 4 * hh < rows_left) *reinterpret_cast<gfloat *>(yb + off) = sv[q];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (rows_left >= GDN_BM) epilogue(std::true_type{});
    else epilogue(std::false_type{});
  }
}

// what the kernel covers: a stand-alone (I)GDN launch (ksize 1, c_in == c_out) of 64 / 128 channels, no activation, no gate
bool gdn_resident_supported(const aivc_conv_params &p) {
  if (p.mode != AIVC_MODE_GDN && p.mode != AIVC_MODE_IGDN) return false;
  if (p.c_in != 64 && p.c_in != 128) return false;  // (192 channels: gamma alone is 96 registers, one workgroup per CU: 4.1 against 3.85 ms generic)
  if (p.c_out != p.c_in || !p.bias || p.mul || p.act1 != AIVC_ACT_NONE || p.act2 != AIVC_ACT_NONE) return false;
  if (((uintptr_t)p.x & 15u) || ((uintptr_t)p.w & 15u)) return false;
  const uint64_t pix = (uint64_t)p.n * p.h_in * p.w_in;
  return pix >= 1 && pix < 0x7FFFFFC0ull;
}

template <int C, bool INV, bool RES>
static int gdn_launch(const GdnArgs &a, hipStream_t s, int n_cu) {
  using Cfg = GdnCfg<C>;
  constexpr size_t lds = 2 * (size_t)Cfg::STAGE;
  static LdsOptIn opt_in;
  if (!opt_in.raise(reinterpret_cast<const void *>(gdn_resident_kernel<C, INV, RES>), lds)) return check_launch("gdn_resident lds attribute");
  unsigned grid = (unsigned)(n_cu * Cfg::WGS);
  if ((unsigned)a.ntiles < grid) grid = (unsigned)a.ntiles;
  hipLaunchKernelGGL((gdn_resident_kernel<C, INV, RES>), dim3(grid), dim3(64 * Cfg::NW), lds, s, a);
  return check_launch("gdn_resident");
}

int gdn_resident(const aivc_conv_params &p, hipStream_t s) {
  if (!gdn_resident_supported(p)) return AIVC_ERR_UNSUPPORTED;
  GdnArgs a;
  a.p = p;
  a.M = p.n * p.h_in * p.w_in;
  a.ntiles = (a.M + GDN_BM - 1) / GDN_BM;
  static std::atomic<int> n_cu{0};
  if (n_cu.load(std::memory_order_relaxed) == 0) {
    int dev = 0, cus = 0;
    n_cu = hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0 ? cus : 256;
  }
  const int cus = n_cu.load(std::memory_order_relaxed);
  const bool inv = p.mode == AIVC_MODE_IGDN, res = p.res != nullptr;
  auto go = [&](auto CC) {
    constexpr int C = decltype(CC)::value;
    if (inv) return res ? gdn_launch<C, true, true>(a, s, cus) : gdn_launch<C, true, false>(a, s, cus);
    return res ? gdn_launch<C, false, true>(a, s, cus) : gdn_launch<C, false, false>(a, s, cus);
  };
  if (p.c_in == 64) return go(std::integral_constant<int, 64>{});
  return go(std::integral_constant<int, 128>{});
}

}  // namespace aivc
