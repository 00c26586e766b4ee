// conv_mfma.hip -- implicit-GEMM convolution family on the gfx950 matrix cores, exact fp32.
//
//   GEMM view      M = output pixels (n, oy, ox), N = output channels, K = (ky, kx, ci)
//   instruction    v_mfma_f32_32x32x2_f32: D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)) with k0 supplied by lanes 0-31
//                  and k1 by lanes 32-63: a fixed-order fp32 fmaf chain (no split-K, no atomics, one accumulator
//                  per output).
//   accumulation   THE ARITHMETIC CONTRACT of the conv family (include/aivc_hip.h): the reduction index
//   order          kk = tap * c_in + ci (taps in (ky, kx) order; transposed conv: the taps of the output's parity
//                  class) is walked in groups of 8, inside a group in the order 0, 4, 1, 5, 2, 6, 3, 7.  That is
//                  what the MFMA does when an LDS row holds K in natural order and lane half h reads the 16 bytes
//                  at column 8 o + 4 h: step s multiplies k = 8 o + s (half 0) then k = 8 o + 4 + s (half 1).
//                  Staging is therefore a plain copy global -> registers -> LDS (a permuted LDS layout that
//                  reproduced an ascending chain cost 32 v_mov per K-tile and thread; measured: every issued
//                  instruction costs the fp32 matrix pipe ~4.4 cycles).  The scalar kernel, the thin kernels and
//                  the CPU oracle walk K in the same order, so all of them agree bit for bit.
//   roofline       fp32 MFMA = 157.3 TFLOP/s dense (256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz).
//   data layout    activations NHWC, weights OHWI: a K-slice of one pixel / one output channel is a
//                  contiguous run of floats -> every global access is a 16-byte load.
//   staging        global -> registers (issued one K-tile ahead, in flight during the MFMAs) ->
//                  LDS (one buffer, 2 barriers per K-tile; 2-3 workgroups per CU hide them).
//                  LDS rows hold BK = 32 K-values (+4 pad -> conflict-free ds_read_b128).
//   im2col         done in the loader's address arithmetic: replicate padding = clamp of the input
//                  coordinate; transposed conv = 4 output-parity classes (slowest tile index), each a small
//                  dense conv over the taps of that parity with zero fill outside the image.
//   epilogue       bias / GDN division / activation / gate / residual fused, straight from the
//                  accumulators (lanes 0-31 of a row write 128 contiguous bytes).
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"

namespace aivc {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct MfmaArgs {
  aivc_conv_params p;
  int M;               // GEMM rows per z-slice
  uint32_t cin_magic;  // ceil(2^32 / c_in): k / c_in == (k * magic) >> 32 for k < 2^16
  uint32_t w_magic, h_magic;  // floor(2^32 / w_in), floor(2^32 / h_in): quotient low by at most one
  int gx, gy;                  // pixel tiles, c_out tiles (the grid is 1-D: gx x gy x parity classes)
  int tc_order;                // transposed conv: 1 = XCD-contiguous tile runs inside a parity class
};

constexpr int BK = 32;
constexpr int OCT = BK / 8;  // 8-float units per LDS row
constexpr int LDS_STRIDE = BK + 4;

#ifdef AIVC_TUNING  // per-workgroup phase timestamps (tools/phase_probe.py builds a copy with -DAIVC_TUNING; never in the product library)
__device__ unsigned long long aivc_dbg_t[8 * 8192];
#define DBG_T(i) if (threadIdx.x == 0 && blockIdx.x < 8192) aivc_dbg_t[blockIdx.x * 8 + (i)] = (i) == 0 || (i) == 5 ? wall_clock64() : clock64()
#else
#define DBG_T(i)
#endif

#ifndef AIVC_TAIL_WAVES
#define AIVC_TAIL_WAVES 2  // waves per SIMD the fused-tail kernel is compiled for (3 spills; measured equal)
#endif
constexpr int TAIL_N = 128;  // output channels of the fused 1x1 tail (the bottleneck blocks: 64 -> 128)

// LDS-DMA of 16 bytes per lane (global_load_lds_dwordx4): LDS destination = lds_dst (wave-uniform, through M0) +
// lane * 16, source = base (SGPR pair) + voff (per-lane byte offset).  Counts on vmcnt like a load; no VGPR result.
__device__ __forceinline__ void glds16(const float *base, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds_dst) : "memory", "m0");
}

// bf16x3: x = h + m + l exactly, each term the bf16 nearest (ties to even) to what is left: 8 + 8 + 8 significant bits.
// Two values at a time (v_cvt_pk_bf16_f32 packs a pair); used by the K loop and by aivc_split_weights_bf16x3.
__device__ __forceinline__ void bf16x3_split2(float x0, float x1, uint32_t &h, uint32_t &m, uint32_t &l) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  auto pk = [](float u, float v) {
    const bf16x2 t = __builtin_convertvector((f32x2){u, v}, bf16x2);  // round to nearest even
    return __builtin_bit_cast(uint32_t, t);
  };
  h = pk(x0, x1);
  const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xFFFF0000u);
  m = pk(r0, r1);
  const float q0 = r0 - __builtin_bit_cast(float, m << 16), q1 = r1 - __builtin_bit_cast(float, m & 0xFFFF0000u);
  l = pk(q0, q1);
}

// PREC: 0 = the fp32 arithmetic contract (v_mfma_f32_32x32x2_f32, fixed-order fmaf chains); 1 = "bf16x3" (round 5, a
// precision MODE, never the default): every fp32 operand is split exactly into three bf16 terms x = h + m + l and a
// product a * b is the six bf16 MFMA products h h', h m', m h', m m', h l', l h' with fp32 accumulation
// (v_mfma_f32_32x32x16_bf16: 16x the fp32 MFMA rate per instruction) -- the dropped terms are below 2^-24 |a b|.
// Same LDS image, loader, epilogues and fused phases; only the K loop's fragment reads and MFMAs differ.  Results are
// NOT the contract's bits (other summation tree): parity per mode is reported by tests/test_gpu_precision.py.
// PREC 2 = PREC 1 with the weights split ahead of the launch (aivc_conv_params.w_bf16x3): the B side of a stage is three
// bf16 planes of [BN rows][32 k] (64 bytes per row and plane), fetched by the same LDS-DMA, and a fragment is one
// ds_read_b128 per term; the same terms in the same products as PREC 1, so the same bits.
template <int MODE, int WM, int WN, int TM, int TN, bool FUSE, bool FASTK, bool TAIL = false, bool GLDS = false, int PREC = 0>
__global__ __launch_bounds__(256, (PREC ? 1 : (TAIL ? AIVC_TAIL_WAVES : (TM * TN >= 8 ? 2 : (FUSE && TM * TN == 2 && WN == 2 && TM == 2 ? 3 : 1))))) void conv_mfma_kernel(MfmaArgs a) {
  constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
  static_assert(!TAIL || (MODE == AIVC_MODE_CONV && !FUSE && FASTK && BN == 64 && BN % BK == 0), "fused tail: conv, c_out 64");
  static_assert(PREC == 0 || (GLDS && FASTK), "bf16x3: the LDS-DMA loop with c_in % 32 == 0");
  constexpr int UA = BM * OCT / 256;          // (row, octet) units per thread for A
  constexpr int UB = (BN * OCT + 255) / 256;  // ... for B
  constexpr bool B_FULL = (BN * OCT) % 256 == 0;  // every thread stages a B unit: no exec masking
  constexpr bool TCONV = MODE == AIVC_MODE_TCONV;
  constexpr bool GDN = MODE == AIVC_MODE_GDN;  // covers IGDN (runtime mode in the epilogue)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  DBG_T(0);
  DBG_T(1);
#ifdef AIVC_TUNING  // per-workgroup phase timestamps (tools/phase_probe.py builds a copy with -DAIVC_TUNING; never in the product library)
  if (threadIdx.x == 0 && blockIdx.x < 8192) aivc_dbg_t[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);
#endif
  float *As = smem;
  float *Bs = smem + BM * LDS_STRIDE;

  const aivc_conv_params &p = a.p;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // XCD-aware tile order (stride-1/2 conv and GDN).  The dispatcher deals consecutive workgroup ids
  // round-robin to the 8 XCDs (each with a private 4 MiB L2); remap (bijectively) so that one XCD works on
  // a contiguous run of tiles, ordered so that neighbours share input: the c_out tiles of one pixel tile
  // first (same A rows), then the next pixel tile (shared halo rows).
  // Transposed conv: parity class slowest in dispatch order (its 4 classes have reductions of different length,
  // 9/6/6/4 taps for k = 5: whole-grid contiguous runs per XCD would unbalance the XCDs), and INSIDE a class every
  // XCD gets a contiguous run of the class's tiles (a.tc_order = 1, round 3: + 1 ... 2 % on the transposed layers;
  // the vertical halo rows of neighbouring tiles meet in one L2).  Class fastest -- the four classes of a pixel tile
  // back to back on one XCD -- measured 25 % slower, with the round-robin XCD deal and with contiguous runs alike.
  uint32_t tile_id = blockIdx.x;
  const uint32_t per_class = (uint32_t)a.gx * (uint32_t)a.gy;
  int bz = 0;
  if (MODE != AIVC_MODE_TCONV) {
    const uint32_t nwg = gridDim.x, bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  } else {
    bz = (int)(tile_id / per_class);
    tile_id -= (uint32_t)bz * per_class;
    if (a.tc_order == 1) {
      // the workgroups of one XCD inside this class are those with the same (id within the class) & 7
      const uint32_t l = tile_id, q = per_class >> 3, r = per_class & 7, v = l & 7;
      tile_id = (v < r ? v * (q + 1) : r * (q + 1) + (v - r) * q) + (l >> 3);
    }
  }
  const int by = (int)(tile_id % (uint32_t)a.gy), bx = (int)(tile_id / (uint32_t)a.gy);
  const int m0 = bx * BM, n0 = by * BN;
  const int ks = p.ksize, Cin = p.c_in, H = p.h_in, W = p.w_in, Cout = p.c_out;
  const int M = a.M;

  int pyc = 0, pxc = 0, ky0 = 0, kx0 = 0, nky = ks, nkx = ks, tpad = 0;
  if (TCONV) {
    tpad = (ks + 1) / 2 - 1;
    pyc = bz >> 1;
    pxc = bz & 1;
    ky0 = (pyc + tpad) & 1;
    kx0 = (pxc + tpad) & 1;
    nky = (ks - ky0 + 1) / 2;
    nkx = (ks - kx0 + 1) / 2;
  }
  const int K = nky * nkx * Cin;
  const int nkt = (K + BK - 1) / BK;
  const uint32_t inv_nkx = (65536u + nkx - 1) / nkx;

  // ---- per-thread loader units ------------------------------------------------------------
  int a_by[UA], a_bx[UA];
  uint32_t a_nb[UA];
#pragma unroll
  for (int j = 0; j < UA; ++j) {
    const int row = (tid + 256 * j) / OCT;
    int m = m0 + row;
    m = m < M ? m : M - 1;
    if (TCONV) {
      const int qx = m % W, t = m / W;
      a_bx[j] = qx;
      a_by[j] = t % H;
      a_nb[j] = (uint32_t)(t / H) * (uint32_t)(H * W);
    } else {
      const int ox = m % p.w_out, t = m / p.w_out;
      a_bx[j] = ox * p.stride - p.pad;
      a_by[j] = (t % p.h_out) * p.stride - p.pad;
      a_nb[j] = (uint32_t)(t / p.h_out) * (uint32_t)(H * W);
    }
  }

  float4 ra[UA][2], rb[UB][2];

  auto tap_of = [&](int kk, int &ty, int &tx, int &ci) {
    const int tap = (int)(((uint64_t)(uint32_t)kk * a.cin_magic) >> 32);
    ci = kk - tap * Cin;
    ty = (int)(((uint32_t)tap * inv_nkx) >> 16);
    tx = tap - ty * nkx;
  };

  // FASTK (c_in % 32 == 0): a K-tile lies inside one kernel tap, K % 32 == 0, so the loader is
  // branch-free: one clamped pixel address per unit, two 16-byte loads off it, weights by pointer
  // bump.  Out-of-image taps of the transposed conv are loaded from a clamped address and zeroed.
  uint32_t b_off[UB];
  bool b_ok[UB];
#pragma unroll
  for (int j = 0; j < UB; ++j) {
    const int u = tid + 256 * j;
    const int co = n0 + u / OCT;
    b_ok[j] = B_FULL || u < BN * OCT;
    const int coc = co < Cout ? co : Cout - 1;  // clamped: rows beyond c_out are never stored
    b_off[j] = TCONV ? (uint32_t)coc * (uint32_t)(ks * ks * Cin) + (uint32_t)((u % OCT) * 8)
                     : (uint32_t)coc * (uint32_t)K + (uint32_t)((u % OCT) * 8);
  }
  // The element offset of the unit's input pixel only changes when the K-tile enters a new kernel tap (every
  // c_in / 32 tiles): it is kept in a register and recomputed behind a wave-uniform branch, every other tile
  // costs one add per unit (instruction count is what the matrix pipe pays for, see the header).
  uint32_t a_pix[UA];
  bool a_in[UA];
#pragma unroll
  for (int j = 0; j < UA; ++j) {
    a_pix[j] = 0;
    a_in[j] = true;
  }
  auto load_tile_fast = [&](int kt) {
    const int kbase = kt * BK;
    int ty, tx, ci0;
    tap_of(kbase, ty, tx, ci0);  // wave-uniform
    if (ci0 == 0) {
      const int dyt = (pyc + tpad - (ky0 + 2 * ty)) >> 1, dxt = (pxc + tpad - (kx0 + 2 * tx)) >> 1;
#pragma unroll
      for (int j = 0; j < UA; ++j) {
        int iy = a_by[j] + (TCONV ? dyt : ty), ix = a_bx[j] + (TCONV ? dxt : tx);
        if (TCONV) a_in[j] = iy >= 0 && iy < H && ix >= 0 && ix < W;
        iy = max(min(iy, H - 1), 0);
        ix = max(min(ix, W - 1), 0);
        a_pix[j] = (a_nb[j] + (uint32_t)(iy * W + ix)) * (uint32_t)Cin + (uint32_t)(((tid + 256 * j) % OCT) * 8);
      }
    }
#pragma unroll
    for (int j = 0; j < UA; ++j) {
      const float *src = p.x + (a_pix[j] + (uint32_t)ci0);
      float4 v0 = *reinterpret_cast<const float4 *>(src);
      float4 v1 = *reinterpret_cast<const float4 *>(src + 4);
      if (TCONV) {  // select, not multiply by 0/1: avoids NaN * 0
        v0 = a_in[j] ? v0 : make_float4(0.f, 0.f, 0.f, 0.f);
        v1 = a_in[j] ? v1 : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (GDN) {
        v0.x *= v0.x; v0.y *= v0.y; v0.z *= v0.z; v0.w *= v0.w;
        v1.x *= v1.x; v1.y *= v1.y; v1.z *= v1.z; v1.w *= v1.w;
      }
      ra[j][0] = v0;
      ra[j][1] = v1;
    }
    const uint32_t wk = TCONV ? (uint32_t)(((ky0 + 2 * ty) * ks + kx0 + 2 * tx) * Cin + ci0) : (uint32_t)kbase;
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      if (B_FULL || b_ok[j]) {
        const float *src = p.w + (b_off[j] + wk);
        rb[j][0] = *reinterpret_cast<const float4 *>(src);
        rb[j][1] = *reinterpret_cast<const float4 *>(src + 4);
      }
    }
  };

  // generic path (small c_in: a K-tile straddles taps, K has a zero-padded tail): still branch-free --
  // every address is clamped to something valid and the value is zeroed by a select.
  auto load_tile_generic = [&](int kt) {
    const int kbase = kt * BK;
#pragma unroll
    for (int j = 0; j < UA; ++j) {
      const int oct = (tid + 256 * j) % OCT;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int kk = kbase + oct * 8 + q * 4;
        const int kc = kk < K ? kk : K - 4;
        int ty, tx, ci;
        tap_of(kc, ty, tx, ci);
        int iy, ix;
        bool ok = kk < K;
        if (TCONV) {
          iy = a_by[j] + ((pyc + tpad - (ky0 + 2 * ty)) >> 1);
          ix = a_bx[j] + ((pxc + tpad - (kx0 + 2 * tx)) >> 1);
          ok = ok && iy >= 0 && iy < H && ix >= 0 && ix < W;
        } else {
          iy = a_by[j] + ty;
          ix = a_bx[j] + tx;
        }
        iy = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy);
        ix = ix < 0 ? 0 : (ix > W - 1 ? W - 1 : ix);
        const uint32_t off = (a_nb[j] + (uint32_t)(iy * W + ix)) * (uint32_t)Cin + (uint32_t)ci;
        float4 v = *reinterpret_cast<const float4 *>(p.x + off);
        v = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        if (GDN) {
          v.x = v.x * v.x;
          v.y = v.y * v.y;
          v.z = v.z * v.z;
          v.w = v.w * v.w;
        }
        ra[j][q] = v;
      }
    }
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      const int u = tid + 256 * j;
      const int oct = u % OCT;
      const int co = n0 + u / OCT;
      const int coc = co < Cout ? co : Cout - 1;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int kk = kbase + oct * 8 + q * 4;
        const int kc = kk < K ? kk : K - 4;
        uint32_t off;
        if (TCONV) {
          int ty, tx, ci;
          tap_of(kc, ty, tx, ci);
          off = ((uint32_t)coc * (uint32_t)(ks * ks) + (uint32_t)((ky0 + 2 * ty) * ks + kx0 + 2 * tx)) * (uint32_t)Cin +
                (uint32_t)ci;
        } else {
          off = (uint32_t)coc * (uint32_t)K + (uint32_t)kc;
        }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (B_FULL || b_ok[j]) v = *reinterpret_cast<const float4 *>(p.w + off);
        rb[j][q] = (kk < K && co < Cout) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto load_tile = [&](int kt) {
    if constexpr (FASTK) load_tile_fast(kt);
    else load_tile_generic(kt);
  };

  auto store_tile_a = [&](int buf_off = 0) {
#pragma unroll
    for (int j = 0; j < UA; ++j) {
      const int u = tid + 256 * j;
      float *dst = As + buf_off + (u / OCT) * LDS_STRIDE + (u % OCT) * 8;
      *reinterpret_cast<float4 *>(dst) = make_float4(ra[j][0].x, ra[j][0].y, ra[j][0].z, ra[j][0].w);
      *reinterpret_cast<float4 *>(dst + 4) = make_float4(ra[j][1].x, ra[j][1].y, ra[j][1].z, ra[j][1].w);
    }
  };
  auto store_tile_b = [&](int buf_off = 0) {
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      const int u = tid + 256 * j;
      if (B_FULL || u < BN * OCT) {
        float *dst = Bs + buf_off + (u / OCT) * LDS_STRIDE + (u % OCT) * 8;
        *reinterpret_cast<float4 *>(dst) = make_float4(rb[j][0].x, rb[j][0].y, rb[j][0].z, rb[j][0].w);
        *reinterpret_cast<float4 *>(dst + 4) = make_float4(rb[j][1].x, rb[j][1].y, rb[j][1].z, rb[j][1].w);
      }
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const float *a_frag = As + (wm * TM * 32 + (lane & 31)) * LDS_STRIDE + (lane >> 5) * 4;
  const float *b_frag = Bs + (wn * TN * 32 + (lane & 31)) * LDS_STRIDE + (lane >> 5) * 4;

  // one K-tile of MFMAs out of LDS (measured in isolation, tools/mfma_probe.hip: this loop keeps the
  // matrix pipe 98% busy, i.e. what is lost in the whole kernel is lost outside of it)
  // AIVC_CONV_SPARSE4 (3-channel images stored as 4 channels): reduction indices kk with kk % 4 == 3 multiply a
  // zero input -- step s = 3 of every octet (k = 8 o + 3 and 8 o + 7) is an exact no-op and is not issued.
  const bool skip3 = !FASTK && (p.flags & AIVC_CONV_SPARSE4) != 0;
  auto mma_octs = [&](floatx16 (&c)[TM][TN], int buf_off, int o_lo, int o_hi, bool skip) {
#pragma unroll
    for (int o = o_lo; o < o_hi; ++o) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4 *>(a_frag + buf_off + i * 32 * LDS_STRIDE + o * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4 *>(b_frag + buf_off + j * 32 * LDS_STRIDE + o * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s == 3 && skip) continue;  // wave-uniform
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float av = s == 0 ? af[i].x : (s == 1 ? af[i].y : (s == 2 ? af[i].z : af[i].w));
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const float bv = s == 0 ? bf[j].x : (s == 1 ? bf[j].y : (s == 2 ? bf[j].z : bf[j].w));
            c[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c[i][j], 0, 0, 0);
          }
        }
      }
    }
  };
  auto mma_tile = [&](floatx16 (&c)[TM][TN], bool skip) { mma_octs(c, 0, 0, OCT, skip); };

  if constexpr (GLDS) {
    // ---- LDS-DMA K loop (round 3) -----------------------------------------------------------------------------
    // Operand tiles go global -> LDS by global_load_lds_dwordx4: no staging registers, no ds_write pass, no per-K-tile
    // address arithmetic on the vector unit (the K-tile's channel offset sits in the scalar base, the per-lane pixel
    // offset changes only with the kernel tap).  What that buys on this part: the sustained
    // fp32 MFMA rate is capped near 135 TFLOP/s by a limiter that trades clock against pipe use, so the way up is
    // less data movement per MFMA -- 0.27 instead of 0.39 LDS instructions per MFMA, no VGPR round trip.
    //   LDS image   two stages of (BM + BN) rows x 128 bytes, UNPADDED (a DMA writes 64 x 16 contiguous bytes: 8
    //               rows), XOR-swizzled instead: the 16-byte slot s of row R holds data chunk s ^ swz(R),
    //               swz(R) = (R & 7) ^ ((R >> 3) & 3) -- applied on the SOURCE address of the DMA and on the
    //               fragment reads alike; conflict-free ds_read_b128 for 8- and 16-lane groups.
    //   schedule    tile kt+1 is in flight during the MFMAs of tile kt; the wait for it, the one barrier per K-tile
    //               and the first fragment reads of tile kt+1 sit in the shadow of tile kt's last 16 MFMAs, after
    //               which this stage is refilled with tile kt+2 (its last readers passed the barrier with their
    //               fragments in registers).  Accumulation order unchanged: octets ascending, AIVC_K_ORDER inside.
    //   transposed  out-of-image taps (zero fill): such lanes take no part in the DMA and write 16 zero bytes to their
    //   conv        slot instead; tiles away from the image border never see the branch (wave-uniform test per tap).
    //   generic K   (c_in % 32 != 0: the image layers, c_in of 4 / 8 / 12; conv only) a 16-byte chunk is one (tap, 4 input
    //               channels) quad: every lane decodes the quad of its chunk once per K-tile -- the same for all of its
    //               rows -- and clamps per row; the quads beyond K of the last tile write zeros instead of loading.
    static_assert(!GDN && (FASTK || MODE == AIVC_MODE_CONV), "LDS-DMA loop: conv / transposed conv (c_in % 32 == 0), conv (any c_in % 4 == 0)");
    // (PREC 2: B rows are 3 planes x 64 bytes; one DMA instruction covers 16 rows of one plane, 3 BN / 16 of them per stage)
    constexpr int ROWB = BK * 4, BPLANE = BN * 64, STAGE_B = PREC == 2 ? BM * ROWB + 3 * BPLANE : (BM + BN) * ROWB;
    constexpr int GA = BM / 32, GB = PREC == 2 ? 3 * BN / 64 : BN / 32;
    char *ring = reinterpret_cast<char *>(smem);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)ring;
    const int l3 = lane >> 3;
    const uint32_t chunk_b = (uint32_t)(((lane & 7) ^ l3 ^ wave) << 4);  // this lane's data chunk (bytes) in a K row
    int g_by[GA], g_bx[GA];
    uint32_t g_nb[GA], g_avo[GA], g_bvo[GB];
    bool g_in[GA];
    bool tap_all_in = true;  // wave-uniform: no lane of this wave samples outside the image at the current tap
#pragma unroll
    for (int j = 0; j < GA; ++j) {
      int m = m0 + 32 * j + 8 * wave + l3;
      m = m < M ? m : M - 1;
      if (TCONV) {
        const int t = m / W;
        g_bx[j] = m % W;
        g_by[j] = t % H;
        g_nb[j] = (uint32_t)(t / H) * (uint32_t)(H * W);
      } else {
        const int ox = m % p.w_out, t = m / p.w_out;
        g_bx[j] = ox * p.stride - p.pad;
        g_by[j] = (t % p.h_out) * p.stride - p.pad;
        g_nb[j] = (uint32_t)(t / p.h_out) * (uint32_t)(H * W);
      }
      g_avo[j] = 0;
      g_in[j] = true;
    }
#pragma unroll
    for (int j = 0; j < GB; ++j) {
      if constexpr (PREC == 2) {
        // instruction q of the stage's 3 BN / 16: plane q / (BN / 16), rows 16 (q % (BN / 16)) ..; lane -> row + (lane >> 2),
        // LDS chunk slot lane & 3 holds data chunk slot ^ ((row >> 2) & 3): the 16 lanes of a ds_read_b128 phase then
        // hit 16 different bank quads (rows of 64 bytes)
        const int q = wave * GB + j, pl = q / (BN / 16), r = 16 * (q % (BN / 16)) + (lane >> 2);
        const int co = n0 + r, coc = co < Cout ? co : Cout - 1;
        g_bvo[j] = (uint32_t)coc * (uint32_t)(ks * ks * Cin * 6) + (uint32_t)(pl * 64) + (uint32_t)((((lane & 3) ^ ((r >> 2) & 3))) << 4);
      } else {
        const int co = n0 + 32 * j + 8 * wave + l3;
        const int coc = co < Cout ? co : Cout - 1;  // rows beyond c_out are never stored
        g_bvo[j] = (uint32_t)coc * (uint32_t)((TCONV ? ks * ks * Cin : K) * 4) + (FASTK ? chunk_b : 0u);
      }
    }
    const uint32_t wdst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)wave * 1024u);
    const uint32_t bwdst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(BM * ROWB) + (uint32_t)wave * (uint32_t)(GB * 1024));  // PREC 2: this wave's B instructions
    // tiles are issued in K order: the (tap, channel offset) of the next one is kept as running scalars
    int ty = 0, tx = 0, ci0 = 0;
    const float *wrun = p.w;  // conv: weight row offset of the next tile
    int kt_next = 0;           // generic K: index of the next tile to issue
    auto issue_tile_generic = [&](int stage) {
      const int kk = kt_next * BK + (int)(chunk_b >> 2);  // first reduction index of this lane's quad
      const bool qok = kk < K;                            // beyond K (last tile only): zeros
      int qty, qtx, qci;
      tap_of(qok ? kk : 0, qty, qtx, qci);
      const bool tail = (kt_next + 1) * BK > K;           // wave-uniform
      const uint32_t dst = wdst + (uint32_t)stage * STAGE_B;
#pragma unroll
      for (int j = 0; j < GA; ++j) {
        const int iy = max(min(g_by[j] + qty, H - 1), 0), ix = max(min(g_bx[j] + qtx, W - 1), 0);
        const uint32_t off = ((g_nb[j] + (uint32_t)(iy * W + ix)) * (uint32_t)Cin + (uint32_t)qci) * 4u;
        if (!tail || qok) glds16(p.x, off, dst + j * 4096);
        else *reinterpret_cast<float4 *>(ring + stage * STAGE_B + (32 * j + 8 * wave) * ROWB + lane * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int j = 0; j < GB; ++j) {
        if (!tail || qok) glds16(p.w, g_bvo[j] + (uint32_t)kk * 4u, dst + BM * ROWB + j * 4096);
        else *reinterpret_cast<float4 *>(ring + stage * STAGE_B + (BM + 32 * j + 8 * wave) * ROWB + lane * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      ++kt_next;
    };
    auto issue_tile_fast = [&](int stage) {
      if (ci0 == 0) {              // new kernel tap: the per-lane pixel offsets change
        const int dyt = (pyc + tpad - (ky0 + 2 * ty)) >> 1, dxt = (pxc + tpad - (kx0 + 2 * tx)) >> 1;
        bool all_in = true;
#pragma unroll
        for (int j = 0; j < GA; ++j) {
          int iy = g_by[j] + (TCONV ? dyt : ty), ix = g_bx[j] + (TCONV ? dxt : tx);
          if (TCONV) {
            g_in[j] = iy >= 0 && iy < H && ix >= 0 && ix < W;
            all_in = all_in && g_in[j];
          }
          iy = max(min(iy, H - 1), 0);
          ix = max(min(ix, W - 1), 0);
          g_avo[j] = (g_nb[j] + (uint32_t)(iy * W + ix)) * (uint32_t)(Cin * 4) + chunk_b;
        }
        if (TCONV) tap_all_in = __builtin_amdgcn_ballot_w64(all_in) == ~0ull;
      }
      const float *ab = p.x + ci0;
      const float *bb = TCONV ? p.w + (((ky0 + 2 * ty) * ks + kx0 + 2 * tx) * Cin + ci0) : wrun;
      const uint32_t dst = wdst + (uint32_t)stage * STAGE_B;
      if (!TCONV || tap_all_in) {
#pragma unroll
        for (int j = 0; j < GA; ++j) glds16(ab, g_avo[j], dst + j * 4096);
      } else {
#pragma unroll
        for (int j = 0; j < GA; ++j) {
          if (g_in[j]) glds16(ab, g_avo[j], dst + j * 4096);
          else *reinterpret_cast<float4 *>(ring + stage * STAGE_B + (32 * j + 8 * wave) * ROWB + lane * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      if constexpr (PREC == 2) {
        // tile t of the split weights: 192 bytes per row and tile (aivc_split_weights_bf16x3)
        const int kidx = TCONV ? ((ky0 + 2 * ty) * ks + kx0 + 2 * tx) * Cin + ci0 : (int)(wrun - p.w);
        const float *bs = reinterpret_cast<const float *>(reinterpret_cast<const char *>(p.w_bf16x3) + (size_t)(kidx / BK) * 192);
        const uint32_t bdst = bwdst + (uint32_t)stage * STAGE_B;
#pragma unroll
        for (int j = 0; j < GB; ++j) glds16(bs, g_bvo[j], bdst + j * 1024);
      } else {
#pragma unroll
        for (int j = 0; j < GB; ++j) glds16(bb, g_bvo[j], dst + BM * ROWB + j * 4096);
      }
      wrun += BK;
      ci0 += BK;
      if (ci0 == Cin) {
        ci0 = 0;
        if (++tx == nkx) {
          tx = 0;
          ++ty;
        }
      }
    };
    auto issue_tile = [&](int stage) {
      if constexpr (FASTK) issue_tile_fast(stage);
      else issue_tile_generic(stage);
    };
    // fragment reads: lane reads row (lane & 31) of its 32-row blocks, data chunk 2 o + (lane >> 5)
    const int sw = (lane & 7) ^ ((lane >> 3) & 3);
    const char *a_rd[OCT], *b_rd[OCT];
#pragma unroll
    for (int o = 0; o < OCT; ++o) {
      const int off = ((2 * o + (lane >> 5)) ^ sw) << 4;
      a_rd[o] = ring + (wm * TM * 32 + (lane & 31)) * ROWB + off;
      b_rd[o] = ring + BM * ROWB + (wn * TN * 32 + (lane & 31)) * ROWB + off;
    }
    float4 fa[2][TM], fb[2][TN];
    auto read_oct = [&](auto SET, auto STAGE, auto O) {
      constexpr int set = decltype(SET)::value, stage = decltype(STAGE)::value, o = decltype(O)::value;
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[set][i] = *reinterpret_cast<const float4 *>(a_rd[o] + stage * STAGE_B + i * 32 * ROWB);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[set][j] = *reinterpret_cast<const float4 *>(b_rd[o] + stage * STAGE_B + j * 32 * ROWB);
    };
    auto mfma_step = [&](auto SET, auto S) {
      constexpr int set = decltype(SET)::value, st = decltype(S)::value;
      if constexpr (!FASTK && st == 3) {
        if (skip3) return;  // AIVC_CONV_SPARSE4: k % 4 == 3 multiplies the zero pad channel of an image (exact no-op)
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float av = st == 0 ? fa[set][i].x : (st == 1 ? fa[set][i].y : (st == 2 ? fa[set][i].z : fa[set][i].w));
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float bv = st == 0 ? fb[set][j].x : (st == 1 ? fb[set][j].y : (st == 2 ? fb[set][j].z : fb[set][j].w));
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
        }
      }
    };
    if constexpr (PREC != 0) {
      // ---- bf16x3 K loop: a K tile of 32 is two slabs of 16; a fragment = 8 consecutive k of one row (two 16-byte
      // chunks), lanes 0-31 take k 0-7 of the slab, lanes 32-63 k 8-15 (the operand layout of the 32x32x16 MFMA).
      // Per slab a wave splits its TM + TN raw fragments (44 vector instructions each) and issues 6 x TM x TN MFMAs.
      // The two phases do not overlap on this part: a vector instruction costs the matrix pipe its 4 issue cycles whoever
      // issues it (the law of round 2) -- dealing the split out behind the MFMAs in source order (volatile
      // asm; hipcc's schedulers otherwise gather the splits in front of the MFMAs of a block whatever fences or
      // sched_group_barrier ask) measured 133 instead of 143 TFLOP/s fp32-equivalent (experiments/r05.md 7).
      typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      struct Raw { float4 lo, hi; };       // k 0-3, k 4-7 of the lane's 8
      struct Tri { u32x4 h, m, l; };       // the three bf16 terms, packed in k order
      const char *a_rs[2][2], *b_rs[2][2];  // [slab][chunk of the pair]
#pragma unroll
      for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int off = ((4 * sl + 2 * (lane >> 5) + c) ^ sw) << 4;
          a_rs[sl][c] = ring + (wm * TM * 32 + (lane & 31)) * ROWB + off;
          b_rs[sl][c] = ring + BM * ROWB + (wn * TN * 32 + (lane & 31)) * ROWB + off;
        }
      // PREC 2: the lane's fragment of plane 0 (8 consecutive k of its row: chunk 2 slab + (lane >> 5), swizzled as the loader wrote it)
      const char *b_rp[2];
#pragma unroll
      for (int sl = 0; sl < 2; ++sl)
        b_rp[sl] = ring + BM * ROWB + (wn * TN * 32 + (lane & 31)) * 64 + (((2 * sl + (lane >> 5)) ^ ((lane >> 2) & 3)) << 4);
      Raw ra[2][TM], rb[2][PREC == 2 ? 1 : TN];
      Tri tbr[2][PREC == 2 ? TN : 1];
      auto read_slab = [&](auto SET, auto STAGE, auto SLAB) {
        constexpr int set = decltype(SET)::value, stage = decltype(STAGE)::value, sl = decltype(SLAB)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          ra[set][i].lo = *reinterpret_cast<const float4 *>(a_rs[sl][0] + stage * STAGE_B + i * 32 * ROWB);
          ra[set][i].hi = *reinterpret_cast<const float4 *>(a_rs[sl][1] + stage * STAGE_B + i * 32 * ROWB);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (PREC == 2) {
            tbr[set][j].h = *reinterpret_cast<const u32x4 *>(b_rp[sl] + stage * STAGE_B + j * 32 * 64);
            tbr[set][j].m = *reinterpret_cast<const u32x4 *>(b_rp[sl] + stage * STAGE_B + BPLANE + j * 32 * 64);
            tbr[set][j].l = *reinterpret_cast<const u32x4 *>(b_rp[sl] + stage * STAGE_B + 2 * BPLANE + j * 32 * 64);
          } else {
            rb[set][j].lo = *reinterpret_cast<const float4 *>(b_rs[sl][0] + stage * STAGE_B + j * 32 * ROWB);
            rb[set][j].hi = *reinterpret_cast<const float4 *>(b_rs[sl][1] + stage * STAGE_B + j * 32 * ROWB);
          }
        }
      };
      auto split = [&](const Raw &r) {
        uint32_t h[4], m[4], l[4];
        bf16x3_split2(r.lo.x, r.lo.y, h[0], m[0], l[0]);
        bf16x3_split2(r.lo.z, r.lo.w, h[1], m[1], l[1]);
        bf16x3_split2(r.hi.x, r.hi.y, h[2], m[2], l[2]);
        bf16x3_split2(r.hi.z, r.hi.w, h[3], m[3], l[3]);
        return Tri{(u32x4){h[0], h[1], h[2], h[3]}, (u32x4){m[0], m[1], m[2], m[3]}, (u32x4){l[0], l[1], l[2], l[3]}};
      };
      auto mm = [&](floatx16 &c, const u32x4 &x, const u32x4 &y) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0);
      };
      auto slab_mfmas = [&](auto SET) {
        constexpr int set = decltype(SET)::value;
        Tri ta[TM], tb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) ta[i] = split(ra[set][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (PREC == 2) tb[j] = tbr[set][j];
          else tb[j] = split(rb[set][j]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {  // small terms first
            mm(acc[i][j], ta[i].l, tb[j].h);
            mm(acc[i][j], ta[i].h, tb[j].l);
            mm(acc[i][j], ta[i].m, tb[j].m);
            mm(acc[i][j], ta[i].m, tb[j].h);
            mm(acc[i][j], ta[i].h, tb[j].m);
            mm(acc[i][j], ta[i].h, tb[j].h);
          }
      };
      using std::integral_constant;
      using J0 = integral_constant<int, 0>;
      using J1 = integral_constant<int, 1>;
      auto body_bf = [&](auto STAGE, int kt) {
        constexpr int stage = decltype(STAGE)::value;
        using NEXT = integral_constant<int, 1 - stage>;
        read_slab(J1{}, STAGE, J1{});   // slab 1 of this tile on its way
        slab_mfmas(J0{});               // slab 0 (read at the end of the previous tile)
        if (kt + 1 < nkt) {
          // every read of this stage is in registers, this wave's DMAs (and zero fills) of tile kt + 1 have landed ...
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();  // ... and everybody else's
          read_slab(J0{}, NEXT{}, J0{});
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (kt + 2 < nkt) issue_tile(stage);
        slab_mfmas(J1{});
      };
      issue_tile(0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (nkt > 1) issue_tile(1);
      read_slab(J0{}, J0{}, J0{});
      for (int kt = 0; kt < nkt; kt += 2) {
        body_bf(J0{}, kt);
        if (kt + 1 < nkt) body_bf(J1{}, kt + 1);
      }
      __syncthreads();  // the ring is reused (padded layout) by the fused phases below
    } else {
    using std::integral_constant;
    using I0 = integral_constant<int, 0>;
    using I1 = integral_constant<int, 1>;
    using I2 = integral_constant<int, 2>;
    using I3 = integral_constant<int, 3>;
#define AIVC_SB() __builtin_amdgcn_sched_barrier(0)
    auto oct_mfmas = [&](auto SET) { mfma_step(SET, I0{}); mfma_step(SET, I1{}); mfma_step(SET, I2{}); mfma_step(SET, I3{}); };
    auto body = [&](auto STAGE, auto CHECK, int kt) {  // CHECK false: the caller guarantees kt + 2 < nkt
      constexpr int stage = decltype(STAGE)::value;
      constexpr bool chk = decltype(CHECK)::value;
      using NEXT = integral_constant<int, 1 - stage>;
      read_oct(I1{}, STAGE, I1{});
      AIVC_SB();
      oct_mfmas(I0{});
      AIVC_SB();
      read_oct(I0{}, STAGE, I2{});
      AIVC_SB();
      oct_mfmas(I1{});
      AIVC_SB();
      read_oct(I1{}, STAGE, I3{});
      AIVC_SB();
      oct_mfmas(I0{});
      AIVC_SB();
      mfma_step(I1{}, I0{});
      AIVC_SB();
      if (!chk || kt + 1 < nkt) {
        // this wave's DMAs (and zero fills) of tile kt + 1 have landed ...
        if (TCONV || !FASTK) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // ... and everybody else's; everybody is done reading this stage
      }
      AIVC_SB();
      mfma_step(I1{}, I1{});
      AIVC_SB();
      if (!chk || kt + 1 < nkt) read_oct(I0{}, NEXT{}, I0{});
      AIVC_SB();
      mfma_step(I1{}, I2{});
      AIVC_SB();
      if (!chk || kt + 2 < nkt) issue_tile(stage);
      AIVC_SB();
      mfma_step(I1{}, I3{});
      AIVC_SB();
    };
    issue_tile(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (nkt > 1) issue_tile(1);
    read_oct(I0{}, I0{}, I0{});
    using std::false_type;
    using std::true_type;
    int kt = 0;
    for (; kt + 3 < nkt; kt += 2) {  // both tiles of the pair have two successors: no end-of-reduction tests
      if (kt == 2) { DBG_T(2); }
      body(I0{}, false_type{}, kt);
      body(I1{}, false_type{}, kt + 1);
    }
    for (; kt < nkt; kt += 2) {
      body(I0{}, true_type{}, kt);
      if (kt + 1 < nkt) body(I1{}, true_type{}, kt + 1);
    }
#undef AIVC_SB
    __syncthreads();  // the ring is reused (padded layout) by the fused phases below
    }  // PREC
  } else {
  load_tile(0);
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt == 1) { DBG_T(2); }
    __syncthreads();
    store_tile_a();
    store_tile_b();
    __syncthreads();
    if (kt + 1 < nkt) load_tile(kt + 1);
    mma_tile(acc, skip3);
  }
  }


  // ---- fused (I)GDN: second, small GEMM  s[m][i] = sum_j x[m][j]^2 * gamma[i][j]  -------------
  // The biased conv outputs x stay in `acc`; their squares go through LDS (the A tile buffer) one
  // 32-channel chunk at a time, gamma streams through the B tile buffer.  BN == c_out here, so a
  // workgroup owns every channel of its pixels.
  DBG_T(3);
  floatx16 acc2[FUSE ? TM : 1][FUSE ? TN : 1];
  if constexpr (FUSE) {
    // gamma chunk kt2 -> registers (the weight staging registers are free now), one chunk ahead of its use
    auto load_gamma = [&](int kt2) {
#pragma unroll
      for (int j = 0; j < UB; ++j) {
        const int u = tid + 256 * j;
        if (u < BN * OCT) {
          const float *src = p.gdn_gamma + (size_t)(u / OCT) * Cout + kt2 * BK + (u % OCT) * 8;
          rb[j][0] = *reinterpret_cast<const float4 *>(src);
          rb[j][1] = *reinterpret_cast<const float4 *>(src + 4);
        }
      }
    };
    load_gamma(0);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int co = (wn * TN + j) * 32 + (lane & 31);
      const float b = p.bias ? p.bias[co] : 0.0f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (p.bias) acc[i][j][r] = acc[i][j][r] + b;
          acc2[i][j][r] = 0.0f;
        }
    }
    const int k_in = lane & 31;
    const int pos = k_in;  // natural channel order in the LDS row (see the accumulation order in the header)
    constexpr int NB = BK / 32;  // 32-channel accumulator blocks per K chunk
    const int nk2 = Cout / BK;
    for (int kt2 = 0; kt2 < nk2; ++kt2) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int blk = wn * TN + j - kt2 * NB;
        if (blk >= 0 && blk < NB) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
              float xv = acc[i][j][r];
              asm volatile("" : "+v"(xv));  // keeps the squares inside the loop (hoisted, they cost 64 registers)
              As[row * LDS_STRIDE + blk * 32 + pos] = xv * xv;
            }
        }
      }
      store_tile_b();
      __syncthreads();
      if (kt2 + 1 < nk2) load_gamma(kt2 + 1);
      mma_tile(acc2, false);  // the GDN reduction runs over all (real) channels
    }
  }

  // ---- fused 1x1 tail: y = act2(W3 . act1(acc + bias) + b3 (+ res)), TAIL_N output channels ---------------
  // Same mechanism as the fused GDN: the activated outputs t of this conv (BN == c_out: a workgroup owns every
  // channel of its pixels) go through the A tile buffer 32 channels at a time, the 1x1 weights stream through
  // the B buffer, a second MFMA GEMM [BM x c_out] x [c_out x TAIL_N] accumulates in the order of a stand-alone
  // 1x1 launch (K tiles of 32, octets in AIVC_K_ORDER) -- bit identical to the two launches, and the c_out-wide
  // intermediate never goes to memory.
  if constexpr (TAIL) {
    constexpr int TN2 = TAIL_N / (32 * WN);
    constexpr int UB2 = TAIL_N * OCT / 256;
    float4 rt[UB2][2];
    auto load_w3 = [&](int kt2) {
#pragma unroll
      for (int j = 0; j < UB2; ++j) {
        const int u = tid + 256 * j;
        const float *src = p.tail_w + (size_t)(u / OCT) * Cout + kt2 * BK + (u % OCT) * 8;
        rt[j][0] = *reinterpret_cast<const float4 *>(src);
        rt[j][1] = *reinterpret_cast<const float4 *>(src + 4);
      }
    };
    load_w3(0);
    // The residual operand is fetched while the tail GEMM runs, 32 output rows (one accumulator row block) at a
    // time into the registers the K loop no longer needs: fetched in the epilogue, each output row was one
    // dependent HBM round trip (measured: 30 % of a workgroup's time, 25 us for 128 KB).
    const int col = wn * TN2 * 32 + (lane & 31);
    const int lrow = 4 * (lane >> 5);
    const int wrow = wm * TM * 32 + lrow;    // first tile row of this lane
    const int rows_left = M - m0 - wrow;     // rows of this lane that exist, counted from wrow
    const bool whole = m0 + BM <= M;
    const bool has_res = p.res != nullptr;
    float rv[TM][16][TN2];
    // residual and output addresses: ONE wave-uniform base per tile (scalar registers) + a 32-bit element offset per lane,
    // rows and channel blocks as constants off it -- as 64-bit per-element pointers every access cost four vector
    // instructions and two hazard nops of address arithmetic (this kernel runs two waves per SIMD: nothing hides them)
    typedef __attribute__((address_space(1))) char gchar;
    typedef __attribute__((address_space(1))) float gfloat;
    typedef __attribute__((address_space(1))) const float cgfloat;
    const uint32_t lane_b = (uint32_t)(wrow * TAIL_N + col) * 4u;  // byte offset of this lane's first element in the tile
    // rows k = 8 g + q (g = 0 .. 4 TM - 1, q = 0 .. 3) of the lane: one 32-bit offset per group g (4 KB apart), row and
    // channel block inside the instruction's 12-bit immediate
    auto group_off = [&](int g) {
      uint32_t vo = lane_b + (uint32_t)(g * 8 * TAIL_N * 4);
      asm volatile("" : "+v"(vo));  // (kept as a register: folded back into 64-bit address arithmetic otherwise)
      return vo;
    };
    auto fetch_res = [&](int i) {
      if (!has_res) return;
      const gchar *rbase = (const gchar *)(uintptr_t)(p.res + (size_t)m0 * TAIL_N);
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const uint32_t vo = group_off(i * 4 + rq);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int r = rq * 4 + rr, k = i * 32 + rr + 8 * rq;
          // rows beyond M: any valid address, never stored
          const uint32_t off = (whole || k < rows_left) ? vo + (uint32_t)(rr * TAIL_N * 4) : (uint32_t)col * 4u;
#pragma unroll
          for (int j = 0; j < TN2; ++j) rv[i][r][j] = *reinterpret_cast<cgfloat *>(rbase + off + (uint32_t)(128 * j));
        }
      }
    };
    {
      const int a1 = p.act1;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const float b = p.bias[(wn * TN + j) * 32 + (lane & 31)];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[i][j][r] + b;
            const float neg = a1 == AIVC_ACT_LEAKY ? v * 0.01f : (a1 == AIVC_ACT_RELU ? 0.0f : v);
            acc[i][j][r] = v > 0.0f ? v : neg;
          }
      }
    }
    fetch_res(0);
    floatx16 acc3[TM][TN2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[i][j][r] = 0.0f;
    const float *b2_frag = Bs + (wn * TN2 * 32 + (lane & 31)) * LDS_STRIDE + (lane >> 5) * 4;
    constexpr int nk2 = BN / BK;
#pragma unroll
    for (int kt2 = 0; kt2 < nk2; ++kt2) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (wn * TN + j == kt2) {  // wave-uniform: this wave holds the 32 channels of chunk kt2
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
              As[row * LDS_STRIDE + (lane & 31)] = acc[i][j][r];
            }
        }
      }
#pragma unroll
      for (int j = 0; j < UB2; ++j) {
        const int u = tid + 256 * j;
        float *dst = Bs + (u / OCT) * LDS_STRIDE + (u % OCT) * 8;
        *reinterpret_cast<float4 *>(dst) = make_float4(rt[j][0].x, rt[j][0].y, rt[j][0].z, rt[j][0].w);
        *reinterpret_cast<float4 *>(dst + 4) = make_float4(rt[j][1].x, rt[j][1].y, rt[j][1].z, rt[j][1].w);
      }
      __syncthreads();
      if (kt2 + 1 < nk2) load_w3(kt2 + 1);
      if (kt2 + 1 == nk2) {  // the K-loop accumulators are dead now: their registers take the other row blocks
#pragma unroll
        for (int i = 1; i < TM; ++i) fetch_res(i);
      }
#pragma unroll
      for (int o = 0; o < OCT; ++o) {
        float4 af[TM], bf[TN2];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4 *>(a_frag + i * 32 * LDS_STRIDE + o * 8);
#pragma unroll
        for (int j = 0; j < TN2; ++j) bf[j] = *reinterpret_cast<const float4 *>(b2_frag + j * 32 * LDS_STRIDE + o * 8);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const float av = s == 0 ? af[i].x : (s == 1 ? af[i].y : (s == 2 ? af[i].z : af[i].w));
#pragma unroll
            for (int j = 0; j < TN2; ++j) {
              const float bv = s == 0 ? bf[j].x : (s == 1 ? bf[j].y : (s == 2 ? bf[j].z : bf[j].w));
              acc3[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc3[i][j], 0, 0, 0);
            }
          }
      }
    }
    DBG_T(4);
    // epilogue over TAIL_N channels: bias, residual, activation (arithmetic of Epilogue::finish, common.h); the
    // combination is chosen once per workgroup, rows beyond M are masked in the last pixel tile only
    {
      float cb[TN2];
#pragma unroll
      for (int j = 0; j < TN2; ++j) cb[j] = p.tail_bias[col + 32 * j];
      gchar *yb = (gchar *)(uintptr_t)(p.y + (size_t)m0 * TAIL_N);
      auto emit = [&](auto KIND, auto WHOLE) {
        constexpr int KD = decltype(KIND)::value;  // 0-2: act2 none/relu/leaky, no residual; 3-5: same after the residual
        constexpr bool WH = decltype(WHOLE)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const uint32_t vo = group_off(i * 4 + rq);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const int r = rq * 4 + rr, k = i * 32 + rr + 8 * rq;
              if (!WH && k >= rows_left) continue;
#pragma unroll
              for (int j = 0; j < TN2; ++j) {
                float v = acc3[i][j][r] + cb[j];
                if constexpr (KD >= 3) v = v + rv[i][r][j];
                if constexpr (KD % 3 == 1) v = v > 0.0f ? v : 0.0f;
                // leaky as a maximum: the same bits as the select for every input (v and 0.01 v have one sign, so no
                // +0 / -0 question arises), one instruction and no condition-register hazard
                if constexpr (KD % 3 == 2) v = __builtin_fmaxf(v, v * 0.01f);
                *reinterpret_cast<gfloat *>(yb + vo + (uint32_t)(rr * TAIL_N * 4 + 128 * j)) = v;
              }
            }
          }
      };
      using std::integral_constant;
      const int kind = (has_res ? 3 : 0) + (p.act2 == AIVC_ACT_RELU ? 1 : (p.act2 == AIVC_ACT_LEAKY ? 2 : 0));
      auto emit_k = [&](auto WHOLE) {
        switch (kind) {
          case 0: emit(integral_constant<int, 0>{}, WHOLE); break;
          case 1: emit(integral_constant<int, 1>{}, WHOLE); break;
          case 2: emit(integral_constant<int, 2>{}, WHOLE); break;
          case 3: emit(integral_constant<int, 3>{}, WHOLE); break;
          case 4: emit(integral_constant<int, 4>{}, WHOLE); break;
          default: emit(integral_constant<int, 5>{}, WHOLE); break;
        }
      };
      if (whole) emit_k(integral_constant<bool, true>{});
      else emit_k(integral_constant<bool, false>{});
    }
    DBG_T(6);
    DBG_T(5);
    return;
  }

  DBG_T(4);
  // ---- lean epilogue for whole tiles ------------------------------------------------------------
  // Every instruction a wave issues costs the matrix pipe ~4.4 cycles (measured: dummy VALU or SALU
  // instructions in the K loop cost the same), so the epilogue is written for instruction count: one uniform
  // 64-bit base per tile, one 32-bit byte offset per lane and output row, the channel blocks of a row as
  // immediate offsets; the activation / residual combination is selected once per workgroup (uniform branch)
  // instead of per element.  Same arithmetic, in the same order, as Epilogue::store/finish (common.h).
  // Partial tiles, missing bias, gates and the sigmoid take the general path below.
  {
    const int a1 = p.act1, a2 = p.act2;
    int kind = -1;  // 0-2: act1 none/leaky/relu, no residual; 3-5: residual then act2 none/relu/leaky; 6: leaky, residual
    if (p.mul == nullptr && p.bias != nullptr) {
      if (p.res == nullptr && a2 == AIVC_ACT_NONE && a1 != AIVC_ACT_SIGMOID) kind = a1 == AIVC_ACT_NONE ? 0 : (a1 == AIVC_ACT_LEAKY ? 1 : 2);
      if (p.res != nullptr && a1 == AIVC_ACT_NONE && a2 != AIVC_ACT_SIGMOID) kind = a2 == AIVC_ACT_NONE ? 3 : (a2 == AIVC_ACT_RELU ? 4 : 5);
      if (p.res != nullptr && a1 == AIVC_ACT_LEAKY && a2 == AIVC_ACT_NONE) kind = 6;
    }
    // 7: the attention gate x + trunk * sigmoid(conv) -- 64x64 tiles only (16 outputs per thread: sixteen inlined
    // copies of the fp64-polynomial sigmoid fit the instruction cache, the 128 of a 128x128 tile did not)
    if (TM * TN == 1 && p.mul != nullptr && p.res != nullptr && p.bias != nullptr && a1 == AIVC_ACT_SIGMOID && a2 == AIVC_ACT_NONE)
      kind = 7;
    if ((FUSE || GDN) && kind != 0 && kind != 3) kind = -1;
    const bool whole = m0 + BM <= M && n0 + BN <= Cout && (!TCONV || W >= BM);
    if (whole && kind >= 0) {
      // an opaque zero: nothing below may be scheduled / hoisted above this point (the fused GDN phase before
      // it is at the register limit of two waves per SIMD)
      int opq = 0;
      asm volatile("" : "+s"(opq) : : "memory");
      const int col = n0 + wn * TN * 32 + (lane & 31) + opq;
      const int lrow = 4 * (lane >> 5);
      float cb[TN], cbeta[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        cb[j] = FUSE ? 0.0f : p.bias[col + 32 * j];
        cbeta[j] = FUSE ? p.gdn_beta[col + 32 * j] : 0.0f;
      }
      // element (row k of this wave's sub-tile, channel block j) lives at  base + k * kstep + roff(k) + 128 j  bytes
      const uint32_t mw = (uint32_t)(m0 + wm * TM * 32 + opq);  // first row of the wave's sub-tile (uniform)
      ptrdiff_t base_elems;
      uint32_t lane_off, qx0 = 0;
      size_t kstep;
      if constexpr (!TCONV) {
        base_elems = (ptrdiff_t)mw * Cout;
        lane_off = (uint32_t)(lrow * Cout + col) * 4u;
        kstep = (size_t)Cout * 4;
      } else {
        // output pixel of GEMM row m: 4 m - 2 (m mod W) + pyc * w_out + pxc   (h_out = 2 H, w_out = 2 W)
        const uint32_t ml = mw + (uint32_t)lrow;
        const uint32_t t = __umulhi(ml, a.w_magic);
        qx0 = ml - t * (uint32_t)W;
        qx0 = qx0 >= (uint32_t)W ? qx0 - (uint32_t)W : qx0;
        base_elems = ((ptrdiff_t)4 * mw + pyc * p.w_out + pxc - 2 * W) * Cout;
        lane_off = (uint32_t)(4 * lrow * Cout + col) * 4u;
        kstep = (size_t)Cout * 16;
      }
      const uint32_t cout8 = (uint32_t)Cout * 8u;
      char *yb = reinterpret_cast<char *>(p.y + base_elems);
      const char *rb_ = reinterpret_cast<const char *>(p.res + base_elems);
      const char *xb = reinterpret_cast<const char *>(p.x + base_elems);
      const char *mb_ = reinterpret_cast<const char *>(p.mul + base_elems);
      const bool inv = FUSE ? p.gdn == 2 : p.mode == AIVC_MODE_IGDN;
      auto emit = [&](auto KIND, auto INV) {
        constexpr int KD = decltype(KIND)::value;
        constexpr bool IV = decltype(INV)::value;
        auto row_of = [&](int idx, int &k, uint32_t &off) {
          k = (idx >> 4) * 32 + (idx & 3) + 8 * ((idx & 15) >> 2);
          off = lane_off;
          if constexpr (TCONV) {
            uint32_t qx = qx0 + (uint32_t)k;
            const uint32_t qw = qx - (uint32_t)W;
            qx = qx < qw ? qx : qw;  // one wrap at most (W >= BM): the unsigned difference is huge when there is none
            off += ((uint32_t)W - qx) * cout8;
          }
        };
        // Fused GDN + residual (the closing conv of the residual blocks): rows are kept apart by scheduling barriers
        // (registers), which made every row one dependent round trip for its residual (46 k instead of 24 k cycles
        // per tile, tools/phase_probe.py): the residual of row idx + 4 is requested while row idx is computed.
        constexpr int RD = ((FUSE && KD >= 3 && !TCONV) || KD == 7) ? 4 : 0;  // (transposed: the extra row offsets cost the second wave per SIMD)
        float rq[RD ? RD : 1][TN], mq[KD == 7 ? RD : 1][TN];
        auto load_res = [&](int idx, float (&dst)[TN]) {
          int k;
          uint32_t off;
          row_of(idx, k, off);
          const char *rrow = rb_ + (size_t)k * kstep;
#pragma unroll
          for (int j = 0; j < TN; ++j) dst[j] = *reinterpret_cast<const float *>(rrow + off + 128 * j);
          if constexpr (KD == 7) {  // the gate's multiplicand travels with the residual
            const char *mrow = mb_ + (size_t)k * kstep;
#pragma unroll
            for (int j = 0; j < TN; ++j) mq[idx % RD][j] = *reinterpret_cast<const float *>(mrow + off + 128 * j);
          }
        };
        if constexpr (RD > 0) {
#pragma unroll
          for (int d = 0; d < RD; ++d) load_res(d, rq[d]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int idx = i * 16 + r;
            int k;
            uint32_t off;
            row_of(idx, k, off);
            char *yrow = yb + (size_t)k * kstep;
            const char *rrow = rb_ + (size_t)k * kstep;
            const char *xrow = xb + (size_t)k * kstep;
            float rv[TN], xv[TN], mv[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              if constexpr (KD == 7) mv[j] = mq[idx % RD][j];
              if constexpr (RD > 0) rv[j] = rq[idx % RD][j];
              else if constexpr (KD >= 3) rv[j] = *reinterpret_cast<const float *>(rrow + off + 128 * j);
              if constexpr (GDN) xv[j] = *reinterpret_cast<const float *>(xrow + off + 128 * j);
            }
            if constexpr (RD > 0) {
              if (idx + RD < TM * 16) load_res(idx + RD, rq[idx % RD]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              float v;
              if constexpr (FUSE) {
                const float nrm = __builtin_sqrtf(acc2[i][j][r] + cbeta[j]);
                v = IV ? acc[i][j][r] * nrm : acc[i][j][r] / nrm;
              } else {
                v = acc[i][j][r] + cb[j];
                if constexpr (GDN) {
                  const float nrm = __builtin_sqrtf(v);
                  v = IV ? xv[j] * nrm : xv[j] / nrm;
                }
              }
              if constexpr (KD == 7) v = mv[j] * aivc_sigmoidf_det(v);
              if constexpr (KD == 1 || KD == 6) v = v > 0.0f ? v : v * 0.01f;
              if constexpr (KD == 2) v = v > 0.0f ? v : 0.0f;
              if constexpr (KD >= 3) v = v + rv[j];
              if constexpr (KD == 4) v = v > 0.0f ? v : 0.0f;
              if constexpr (KD == 5) v = v > 0.0f ? v : v * 0.01f;
              *reinterpret_cast<float *>(yrow + off + 128 * j) = v;
            }
            // keep the rows apart: interleaving the sqrt / division sequences of many rows costs registers
            // (the fused kernels sit at the 256-register limit of two waves per SIMD)
            if constexpr (FUSE || GDN) __builtin_amdgcn_sched_barrier(0);
          }
      };
      using std::integral_constant;
      if constexpr (FUSE || GDN) {
        if (kind == 0) { if (inv) emit(integral_constant<int, 0>{}, integral_constant<bool, true>{}); else emit(integral_constant<int, 0>{}, integral_constant<bool, false>{}); }
        else { if (inv) emit(integral_constant<int, 3>{}, integral_constant<bool, true>{}); else emit(integral_constant<int, 3>{}, integral_constant<bool, false>{}); }
      } else {
        switch (kind) {
          case 0: emit(integral_constant<int, 0>{}, integral_constant<bool, false>{}); break;
          case 1: emit(integral_constant<int, 1>{}, integral_constant<bool, false>{}); break;
          case 2: emit(integral_constant<int, 2>{}, integral_constant<bool, false>{}); break;
          case 3: emit(integral_constant<int, 3>{}, integral_constant<bool, false>{}); break;
          case 4: emit(integral_constant<int, 4>{}, integral_constant<bool, false>{}); break;
          case 5: emit(integral_constant<int, 5>{}, integral_constant<bool, false>{}); break;
          case 7:
            if constexpr (TM * TN == 1) emit(integral_constant<int, 7>{}, integral_constant<bool, false>{});
            break;
          default: emit(integral_constant<int, 6>{}, integral_constant<bool, false>{}); break;
        }
      }
      DBG_T(6);
      DBG_T(5);
      return;
    }
  }
  // ---- general epilogue -------------------------------------------------------------------------
  // Same arithmetic and order as Epilogue::store/finish (common.h), organised for the instruction cache:
  // the unrolled per-accumulator code holds only branch-free work (bias, GDN division, leaky/relu as a
  // select, gate, residual); the sigmoid activation (an fp64 polynomial, include/aivc_detmath.h) would be
  // inlined 128 times there -- 150 KB of code that even when skipped made every output an instruction
  // cache miss and stretched the epilogue to a fifth of the kernel.  Layers that use it (attention gates)
  // store the pre-activation value and finish in a rolled second pass over the thread's own outputs.
  // Per-channel constants are read once; per-pixel operands of 4 rows are fetched before those rows are
  // stored so no load queues behind a store.
  {
    const float *__restrict__ g_mul = p.mul;
    const float *__restrict__ g_res = p.res;
    const float *__restrict__ g_x = p.x;
    float *__restrict__ g_y = p.y;
    const int act1 = p.act1, act2 = p.act2;
    const bool heavy = act1 == AIVC_ACT_SIGMOID || act2 == AIVC_ACT_SIGMOID;
    const bool has_bias = p.bias != nullptr, has_mul = p.mul != nullptr && !heavy, has_res = p.res != nullptr && !heavy;
    const bool gdn_mode = GDN;  // stand-alone (I)GDN launch: normalise the input by the accumulator
    auto act_cheap = [](int act, float v) {  // NONE / LEAKY / RELU of act_apply() without branches
      const float neg = act == AIVC_ACT_LEAKY ? v * 0.01f : (act == AIVC_ACT_RELU ? 0.0f : v);
      return v > 0.0f ? v : neg;
    };
    auto out_pixel = [&](uint32_t mc) -> size_t {
      if (!TCONV) return (size_t)mc;
      uint32_t t = __umulhi(mc, a.w_magic), qx = mc - t * (uint32_t)W;
      if (qx >= (uint32_t)W) { ++t; qx -= (uint32_t)W; }
      uint32_t n = __umulhi(t, a.h_magic), qy = t - n * (uint32_t)H;
      if (qy >= (uint32_t)H) { ++n; qy -= (uint32_t)H; }
      return ((size_t)n * p.h_out + (2 * qy + pyc)) * p.w_out + (2 * qx + pxc);
    };
    float cb[TN], cbeta[TN];
    int cch[TN];
    bool cok[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int co = n0 + (wn * TN + j) * 32 + (lane & 31);
      cok[j] = co < Cout;
      cch[j] = cok[j] ? co : Cout - 1;
      cb[j] = (!FUSE && has_bias) ? p.bias[cch[j]] : 0.0f;
      cbeta[j] = FUSE ? p.gdn_beta[cch[j]] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        size_t base[4];
        bool ok[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int m = m0 + (wm * TM + i) * 32 + rr + 8 * rg + 4 * (lane >> 5);
          ok[rr] = m < M;
          base[rr] = out_pixel((uint32_t)(ok[rr] ? m : M - 1)) * (size_t)Cout;
        }
        float vm[4][TN], vr[4][TN], vx[4][TN];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const size_t o = base[rr] + cch[j];
            vm[rr][j] = has_mul ? g_mul[o] : 1.0f;
            vr[rr][j] = has_res ? g_res[o] : 0.0f;
            vx[rr][j] = gdn_mode ? g_x[o] : 0.0f;
          }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int r = rg * 4 + rr;
            float v;
            if constexpr (FUSE) {
              const float nrm = __builtin_sqrtf(acc2[i][j][r] + cbeta[j]);
              const float xv = acc[i][j][r];
              v = p.gdn == 2 ? xv * nrm : xv / nrm;
            } else {
              v = acc[i][j][r];
              if (has_bias) v = v + cb[j];
              if (gdn_mode) {
                const float nrm = __builtin_sqrtf(v);
                v = p.mode == AIVC_MODE_IGDN ? vx[rr][j] * nrm : vx[rr][j] / nrm;
              }
            }
            if (!heavy) {
              v = act_cheap(act1, v);
              if (has_mul) v = vm[rr][j] * v;
              if (has_res) v = v + vr[rr][j];
              v = act_cheap(act2, v);
            }
            if (ok[rr] && cok[j]) g_y[base[rr] + cch[j]] = v;
          }
      }
    }
    if (heavy) {
      __threadfence_block();  // the pass below re-reads this thread's own stores
      float *y2 = p.y;
      // 4 outputs per trip: their loads overlap (one per trip left the pass latency-bound: a dependent
      // load - sigmoid - store chain per output), the code still holds only 4 copies of the fp64 sigmoid
#pragma unroll 4
      for (int q = 0; q < TM * 16 * TN; ++q) {
        const int j = q % TN, r = (q / TN) & 15, i = q / (TN * 16);
        const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int co = n0 + (wn * TN + j) * 32 + (lane & 31);
        if (m < M && co < Cout) {
          const size_t o = out_pixel((uint32_t)m) * (size_t)Cout + co;
          float v = act_apply(act1, y2[o]);
          if (p.mul) v = p.mul[o] * v;
          if (p.res) v = v + p.res[o];
          y2[o] = act_apply(act2, v);
        }
      }
    }
  }
  DBG_T(6);
  DBG_T(5);
}

#ifdef AIVC_TUNING  // per-workgroup phase timestamps (tools/phase_probe.py builds a copy with -DAIVC_TUNING; never in the product library)
extern "C" __attribute__((visibility("default"))) int aivc_dbg_dump(unsigned long long *host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(aivc_dbg_t), sizeof(unsigned long long) * (size_t)n);
}
#endif

template <int MODE, int WM, int WN, int TM, int TN, bool FUSE, bool FASTK, bool TAIL = false, bool GLDS = false, int PREC = 0>
static int launch_cfg2(const aivc_conv_params &p, hipStream_t s) {
  constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
  MfmaArgs a;
  a.p = p;
  a.M = MODE == AIVC_MODE_TCONV ? p.n * p.h_in * p.w_in : p.n * p.h_out * p.w_out;
  a.cin_magic = (uint32_t)((0x100000000ull + (uint64_t)p.c_in - 1) / (uint64_t)p.c_in);
  a.w_magic = p.w_in > 1 ? (uint32_t)(0x100000000ull / (uint64_t)p.w_in) : 0xFFFFFFFFu;
  a.h_magic = p.h_in > 1 ? (uint32_t)(0x100000000ull / (uint64_t)p.h_in) : 0xFFFFFFFFu;
  a.gx = (a.M + BM - 1) / BM;
  a.gy = (p.c_out + BN - 1) / BN;
  static const int tc_order = getenv("AIVC_TCONV_ORDER") ? atoi(getenv("AIVC_TCONV_ORDER")) : 1;  // tuning aid: 0 = plain dispatch order
  a.tc_order = tc_order;
  dim3 grid((unsigned)a.gx * (unsigned)a.gy * (MODE == AIVC_MODE_TCONV ? 4u : 1u), 1, 1);
  size_t lds = (size_t)(BM + (TAIL && TAIL_N > BN ? TAIL_N : BN)) * LDS_STRIDE * sizeof(float);
  if (GLDS && lds < (size_t)2 * (BM + BN) * BK * sizeof(float)) lds = (size_t)2 * (BM + BN) * BK * sizeof(float);
  if (PREC == 2 && lds < (size_t)2 * (BM * BK * 4 + BN * 192)) lds = (size_t)2 * (BM * BK * 4 + BN * 192);
  if (lds > 64 * 1024) {
    static LdsOptIn opt_in;  // per instantiation, per device
    if (!opt_in.raise(reinterpret_cast<const void *>(&conv_mfma_kernel<MODE, WM, WN, TM, TN, FUSE, FASTK, TAIL, GLDS, PREC>), lds))
      return check_launch("conv_mfma lds attribute");
  }
  hipLaunchKernelGGL((conv_mfma_kernel<MODE, WM, WN, TM, TN, FUSE, FASTK, TAIL, GLDS, PREC>), grid, dim3(256), lds, s, a);
  return check_launch("conv_mfma");
}

#ifndef AIVC_CONV_BF16X3
// LDS-DMA K loop: conv with c_in % 32 == 0 on the tiles it is instantiated for; per-lane BYTE offsets are 32 bits
static bool use_glds(const aivc_conv_params &p) {
  static const int off = getenv("AIVC_NO_GLDS") ? atoi(getenv("AIVC_NO_GLDS")) : 0;  // tuning aid: 1 = the register-staged loop everywhere, 2 = for transposed conv
  if (off == 1 || (off == 2 && p.mode == AIVC_MODE_TCONV) || (p.mode != AIVC_MODE_CONV && p.mode != AIVC_MODE_TCONV)) return false;
  if (p.c_in % BK != 0 && (p.mode != AIVC_MODE_CONV || off == 3)) return false;  // generic K: conv only (3 = tuning aid: off)
  return (uint64_t)p.n * p.h_in * p.w_in * p.c_in * 4ull < 0xFFFFFFFFull && (uint64_t)p.c_out * p.ksize * p.ksize * p.c_in * 4ull < 0xFFFFFFFFull;
}

template <int MODE, int WM, int WN, int TM, int TN, bool FUSE>
static int launch_cfg(const aivc_conv_params &p, hipStream_t s) {
  // every tile of the menu except 256x128 (96 KB of ring: one workgroup per CU)
  if constexpr ((MODE == AIVC_MODE_CONV || MODE == AIVC_MODE_TCONV) && 32 * WM * TM + 32 * WN * TN <= 320) {
    if (use_glds(p)) {
      if (p.c_in % BK == 0) return launch_cfg2<MODE, WM, WN, TM, TN, FUSE, true, false, true>(p, s);
      if constexpr (MODE == AIVC_MODE_CONV && !(WM == 4 && TM == 2)) return launch_cfg2<MODE, WM, WN, TM, TN, FUSE, false, false, true>(p, s);
    }
  }
  if (p.c_in % BK == 0) return launch_cfg2<MODE, WM, WN, TM, TN, FUSE, true>(p, s);
  if constexpr (WM == 4 && TM == 2) return AIVC_ERR_UNSUPPORTED;  // pick_tile never sends a generic reduction here
  else return launch_cfg2<MODE, WM, WN, TM, TN, FUSE, false>(p, s);
}

// tile menu {BM x BN}: id 0 = 128x128, 1 = 64x64, 2 = 256x64, 3 = 128x32, (4 = 256x128: retired), 5 = 64x128, 6 = 128x64
static int pick_tile_auto(const aivc_conv_params &p);
static int pick_tile(const aivc_conv_params &p) {
  // tuning aid: AIVC_FORCE_TILE=<id> overrides the choice when that tile can run the shape
  if (const char *e = getenv("AIVC_FORCE_TILE")) {
    const int t = atoi(e);
    const int bn = t == 0 || t == 5 ? 128 : (t == 3 ? 32 : 64);
    // (id 4 = 256x128 left the menu in round 3: never chosen since round 2, and it spilled; the 256-row tile is
    // instantiated for c_in % 32 == 0 only -- its generic loader spilled 700 bytes)
    if (t >= 0 && t <= 6 && t != 4 && (t != 2 || p.c_in % BK == 0) && (!p.gdn || bn == p.c_out)) return t;
  }
  return pick_tile_auto(p);
}
static int pick_tile_auto(const aivc_conv_params &p) {
  const bool t = p.mode == AIVC_MODE_TCONV;
  const long M = t ? (long)p.n * p.h_in * p.w_in : (long)p.n * p.h_out * p.w_out;
  const int z = t ? 4 : 1;
  const int co = p.c_out;
  auto blocks = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((co + bn - 1) / bn) * z; };
  const int taps = t ? (p.ksize * p.ksize + 3) / 4 : p.ksize * p.ksize;
  const long kred = (long)taps * p.c_in;
  // Measured on MI355X (tools/bench_conv.py, batch 8): every tile saturates at 105-117 TFLOP/s on
  // long reductions; what differs is how well short reductions / few blocks are hidden, where
  // the small 64x64 tile (6 workgroups per CU) wins.  Rules:
  //   c_out <= 64            -> 64x64 (also with fused GDN: BN == c_out)
  //   c_out  = 32            -> 128x32
  //   fused GDN, c_out = 128 -> 128x128, or 64x128 for the short reductions of a transposed 3x3
  //   otherwise score the candidates by efficiency class x block-count balance
  if (co <= 32) return 3;
  // attention gates (sigmoid): their per-output pass wants many small tiles (16 outputs per thread)
  if ((p.act1 == AIVC_ACT_SIGMOID || p.act2 == AIVC_ACT_SIGMOID) && !p.gdn) return 1;
  // c_out = 64: 256x64 (four waves stacked along M, 4 accumulators each) once there are >= ~1000 such tiles and
  // the reduction is long; else the small tile (r02 sweep, tools/_tile_sweep.sh: 109 vs 107, 94 vs 91 TFLOP/s)
  // 128x64 (id 6: 126 registers, four waves per SIMD) is ahead where the epilogue weighs most against a short or
  // loader-heavy reduction: the image layers (c_in of 4 / 8 / 12), 1x1 and stride-2 convs (r02: 4080 vs 4232 us,
  // 114 vs 119, 548 vs 582); transposed convs and the 3x3 stay on 256x64 / 64x64
  // Round 3, LDS-DMA loop (c_in % 32 == 0; tools/bench_conv.py with AIVC_FORCE_TILE, same box): 64x128 (three
  // workgroups per CU: 48 KB of ring, <= 172 registers) beats 128x128 (two) wherever BN = 128 fits -- 5x5 s2 64->128
  // + GDN 135.3 -> 137.8, 3x3 128->128 138.4 -> 140.3 (+ GDN 131.9 -> 133.5), transposed 5x5 128->128 122.6 -> 132.9
  // (+ GDN 114.6 -> 127.4), transposed 3x3 115.0 -> 118.5 TFLOP/s; for c_out = 64 the transposed 5x5 + GDN runs
  // 121.5 on 256x64, 125.1 on 128x64, 125.6 on 64x64.  AIVC_TILE_RULES_R2 restores the round-2 rules below.
  static const bool r2_rules = getenv("AIVC_TILE_RULES_R2") != nullptr;
  if (!r2_rules && p.c_in % BK == 0 && (p.mode == AIVC_MODE_CONV || t)) {
    // Round 4: few tiles (a single frame's 1/16-resolution layers: 8160 pixels = 128 tiles of 64x128 for 256 CUs): the 64x64
    // tile doubles the workgroup count -- 3x3 128->128 at 68x120, batch 1: 53 -> 90 TFLOP/s, transposed 5x5 94 -> 104, equal
    // from ~512 tiles on (tools/_ab_tiles_n4.sh at BATCH=1 / 4).  Not with a fused GDN (its tile must hold all channels).
    if (co % 128 == 0) return (!p.gdn && blocks(64, 128) <= 512) ? 1 : 5;
    if (co <= 64 && t) return 1;
  }
  if (co <= 64 && !t && M >= 65536 && (p.c_in % BK != 0 || p.ksize == 1 || p.stride == 2)) return 6;
  if (co <= 64) return (M >= 250000 && kred >= 96 && p.c_in % BK == 0) ? 2 : 1;
  if (p.gdn) return (t && p.ksize == 3) ? 5 : 0;  // BN must equal c_out = 128
  // c_out above 64 that is no multiple of 128 (models of other widths: 96, 144, 192 ...; round 6, bench.py --widths): the
  // score also counts the columns a tile pads -- 192 channels are 1.5 tiles of 128 (a quarter of the matrix work wasted)
  // but exactly three of 64 -- and the 64-column tiles with four accumulators per wave (256x64 / 128x64) and the 64x128
  // LDS-DMA tile join the candidates.
  auto score = [&](int bm, int bn, int slots, double base) {
    const long b = blocks(bm, bn);
    const long rounds = (b + slots - 1) / slots;
    const double cols = (double)co / (double)(((co + bn - 1) / bn) * bn);
    return base * cols * (double)b / (double)(rounds * slots);
  };
  // stand-alone (I)GDN launch (K = C: bound by its memory traffic): with 128 channels the 128-column tile reads the input
  // once as the GEMM operand instead of once per 64-column tile (round 6: the second launch of a Winograd-covered layer,
  // 4.0 -> see experiments/r06.md); AIVC_GDN_TILE overrides (tuning aid)
  if (p.mode == AIVC_MODE_GDN || p.mode == AIVC_MODE_IGDN) {
    static const int gdn_tile = getenv("AIVC_GDN_TILE") ? atoi(getenv("AIVC_GDN_TILE")) : -1;
    if (gdn_tile >= 0) return gdn_tile;
    return co == 128 ? 5 : 1;
  }
  double best = score(128, 128, 512, 0.80);  // 176 registers: two workgroups per CU
  int tile = 0;
  const double s1 = score(64, 64, 1536, kred <= 256 ? 0.85 : 0.74);
  if (s1 > best) best = s1, tile = 1;
  if ((t && p.ksize == 3) || (!r2_rules && p.c_in % BK == 0 && co % 128 != 0)) {
    const double s5 = score(64, 128, 1024, t && p.ksize == 3 ? 0.75 : 0.80);
    if (s5 > best) best = s5, tile = 5;
  }
  if (!r2_rules && co % 128 != 0) {
    const double s6 = score(128, 64, 1024, 0.76);
    if (s6 > best) best = s6, tile = 6;
    if (p.c_in % BK == 0) {
      const double s2 = score(256, 64, 512, 0.78);
      if (s2 > best) best = s2, tile = 2;
    }
  }
  return tile;
}

template <int MODE>
static int launch_mode(const aivc_conv_params &p, hipStream_t s) {
  const int tile = pick_tile(p);
  if constexpr (MODE != AIVC_MODE_GDN) {
    if (p.gdn) {
      switch (tile) {
        case 0: return launch_cfg<MODE, 2, 2, 2, 2, true>(p, s);
        case 1: return launch_cfg<MODE, 2, 2, 1, 1, true>(p, s);
        case 2: return launch_cfg<MODE, 4, 1, 2, 2, true>(p, s);
        case 5: return launch_cfg<MODE, 2, 2, 1, 2, true>(p, s);
        case 6: return launch_cfg<MODE, 2, 2, 2, 1, true>(p, s);
        default: return launch_cfg<MODE, 4, 1, 1, 1, true>(p, s);
      }
    }
  }
  switch (tile) {
    case 0: return launch_cfg<MODE, 2, 2, 2, 2, false>(p, s);
    case 1: return launch_cfg<MODE, 2, 2, 1, 1, false>(p, s);
    case 2: return launch_cfg<MODE, 4, 1, 2, 2, false>(p, s);
    case 5: return launch_cfg<MODE, 2, 2, 1, 2, false>(p, s);
    case 6: return launch_cfg<MODE, 2, 2, 2, 1, false>(p, s);
    default: return launch_cfg<MODE, 4, 1, 1, 1, false>(p, s);
  }
}

// fused 1x1 tail: a conv with c_out = 64 (one 128x64 tile owns every channel of its pixels), TAIL_N tail channels,
// bias on both, cheap activations
bool conv2d_mfma_tail_supported(const aivc_conv_params &p) {
  return p.mode == AIVC_MODE_CONV && !p.gdn && !p.mul && p.c_out == 64 && p.tail_c_out == TAIL_N && p.c_in % BK == 0 &&
         p.bias && p.tail_bias && p.tail_w && p.act1 != AIVC_ACT_SIGMOID && p.act2 != AIVC_ACT_SIGMOID &&
         conv2d_mfma_supported(p);
}

int conv2d_mfma_variant(const aivc_conv_params &p) {
  if (p.tail_c_out) return 190;
  const int mode = p.mode == AIVC_MODE_TCONV ? 1 : (p.mode == AIVC_MODE_CONV ? 0 : 2);
  return 100 + 10 * mode + pick_tile(p) + (p.gdn ? 50 : 0);
}

bool conv2d_mfma_supported(const aivc_conv_params &p) {
  if (p.gdn && p.c_out != 32 && p.c_out != 64 && p.c_out != 128) return false;
  // thin outputs (c_out of 3 / 6): N is padded to 32, still ~6x faster than the scalar kernel once
  // the reduction is long; tiny reductions stay scalar
  if (p.c_out < 16 && p.c_in * p.ksize * p.ksize < 256) return false;
  // 32-bit element offsets inside the kernel
  const uint64_t in_elems = (uint64_t)p.n * p.h_in * p.w_in * p.c_in;
  const uint64_t w_elems = (uint64_t)p.c_out * p.ksize * p.ksize * p.c_in;
  if (in_elems >= 0xFFFFFFFFull || w_elems >= 0xFFFFFFFFull) return false;
  if ((uint64_t)p.ksize * p.ksize * p.c_in >= 65536ull) return false;
  return true;
}

// byte size of the input tensor / of the weights as the LDS-DMA loader addresses them (32-bit byte offsets)
static bool glds_sizes_ok(const aivc_conv_params &p) {
  return (uint64_t)p.n * p.h_in * p.w_in * p.c_in * 4ull < 0xFFFFFFFFull;
}

int conv2d_mfma(const aivc_conv_params &p, hipStream_t s) {
  // a batch whose input exceeds the 4 GB the LDS-DMA loader can address goes out as several launches over
  // sub-batches (images are independent; same kernels, same results)
  if ((p.mode == AIVC_MODE_CONV || (p.mode == AIVC_MODE_TCONV && p.c_in % BK == 0)) && p.n > 1 && !glds_sizes_ok(p) &&
      !getenv("AIVC_NO_GLDS")) {
    const uint64_t per_image = (uint64_t)p.h_in * p.w_in * p.c_in * 4ull;
    int chunk = (int)(0xFFFFFFF0ull / per_image);
    if (chunk >= 1) {
      const int c_y = p.tail_c_out ? p.tail_c_out : p.c_out;
      for (int n0 = 0; n0 < p.n; n0 += chunk) {
        aivc_conv_params q = p;
        q.n = p.n - n0 < chunk ? p.n - n0 : chunk;
        const size_t in_off = (size_t)n0 * p.h_in * p.w_in * p.c_in, out_off = (size_t)n0 * p.h_out * p.w_out * c_y;
        q.x = p.x + in_off;
        q.y = p.y + out_off;
        if (p.res) q.res = p.res + out_off;
        if (p.mul) q.mul = p.mul + out_off;
        if (const int rc = conv2d_mfma(q, s)) return rc;
      }
      return AIVC_OK;
    }
  }
  if (p.tail_c_out) {
    if (!conv2d_mfma_tail_supported(p)) return AIVC_ERR_UNSUPPORTED;
    if (use_glds(p)) {
      static const int tail_tile = getenv("AIVC_TAIL_TILE") ? atoi(getenv("AIVC_TAIL_TILE")) : 6;  // tuning aid: 1 = 64x64
      if (tail_tile == 1) return launch_cfg2<AIVC_MODE_CONV, 2, 2, 1, 1, false, true, true, true>(p, s);
      return launch_cfg2<AIVC_MODE_CONV, 2, 2, 2, 1, false, true, true, true>(p, s);
    }
    return launch_cfg2<AIVC_MODE_CONV, 2, 2, 2, 1, false, true, true>(p, s);
  }
  switch (p.mode) {
    case AIVC_MODE_CONV: return launch_mode<AIVC_MODE_CONV>(p, s);
    case AIVC_MODE_TCONV: return launch_mode<AIVC_MODE_TCONV>(p, s);
    case AIVC_MODE_GDN:
    case AIVC_MODE_IGDN: return launch_mode<AIVC_MODE_GDN>(p, s);
    default: return AIVC_ERR_UNSUPPORTED;
  }
}

#else  // AIVC_CONV_BF16X3: this translation unit (conv_bf16x3.hip includes this file) holds the bf16x3 instantiations only

// The precision mode covers the layers that carry the FLOPs: conv / transposed conv with c_in % 32 == 0 and c_out of 64
// or a multiple of 128, with or without fused (I)GDN (its second GEMM stays fp32), no fused 1x1 tail.  Wave tile 64x64
// (128x128 / 256x64 workgroup tiles): the six products of a 64x64x16 slab are 24 MFMAs of 32 cycles against ~180 vector
// instructions of operand splitting -- smaller wave tiles are bound by the splitting.
bool conv2d_bf16x3_supported(const aivc_conv_params &p) {
  if (p.mode != AIVC_MODE_CONV && p.mode != AIVC_MODE_TCONV) return false;
  if (p.c_in % BK != 0) return false;
  // fused 1x1 tail (its GEMM stays fp32, like the fused GDN's): the bottleneck blocks' 3x3 64 -> 64 + 1x1 64 -> 128
  if (p.tail_c_out && (p.tail_c_out != TAIL_N || p.c_out != 64 || p.mode != AIVC_MODE_CONV || p.gdn || p.mul || !p.bias || !p.tail_bias)) return false;
  if (p.c_out != 64 && p.c_out % 128 != 0) return false;
  if (p.gdn && p.c_out != 64 && p.c_out != 128) return false;
  // short reductions (the 1x1 convs: two to four K tiles) are prologue / epilogue work on the mode's big tiles: they stay
  // on the fp32 kernels' small tiles (measured: 109-121 TFLOP/s fp32-equivalent against 125-133 there)
  const int taps = p.mode == AIVC_MODE_TCONV ? (p.ksize * p.ksize + 3) / 4 : p.ksize * p.ksize;
  if (taps * p.c_in < 512) return false;
  if ((uint64_t)p.c_out * p.ksize * p.ksize * p.c_in * 4ull >= 0xFFFFFFFFull || (uint64_t)p.ksize * p.ksize * p.c_in >= 65536ull) return false;
  return (uint64_t)p.h_in * p.w_in * p.c_in * 4ull < 0xFFFFFFF0ull;  // one image inside the loader's 32-bit byte offsets
}

template <int MODE, int PREC>
static int launch_bf16x3_prec(const aivc_conv_params &p, hipStream_t s) {
  if (p.c_out == 64) return p.gdn ? launch_cfg2<MODE, 4, 1, 2, 2, true, true, false, true, PREC>(p, s)
                                  : launch_cfg2<MODE, 4, 1, 2, 2, false, true, false, true, PREC>(p, s);
  return p.gdn ? launch_cfg2<MODE, 2, 2, 2, 2, true, true, false, true, PREC>(p, s)
               : launch_cfg2<MODE, 2, 2, 2, 2, false, true, false, true, PREC>(p, s);
}
// Tile of a launch of the mode (the ids of aivc_conv2d_variant: 0 = 128x128, 2 = 256x64, 5 = 64x128, 6 = 128x64).  Weights
// split in the K loop: wave tile 64x64 (the split is 44 vector instructions per fragment: smaller wave tiles are bound
// by it).  Weights split ahead (w_bf16x3): measured per layer class on the bench's shapes (tools/bf16x3_probe.py,
// TFLOP/s fp32-equivalent, in-loop | 64x64 wave tile | 32x64 wave tile): conv to 128 channels 163-181 | 174-204 | 161-182,
// conv to 64 160 | 162 | 173, transposed to 128 153 | 153 | 162, transposed to 64 149 | 139 | 151 (the 256x64 tile's ring
// grows to 88 KB with the three weight planes: one workgroup per CU).
int conv2d_bf16x3_tile(const aivc_conv_params &p) {
  static const int force = getenv("AIVC_BF16X3_TILE") ? atoi(getenv("AIVC_BF16X3_TILE")) : 0;  // tuning aid: 1 = wave tile 64x64 everywhere
  const bool ahead = p.w_bf16x3 != nullptr && (uint64_t)p.c_out * p.ksize * p.ksize * p.c_in * 6ull < 0xFFFFFFFFull;
  if (p.tail_c_out) return 6;  // fused tail: 128x64 either way (64 rows of 128 tail channels per wave would not fit the registers)
  if (!ahead || force == 1) return p.c_out == 64 ? 2 : 0;
  if (p.c_out == 64) return 6;
  return p.mode == AIVC_MODE_TCONV ? 5 : 0;
}

template <int MODE>
static int launch_bf16x3(const aivc_conv_params &p, hipStream_t s) {
  // weights split ahead of the launch (aivc_split_weights_bf16x3) or by the K loop: the same terms, the same bits
  const bool ahead = p.w_bf16x3 != nullptr && (uint64_t)p.c_out * p.ksize * p.ksize * p.c_in * 6ull < 0xFFFFFFFFull;
  if constexpr (MODE == AIVC_MODE_CONV) {
    if (p.tail_c_out) return ahead ? launch_cfg2<MODE, 4, 1, 1, 2, false, true, true, true, 2>(p, s)
                                   : launch_cfg2<MODE, 4, 1, 1, 2, false, true, true, true, 1>(p, s);
  }
  if (!ahead) return launch_bf16x3_prec<MODE, 1>(p, s);
  switch (conv2d_bf16x3_tile(p)) {
    case 6: return p.gdn ? launch_cfg2<MODE, 4, 1, 1, 2, true, true, false, true, 2>(p, s)
                         : launch_cfg2<MODE, 4, 1, 1, 2, false, true, false, true, 2>(p, s);
    case 5: return p.gdn ? launch_cfg2<MODE, 2, 2, 1, 2, true, true, false, true, 2>(p, s)
                         : launch_cfg2<MODE, 2, 2, 1, 2, false, true, false, true, 2>(p, s);
    default: return launch_bf16x3_prec<MODE, 2>(p, s);
  }
}

__global__ __launch_bounds__(256) void split_weights_kernel(const float *__restrict__ w, size_t pairs, int k_total, uint32_t *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // pair (k, k + 1) of one row
  if (i >= pairs) return;
  const size_t half = (size_t)k_total / 2, co = i / half;
  const int k = 2 * (int)(i - co * half);
  const float2 x = *reinterpret_cast<const float2 *>(w + 2 * i);
  uint32_t h, m, l;
  bf16x3_split2(x.x, x.y, h, m, l);
  uint32_t *dst = out + ((co * (size_t)(k_total / 32) + (size_t)(k / 32)) * 3) * 16 + (size_t)((k % 32) / 2);
  dst[0] = h;
  dst[16] = m;
  dst[32] = l;
}

int split_weights_bf16x3(const float *w, int c_out, int k_total, void *out, hipStream_t s) {
  const size_t pairs = (size_t)c_out * (size_t)k_total / 2;
  hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, s, w, pairs, k_total, reinterpret_cast<uint32_t *>(out));
  return check_launch("split_weights_bf16x3");
}

int conv2d_bf16x3(const aivc_conv_params &p, hipStream_t s) {
  if (!conv2d_bf16x3_supported(p)) return AIVC_ERR_UNSUPPORTED;
  // a batch beyond the 4 GB the loader addresses goes out as sub-batches (images are independent)
  const uint64_t per_image = (uint64_t)p.h_in * p.w_in * p.c_in * 4ull;
  const int chunk = (int)(0xFFFFFFF0ull / per_image);
  for (int n0 = 0; n0 < p.n; n0 += chunk) {
    aivc_conv_params q = p;
    q.n = p.n - n0 < chunk ? p.n - n0 : chunk;
    const size_t in_off = (size_t)n0 * p.h_in * p.w_in * p.c_in;
    const size_t out_off = (size_t)n0 * p.h_out * p.w_out * (p.tail_c_out ? p.tail_c_out : p.c_out);  // (fused tail: y and res are the tail's)
    q.x = p.x + in_off;
    q.y = p.y + out_off;
    if (p.res) q.res = p.res + out_off;
    if (p.mul) q.mul = p.mul + out_off;
    const int rc = p.mode == AIVC_MODE_CONV ? launch_bf16x3<AIVC_MODE_CONV>(q, s) : launch_bf16x3<AIVC_MODE_TCONV>(q, s);
    if (rc) return rc;
  }
  return AIVC_OK;
}

#endif  // AIVC_CONV_BF16X3

}  // namespace aivc
