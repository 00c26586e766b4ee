// conv_wino.hip -- version 2 of the fp32 arithmetic contract (AIVC_PREC_FP32_WINO, include/aivc_hip.h): the stride-1 3x3
// convolutions with c_in % 32 == 0 and c_out % 64 == 0 as Winograd F(2x2, 3x3) on the gfx950 matrix cores.
//
//   why            the fp32 matrix pipe is the scarce unit of this part (157 TFLOP/s, 1/16 of the bf16 rate) and the codec's
//                  step is 0.75 of it end to end: what is left under the tap chain is ~10 %.  F(2x2, 3x3) issues 16
//                  multiplications per 2x2 output pixels and channel pair instead of 36.
//   first version  (round 6, experiments/r06.md): position-outer K loop, the four input pixels of a position fetched per
//                  position and K-tile -- 12 KB of L2 -> LDS traffic per output pixel at 2.25x the tap kernel's pace
//                  (~11 TB/s asked of the L2): 0.56 of the matrix peak, 1.2x the tap kernel.  This version fetches every
//                  input pixel of a block ONCE per channel chunk.
//   work split     a workgroup = a block of 8 x 8 output tiles (16 x 16 pixels) x 64 output channels, 8 waves: (4 x 8 tiles)
//                  x 32 channels x 8 of the 16 positions each -- 8 accumulator blocks (128 registers) per wave, two waves per
//                  SIMD.  The reduction runs over chunks of 8 input channels (one octet of AIVC_K_ORDER = four
//                  v_mfma_f32_32x32x2_f32 steps per position); M_p = a fixed-order fmaf chain over ci per position, as the
//                  contract says.  After the last chunk a wave folds its 8 positions into the partial sums S0 / S1 of the
//                  four outputs of every tile, the two waves of a pair exchange halves through LDS and each finishes one
//                  output row (a = 0 / a = 1) of the tiles.
//   per chunk      raw patch (18 x 18 pixels x 8 channels, 10 KB: replicate-clamped on the global side of the LDS-DMA, stored
//                  as four parity planes so that the transform's reads are contiguous), U image (16 positions x 64 channels x
//                  8, 32 KB: aivc_winograd_weights lays it out as it is staged, one contiguous copy), V (16 positions x 64
//                  tiles x 8, by a cooperative transform: 8 ds_read_b128 + 32 v_fma + 4 ds_write_b128 per thread).
//                  Everything double-buffered (158 KB of LDS, one workgroup per CU) and ONE barrier per chunk: in chunk c a
//                  wave issues the DMAs of raw(c + 2) and U(c + 1), transforms raw(c + 1) into the other V buffer and
//                  multiplies chunk c -- the two waves of a SIMD in opposite order, so that one's transform runs under the
//                  other's MFMAs.
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"

namespace aivc {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct WinoArgs {
  aivc_conv_params p;
  int TH, TW;    // output tiles per image column / row
  int nby, nbx;  // blocks of 8 x 8 tiles per image
  int gy;        // blocks of 64 output channels
  int total;     // blocks in all: n * nby * nbx * gy
  int poly;      // 1: the 5x5 stride-2 convolution as four stride-1 3x3 convolutions of the input's polyphase components
  int cpp_shift; // log2 of the chunks per phase (c_in / 8)
};

__device__ __forceinline__ void wino_glds16(const float *base, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds_dst) : "memory", "m0");
}

constexpr int WINO_RAW_STAGE = 11 * 1024;  // 648 slots of 16 bytes (4 parity planes x 81 pixels x 2 channel quads), 11 DMA instructions
constexpr int WINO_U_STAGE = 32 * 1024;    // [16 positions][2 quads][64 channels][4 floats]
constexpr int WINO_V_QUAD = 1152;          // [64 tiles][4 floats] + 128: the second quad starts 32 banks further
constexpr int WINO_V_POS = 2 * WINO_V_QUAD;
constexpr int WINO_V_STAGE = 16 * WINO_V_POS;
constexpr int WINO_LDS = 2 * (WINO_RAW_STAGE + WINO_U_STAGE + WINO_V_STAGE);

// 4 KB of zeros: the source of the patch pixels outside the image in the transposed form (zero extension; an LDS-DMA cannot fill)
__device__ __attribute__((aligned(4096))) float wino_zeros[1024];

// MODE 0: stride-1 3x3.  MODE 1 (POLY): the polyphase form of the 5x5 stride-2 convolutions.  MODE 2 (TC): the 5x5 stride-2
// TRANSPOSED convolutions: each of the four output parity classes is a stride-1 3x3 correlation of the (zero-extended) input with
// the class's taps padded with zeros -- a block of the list is (pixel block, class, 64 output channels), the outputs of class
// (pyc, pxc) land on pixels (2 y + pyc, 2 x + pxc); positions with i == 0 (pyc = 1) or j == 0 (pxc = 1) have U = 0 and are not issued.
// Separate instantiations: the stride-1 3x3 kernel pays nothing for the others.
template <int MODE>
__global__ __launch_bounds__(512) void conv_wino_kernel(WinoArgs a) {
  constexpr bool POLY = MODE == 1, TC = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) char wsmem[];
  char *raw = wsmem, *Us = wsmem + 2 * WINO_RAW_STAGE, *Vs = Us + 2 * WINO_U_STAGE;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)wsmem;

  const aivc_conv_params &p = a.p;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ph = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;  // waves w and w + 4 (one SIMD) = the two position halves of a sub-tile
  // H x W: the pixel grid the blocks walk = the OUTPUT's (stride 1: also the input's); Hi x Wi: the input's.  Polyphase form
  // (a.poly, 5x5 stride 2, include/aivc_hip.h): the reduction runs over 4 phases x c_in channels, chunk c belongs to phase
  // c >> cpp_shift = 2 py + px, whose patch pixel (y, x) is input pixel (clamp(2 y + py), clamp(2 x + px)) -- the replicate
  // padding of the ORIGINAL image -- and whose 3x3 kernel is the phase's taps padded with zeros: positions with i == 3 (py = 1)
  // or j == 3 (px = 1) have U = 0 and are not issued (49 instead of 64 of the 4 x 16 position products).
  const int H = TC ? p.h_in : p.h_out, W = TC ? p.w_in : p.w_out, Hi = p.h_in, Wi = p.w_in, Cin = p.c_in, Cout = p.c_out;
  const int gyc = TC ? 4 * a.gy : a.gy;  // entries of the block list per pixel block: channel blocks (x 4 classes)
  constexpr int poly = POLY ? 1 : 0;
  const int cpp_shift = a.cpp_shift, cpp_mask = (1 << a.cpp_shift) - 1;

  // Persistent workgroups (one per CU: a.nwg of them) walk the blocks.  The dispatcher deals consecutive workgroup ids
  // round-robin to the 8 XCDs (each with a private L2): workgroup b sits on XCD b & 7 and takes blocks of that XCD's
  // contiguous eighth of the block list, interleaved with the other workgroups of the XCD -- they advance through one region
  // together; the channel blocks of a pixel block are neighbours in the list (same raw patch).
  const uint32_t total = (uint32_t)a.total, nwg = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const uint32_t per_xcd = (total + 7u) >> 3, wg_per_xcd = (nwg + 7u - xcd) >> 3;  // workgroups on this XCD
  const uint32_t x_lo = xcd * per_xcd, x_hi = min(x_lo + per_xcd, total);
  auto block_of = [&](uint32_t k) -> uint32_t { return x_lo + slot + k * wg_per_xcd; };  // k-th block of this workgroup
  const uint32_t n_mine = x_lo + slot < x_hi ? (x_hi - x_lo - slot + wg_per_xcd - 1u) / wg_per_xcd : 0u;
  if (n_mine == 0u) return;
  // descriptor of a block: what the loader needs (image base, chunk images of its channel block, per-lane patch offsets) and
  // what the epilogue needs (coordinates)
  struct Desc {
    const float *xbase, *ubase;
    uint32_t r_off[2];  // stride-1 form: per-lane byte offsets of the block's raw patch (the polyphase form keeps those of the phase being issued)
    int img, byi, bxi, cb;  // cb: channel block (transposed form: class * gy + channel block)
  };
  // raw: instruction k of 11 writes slots 64 k .. 64 k + 63; wave w issues k = w and, for w < 3, k = w + 8.  slot = (plane *
  // 81 + hy * 9 + hx) * 2 + quad with plane = (py & 1) * 2 + (px & 1), hy = py >> 1, hx = px >> 1 for patch pixel (py, px)
  int s_quad[2], s_py[2], s_px[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    int sl = 64 * (wave + 8 * k) + lane;
    sl = sl < 648 ? sl : 647;  // (pad lanes of the last instruction fetch the last slot again)
    const int q2 = sl >> 1, plane = q2 / 81, rem = q2 - plane * 81, hy = rem / 9, hx = rem - hy * 9;
    s_quad[k] = sl & 1;
    s_py[k] = 2 * hy + (plane >> 1);
    s_px[k] = 2 * hx + (plane & 1);
  }
  const int nch = poly ? 4 * (Cin >> 3) : (Cin >> 3);  // chunks of 8 (virtual) input channels per block
  auto make_desc = [&](uint32_t blk) {
    Desc d;
    d.cb = (int)(blk % (uint32_t)gyc);
    uint32_t rest = blk / (uint32_t)gyc;
    d.bxi = (int)(rest % (uint32_t)a.nbx);
    rest /= (uint32_t)a.nbx;
    d.byi = (int)(rest % (uint32_t)a.nby);
    d.img = (int)(rest / (uint32_t)a.nby);
    d.xbase = p.x + (size_t)d.img * (size_t)Hi * Wi * Cin;                       // this image (32-bit byte offsets inside it)
    d.ubase = p.w_wino + (size_t)d.cb * (size_t)nch * (WINO_U_STAGE / 4);        // this channel block's chunk images
    if constexpr (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int iy = max(min(16 * d.byi - 1 + s_py[k], Hi - 1), 0), ix = max(min(16 * d.bxi - 1 + s_px[k], Wi - 1), 0);
        d.r_off[k] = ((uint32_t)(iy * Wi + ix) * (uint32_t)Cin + (uint32_t)(4 * s_quad[k])) * 4u;
      }
    }
    return d;
  };
  // per-lane byte offsets of the raw patch of block d_ in phase ph2 = 2 py + px (stride 1: phase 0, step 1): recomputed when the
  // issue stream enters a phase (every c_in / 8 chunks), kept in two registers in between
  uint32_t iss_off[2];
  auto patch_offsets = [&](const Desc &d_, int ph2) {
    const int st = poly ? 2 : 1, py = ph2 >> 1, px = ph2 & 1;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int iy = max(min(st * (16 * d_.byi - 1 + s_py[k]) + py, Hi - 1), 0), ix = max(min(st * (16 * d_.bxi - 1 + s_px[k]) + px, Wi - 1), 0);
      iss_off[k] = ((uint32_t)(iy * Wi + ix) * (uint32_t)Cin + (uint32_t)(4 * s_quad[k])) * 4u;
    }
  };
  // transposed form: per-lane 64-bit source addresses of the chunk being issued (a pixel outside the image reads zeros), advanced
  // by the 32 bytes of a chunk after every issue, recomputed when the issue stream enters a block
  uint64_t iss_addr[2];
  auto tc_addresses = [&](const Desc &d_) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int iy = 16 * d_.byi - 1 + s_py[k], ix = 16 * d_.bxi - 1 + s_px[k];
      const bool in = iy >= 0 && iy < Hi && ix >= 0 && ix < Wi;
      const uint64_t a_in = (uint64_t)(uintptr_t)d_.xbase + ((uint64_t)((uint32_t)(iy * Wi + ix)) * (uint32_t)Cin + (uint32_t)(4 * s_quad[k])) * 4u;
      const uint64_t a_z = (uint64_t)(uintptr_t)wino_zeros + (uint32_t)(16 * s_quad[k]);
      iss_addr[k] = in ? a_in : a_z;
    }
  };
  Desc cur = make_desc(block_of(0)), nxt = n_mine > 1u ? make_desc(block_of(1)) : cur;

  const uint32_t r_dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)wave * 1024u);
  const uint32_t u_dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(2 * WINO_RAW_STAGE) + (uint32_t)wave * 4096u);
  const uint32_t u_off = (uint32_t)(wave * 4096 + lane * 16);
  auto uniform_ptr = [](const float *q) -> const float * {  // (wave-uniform by construction: tell the compiler, the DMA wants a scalar base)
    const uint64_t v = (uint64_t)(uintptr_t)q;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const float *>((uintptr_t)(((uint64_t)hi << 32) | lo));
  };
  auto issue_raw = [&](const Desc &d_, int c) {  // chunk c of block d_ -> raw stage c & 1 (chunks are issued in order)
    const uint32_t d = r_dst + (uint32_t)((c & 1) * WINO_RAW_STAGE);
    if constexpr (TC) {
      if (c == 0) tc_addresses(d_);
      const uint32_t du = __builtin_amdgcn_readfirstlane(d);
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(iss_addr[0]), "s"(du) : "memory", "m0");
      if (wave < 3) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(iss_addr[1]), "s"(du + 8192u) : "memory", "m0");
      iss_addr[0] += 32u;
      iss_addr[1] += 32u;
    } else if constexpr (POLY) {
      if ((c & cpp_mask) == 0) patch_offsets(d_, c >> cpp_shift);  // the issue stream enters a phase
      const float *src = uniform_ptr(d_.xbase + 8 * (c & cpp_mask));
      const uint32_t du = __builtin_amdgcn_readfirstlane(d);  // (under register pressure the compiler parks the uniform in a VGPR)
      wino_glds16(src, iss_off[0], du);
      if (wave < 3) wino_glds16(src, iss_off[1], du + 8192u);
    } else {
      const float *src = d_.xbase + 8 * c;
      wino_glds16(src, d_.r_off[0], d);
      if (wave < 3) wino_glds16(src, d_.r_off[1], d + 8192u);
    }
  };
  // zero-by-construction positions (polyphase / transposed forms): is position (i, j) of phase / class `pc` zero?
  auto zero_pos = [&](int pc, int i, int j) -> bool {
    if constexpr (POLY) return ((pc >> 1) && i == 3) || ((pc & 1) && j == 3);
    if constexpr (TC) return ((pc >> 1) && i == 0) || ((pc & 1) && j == 0);
    return false;
  };
  auto phase_or_class = [&](const Desc &d_, int c) -> int { return POLY ? c >> cpp_shift : (TC ? d_.cb / a.gy : 0); };
  auto issue_u = [&](const Desc &d_, int c) {  // chunk c of block d_ -> U stage c & 1: a straight copy of the chunk image
    const float *src = MODE != 0 ? uniform_ptr(d_.ubase + (size_t)c * (WINO_U_STAGE / 4)) : d_.ubase + (size_t)c * (WINO_U_STAGE / 4);
    const uint32_t d = MODE != 0 ? __builtin_amdgcn_readfirstlane(u_dst + (uint32_t)((c & 1) * WINO_U_STAGE)) : u_dst + (uint32_t)((c & 1) * WINO_U_STAGE);
    // four instructions off ONE M0: the instruction offset advances the global and the LDS address alike (the chunk image is
    // contiguous on both sides).  This wave's four pieces are positions 2 wave (two channel quads) and 2 wave + 1: a position whose
    // U is zero by construction is neither multiplied nor fetched
    bool lo = true, hi = true;
    if constexpr (MODE != 0) {
      const int pc = phase_or_class(d_, c), i = wave >> 1, j = 2 * (wave & 1);
      lo = !zero_pos(pc, i, j);
      hi = !zero_pos(pc, i, j + 1);
    }
    if (lo && hi) {
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %0, %1\n\t"
                   "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                   "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
                   "global_load_lds_dwordx4 %0, %1 offset:3072"
                   : : "v"(u_off), "s"(src), "s"(d) : "memory", "m0");
    } else if (lo) {
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %0, %1\n\t"
                   "global_load_lds_dwordx4 %0, %1 offset:1024"
                   : : "v"(u_off), "s"(src), "s"(d) : "memory", "m0");
    } else if (hi) {
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
                   "global_load_lds_dwordx4 %0, %1 offset:3072"
                   : : "v"(u_off), "s"(src), "s"(d) : "memory", "m0");
    }
  };

  // ---- transform plan: thread = (tile, channel quad, row i of the 4 x 4 positions) -----------------------------------------
  // Which (tile, channel quad) unit of the wave's 4 tile rows x 8 tiles x 2 quads a lane transforms is chosen for the LDS
  // (MI355X_MICROARCH.md, LDS):  ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (and
  // + 32), banks of 4 bytes mod 64 -- a group reads ONE tile row of a parity plane: 16 consecutive slots = 256 contiguous
  // bytes, every bank once;  ds_write_b128 is served 8 consecutive lanes at a time, banks mod 32 -- such an octet holds four
  // lanes of each of two read groups: the row of the first group hands it tiles 0-3 (or 4-7) of one quad, the row of the
  // second group tiles 4-7 (0-3) of the same quad: 2 x 64 bytes that differ in address bit 6.  With (quad, tx, ty) read off
  // the lane id directly a third of the kernel's LDS cycles were bank conflicts (profiles/r06_pmc_wino_before_lane_map.json).
  const int l32 = lane & 31;
  const int t_grp = (l32 < 4 || (l32 >= 12 && l32 < 16) || (l32 >= 20 && l32 < 28)) ? 0 : 1;
  const int t_pos = t_grp == 0 ? (l32 < 4 ? l32 : (l32 < 16 ? l32 - 8 : l32 - 12)) : (l32 < 12 ? l32 - 4 : (l32 < 20 ? l32 - 8 : l32 - 16));
  const int t_quad = t_pos >> 3, t_tx = (t_pos & 7) ^ (t_grp << 2), t_ty = ((tid >> 6) & 1) * 4 + (lane >> 5) * 2 + t_grp;
  const int t_i = __builtin_amdgcn_readfirstlane(tid >> 7);
  const int t_ra = t_i == 0 ? 0 : (t_i == 2 ? 2 : 1), t_rb = t_i == 3 ? 3 : (t_i == 2 ? 1 : 2);  // rows A[i], B[i] of the patch
  const float t_si = t_i == 1 ? 1.0f : -1.0f;
  const int t_base = ((t_ty * 9 + t_tx) * 2 + t_quad) * 16;
  // byte offset of patch pixel (2 ty + r, 2 tx + c) relative to t_base
  auto pix_off = [](int r, int c) { return (((r & 1) * 2 + (c & 1)) * 81 + (r >> 1) * 9 + (c >> 1)) * 32; };
  int t_oa[4], t_ob[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    t_oa[c] = t_base + pix_off(t_ra, c);
    t_ob[c] = t_base + pix_off(t_rb, c);
  }
  const int t_vdst = (4 * t_i) * WINO_V_POS + t_quad * WINO_V_QUAD + (t_ty * 8 + t_tx) * 16;
  // the transform of one chunk in 8 steps (interleaved with the 8 positions of the multiply): steps 0-3 combine the rows of
  // patch column cc = step (R[cc] = d[A_i][cc] + S_i d[B_i][cc]), steps 4-7 write V_j = R[A_j] + S_j R[B_j], j = step - 4
  float4 R[4];
  auto comb = [](const float4 &u, const float4 &v, float sgn) {
    float4 o;
    o.x = __builtin_fmaf(sgn, v.x, u.x);
    o.y = __builtin_fmaf(sgn, v.y, u.y);
    o.z = __builtin_fmaf(sgn, v.z, u.z);
    o.w = __builtin_fmaf(sgn, v.w, u.w);
    return o;
  };
  auto transform_step = [&](auto STAGE, auto STEP) {  // raw stage -> V stage of the same parity (a compile-time constant: the
    // chunk loop is unrolled by two, so every LDS access of the loop is a per-thread base + an immediate)
    constexpr int st = decltype(STEP)::value, stage = decltype(STAGE)::value;
    const char *rs = raw + stage * WINO_RAW_STAGE;
    char *vd = Vs + stage * WINO_V_STAGE + t_vdst;
    if constexpr (st < 4) {
      const float4 xa = *reinterpret_cast<const float4 *>(rs + t_oa[st]), xb = *reinterpret_cast<const float4 *>(rs + t_ob[st]);
      R[st] = comb(xa, xb, t_si);
    } else if constexpr (st == 4) {
      *reinterpret_cast<float4 *>(vd) = comb(R[0], R[2], -1.0f);
    } else if constexpr (st == 5) {
      *reinterpret_cast<float4 *>(vd + WINO_V_POS) = comb(R[1], R[2], 1.0f);
    } else if constexpr (st == 6) {
      *reinterpret_cast<float4 *>(vd + 2 * WINO_V_POS) = comb(R[2], R[1], -1.0f);
    } else {
      *reinterpret_cast<float4 *>(vd + 3 * WINO_V_POS) = comb(R[1], R[3], -1.0f);
    }
  };
  auto transform0 = [&]() {  // chunk 0 of the workgroup's first block
    using std::integral_constant;
    using Z = integral_constant<int, 0>;
    transform_step(Z{}, integral_constant<int, 0>{});
    transform_step(Z{}, integral_constant<int, 1>{});
    transform_step(Z{}, integral_constant<int, 2>{});
    transform_step(Z{}, integral_constant<int, 3>{});
    transform_step(Z{}, integral_constant<int, 4>{});
    transform_step(Z{}, integral_constant<int, 5>{});
    transform_step(Z{}, integral_constant<int, 6>{});
    transform_step(Z{}, integral_constant<int, 7>{});
  };

  // ---- multiply plan -------------------------------------------------------------------------------------------------------
  floatx16 acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;
  const int hh = lane >> 5, l31 = lane & 31;
  const int a_rd = (8 * ph) * WINO_V_POS + hh * WINO_V_QUAD + (32 * wm + l31) * 16;
  const int b_rd = (8 * ph) * 2048 + hh * 1024 + (32 * wn + l31) * 16;
  // chunk c multiplied (8 positions x 4 MFMAs) with the transform of chunk c + 1 dealt out between the positions: one
  // instruction stream per wave in which matrix and vector / LDS work alternate, so that the two waves of a SIMD fill each
  // other's gaps without any phase arrangement
  // FIRST: the block's first chunk starts its accumulators from the inline constant 0 (no clearing pass after the fold)
  // pm: bit q set = position q of this wave's half is issued (polyphase form: positions whose U is zero by construction are not)
  // tm: bit s set = step s of the NEXT chunk's transform is run (0: no transform rides along)
  auto multiply = [&](auto STAGE, auto FIRST, uint32_t tm, uint32_t pm) {
    constexpr int stage = decltype(STAGE)::value;
    constexpr bool first = decltype(FIRST)::value;
    using NEXT = std::integral_constant<int, 1 - stage>;
    const char *va = Vs + stage * WINO_V_STAGE + a_rd, *ub = Us + stage * WINO_U_STAGE + b_rd;
    // positions in pairs: the MFMAs of two accumulators alternate (a dependent MFMA waits for its predecessor's last pass),
    // the fragments of the next pair are read while this pair multiplies, the transform steps follow the pair's MFMAs (the
    // first MFMAs behind the chunk's barrier then wait for two LDS round trips only)
    float4 af[2][2], bf[2][2];
    auto rd = [&](int set, int q) {
      af[set][0] = *reinterpret_cast<const float4 *>(va + q * WINO_V_POS);
      bf[set][0] = *reinterpret_cast<const float4 *>(ub + q * 2048);
      af[set][1] = *reinterpret_cast<const float4 *>(va + (q + 1) * WINO_V_POS);
      bf[set][1] = *reinterpret_cast<const float4 *>(ub + (q + 1) * 2048);
    };
    rd(0, 0);
    auto pair = [&](auto Q) {
      constexpr int q = decltype(Q)::value, set = (q >> 1) & 1;
      using std::integral_constant;
      if constexpr (q + 2 < 8) {
        if (MODE == 0 || (first && !TC) || (pm & (0xCu << q))) rd(1 - set, q + 2);
      }
      __builtin_amdgcn_sched_barrier(0);
      const float4 x0 = af[set][0], y0 = bf[set][0], x1 = af[set][1], y1 = bf[set][1];
#ifdef WINO_EXP_NOMFMA
      acc[q][0] += x0.x * y0.x + x0.y * y0.y + x0.z * y0.z + x0.w * y0.w;
      acc[q + 1][0] += x1.x * y1.x + x1.y * y1.y + x1.z * y1.z + x1.w * y1.w;
#else
      if constexpr (first && TC) {
        // the block's first chunk: a position the class does not issue gets a cleared accumulator (one MFMA of zeros), the others
        // start from the inline zero
        const floatx16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if ((pm >> q) & 1u) {
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.x, y0.x, zero, 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.y, y0.y, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.z, y0.z, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.w, y0.w, acc[q], 0, 0, 0);
        } else {
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(0.0f, 0.0f, zero, 0, 0, 0);
        }
        if ((pm >> (q + 1)) & 1u) {
          acc[q + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.x, y1.x, zero, 0, 0, 0);
          acc[q + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.y, y1.y, acc[q + 1], 0, 0, 0);
          acc[q + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.z, y1.z, acc[q + 1], 0, 0, 0);
          acc[q + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.w, y1.w, acc[q + 1], 0, 0, 0);
        } else {
          acc[q + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(0.0f, 0.0f, zero, 0, 0, 0);
        }
      } else if constexpr (first) {
        const floatx16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.x, y0.x, zero, 0, 0, 0);
        acc[q + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.x, y1.x, zero, 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.y, y0.y, acc[q], 0, 0, 0);
        acc[q + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.y, y1.y, acc[q + 1], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.z, y0.z, acc[q], 0, 0, 0);
        acc[q + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.z, y1.z, acc[q + 1], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.w, y0.w, acc[q], 0, 0, 0);
        acc[q + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.w, y1.w, acc[q + 1], 0, 0, 0);
      } else {
        // (wave-uniform tests; the four MFMAs of a position in a row: alternating two accumulators measured the same, experiments/r06.md 2)
        if (MODE == 0 || ((pm >> q) & 1u)) {
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.x, y0.x, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.y, y0.y, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.z, y0.z, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.w, y0.w, acc[q], 0, 0, 0);
        }
        if (MODE == 0 || ((pm >> (q + 1)) & 1u)) {
          acc[q + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.x, y1.x, acc[q + 1], 0, 0, 0);
          acc[q + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.y, y1.y, acc[q + 1], 0, 0, 0);
          acc[q + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.z, y1.z, acc[q + 1], 0, 0, 0);
          acc[q + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.w, y1.w, acc[q + 1], 0, 0, 0);
        }
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
#ifndef WINO_EXP_NOXFORM
      if ((tm >> q) & 1u) transform_step(NEXT{}, integral_constant<int, q>{});
      if ((tm >> (q + 1)) & 1u) transform_step(NEXT{}, integral_constant<int, q + 1>{});
#endif
      __builtin_amdgcn_sched_barrier(0);
    };
    using std::integral_constant;
    pair(integral_constant<int, 0>{});
    pair(integral_constant<int, 2>{});
    pair(integral_constant<int, 4>{});
    pair(integral_constant<int, 6>{});
  };

  // ---- the reduction, pipelined ACROSS blocks: chunk indices run on through the workgroup's blocks (c_in / 8 is even, so
  // the stage parity of a chunk is its index inside its block) -------------------------------------------------------------
  auto issue_raw_at = [&](uint32_t kb, int c) {  // chunk c of this workgroup's block kb, c possibly beyond the block's chunks
    if (c < nch) issue_raw(cur, c);
    else if (kb + 1u < n_mine) issue_raw(nxt, c - nch);
  };
  auto issue_u_at = [&](uint32_t kb, int c) {
    if (c < nch) issue_u(cur, c);
    else if (kb + 1u < n_mine) issue_u(nxt, c - nch);
  };
  issue_raw(cur, 0);
  issue_u(cur, 0);
  issue_raw_at(0u, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  transform0();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const int act1 = p.act1, act2 = p.act2;
  auto act_cheap = [](int act, float v) {  // NONE / LEAKY / RELU of act_apply() without branches
    const float neg = act == AIVC_ACT_LEAKY ? v * 0.01f : (act == AIVC_ACT_RELU ? 0.0f : v);
    return v > 0.0f ? v : neg;
  };
  for (uint32_t kb = 0; kb < n_mine; ++kb) {
    // two chunks per trip (stage parities 0, 1: compile-time LDS offsets); the block's first chunk starts the accumulators
    auto chunk = [&](auto STAGE, auto FIRST, int c) {
      // here: V(c) complete, U(c) and raw(c + 1) in LDS; raw stage c & 1, U stage (c + 1) & 1 and V stage (c + 1) & 1 are free
      // (chunks beyond this block's are the first ones of the next block).  The transform of chunk c + 1 rides along (always,
      // except behind the last chunk of the workgroup's last block: a run-time flag -- as a third instantiation of the chunk
      // the register allocator spilled 213 registers)
#ifndef WINO_EXP_NODMA
      issue_raw_at(kb, c + 2);
      issue_u_at(kb, c + 1);
#endif
      uint32_t pm = 0xFFu;
      if (poly) {  // phase 2 py + px of this chunk: j == 3 (q = 3, 7) is zero for px = 1, i == 3 (the upper half's q = 4 .. 7) for py = 1
        const int ph2 = c >> cpp_shift;
        pm = 0xFFu & ~((ph2 & 1) ? 0x88u : 0u) & ~(((ph2 >> 1) && ph) ? 0xF0u : 0u);
      }
      if constexpr (TC) {  // class 2 pyc + pxc of this block: i == 0 (the lower half's q = 0 .. 3) is zero for pyc = 1, j == 0 (q = 0, 4) for pxc = 1
        // (the block's FIRST chunk issues every position: those multiply by U = 0 and clear their accumulators)
        const int cls = cur.cb / a.gy;
        pm = 0xFFu & ~((cls & 1) ? 0x11u : 0u) & ~(((cls >> 1) && !ph) ? 0x0Fu : 0u);
      }
      // the transform of chunk c + 1 (the next block's chunk 0 behind this block's last): this thread computes the positions
      // (i = t_i, j = 0 .. 3) -- nothing if row i is zero by construction in that chunk's phase / class, not V_j if column j is
      uint32_t tm = (c + 1 < nch || kb + 1u < n_mine) ? 0xFFu : 0u;
      if constexpr (MODE != 0) {
        const int pcn = c + 1 < nch ? phase_or_class(cur, c + 1) : phase_or_class(nxt, 0);
        if (zero_pos(pcn, t_i, 1)) tm = 0u;  // (j = 1 is never zero by construction: the row is)
        else if (zero_pos(pcn, 1, POLY ? 3 : 0)) tm &= POLY ? ~0x80u : ~0x10u;
      }
      multiply(STAGE, FIRST, tm, pm);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    };
    {
      using std::integral_constant;
      using S0 = integral_constant<int, 0>;
      using S1 = integral_constant<int, 1>;
      chunk(S0{}, std::true_type{}, 0);
      chunk(S1{}, std::false_type{}, 1);
      for (int c = 2; c < nch; c += 2) {
        chunk(S0{}, std::false_type{}, c);
        chunk(S1{}, std::false_type{}, c + 1);
      }
    }

#ifdef WINO_EXP_NOEPI
    if (kb + 1u < n_mine) { cur = nxt; if (kb + 2u < n_mine) nxt = make_desc(block_of(kb + 2u)); continue; }
#endif
    // ---- fold: S[a][b] = sum over this wave's positions (ascending, from +0) of T[a][i] T[b][j] M_p -------------------------------
    // keep[b]: output row a = ph (this wave finishes it), give[b]: row 1 - ph (handed to the partner).  The position half is a
    // compile-time constant inside each instantiation: the coefficients are, and the fold is the 18 block additions /
    // subtractions it needs (with ph a run-time value the compiler built every term of both signs and selected: 308 v_cndmask +
    // 224 v_sub + 136 v_pk_add per block and wave)
    floatx16 keep[2], give[2];
    auto fold = [&](auto PH) {
      constexpr int phc = decltype(PH)::value;
#pragma unroll
      for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int r = 0; r < 16; ++r) keep[bb][r] = give[bb][r] = 0.0f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = 2 * phc + (q >> 2), j = q & 3;
        const int t0i = i < 3 ? 1 : 0, t1i = i == 0 ? 0 : (i == 1 ? 1 : -1);
        const int t0j = j < 3 ? 1 : 0, t1j = j == 0 ? 0 : (j == 1 ? 1 : -1);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const int cf = ((o >> 1) ? t1i : t0i) * ((o & 1) ? t1j : t0j);
          floatx16 &dst = (o >> 1) == phc ? keep[o & 1] : give[o & 1];
          if (cf > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[r] = dst[r] + acc[q][r];
          } else if (cf < 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[r] = dst[r] - acc[q][r];
          }
        }
      }
    };
    if (ph) fold(std::integral_constant<int, 1>{});
    else fold(std::integral_constant<int, 0>{});
    // exchange: the wave of half ph finishes output row a = ph of the tiles; it hands its partial sums of the other row to its
    // partner (wave ^ 4), one output column b at a time, through the V stage the reduction does not touch before the next
    // chunk's transform (stage 1: the last chunk's V went to stage (nch - 1) & 1 = 1; the next block's chunk 0 sits in stage 0)
    {
      float *xch = reinterpret_cast<float *>(Vs + WINO_V_STAGE) + wave * 1024;  // [16 registers][64 lanes] per wave: 32 KB
      const float *theirs = reinterpret_cast<const float *>(Vs + WINO_V_STAGE) + (wave ^ 4) * 1024;
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[r * 64 + lane] = give[bb][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) keep[bb][r] = keep[bb][r] + theirs[r * 64 + lane];  // acc[a][b] = S0 + S1 (one fp32 addition:
        // the same bits whichever of the two operands is this wave's)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }

    // ---- epilogue: the contract's order (Epilogue::finish, common.h) on output row a = ph of every tile -------------------
    {
      const float *__restrict__ g_mul = p.mul;
      const float *__restrict__ g_res = p.res;
      float *__restrict__ g_y = p.y;
      // transposed form: this block's class (pyc, pxc); output pixel of grid pixel (y, x) = (2 y + pyc, 2 x + pxc) of a 2 H x 2 W image
      const int cls = TC ? cur.cb / a.gy : 0, pyc = cls >> 1, pxc = cls & 1;
      const int OS = TC ? 2 : 1;                     // output pixels per grid pixel and axis
      const int Wo_ = OS * W, Ho_ = OS * H;
      const int co = (TC ? cur.cb - cls * a.gy : cur.cb) * 64 + 32 * wn + l31;
      const bool has_bias = p.bias != nullptr;
      const float cbias = has_bias ? p.bias[co] : 0.0f;
      const int oy_w = 2 * (8 * cur.byi + 4 * wm) + ph;  // output row of the wave's first tile row (wave-uniform)
      const bool inside_x = 16 * cur.bxi + 16 <= W;
      if (!g_mul && act1 != AIVC_ACT_SIGMOID) {
        // fast path: one wave-uniform 64-bit base per tile row, one 32-bit byte offset per lane, tile column and output column as
        // immediates -- no per-element address arithmetic; lane masks only in the right-edge column of an image whose width is
        // no multiple of 16 (EDGE: the general path below made every output of such a block one dependent residual round
        // trip -- 160 instead of 43 us per block, which is what kept the kernel at the tap kernel's pace on the 68 x 120 layers)
        const int cols_left = W - 16 * cur.bxi;  // output columns of this block that exist (wave-uniform)
        typedef __attribute__((address_space(1))) char gchar;
        typedef __attribute__((address_space(1))) float gfloat;
        typedef __attribute__((address_space(1))) const float cgfloat;
        const uint32_t c4 = (uint32_t)Cout * 4u, sx = (uint32_t)OS * c4;  // bytes per output pixel / per grid pixel along x
        uint32_t lane_b = (uint32_t)(2 * (8 * cur.bxi + 4 * hh)) * sx + (uint32_t)pxc * c4 + (uint32_t)co * 4u;
        asm volatile("" : "+v"(lane_b));
        const size_t img_b = (size_t)cur.img * (size_t)Ho_ * Wo_ * c4;
        auto rows = [&](auto RES, auto KIND, auto EDGE) {
          constexpr bool has_res = decltype(RES)::value;
          constexpr bool edge = decltype(EDGE)::value;
          constexpr int kind = decltype(KIND)::value;  // act1 / act2 combination, see below
          bool ok[8];  // column 2 (4 hh + rr) + b of the block exists
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int b = 0; b < 2; ++b) ok[rr * 2 + b] = !edge || 2 * (4 * hh + rr) + b < cols_left;
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const int oy = oy_w + 2 * rq;
            if (oy >= H) continue;  // wave-uniform
            const size_t row_b = img_b + (size_t)(OS * oy + pyc) * Wo_ * c4;
            gchar *yb = (gchar *)(uintptr_t)(reinterpret_cast<char *>(g_y) + row_b);
            const gchar *rb = (const gchar *)(uintptr_t)(reinterpret_cast<const char *>(g_res) + row_b);
            float rv[8];
            if constexpr (has_res) {
#pragma unroll
              for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                  rv[rr * 2 + b] = ok[rr * 2 + b] ? *reinterpret_cast<cgfloat *>(rb + lane_b + (uint32_t)((2 * rr + b) * 512) * (sx / 512u)) : 0.0f;
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
              for (int b = 0; b < 2; ++b) {
                const int r = rq * 4 + rr;
                float v = keep[b][r];
                if (has_bias) v = v + cbias;
                if constexpr (kind == 1) v = __builtin_fmaxf(v, v * 0.01f);       // act1 leaky (same bits as the select: common.h)
                if constexpr (kind == 2) v = v > 0.0f ? v : 0.0f;                 // act1 relu
                if constexpr (has_res) v = v + rv[rr * 2 + b];
                if constexpr (kind == 3) v = v > 0.0f ? v : 0.0f;                 // act2 relu
                if constexpr (kind == 4) v = __builtin_fmaxf(v, v * 0.01f);       // act2 leaky
#ifdef WINO_EXP_NOSTORE
                if (v == 123.456f)
#endif
                if (ok[rr * 2 + b]) *reinterpret_cast<gfloat *>(yb + lane_b + (uint32_t)((2 * rr + b) * 512) * (sx / 512u)) = v;
              }
          }
        };
        using std::integral_constant;
        // kinds: 0 none; 1 / 2: act1 leaky / relu (act2 none); 3 / 4: act2 relu / leaky (act1 none); anything else -> slow path
        int kind = -1;
        if (act1 == AIVC_ACT_NONE && act2 == AIVC_ACT_NONE) kind = 0;
        else if (act2 == AIVC_ACT_NONE) kind = act1 == AIVC_ACT_LEAKY ? 1 : 2;
        else if (act1 == AIVC_ACT_NONE) kind = act2 == AIVC_ACT_RELU ? 3 : 4;
        if (kind >= 0 && sx % 512u == 0u) {
          auto go = [&](auto RES, auto EDGE) {
            switch (kind) {
              case 0: rows(RES, integral_constant<int, 0>{}, EDGE); break;
              case 1: rows(RES, integral_constant<int, 1>{}, EDGE); break;
              case 2: rows(RES, integral_constant<int, 2>{}, EDGE); break;
              case 3: rows(RES, integral_constant<int, 3>{}, EDGE); break;
              default: rows(RES, integral_constant<int, 4>{}, EDGE); break;
            }
          };
          if (inside_x) {
            if (g_res) go(integral_constant<bool, true>{}, integral_constant<bool, false>{});
            else go(integral_constant<bool, false>{}, integral_constant<bool, false>{});
          } else {
            if (g_res) go(integral_constant<bool, true>{}, integral_constant<bool, true>{});
            else go(integral_constant<bool, false>{}, integral_constant<bool, true>{});
          }
          goto epilogue_done;
        }
      }
      {
        const size_t ybase = (size_t)cur.img * (size_t)Ho_ * Wo_ * Cout + (size_t)co;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int t = (r & 3) + 8 * (r >> 2) + 4 * hh;  // tile of the wave's 4 x 8
          const int oy = oy_w + 2 * (t >> 3), ox0 = 2 * (8 * cur.bxi + (t & 7));
          if (oy >= H) continue;
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int ox = ox0 + b;
            if (ox >= W) continue;
            const size_t o = ybase + ((size_t)(OS * oy + pyc) * Wo_ + (size_t)(OS * ox + pxc)) * (size_t)Cout;
            float v = keep[b][r];
            if (has_bias) v = v + cbias;
            v = act_cheap(act1, v);
            if (g_mul) v = g_mul[o] * v;
            if (g_res) v = v + g_res[o];
            v = act_cheap(act2, v);
#ifdef WINO_EXP_NOSTORE
            if (v == 123.456f)
#endif
            g_y[o] = v;
          }
        }
      }
    epilogue_done:;
    }
    cur = nxt;
    if (kb + 2u < n_mine) nxt = make_desc(block_of(kb + 2u));
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
    for (int j = 0; j < 4; ++j) u[AIVC_WINO_U_INDEX(co, 4 * i + j, ci, c_in)] = (float)uu[i][j];
}

int winograd_weights(const float *w, int c_out, int c_in, float *u, hipStream_t s) {
  hipLaunchKernelGGL(winograd_weights_kernel, dim3(cdiv((size_t)c_out * c_in, 256)), dim3(256), 0, s, w, c_out, c_in, u);
  return check_launch("winograd_weights");
}

// Polyphase form of the 5x5 stride-2 kernel (include/aivc_hip.h): phase 2 py + px holds the taps ky = 2 r + py, kx = 2 l + px
// as a 3x3 kernel g[r][l] (zero where ky or kx would be 5), U = G g G^T as above; virtual input channel phase * c_in + ci of a
// 4 c_in-channel layer in the staging order of AIVC_WINO_U_INDEX.
__global__ void __launch_bounds__(256) winograd_weights_poly5_kernel(const float *w, int c_out, int c_in, float *u) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)c_out * c_in * 4) return;
  const int phase = (int)(idx % 4), ci = (int)((idx / 4) % c_in), co = (int)(idx / ((size_t)4 * c_in));
  const int py = phase >> 1, px = phase & 1;
  auto tap = [&](int r, int l) -> double {
    const int ky = 2 * r + py, kx = 2 * l + px;
    return ky < 5 && kx < 5 ? (double)w[(((size_t)co * 5 + ky) * 5 + kx) * c_in + ci] : 0.0;
  };
  double t[4][3], uu[4][4];
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    double col[4];
    wino_g(tap(0, l), tap(1, l), tap(2, l), col);
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i][l] = col[i];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) wino_g(t[i][0], t[i][1], t[i][2], uu[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) u[AIVC_WINO_U_INDEX(co, 4 * i + j, phase * c_in + ci, 4 * c_in)] = (float)uu[i][j];
}

int winograd_weights_poly5(const float *w, int c_out, int c_in, float *u, hipStream_t s) {
  hipLaunchKernelGGL(winograd_weights_poly5_kernel, dim3(cdiv((size_t)c_out * c_in * 4, 256)), dim3(256), 0, s, w, c_out, c_in, u);
  return check_launch("winograd_weights_poly5");
}

// Transposed 5x5 stride-2 kernel, class by class (include/aivc_hip.h): class 2 pyc + pxc holds the taps ky = pyc + 4 - 2 r,
// kx = pxc + 4 - 2 l as a 3x3 kernel g[r][l] (zero where ky or kx would be 5: r = 0 for pyc = 1), U = G g G^T; output channel
// block class * (c_out / 64) + co / 64 of a layer of 4 c_out "virtual" output channels in the staging order of AIVC_WINO_U_INDEX.
__global__ void __launch_bounds__(256) winograd_weights_tconv5_kernel(const float *w, int c_out, int c_in, float *u) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)c_out * c_in * 4) return;
  const int ci = (int)(idx % c_in), co = (int)((idx / c_in) % c_out), cls = (int)(idx / ((size_t)c_in * c_out));
  const int pyc = cls >> 1, pxc = cls & 1;
  auto tap = [&](int r, int l) -> double {
    const int ky = pyc + 4 - 2 * r, kx = pxc + 4 - 2 * l;
    return ky < 5 && kx < 5 ? (double)w[(((size_t)co * 5 + ky) * 5 + kx) * c_in + ci] : 0.0;
  };
  double t[4][3], uu[4][4];
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    double col[4];
    wino_g(tap(0, l), tap(1, l), tap(2, l), col);
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i][l] = col[i];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) wino_g(t[i][0], t[i][1], t[i][2], uu[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) u[AIVC_WINO_U_INDEX(cls * c_out + co, 4 * i + j, ci, c_in)] = (float)uu[i][j];
}

int winograd_weights_tconv5(const float *w, int c_out, int c_in, float *u, hipStream_t s) {
  hipLaunchKernelGGL(winograd_weights_tconv5_kernel, dim3(cdiv((size_t)c_out * c_in * 4, 256)), dim3(256), 0, s, w, c_out, c_in, u);
  return check_launch("winograd_weights_tconv5");
}

// what the kernel can address: 32-bit byte offsets inside one image
bool conv2d_wino_supported(const aivc_conv_params &p) {
  if (!aivc_winograd_covers(&p) || p.gdn) return false;
  if ((uint64_t)p.h_in * p.w_in * p.c_in * 4u >= 0xFFFF0000ull) return false;
  const bool tc = p.mode == AIVC_MODE_TCONV;
  if (tc && p.c_in > 960) return false;  // (a pixel outside the image reads c_in floats of the 4 KB zero page, one chunk after the other)
  const uint64_t gh = tc ? p.h_in : p.h_out, gw = tc ? p.w_in : p.w_out;  // the pixel grid the blocks walk
  const uint64_t blocks = (uint64_t)p.n * ((gh + 15) / 16) * ((gw + 15) / 16) * ((uint64_t)p.c_out / 64) * (tc ? 4 : 1);
  return blocks < 0x7FFFFFFFull;
}

int conv2d_wino_variant(const aivc_conv_params &p) { return p.mode == AIVC_MODE_TCONV ? 303 : (p.ksize == 5 ? 302 : 301); }

template <int MODE>
static int wino_launch(const WinoArgs &a, unsigned grid, hipStream_t s) {
  static LdsOptIn opt_in;
  if (!opt_in.raise(reinterpret_cast<const void *>(conv_wino_kernel<MODE>), WINO_LDS)) return check_launch("conv_wino lds attribute");
  hipLaunchKernelGGL(conv_wino_kernel<MODE>, dim3(grid), dim3(512), WINO_LDS, s, a);
  return check_launch("conv_wino");
}

int conv2d_wino(const aivc_conv_params &p, hipStream_t s) {
  if (!conv2d_wino_supported(p) || !p.w_wino) return AIVC_ERR_UNSUPPORTED;
  const bool tc = p.mode == AIVC_MODE_TCONV;
  WinoArgs a;
  a.p = p;
  a.poly = !tc && p.ksize == 5 ? 1 : 0;
  a.cpp_shift = 0;
  while ((8 << a.cpp_shift) < p.c_in) ++a.cpp_shift;  // (polyphase form: c_in / 8 is a power of two, aivc_winograd_covers)
  a.TH = ((tc ? p.h_in : p.h_out) + 1) / 2;
  a.TW = ((tc ? p.w_in : p.w_out) + 1) / 2;
  a.nby = (a.TH + 7) / 8;
  a.nbx = (a.TW + 7) / 8;
  a.gy = p.c_out / 64;
  a.total = (int)((size_t)p.n * a.nby * a.nbx * a.gy * (tc ? 4 : 1));
  static std::atomic<int> n_cu{0};
  if (n_cu.load(std::memory_order_relaxed) == 0) {
    int dev = 0, cus = 0;
    n_cu = hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0 ? cus : 256;
  }
  // persistent workgroups, one per CU (158 KB of LDS each); a multiple of 8 so that every XCD gets its share of the list
  unsigned grid = (unsigned)n_cu.load(std::memory_order_relaxed);
  if ((unsigned)a.total < grid) grid = (unsigned)a.total;
  if (tc) return wino_launch<2>(a, grid, s);
  return a.poly ? wino_launch<1>(a, grid, s) : wino_launch<0>(a, grid, s);
}

}  // namespace aivc
