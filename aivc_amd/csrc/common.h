// common.h -- shared host/device helpers of libaivc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/aivc_detmath.h"
#include "../../include/aivc_hip.h"

#define AIVC_EXPORT extern "C" __attribute__((visibility("default")))

namespace aivc {

void set_last_error(const char *msg);
int check_launch(const char *what);  // hipGetLastError() -> AIVC_OK / AIVC_ERR_LAUNCH

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: one process driving several GPUs through the
// C ABI must raise it on each of them.  One bit per device ordinal (mod 64) and kernel instantiation; safe to race
// (the worst case sets the attribute twice).
struct LdsOptIn {
  std::atomic<uint64_t> done{0};
  bool raise(const void *fn, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
    done.fetch_or(bit, std::memory_order_release);
    return true;
  }
};

static inline hipStream_t to_stream(aivc_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

__device__ __forceinline__ float act_apply(int act, float v) {
  switch (act) {
    case AIVC_ACT_LEAKY: return v > 0.0f ? v : v * 0.01f;
    case AIVC_ACT_RELU: return v > 0.0f ? v : 0.0f;
    case AIVC_ACT_SIGMOID: return aivc_sigmoidf_det(v);
    default: return v;
  }
}

// ---- first-layer GDN epilogue (conv_images.hip): lean, correctly rounded sqrt / division ---------------------------------
// __builtin_sqrtf / operator/ compile to the full IEEE sequences: operand scaling for denormals and extreme exponents
// (v_div_scale, a compare + select + multiply in front of v_sqrt), the refinement, then unscaling and special-value
// fix-up (v_div_fmas / v_div_fixup, v_cmp_class + select): 16 + 11 vector instructions (and three s_nop hazards) per
// output, of which the refinement is 8 + 8.  The operands are ordinary numbers: a wavefront checks ONCE that all of its
// normalisation sums s and numerators v lie in [2^-60, 2^60] (two instructions per output, gdn_range) and then runs only
// the refinements below; any wavefront that fails the check takes the compiler's full sequences.  Inside that range no
// scaling step of the full sequence fires and no fix-up applies, so:
//   div_rn_safe   IS the hardware expansion of operator/ without its range handling (v_div_scale returns its operand,
//                 v_div_fmas is a plain fma, v_div_fixup passes the quotient through): same bits by construction;
//   sqrt_rn_safe  is the reciprocal-square-root refinement LLVM uses for an IEEE sqrt when fp32 denormals are flushed
//                 (two coupled Newton steps and a final residual correction): correctly rounded, hence the same bits as
//                 __builtin_sqrtf -- checked EXHAUSTIVELY over every float in the range by aivc_selfcheck_gdn_math
//                 (tests/test_gpu_ops.py), which also compares div_rn_safe with operator/ on 2^39 random operand pairs.
// Measured (round 5, experiments/r05.md 11): the image layer 3.67 -> 3.49 ms (1 image) / 5.47 -> 5.26 ms (2 images) per
// 32 frames of 1080p; the same change in the fused-GDN epilogues of conv_mfma.hip measured NOTHING (+-1 % on every shape,
// even without the range check): those kernels hide their epilogue behind the other workgroups' K loops.  Not applied there.
constexpr float GDN_SAFE_LO = 0x1p-60f, GDN_SAFE_HI = 0x1p60f;
__device__ __forceinline__ float sqrt_rn_safe(float s) {
  const float y = __builtin_amdgcn_rsqf(s);
  float g = s * y, h = 0.5f * y;
  const float e = __builtin_fmaf(-h, g, 0.5f);
  h = __builtin_fmaf(h, e, h);
  g = __builtin_fmaf(g, e, g);
  const float d = __builtin_fmaf(-g, g, s);
  return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float div_rn_safe(float v, float d) {
  float r = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  float q = v * r;
  float t = __builtin_fmaf(-d, q, v);
  q = __builtin_fmaf(t, r, q);
  t = __builtin_fmaf(-d, q, v);
  return __builtin_fmaf(t, r, q);
}
// running maximum / minimum of |a| and s (v_max3_f32 / v_min3_f32: one instruction each for two values; a NaN operand is
// ignored -- it propagates through either sequence alike)
__device__ __forceinline__ void gdn_range(float &mx, float &mn, float a, float s) {
  // (plain builtins: the backend fuses the pairs into v_max3_f32 / v_min3_f32; inline asm made it fence every
  // instruction with an s_nop)
  mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fabsf(a)), s);
  mn = __builtin_fminf(__builtin_fminf(mn, __builtin_fabsf(a)), s);
}
__device__ __forceinline__ void gdn_range_pair(float &mx, float &mn, float s0, float s1) {  // two sums (signs matter)
  mx = __builtin_fmaxf(__builtin_fmaxf(mx, s0), s1);
  mn = __builtin_fminf(__builtin_fminf(mn, s0), s1);
}
__device__ __forceinline__ bool gdn_range_ok(float mx, float mn) {  // wave-uniform: every lane's values are in range
  return __builtin_amdgcn_ballot_w64(mn >= GDN_SAFE_LO && mx <= GDN_SAFE_HI) == __builtin_amdgcn_ballot_w64(true);
}

// Epilogue shared by every conv implementation (order fixed by include/aivc_hip.h).
struct Epilogue {
  const float *bias, *mul, *res, *xin;  // xin: GDN input (same pixel/channel indexing as y)
  float *y;
  int act1, act2, mode;
  __device__ __forceinline__ void store(size_t opix, int co, int c_out, float acc) const {
    float v = acc;
    if (bias) v = v + bias[co];
    if (mode == AIVC_MODE_GDN || mode == AIVC_MODE_IGDN) {
      const float xc = xin[opix * c_out + co];
      const float nrm = __builtin_sqrtf(v);
      v = (mode == AIVC_MODE_IGDN) ? xc * nrm : xc / nrm;
    }
    finish(opix, co, c_out, v);
  }
  // everything after the (I)GDN step
  __device__ __forceinline__ void finish(size_t opix, int co, int c_out, float v) const {
    v = act_apply(act1, v);
    const size_t o = opix * c_out + co;
    if (mul) v = mul[o] * v;
    if (res) v = v + res[o];
    v = act_apply(act2, v);
    y[o] = v;
  }
};

int conv2d_direct(const aivc_conv_params &p, hipStream_t s);
int conv2d_mfma(const aivc_conv_params &p, hipStream_t s);  // AIVC_ERR_UNSUPPORTED if shape not covered
bool conv2d_mfma_supported(const aivc_conv_params &p);
bool conv2d_mfma_tail_supported(const aivc_conv_params &p);
int conv2d_mfma_variant(const aivc_conv_params &p);
bool conv2d_bf16x3_supported(const aivc_conv_params &p);  // conv_bf16x3.hip: the precision mode (aivc_conv_params.precision = 1)
int conv2d_bf16x3(const aivc_conv_params &p, hipStream_t s);
int split_weights_bf16x3(const float *w, int c_out, int k_total, void *out, hipStream_t s);
int conv2d_bf16x3_tile(const aivc_conv_params &p);  // tile id of the mode's launch (aivc_conv2d_variant)  // 100 + 10*mode + tile id (+50 fused gdn); 190 fused 1x1 tail
bool conv2d_wino_supported(const aivc_conv_params &p);  // conv_wino.hip: AIVC_PREC_FP32_WINO, what the kernel can address (no fused gdn)
int conv2d_wino(const aivc_conv_params &p, hipStream_t s);
int conv2d_wino_variant(const aivc_conv_params &p);  // 301: 3x3 stride 1, 302: 5x5 stride 2 in polyphase form, 303: transposed 5x5 stride 2 by classes
int winograd_weights(const float *w, int c_out, int c_in, float *u, hipStream_t s);
int winograd_weights_poly5(const float *w, int c_out, int c_in, float *u, hipStream_t s);  // 5x5 stride 2 in polyphase form
int winograd_weights_tconv5(const float *w, int c_out, int c_in, float *u, hipStream_t s);  // transposed 5x5 stride 2, class by class
bool gdn_resident_supported(const aivc_conv_params &p);  // gdn.hip: stand-alone (I)GDN of 64 / 128 channels, gamma resident in registers
int gdn_resident(const aivc_conv_params &p, hipStream_t s);  // variant 400
bool conv_images_supported(const aivc_image_src *src, int n_img, const aivc_conv_params &p);
int conv_images(const aivc_image_src *src, int n_img, const aivc_conv_params &p, hipStream_t s);  // conv_images.hip
bool conv2d_thin_supported(const aivc_conv_params &p);
int conv2d_thin(const aivc_conv_params &p, hipStream_t s);  // c_out of 3 / 6 (transposed conv): 16x16x4 MFMA or VALU kernel
int conv2d_thin_variant(const aivc_conv_params &p);          // 2 = thin_mfma_kernel, 1 = thin_tconv_kernel

}  // namespace aivc
