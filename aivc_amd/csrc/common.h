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
bool conv_images_supported(const aivc_image_src *src, int n_img, const aivc_conv_params &p);
int conv_images(const aivc_image_src *src, int n_img, const aivc_conv_params &p, hipStream_t s);  // conv_images.hip
bool conv2d_thin_supported(const aivc_conv_params &p);
int conv2d_thin(const aivc_conv_params &p, hipStream_t s);  // c_out of 3 / 6 (transposed conv): 16x16x4 MFMA or VALU kernel
int conv2d_thin_variant(const aivc_conv_params &p);          // 2 = thin_mfma_kernel, 1 = thin_tconv_kernel

}  // namespace aivc
