// api.hip -- library identity, error plumbing and the conv dispatcher.
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace aivc {
static thread_local char g_err[256] = "";
void set_last_error(const char *msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}
int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return AIVC_OK;
  char buf[256];
  snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
  set_last_error(buf);
  return AIVC_ERR_LAUNCH;
}
}  // namespace aivc

AIVC_EXPORT int aivc_abi_version(void) { return AIVC_ABI_VERSION; }
AIVC_EXPORT const char *aivc_last_error(void) { return aivc::g_err; }

// tuning aid: AIVC_GDN_RESIDENT=0 sends stand-alone (I)GDN launches to the generic kernel again (bit identical)
static bool gdn_resident_on() {
  static const bool on = !(getenv("AIVC_GDN_RESIDENT") && atoi(getenv("AIVC_GDN_RESIDENT")) == 0);
  return on;
}

static int validate_conv(const aivc_conv_params *p) {
  if (!p || !p->x || !p->w || !p->y) return AIVC_ERR_ARG;
  const int k = p->ksize, s = p->stride;
  if (k != 1 && k != 3 && k != 5) return AIVC_ERR_UNSUPPORTED;
  if (p->n <= 0 || p->h_in <= 0 || p->w_in <= 0 || p->c_in <= 0 || p->c_out <= 0) return AIVC_ERR_ARG;
  if (p->c_in % 4) return AIVC_ERR_ARG;
  switch (p->mode) {
    case AIVC_MODE_CONV:
      if (s != 1 && s != 2) return AIVC_ERR_UNSUPPORTED;
      if (p->pad != 0 && p->pad != k / 2) return AIVC_ERR_UNSUPPORTED;
      if (p->h_out != (p->h_in + 2 * p->pad - k) / s + 1 || p->w_out != (p->w_in + 2 * p->pad - k) / s + 1)
        return AIVC_ERR_ARG;
      break;
    case AIVC_MODE_TCONV:
      if (s != 2 || k == 1 || p->h_out != 2 * p->h_in || p->w_out != 2 * p->w_in) return AIVC_ERR_ARG;
      break;
    case AIVC_MODE_GDN:
    case AIVC_MODE_IGDN:
      if (k != 1 || p->h_out != p->h_in || p->w_out != p->w_in || p->c_out != p->c_in) return AIVC_ERR_ARG;
      break;
    default:
      return AIVC_ERR_UNSUPPORTED;
  }
  if (p->gdn) {
    if (p->gdn < 0 || p->gdn > 2 || p->mode == AIVC_MODE_GDN || p->mode == AIVC_MODE_IGDN) return AIVC_ERR_ARG;
    if (!p->gdn_beta || !p->gdn_gamma) return AIVC_ERR_ARG;
  }
  if (p->precision != AIVC_PREC_FP32 && p->precision != AIVC_PREC_BF16X3 && p->precision != AIVC_PREC_FP32_WINO) return AIVC_ERR_ARG;
  if (p->tail_c_out) {
    if (p->tail_c_out < 0 || !p->tail_w || p->mode != AIVC_MODE_CONV || p->gdn || p->mul) return AIVC_ERR_ARG;
  }
  return AIVC_OK;
}

AIVC_EXPORT int aivc_conv2d_variant(const aivc_conv_params *p) {
  int rc = validate_conv(p);
  if (rc != AIVC_OK) return rc;
  // version 2 of the contract: the layers it covers run the Winograd chain or nothing (a fused gdn is two launches there)
  if (p->precision == AIVC_PREC_FP32_WINO && aivc_winograd_covers(p))
    return aivc::conv2d_wino_supported(*p) ? aivc::conv2d_wino_variant(*p) : AIVC_ERR_UNSUPPORTED;
  if (p->gdn && (p->algo == AIVC_ALGO_DIRECT || !aivc::conv2d_mfma_supported(*p))) return AIVC_ERR_UNSUPPORTED;
  // 1000 + the fp32 code with the mode's tile: the precision mode takes this launch
  if (p->precision == AIVC_PREC_BF16X3 && p->algo != AIVC_ALGO_DIRECT && aivc::conv2d_bf16x3_supported(*p) &&
      aivc::conv2d_mfma_supported(*p) && (!p->tail_c_out || aivc::conv2d_mfma_tail_supported(*p)))
    return p->tail_c_out ? 1190 : 1000 + 100 + 10 * (p->mode == AIVC_MODE_TCONV ? 1 : 0) + aivc::conv2d_bf16x3_tile(*p) + (p->gdn ? 50 : 0);
  if (p->tail_c_out) {
    if (p->algo == AIVC_ALGO_DIRECT || !aivc::conv2d_mfma_tail_supported(*p)) return AIVC_ERR_UNSUPPORTED;
    return aivc::conv2d_mfma_variant(*p);
  }
  if (p->algo == AIVC_ALGO_DIRECT) return 0;
  if (p->algo == AIVC_ALGO_AUTO && gdn_resident_on() && aivc::gdn_resident_supported(*p)) return 400;
  if (p->algo == AIVC_ALGO_AUTO && aivc::conv2d_thin_supported(*p)) return aivc::conv2d_thin_variant(*p);
  if (p->algo == AIVC_ALGO_MFMA || aivc::conv2d_mfma_supported(*p)) return aivc::conv2d_mfma_variant(*p);
  return 0;
}

AIVC_EXPORT int aivc_split_weights_bf16x3(const float *w, int32_t c_out, int32_t k_total, void *out, aivc_stream_t stream) {
  if (!w || !out || c_out <= 0 || k_total <= 0 || k_total % 32 || ((uintptr_t)out & 15u) || ((uintptr_t)w & 7u)) return AIVC_ERR_ARG;
  return aivc::split_weights_bf16x3(w, c_out, k_total, out, aivc::to_stream(stream));
}

AIVC_EXPORT int aivc_winograd_weights(const float *w, int32_t c_out, int32_t c_in, float *u, aivc_stream_t stream) {
  if (!w || !u || c_out <= 0 || c_in <= 0 || c_out % 64 || c_in % 8) return AIVC_ERR_ARG;
  return aivc::winograd_weights(w, c_out, c_in, u, aivc::to_stream(stream));
}

AIVC_EXPORT int aivc_winograd_weights_poly5(const float *w, int32_t c_out, int32_t c_in, float *u, aivc_stream_t stream) {
  if (!w || !u || c_out <= 0 || c_in <= 0 || c_out % 64 || c_in % 8) return AIVC_ERR_ARG;
  return aivc::winograd_weights_poly5(w, c_out, c_in, u, aivc::to_stream(stream));
}

AIVC_EXPORT int aivc_winograd_weights_tconv5(const float *w, int32_t c_out, int32_t c_in, float *u, aivc_stream_t stream) {
  if (!w || !u || c_out <= 0 || c_in <= 0 || c_out % 64 || c_in % 8) return AIVC_ERR_ARG;
  return aivc::winograd_weights_tconv5(w, c_out, c_in, u, aivc::to_stream(stream));
}

AIVC_EXPORT int aivc_conv_images(const aivc_image_src *src, int32_t n_img, const aivc_conv_params *p, aivc_stream_t stream) {
  if (!p || !p->w || !p->y || !src || n_img < 1 || n_img > AIVC_MAX_IMAGES) return AIVC_ERR_ARG;
  if (p->n <= 0 || p->h_in <= 0 || p->w_in <= 0) return AIVC_ERR_ARG;
  for (int i = 0; i < n_img; ++i) {
    if (src[i].y && (!src[i].u || !src[i].v)) return AIVC_ERR_ARG;
    if (!src[i].y && src[i].f && src[i].f_channels < 3) return AIVC_ERR_ARG;
  }
  if (!aivc::conv_images_supported(src, n_img, *p)) return AIVC_ERR_UNSUPPORTED;
  if (p->h_out != (p->h_in + 2 * p->pad - p->ksize) / p->stride + 1 || p->w_out != (p->w_in + 2 * p->pad - p->ksize) / p->stride + 1)
    return AIVC_ERR_ARG;
  if (p->gdn && (p->gdn < 0 || p->gdn > 2 || !p->gdn_beta || !p->gdn_gamma)) return AIVC_ERR_ARG;
  return aivc::conv_images(src, n_img, *p, aivc::to_stream(stream));
}

AIVC_EXPORT int aivc_conv2d(const aivc_conv_params *p, aivc_stream_t stream) {
  int rc = validate_conv(p);
  if (rc != AIVC_OK) return rc;
  hipStream_t s = aivc::to_stream(stream);
  // version 2 of the fp32 contract: a covered layer is computed by the Winograd chain or not at all (never silently by
  // the tap chain: the two differ in the last bits and an encoder / decoder pair must agree)
  if (p->precision == AIVC_PREC_FP32_WINO && aivc_winograd_covers(p)) {
    if (!p->w_wino || ((uintptr_t)p->w_wino & 15u) || ((uintptr_t)p->x & 15u)) return AIVC_ERR_ARG;  // (16-byte LDS-DMA loads)
    return aivc::conv2d_wino(*p, s);
  }
  // precision mode (never the default): the shapes it covers; everything else runs the fp32 contract
  if (p->precision == AIVC_PREC_BF16X3 && p->algo != AIVC_ALGO_DIRECT && aivc::conv2d_bf16x3_supported(*p) &&
      aivc::conv2d_mfma_supported(*p) && (!p->tail_c_out || aivc::conv2d_mfma_tail_supported(*p)))
    return aivc::conv2d_bf16x3(*p, s);
  if (p->gdn) {  // fused (I)GDN exists on the MFMA path only
    if (p->algo == AIVC_ALGO_DIRECT || !aivc::conv2d_mfma_supported(*p)) return AIVC_ERR_UNSUPPORTED;
    return aivc::conv2d_mfma(*p, s);
  }
  if (p->tail_c_out) {  // fused 1x1 tail: MFMA path only
    if (p->algo == AIVC_ALGO_DIRECT || !aivc::conv2d_mfma_tail_supported(*p)) return AIVC_ERR_UNSUPPORTED;
    return aivc::conv2d_mfma(*p, s);
  }
  if (p->algo == AIVC_ALGO_DIRECT) return aivc::conv2d_direct(*p, s);
  if (p->algo == AIVC_ALGO_MFMA) return aivc::conv2d_mfma(*p, s);
  if (gdn_resident_on() && aivc::gdn_resident_supported(*p)) return aivc::gdn_resident(*p, s);  // stand-alone (I)GDN: same bits as the GDN-mode launch below
  if (aivc::conv2d_thin_supported(*p)) return aivc::conv2d_thin(*p, s);
  if (aivc::conv2d_mfma_supported(*p)) return aivc::conv2d_mfma(*p, s);
  return aivc::conv2d_direct(*p, s);
}
