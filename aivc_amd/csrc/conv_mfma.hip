// conv_mfma.hip -- placeholder until the MFMA implicit-GEMM kernels land (next milestone).
#include "common.h"
namespace aivc {
bool conv2d_mfma_supported(const aivc_conv_params &) { return false; }
int conv2d_mfma(const aivc_conv_params &, hipStream_t) { return AIVC_ERR_UNSUPPORTED; }
}  // namespace aivc
