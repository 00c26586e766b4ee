// conv_bf16x3.hip -- the "bf16x3" precision mode of the conv family (aivc_conv_params.precision = 1): the kernels of
// conv_mfma.hip instantiated with PREC = 1 (fp32 operands split exactly into three bf16 terms, six bf16 MFMA products
// per fp32 product, fp32 accumulation).  A translation unit of its own so that it compiles beside conv_mfma.hip.
#define AIVC_CONV_BF16X3 1
#include "conv_mfma.hip"
