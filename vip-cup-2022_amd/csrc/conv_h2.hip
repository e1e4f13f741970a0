// The GEMM / convolution kernel family of conv_igemm.hip instantiated for the packed STRICT storage (fp16 (hi, lo) pairs, three MFMAs per
// fragment pair; common.hpp): vip_conv2d_nhwc_h2, vip_conv2d_kernel_name_h2.  One source, two arithmetic modes -
// the tiling, staging, XCD mapping and dispatch that were tuned on the fp16 path are the strict path's too.
#define VIP_GEMM_H2 1
#include "conv_igemm.hip"
