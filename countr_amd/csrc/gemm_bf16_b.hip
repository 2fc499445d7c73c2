// bf16 instantiations of gemm_kernel (gemm_kernel.hpp): (COL, ROW), the 3x3 convolutions' forward / dgrad (IM2ROW, ROW) and weight gradient (COL, IM2COL) on the shapes the lean kernels refuse
#include "gemm_kernel.hpp"

int countr_gemm_bf16_b(const countr_gemm_args& a, int ma, int mb, hipStream_t s) {
  if (ma == COUNTR_OP_COL && mb == COUNTR_OP_ROW) return launch<bf16_t, COUNTR_OP_COL, COUNTR_OP_ROW>(a, s);
  if (ma == COUNTR_OP_IM2ROW && mb == COUNTR_OP_ROW) return launch<bf16_t, COUNTR_OP_IM2ROW, COUNTR_OP_ROW>(a, s);
  if (ma == COUNTR_OP_COL && mb == COUNTR_OP_IM2COL) return launch<bf16_t, COUNTR_OP_COL, COUNTR_OP_IM2COL>(a, s);
  return 1;
}
