// fp32 (parity mode) instantiations of gemm_kernel (gemm_kernel.hpp): exact v_mfma_f32_16x16x4_f32 chains, all six operand-mode pairs
#include "gemm_kernel.hpp"

int countr_gemm_f32(const countr_gemm_args& a, int ma, int mb, hipStream_t s) {
  if (ma == COUNTR_OP_ROW && mb == COUNTR_OP_ROW) return launch<float, COUNTR_OP_ROW, COUNTR_OP_ROW>(a, s);
  if (ma == COUNTR_OP_ROW && mb == COUNTR_OP_COL) return launch<float, COUNTR_OP_ROW, COUNTR_OP_COL>(a, s);
  if (ma == COUNTR_OP_COL && mb == COUNTR_OP_COL) return launch<float, COUNTR_OP_COL, COUNTR_OP_COL>(a, s);
  if (ma == COUNTR_OP_COL && mb == COUNTR_OP_ROW) return launch<float, COUNTR_OP_COL, COUNTR_OP_ROW>(a, s);
  if (ma == COUNTR_OP_IM2ROW && mb == COUNTR_OP_ROW) return launch<float, COUNTR_OP_IM2ROW, COUNTR_OP_ROW>(a, s);
  if (ma == COUNTR_OP_COL && mb == COUNTR_OP_IM2COL) return launch<float, COUNTR_OP_COL, COUNTR_OP_IM2COL>(a, s);
  return 1;
}
