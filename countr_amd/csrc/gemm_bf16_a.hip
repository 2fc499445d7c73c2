// bf16 instantiations of gemm_kernel (gemm_kernel.hpp): nn.Linear forward shapes the lean kernels refuse, (ROW, COL) input gradients, (COL, COL) weight gradients / batched attention products
#include "gemm_kernel.hpp"

int countr_gemm_bf16_a(const countr_gemm_args& a, int ma, int mb, hipStream_t s) {
  if (ma == COUNTR_OP_ROW && mb == COUNTR_OP_ROW) return launch<bf16_t, COUNTR_OP_ROW, COUNTR_OP_ROW>(a, s);
  if (ma == COUNTR_OP_ROW && mb == COUNTR_OP_COL) return launch<bf16_t, COUNTR_OP_ROW, COUNTR_OP_COL>(a, s);
  if (ma == COUNTR_OP_COL && mb == COUNTR_OP_COL) return launch<bf16_t, COUNTR_OP_COL, COUNTR_OP_COL>(a, s);
  return 1;
}
