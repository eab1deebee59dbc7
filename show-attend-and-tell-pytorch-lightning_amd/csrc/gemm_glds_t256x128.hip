// direct-to-LDS GEMM / implicit GEMM, tile form 256 x 128 (kernel: gemm_glds_kernel.h; dispatcher: gemm_glds.hip)
#include "gemm_glds_kernel.h"

namespace sat {
template <int AM, int BMo, typename TC> int glds_run_256x128(const BArgs& k, hipStream_t st) { return rung<256, 128, 4, 2, AM, BMo, TC>(k, st); }
SAT_GLDS_INSTANTIATE(glds_run_256x128)
}  // namespace sat
