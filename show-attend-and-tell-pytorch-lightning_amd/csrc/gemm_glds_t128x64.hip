// direct-to-LDS GEMM / implicit GEMM, tile form 128 x 64 (kernel: gemm_glds_kernel.h; dispatcher: gemm_glds.hip)
#include "gemm_glds_kernel.h"

namespace sat {
template <int AM, int BMo, typename TC> int glds_run_128x64(const BArgs& k, hipStream_t st) { return rung<128, 64, 2, 2, AM, BMo, TC>(k, st); }
SAT_GLDS_INSTANTIATE(glds_run_128x64)
}  // namespace sat
