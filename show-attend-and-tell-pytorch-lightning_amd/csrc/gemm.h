// MFMA GEMM / implicit-GEMM family used by every contraction on the SAT hot
// path: decoder Linears (model.py:72-73,90-92,119-123,175-180,188), the 1x1
// projection conv (model.py:53) and the ResNet conv stack fwd/dgrad/wgrad.
#pragma once
#include "common.h"

namespace sat {

// How the A operand (M x K) is laid out / gathered.
enum AMode : int {
    A_ROW = 0,         // A[m][k], k contiguous (activations x Linear)        optional a_rows gather
    A_KMAJOR = 1,      // A stored [k][m], m contiguous (dY for weight grads)
    A_CONV_FWD = 2,    // im2col over NHWC input : m=(n,p,q)  k=(r,s,c)
    A_CONV_DGRAD = 3   // im2col over NHWC dY    : m=(n,h,w)  k=(r,s,ko) with stride predicate
};
// How the B operand (K x N) is laid out / gathered.
enum BMode : int {
    B_ROW = 0,           // B stored [n][k], k contiguous (Linear weight (out,in), conv weight KRSC)
    B_KMAJOR = 1,        // B stored [k][n], n contiguous
    B_CONV_WGRAD = 2,    // k=(n,p,q) pixel, n=(r,s,c): gathered NHWC input
    B_CONV_DGRAD_W = 3   // k=(r,s,ko), n=c : KRSC weight read as [(r,s,ko)][c]
};
enum Epi : int {
    EPI_NONE = 0,
    EPI_BIAS = 1,               // + bias[col]
    EPI_BIAS_SIGMOID_RANGE = 2, // + bias[col] (if bias), sigmoid on cols [c0,c1)
    EPI_ADD_TANH = 3,           // tanh(v + e0[arow][col])   (arow = a_rows[row] if gather)
    EPI_MUL_DTANH = 4,          // v * (1 - e0[row][col]^2)
    EPI_BIAS_RELU = 5           // max(0, v + bias[col])
};

struct ConvGeom {
    int N, H, W, C;      // input  NHWC
    int K, R, S;         // filters KRSC
    int P, Q;            // output NPQK
    int stride, pad;
    int sw = 0;          // horizontal stride when it differs from `stride` (0 = the same): the stem's tap-pair view, forward and weight gradient only
    // stride-2 data gradient by output parity class (bf16 kernel): rows are the input pixels with h % 2 == ph and
    // w % 2 == pw (Hc x Wc of them per image); only the filter taps of matching parity enter the reduction
    int cls = 0, ph = 0, pw = 0, Hc = 0, Wc = 0, r0 = 0, rc = 0, s0 = 0, sc = 0;
};

struct GemmArgs {
    const void* A = nullptr; long lda = 0; const int* a_rows = nullptr;
    const void* B = nullptr; long ldb = 0;
    void* C = nullptr;        long ldc = 0; const int* c_rows = nullptr;
    int a_bf16 = 0, b_bf16 = 0, c_bf16 = 0;   // element type in HBM (0 = fp32, 1 = bf16)
    int bf16_mfma = 0;                        // 1: v_mfma_f32_32x32x16_bf16 (gemm_bf16.hip), 0: exact fp32 MFMA
    int M = 0, N = 0, K = 0;
    int amode = A_ROW, bmode = B_ROW;
    int accumulate = 0;                       // C += result (applied before the epilogue function)
    int epi = EPI_NONE;
    const float* bias = nullptr;
    const float* e0 = nullptr; long lde0 = 0;
    int c0 = 0, c1 = 0;
    ConvGeom g = {};
    float* slab = nullptr; long slab_elems = 0;   // split-K scratch (optional)
    // bf16 data-gradient launches: BatchNorm-backward form of the tile statistics (see BArgs::bn_x in gemm_bf16_common.h)
    const void* bn_x = nullptr; const unsigned char* bn_mask = nullptr; const float* bn_mean = nullptr; const float* bn_invstd = nullptr;
    const void* add_src = nullptr; const unsigned char* add_mask = nullptr;      // accumulate from this tensor (C's layout), gated by mask bits (see BArgs::add_src)
    float* tile_stats = nullptr; int* tile_rows = nullptr;   // bf16 kernels: per-row-tile column statistics of the stored result (see gemm_bf16_common.h); *tile_rows = rows per tile chosen, 0 if not produced
};

// Launches on `stream`; returns SAT_OK or sets last_error.  bf16_mfma requests fall back to the exact fp32
// kernel when the operands are fp32 but not 16-byte gatherable (tiny / odd test shapes).
int launch_gemm(const GemmArgs& a, hipStream_t stream);
int launch_gemm_bf16(const GemmArgs& a, hipStream_t stream);
int gemm_bf16_eligible(const GemmArgs& a);
// dev switches (environment variable = initial value; sat_debug_option(name, value) changes them at run time so that variants can be
// timed alternately inside ONE process): index into dev_switch()
enum { SW_UNUSED0 = 0, SW_WGRAD3X3 = 1, SW_REDUCE_Z16 = 2, SW_WIDE_TILES = 3, SW_BN_ONEPASS = 4, SW_BN_VPT = 5, SW_ACC_PREFETCH = 6, SW_COUNT = 7 };
int& dev_switch(int which);

// wgrad3x3.hip: weight gradient of a 3x3 / stride 1 / pad 1 convolution with all nine taps per workgroup (bf16 activations, fp32 result)
int wgrad3x3_eligible(const ConvGeom& g);
int launch_wgrad3x3(const void* dy, const void* x, float* dw, const ConvGeom& g, float* slab, long slab_elems, hipStream_t st);
// Bytes of split-K scratch that lets launch_gemm fill the chip for this shape (0 = none needed).
size_t gemm_slab_bytes(int M, int N, int K);

}  // namespace sat
