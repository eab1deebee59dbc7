// Shared pieces of the bf16-MFMA GEMM kernels (gemm_bf16.hip: register-staged operands; gemm_glds.hip: operands
// copied HBM -> LDS by the load unit): argument block, epilogue functions, accumulator write-out.
#pragma once
#include <stdlib.h>

#include "gemm.h"

namespace sat {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NT = 256;

struct BArgs {
    const void* A; long lda; const int* a_rows;
    const void* B; long ldb;
    void* C; long ldc; const int* c_rows;
    int M, N, K;
    int accumulate, epi;
    const float* bias; const float* e0; long lde0; int c0, c1;
    ConvGeom g;
    int kchunk, nsplit;
    float* slab;
    int wide_store;       // output tile staged through LDS and written 16 bytes per lane
    int wide_slab = 0;    // the same for split-K partials
    int nstage = 2;       // gemm_glds.hip: LDS ring depth
    int rotate = 1 << 20; // gemm_glds.hip: per-workgroup rotation of the k-tile sequence, window in k-tiles (0 = off)
    float* tile_stats = nullptr;   // optional (wide_store only): per row tile and output column (sum, sum of squares) of the stored bf16 values
    // BatchNorm-backward form of the tile statistics (data-gradient launches): with bn_x set, the per-tile pair is
    // (sum g, sum g * xhat) with g = stored value (masked by the ReLU sign bits when bn_mask is set) and xhat = (bn_x - mean) * invstd,
    // i.e. the partials of dbeta / dgamma of the BatchNorm whose OUTPUT gradient this launch writes.  bn_x / bn_mask share C's layout.
    const __bf16* bn_x = nullptr; const unsigned char* bn_mask = nullptr; const float* bn_mean = nullptr; const float* bn_invstd = nullptr;
    // accumulate from ANOTHER tensor of C's layout (wide bf16 store only): C = result + add_src, add_src gated per element by the bits of add_mask
    // when given - the identity path of a residual block joins its data gradient here without ever being written out masked
    const __bf16* add_src = nullptr; const unsigned char* add_mask = nullptr;
    int acc_prefetch = 1;      // dev switch: accumulating bf16 launches take the instantiation that fetches the old values before the write loop
    int y_fastest = 0;         // tile order inside an XCD's run of tiles: 0 = column tiles fastest (tiles of one row panel share an L2), 1 = row tiles fastest
};

template <typename T> struct VecN { static constexpr int n = 16 / sizeof(T); };

__device__ __forceinline__ float ep_value(const BArgs& a, int row, int col, float v) {
    switch (a.epi) {
        case EPI_BIAS: v += a.bias[col]; break;
        case EPI_BIAS_SIGMOID_RANGE:
            if (a.bias) v += a.bias[col];
            if (col >= a.c0 && col < a.c1) v = fast_sigmoid(v);
            break;
        case EPI_ADD_TANH: { long er = a.a_rows ? (long)a.a_rows[row] : (long)row; v = fast_tanh(v + a.e0[er * a.lde0 + col]); } break;
        case EPI_MUL_DTANH: { float u = a.e0[(long)row * a.lde0 + col]; v *= (1.0f - u * u); } break;
        case EPI_BIAS_RELU: v += a.bias[col]; v = v < 0.f ? 0.f : v; break;        // NaN stays NaN
        default: break;
    }
    return v;
}

// parity-class data gradient: GEMM row (n, h', w') -> NHWC pixel (n, 2h'+ph, 2w'+pw)
__device__ __forceinline__ long class_row(const BArgs& a, int row) {
    const ConvGeom& g = a.g;
    int hw = g.Hc * g.Wc; int n = row / hw; int r = row - n * hw; int hc = r / g.Wc, wc = r - hc * g.Wc;
    return ((long)n * g.H + 2 * hc + g.ph) * g.W + 2 * wc + g.pw;
}

template <typename TC>
__device__ __forceinline__ void put(const BArgs& a, int row, int col, float v) {
    long orow = row;
    if (a.g.cls) orow = class_row(a, row);
    if (a.c_rows) { int r = a.c_rows[row]; if (r < 0) return; orow = r; }
    TC* p = reinterpret_cast<TC*>(a.C) + orow * a.ldc + col;
    if (a.accumulate) v += (float)*p;
    *p = (TC)ep_value(a, row, col, v);
}


// BatchNorm-backward statistics in the write phase of a data-gradient tile (BArgs::bn_x).  Every thread of the write loop owns one
// 8-column group of the tile (NTHR % (BN / 8) == 0) over BM * (BN / 8) / NTHR rows: it adds g and g * xhat of the values it has just
// stored (the bf16-rounded, accumulated ones: exactly what the BatchNorm backward will read back), the threads of a column group are
// combined by wave shuffles and through `sbuf` ([waves][BN][2] floats behind the staged tile) in a fixed order, and the tile's
// (sum g, sum g * xhat) per column lands in tile_stats[tile][column] like the forward statistics do.
// EN = false (the forward forms of the direct-to-LDS kernel, which never carry bn_x): nothing of this is compiled in - as a run-time "off" the
// struct still cost those launches its ~40 VGPRs, i.e. a resident workgroup per CU.
// PF = false (-DSAT_GLDS_BN_PREFETCH=0): the BatchNorm input and its sign bits are read inside the write loop instead of ahead of the k loop -
// 94 instead of 110 VGPRs, a fourth resident workgroup per CU for the 128 x 64 data-gradient launches.  Measured on one box, both builds: C2
// 20.30 (prefetch) vs 20.41 ms, C4 shard 33.30 vs 33.08: a wash, the prefetch stays.
template <int BM, int BN, int NTHR, bool EN = true, bool PF = true>
struct BnAcc {
    static constexpr int VPR = BN / 8, NWV = NTHR / 64, NJ0 = BM * VPR / NTHR;
    static constexpr bool BUILT = EN && NJ0 <= 8;          // the 256x256 form (16 segments per thread on top of 128 accumulator registers) is not: the launcher never asks it
    static constexpr int NJ = BUILT ? NJ0 : 1;
    static_assert(NTHR % VPR == 0 && 64 % VPR == 0, "a thread keeps one column group");
    float s1[8], s2[8], mu[8];          // sum g, sum g * (x - mean): the scale 1 / std is applied once per column in finish() (8 registers less in the write loop)
    bf16x8 xv[PF ? NJ : 1]; unsigned mb[PF ? NJ : 1];          // the thread's segments of the BatchNorm input and their sign bits, fetched before the write phase
    bool on;
    const BArgs* ap; int bm0, bn0, tid0;          // PF = false: where to read them from in add()
    __device__ __forceinline__ BnAcc(const BArgs& a, int bn, int tid) { on = BUILT && a.bn_x != nullptr && a.tile_stats != nullptr; }
    // the per-column constants and the accumulators come to life in front of the write loop, not in front of the k loop (32 registers that the
    // main loop does not have to carry)
    __device__ __forceinline__ void begin(const BArgs& a, int bn, int tid) {
        const int col = bn + (tid % VPR) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; mu[e] = (on && col + e < a.N) ? a.bn_mean[col + e] : 0.f; }
    }
    // issue the loads of every row segment this thread will write (rows past M: nothing to add, sign bits 0)
    __device__ __forceinline__ void prefetch(const BArgs& a, int bm, int bn, int tid) {
        if (!on) return;
        if constexpr (!PF) { ap = &a; bm0 = bm; bn0 = bn; tid0 = tid; return; }
#pragma unroll
        for (int j = 0; j < (PF ? NJ : 0); ++j) {
            const int v = tid + j * NTHR, row = bm + v / VPR, col = bn + (v % VPR) * 8;
            const bool ok = row < a.M && col < a.N;
            const long elem = ok ? (long)row * a.ldc + col : 0;
            xv[j] = *reinterpret_cast<const bf16x8*>(a.bn_x + elem);
            mb[j] = ok ? (a.bn_mask ? (unsigned)a.bn_mask[elem >> 3] : 0xFFu) : 0u;
        }
    }
    __device__ __forceinline__ void add(int j, const bf16x8& o) {
        if (!on || j >= NJ) return;
        bf16x8 x; unsigned m;
        if constexpr (PF) { x = xv[j]; m = mb[j]; }
        else {
            const BArgs& a = *ap;
            const int v = tid0 + j * NTHR, row = bm0 + v / VPR, col = bn0 + (v % VPR) * 8;
            const bool ok = row < a.M && col < a.N;
            const long elem = ok ? (long)row * a.ldc + col : 0;
            x = *reinterpret_cast<const bf16x8*>(a.bn_x + elem);
            m = ok ? (a.bn_mask ? (unsigned)a.bn_mask[elem >> 3] : 0xFFu) : 0u;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float g = ((m >> e) & 1u) ? (float)o[e] : 0.f;
            s1[e] += g; s2[e] = fmaf(g, (float)x[e] - mu[e], s2[e]);
        }
    }
    __device__ __forceinline__ void finish(const BArgs& a, float* sbuf, int tile, int bn, int tid) {
        if (!on) return;
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int o = VPR; o < 64; o <<= 1) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
        }
        if (lane < VPR) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { float* q = sbuf + ((long)wave * BN + lane * 8 + e) * 2; q[0] = s1[e]; q[1] = s2[e]; }
        }
        __syncthreads();
        if (tid < BN && bn + tid < a.N) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < NWV; ++w) { t1 += sbuf[((long)w * BN + tid) * 2]; t2 += sbuf[((long)w * BN + tid) * 2 + 1]; }
            float* o = a.tile_stats + ((long)tile * a.N + bn + tid) * 2;
            o[0] = t1; o[1] = t2 * a.bn_invstd[bn + tid];          // sum g * xhat = invstd * sum g * (x - mean)
        }
    }
};

// Accumulator tile -> C.  acc[i][j] is the 32x32 MFMA block (i, j) of this wave's (BM/2 x BN/2) quadrant at (wm, wn);
// C/D layout of v_mfma_f32_32x32x16_bf16: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
// `smem` is the kernel's operand staging area (free once the k loop is over), `smem_elems` its size in bf16.
template <int BM, int BN, typename TC, int SMEM_ELEMS, bool APF = false, bool BNS = true, bool BPF = true>
__device__ __forceinline__ void store_tile(const BArgs& a, f32x16 (&acc)[BM / 64][BN / 64], __bf16* smem, int bm, int bn, int bz, int wm, int wn,
                                           int tid, int lane, BnAcc<BM, BN, NT, BNS, BPF>& bacc, bool prefetched /* statistics operands fetched by the caller already */) {
    constexpr int TM = BM / 64, TN = BN / 64;
    const int li = lane & 31, lh = lane >> 5;
    if (sizeof(TC) == 2 && a.wide_store) {
        // bf16 result: the MFMA C layout gives each lane one column, i.e. 2-byte stores.  Stage the tile in LDS
        // (the operand buffers are free now) and write whole 16-byte row segments instead.
        constexpr int LDC = BN + 8;
        static_assert(BM * LDC <= SMEM_ELEMS, "C tile must fit the staging buffers");
        __bf16* cs = smem;
        if (!prefetched) bacc.prefetch(a, bm, bn, tid);           // in flight while the tile is staged
        bacc.begin(a, bn, tid);
        __syncthreads();
        if (a.tile_stats && !a.bn_x) {
            // BatchNorm statistics of the tile while it is still in registers (one HBM pass less for the layer that follows):
            // per output column the sum and the sum of squares of the ROUNDED (stored) values over this tile's rows, in fp32
            // (<= 128 terms each); the tiles are combined in double by bn_tile_reduce_kernel (encoder.hip).
            float* sbuf = reinterpret_cast<float*>(smem + BM * LDC);           // [2 wave rows][BN][2], behind the staged tile
            static_assert(BM * LDC * 2 + 4 * BN * 2 * 4 <= SMEM_ELEMS * 2 || SMEM_ELEMS == 0, "tile statistics must fit behind the staged C tile");
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = bm + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const float vb = (float)(__bf16)acc[i][j][r];
                        if (row < a.M) { s1 += vb; s2 = fmaf(vb, vb, s2); }
                    }
                s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                if (lh == 0) { float* q = sbuf + ((wm ? 1 : 0) * BN + wn + j * 32 + li) * 2; q[0] = s1; q[1] = s2; }
            }
            __syncthreads();
            if (tid < BN && bn + tid < a.N) {
                const float s1 = sbuf[tid * 2] + sbuf[(BN + tid) * 2], s2 = sbuf[tid * 2 + 1] + sbuf[(BN + tid) * 2 + 1];
                float* o = a.tile_stats + ((long)(bm / BM) * a.N + bn + tid) * 2;
                o[0] = s1; o[1] = s2;
            }
        }
        // Neighbouring lanes hold neighbouring columns.  Registers r, r+1 are rows R, R+1: the even lane of a pair
        // collects both columns of row R, the odd lane both columns of row R+1 (one DPP swap), so every lane writes
        // one packed 4-byte LDS word per register pair instead of two 2-byte ones.
        const bool odd = lane & 1;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int lr0 = wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, lc = wn + j * 32 + li;
                    float v0 = acc[i][j][r], v1 = acc[i][j][r + 1];
                    if (a.epi != EPI_NONE && bn + lc < a.N) {
                        if (bm + lr0 < a.M) v0 = ep_value(a, bm + lr0, bn + lc, v0);
                        if (bm + lr0 + 1 < a.M) v1 = ep_value(a, bm + lr0 + 1, bn + lc, v1);
                    }
                    const float give = odd ? v0 : v1;
                    const float got = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(give), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
                    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                    bf16x2 pk;
                    if (odd) { pk[0] = (__bf16)got; pk[1] = (__bf16)v1; } else { pk[0] = (__bf16)v0; pk[1] = (__bf16)got; }
                    *reinterpret_cast<bf16x2*>(cs + (lr0 + (odd ? 1 : 0)) * LDC + (lc & ~1)) = pk;
                }
        constexpr int VPR = BN / 8;
        // APF (accumulating launches, their own instantiation): the old values of this thread's segments, all in flight before the write loop -
        // inside it every load waits behind the previous iteration's store to the same array (the compiler must assume they alias).  The
        // registers cost the third workgroup per CU, which is why the non-accumulating launches keep the plain instantiation.
        constexpr int NJO = APF ? BM * VPR / NT : 1;
        bf16x8 oldv[NJO]; unsigned mbv[NJO];
        if constexpr (APF) {
            const __bf16* osrc = a.add_src ? a.add_src : reinterpret_cast<const __bf16*>(a.C);
#pragma unroll
            for (int j = 0; j < NJO; ++j) {
                const int v = tid + j * NT, row = bm + v / VPR, col = bn + (v % VPR) * 8;
                const bool ok = row < a.M && col < a.N;
                const long off = ok ? (a.g.cls ? class_row(a, row) : (long)row) * a.ldc + col : 0;
                oldv[j] = *reinterpret_cast<const bf16x8*>(osrc + off);
                mbv[j] = a.add_mask ? (unsigned)a.add_mask[off >> 3] : 0xFFu;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < BM * VPR / NT; ++j) {
            int v = tid + j * NT, lr = v / VPR, lc = (v % VPR) * 8;
            int row = bm + lr, col = bn + lc;
            if (row < a.M && col < a.N) {
                bf16x8 o = *reinterpret_cast<const bf16x8*>(cs + lr * LDC + lc);
                const long orow = a.g.cls ? class_row(a, row) : (long)row;
                __bf16* dst = reinterpret_cast<__bf16*>(a.C) + orow * a.ldc + col;
                if (a.accumulate) {
                    const long off = orow * a.ldc + col;
                    bf16x8 old; unsigned mb;
                    if constexpr (APF) { old = oldv[j]; mb = mbv[j]; }
                    else {
                        old = *reinterpret_cast<const bf16x8*>((a.add_src ? a.add_src : reinterpret_cast<const __bf16*>(a.C)) + off);
                        mb = a.add_mask ? (unsigned)a.add_mask[off >> 3] : 0xFFu;
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (__bf16)((float)o[e] + (((mb >> e) & 1u) ? (float)old[e] : 0.f));
                }
                *reinterpret_cast<bf16x8*>(dst) = o;
                bacc.add(j, o);
            }
        }
        bacc.finish(a, reinterpret_cast<float*>(smem + BM * LDC), bm / BM, bn, tid);
        return;
    }
    if ((a.nsplit > 1) ? a.wide_slab : (sizeof(TC) == 4 && a.wide_store)) {
        // fp32 result or split-K partial: the C layout gives each lane one column (4-byte stores).  Stage half the tile at a
        // time in LDS as fp32 and write 16-byte row segments; accumulate and the epilogue function run in the write phase
        // (same order as put(): C = f(C_old + acc)).
        constexpr int LDF = BN + 4, HALF = BM / 2, VPR = BN / 4;
        static_assert(HALF * LDF * 4 <= SMEM_ELEMS * 2, "half a fp32 C tile must fit the staging buffers");
        float* fs = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __syncthreads();
            if (wm == h * HALF) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            fs[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDF + wn + j * 32 + li] = acc[i][j][r];
            }
            __syncthreads();
#pragma unroll
            for (int jj = 0; jj < HALF * VPR / NT; ++jj) {
                const int v = tid + jj * NT, lr = v / VPR, lc = (v % VPR) * 4;
                const int row = bm + h * HALF + lr, col = bn + lc;
                if (row < a.M && col < a.N) {
                    float4 o = *reinterpret_cast<const float4*>(fs + lr * LDF + lc);
                    if (a.nsplit > 1) { *reinterpret_cast<float4*>(a.slab + ((long)bz * a.M + row) * a.N + col) = o; continue; }
                    const long orow = a.g.cls ? class_row(a, row) : (long)row;
                    float* dst = reinterpret_cast<float*>(a.C) + orow * a.ldc + col;
                    if (a.accumulate) { const float4 old = *reinterpret_cast<const float4*>(dst); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
                    if (a.epi != EPI_NONE) {
                        o.x = ep_value(a, row, col, o.x); o.y = ep_value(a, row, col + 1, o.y);
                        o.z = ep_value(a, row, col + 2, o.z); o.w = ep_value(a, row, col + 3, o.w);
                    }
                    *reinterpret_cast<float4*>(dst) = o;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = bm + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                int col = bn + wn + j * 32 + li;
                if (row < a.M && col < a.N) {
                    if (a.nsplit > 1) a.slab[((long)bz * a.M + row) * a.N + col] = acc[i][j][r];
                    else put<TC>(a, row, col, acc[i][j][r]);
                }
            }
}

// The same write-out for a workgroup of WR x WC waves (wave (wr, wc) owns the (BM / WR) x (BN / WC) block at (wm, wn) of the tile;
// acc[i][j] its 32x32 MFMA blocks): gemm_glds.hip's 8-wave tiles.  `smem` must hold the staged tile (see store_lds_bytes).
template <int BM, int BN, int WR, int WC, typename TC> constexpr size_t store_lds_bytes() {
    return sizeof(TC) == 2 ? (size_t)BM * (BN + 8) * 2 + (size_t)WR * WC * BN * 2 * 4 : (size_t)(BM / WR) * (BN + 4) * 4;
}
template <int BM, int BN, int WR, int WC, typename TC, bool BNS = true, bool BPF = true>
__device__ __forceinline__ void store_tile_w(const BArgs& a, f32x16 (&acc)[BM / WR / 32][BN / WC / 32], __bf16* smem, int bm, int bn, int bz, int wm, int wn,
                                             int wrow, int tid, int lane, BnAcc<BM, BN, 64 * WR * WC, BNS, BPF>& bacc, bool prefetched) {
    constexpr int TM = BM / WR / 32, TN = BN / WC / 32, NTH = 64 * WR * WC;
    const int li = lane & 31, lh = lane >> 5;
    if (sizeof(TC) == 2 && a.wide_store) {
        constexpr int LDC = BN + 8;
        __bf16* cs = smem;
        if (!prefetched) bacc.prefetch(a, bm, bn, tid);           // in flight while the tile is staged
        bacc.begin(a, bn, tid);
        __syncthreads();
        if (a.tile_stats && !a.bn_x) {
            float* sbuf = reinterpret_cast<float*>(smem + BM * LDC);           // [WR wave rows][BN][2], behind the staged tile
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = bm + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const float vb = (float)(__bf16)acc[i][j][r];
                        if (row < a.M) { s1 += vb; s2 = fmaf(vb, vb, s2); }
                    }
                s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                if (lh == 0) { float* q = sbuf + (wrow * BN + wn + j * 32 + li) * 2; q[0] = s1; q[1] = s2; }
            }
            __syncthreads();
            if (tid < BN && bn + tid < a.N) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int w = 0; w < WR; ++w) { s1 += sbuf[(w * BN + tid) * 2]; s2 += sbuf[(w * BN + tid) * 2 + 1]; }
                float* o = a.tile_stats + ((long)(bm / BM) * a.N + bn + tid) * 2;
                o[0] = s1; o[1] = s2;
            }
        }
        const bool odd = lane & 1;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int lr0 = wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, lc = wn + j * 32 + li;
                    float v0 = acc[i][j][r], v1 = acc[i][j][r + 1];
                    if (a.epi != EPI_NONE && bn + lc < a.N) {
                        if (bm + lr0 < a.M) v0 = ep_value(a, bm + lr0, bn + lc, v0);
                        if (bm + lr0 + 1 < a.M) v1 = ep_value(a, bm + lr0 + 1, bn + lc, v1);
                    }
                    const float give = odd ? v0 : v1;
                    const float got = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(give), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
                    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                    bf16x2 pk;
                    if (odd) { pk[0] = (__bf16)got; pk[1] = (__bf16)v1; } else { pk[0] = (__bf16)v0; pk[1] = (__bf16)got; }
                    *reinterpret_cast<bf16x2*>(cs + (lr0 + (odd ? 1 : 0)) * LDC + (lc & ~1)) = pk;
                }
        __syncthreads();
        constexpr int VPR = BN / 8;
#pragma unroll
        for (int j = 0; j < BM * VPR / NTH; ++j) {
            int v = tid + j * NTH, lr = v / VPR, lc = (v % VPR) * 8;
            int row = bm + lr, col = bn + lc;
            if (row < a.M && col < a.N) {
                bf16x8 o = *reinterpret_cast<const bf16x8*>(cs + lr * LDC + lc);
                const long orow = a.g.cls ? class_row(a, row) : (long)row;
                __bf16* dst = reinterpret_cast<__bf16*>(a.C) + orow * a.ldc + col;
                if (a.accumulate) {
                    const long off = orow * a.ldc + col;
                    bf16x8 old = *reinterpret_cast<const bf16x8*>((a.add_src ? a.add_src : reinterpret_cast<const __bf16*>(a.C)) + off);
                    const unsigned mb = a.add_mask ? (unsigned)a.add_mask[off >> 3] : 0xFFu;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (__bf16)((float)o[e] + (((mb >> e) & 1u) ? (float)old[e] : 0.f));
                }
                *reinterpret_cast<bf16x8*>(dst) = o;
                bacc.add(j, o);
            }
        }
        bacc.finish(a, reinterpret_cast<float*>(smem + BM * LDC), bm / BM, bn, tid);
        return;
    }
    if ((a.nsplit > 1) ? a.wide_slab : (sizeof(TC) == 4 && a.wide_store)) {
        // fp32 result or split-K partial: one wave row of the tile at a time through LDS, 16-byte row segments out
        constexpr int LDF = BN + 4, RB = BM / WR, VPR = BN / 4;
        float* fs = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int h = 0; h < WR; ++h) {
            __syncthreads();
            if (wrow == h) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            fs[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDF + wn + j * 32 + li] = acc[i][j][r];
            }
            __syncthreads();
#pragma unroll
            for (int jj = 0; jj < RB * VPR / NTH; ++jj) {
                const int v = tid + jj * NTH, lr = v / VPR, lc = (v % VPR) * 4;
                const int row = bm + h * RB + lr, col = bn + lc;
                if (row < a.M && col < a.N) {
                    float4 o = *reinterpret_cast<const float4*>(fs + lr * LDF + lc);
                    if (a.nsplit > 1) { *reinterpret_cast<float4*>(a.slab + ((long)bz * a.M + row) * a.N + col) = o; continue; }
                    const long orow = a.g.cls ? class_row(a, row) : (long)row;
                    float* dst = reinterpret_cast<float*>(a.C) + orow * a.ldc + col;
                    if (a.accumulate) { const float4 old = *reinterpret_cast<const float4*>(dst); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
                    if (a.epi != EPI_NONE) {
                        o.x = ep_value(a, row, col, o.x); o.y = ep_value(a, row, col + 1, o.y);
                        o.z = ep_value(a, row, col + 2, o.z); o.w = ep_value(a, row, col + 3, o.w);
                    }
                    *reinterpret_cast<float4*>(dst) = o;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = bm + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                int col = bn + wn + j * 32 + li;
                if (row < a.M && col < a.N) {
                    if (a.nsplit > 1) a.slab[((long)bz * a.M + row) * a.N + col] = acc[i][j][r];
                    else put<TC>(a, row, col, acc[i][j][r]);
                }
            }
}

template <typename TC>
__global__ void splitk_reduce_b_kernel(BArgs a) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)a.M * a.N;
    if (idx >= total) return;
    float v = 0.f;
    for (int z = 0; z < a.nsplit; ++z) v += a.slab[(long)z * total + idx];
    put<TC>(a, (int)(idx / a.N), (int)(idx % a.N), v);
}
// the same, four consecutive columns per thread (N % 4 == 0, slab 16-byte aligned): fixed z order, 16-byte loads
template <typename TC>
__global__ void splitk_reduce_b4_kernel(BArgs a) {
    const long idx4 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total4 = (long)a.M * a.N / 4;
    if (idx4 >= total4) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* s4 = reinterpret_cast<const float4*>(a.slab);
#pragma unroll 4
    for (int z = 0; z < a.nsplit; ++z) { const float4 t = s4[(long)z * total4 + idx4]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    const long e = idx4 * 4; const int row = (int)(e / a.N), col = (int)(e % a.N);
    if (sizeof(TC) == 4 && a.wide_store && !a.g.cls) {          // wide_store: fp32 C, no row scatter, 16-byte addressable rows
        float* dst = reinterpret_cast<float*>(a.C) + (long)row * a.ldc + col;
        if (a.accumulate) { const float4 old = *reinterpret_cast<const float4*>(dst); v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w; }
        if (a.epi != EPI_NONE) { v.x = ep_value(a, row, col, v.x); v.y = ep_value(a, row, col + 1, v.y); v.z = ep_value(a, row, col + 2, v.z); v.w = ep_value(a, row, col + 3, v.w); }
        *reinterpret_cast<float4*>(dst) = v;
    } else {
        put<TC>(a, row, col, v.x); put<TC>(a, row, col + 1, v.y); put<TC>(a, row, col + 2, v.z); put<TC>(a, row, col + 3, v.w);
    }
}

// Many splits into a small result (the stage-1 weight gradients: 100 - 250 slabs of 16 K elements): the loop above leaves a few
// dozen workgroups walking the slabs one after the other.  Here 16 lanes share a 16-byte output vector: lane l adds slabs l, l + 16,
// ... (in that order), the 16 partial sums are combined in lane order through LDS.  Fixed order, so still deterministic.
template <typename TC>
__global__ __launch_bounds__(256) void splitk_reduce_z16_kernel(BArgs a) {
    __shared__ float4 part[16][16];
    const int ev = threadIdx.x & 15, zl = threadIdx.x >> 4;
    const long idx4 = (long)blockIdx.x * 16 + ev;
    const long total4 = (long)a.M * a.N / 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx4 < total4) {
        const float4* s4 = reinterpret_cast<const float4*>(a.slab);
#pragma unroll 4
        for (int z = zl; z < a.nsplit; z += 16) { const float4 t = s4[(long)z * total4 + idx4]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    }
    part[zl][ev] = v;
    __syncthreads();
    if (zl != 0 || idx4 >= total4) return;
#pragma unroll
    for (int l = 1; l < 16; ++l) { const float4 t = part[l][ev]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    const long e = idx4 * 4; const int row = (int)(e / a.N), col = (int)(e % a.N);
    if (sizeof(TC) == 4 && a.wide_store && !a.g.cls) {
        float* dst = reinterpret_cast<float*>(a.C) + (long)row * a.ldc + col;
        if (a.accumulate) { const float4 old = *reinterpret_cast<const float4*>(dst); v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w; }
        if (a.epi != EPI_NONE) { v.x = ep_value(a, row, col, v.x); v.y = ep_value(a, row, col + 1, v.y); v.z = ep_value(a, row, col + 2, v.z); v.w = ep_value(a, row, col + 3, v.w); }
        *reinterpret_cast<float4*>(dst) = v;
    } else {
        put<TC>(a, row, col, v.x); put<TC>(a, row, col + 1, v.y); put<TC>(a, row, col + 2, v.z); put<TC>(a, row, col + 3, v.w);
    }
}

// the reduction launch after a split-K GEMM
template <typename TC>
inline int launch_splitk_reduce(const BArgs& k, hipStream_t st) {
    const long total = (long)k.M * k.N;
    const int z16 = dev_switch(SW_REDUCE_Z16);
    if (k.wide_slab && z16 && k.nsplit >= 32 && total / 4 / 256 < 256)
        hipLaunchKernelGGL(splitk_reduce_z16_kernel<TC>, dim3(cdiv(total / 4, 16)), dim3(256), 0, st, k);
    else if (k.wide_slab) hipLaunchKernelGGL(splitk_reduce_b4_kernel<TC>, dim3(cdiv(total / 4, 256)), dim3(256), 0, st, k);
    else hipLaunchKernelGGL(splitk_reduce_b_kernel<TC>, dim3(cdiv(total, 256)), dim3(256), 0, st, k);
    return launch_ok("splitk_reduce_b");
}

// XCD-aware tile order: the dispatcher deals workgroups round-robin over the 8 XCDs (each with its own L2), so
// workgroups i and i+8 share an L2.  Give every XCD a contiguous run of logical tiles (x fastest): the column
// tiles that re-read the same activation rows then hit one L2 instead of eight.  Bijective for any grid size;
// placement only affects speed, never results.
// y_fastest: a few row tiles against many column tiles (the decoder's per-step products: 640 caption rows x 2048 - 2688 weight rows).  With the
// column tiles fastest an XCD's run of ~50 tiles spans EVERY column tile of one or two row panels: each of the eight L2s pulls the whole weight
// matrix (8 x 2.75 MB for a 3.4 MB problem).  Row tiles fastest gives an XCD a few column tiles of all row panels: its slice of the weights
// plus the (small) activation matrix (16.5 -> 15.6 us and 14.0 -> 13.0 us for the two forward products of a C2 decode step).
__device__ __forceinline__ void xcd_tile(int& bx, int& by, int& bz, int y_fastest = 0) {
    const unsigned total = gridDim.x * gridDim.y * gridDim.z;
    const unsigned orig = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const unsigned q = total >> 3, r = total & 7, xcd = orig & 7, idx = orig >> 3;
    const unsigned lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    if (y_fastest) { by = lin % gridDim.y; const unsigned t2 = lin / gridDim.y; bx = t2 % gridDim.x; bz = t2 / gridDim.x; }
    else { bx = lin % gridDim.x; const unsigned t2 = lin / gridDim.x; by = t2 % gridDim.y; bz = t2 / gridDim.y; }
}

// gemm_glds.hip: direct-to-LDS operand staging; returns -1 when the problem does not fit its forms (caller falls through)
// *bm_used (if given): rows per tile of the kernel that ran (what the per-tile statistics are indexed by)
int launch_gemm_glds(const BArgs& k, int amode, int bmode, int c_bf16, int BMt, hipStream_t st, int* bm_used = nullptr);
int& glds_force_tile();      // dev switches (sat_debug_option)
int& glds_stages8();
int& glds_tall_k();
int& glds_tall_conv();
int& glds_ablate();
int& gemm_tile_override();

}  // namespace sat
