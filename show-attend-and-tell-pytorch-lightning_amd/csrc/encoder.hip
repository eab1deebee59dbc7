// Encoder (ResNet trunk, reference model.py:16-63 via torchvision) on gfx950: NHWC activations,
// KRSC filters.  Convolutions are implicit GEMMs on the MFMA kernel of gemm.hip (forward, data
// gradient, weight gradient); this file adds the memory-bound layers around them: input
// normalisation (model.py:59), train-mode BatchNorm (+ReLU, +residual) forward/backward,
// 3x3/2 max-pool, adaptive average pool / bilinear resize of the `encoder_size` option (readme.md:118-121).
#include <stdlib.h>
#include "../../include/sat_hip.h"
#include "common.h"
#include "gemm.h"
#include "profile.h"

namespace sat {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// 4 consecutive activation elements as fp32, whatever the storage type (fp32: 16 B, bf16: 8 B)
template <typename T> __device__ __forceinline__ float4 ld4(const T* p, long i4);
template <> __device__ __forceinline__ float4 ld4<float>(const float* p, long i4) { return reinterpret_cast<const float4*>(p)[i4]; }
template <> __device__ __forceinline__ float4 ld4<__bf16>(const __bf16* p, long i4) {
    bf16x4 v = reinterpret_cast<const bf16x4*>(p)[i4];
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
template <typename T> __device__ __forceinline__ void st4(T* p, long i4, float4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, long i4, float4 v) { reinterpret_cast<float4*>(p)[i4] = v; }
template <> __device__ __forceinline__ void st4<__bf16>(__bf16* p, long i4, float4 v) {
    bf16x4 o; o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    reinterpret_cast<bf16x4*>(p)[i4] = o;
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long n4) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n4) st4<__bf16>(dst, e, reinterpret_cast<const float4*>(src)[e]);
}
// NCHW [0,1] fp32 -> normalised NHWC bf16 with C padded 3 -> 8 (16-byte pixels for the bf16 implicit GEMM)
__global__ void normalize_nhwc8_bf16_kernel(const float* __restrict__ img, __bf16* __restrict__ out, int H, int W, long total,
                                            float m0, float m1, float m2, float s0, float s1, float s2) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    long hw = (long)H * W; long n = e / hw, p = e - n * hw;
    const float* src = img + n * 3 * hw + p;
    st4<__bf16>(out, 2 * e, make_float4((src[0] - m0) / s0, (src[hw] - m1) / s1, (src[2 * hw] - m2) / s2, 0.f));
    st4<__bf16>(out, 2 * e + 1, make_float4(0.f, 0.f, 0.f, 0.f));
}
// stem filters (K,R,S,3) fp32 -> (K,R,S,8) bf16, and the fp32 (K,R,S,8) gradient back to (K,R,S,3)
__global__ void pad_c3_to_c8_bf16_kernel(const float* __restrict__ w3, __bf16* __restrict__ w8, long n) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    st4<__bf16>(w8, 2 * e, make_float4(w3[3 * e], w3[3 * e + 1], w3[3 * e + 2], 0.f));
    st4<__bf16>(w8, 2 * e + 1, make_float4(0.f, 0.f, 0.f, 0.f));
}
__global__ void unpad_c8_to_c3_kernel(const float* __restrict__ w8, float* __restrict__ w3, long n) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    w3[3 * e] = w8[8 * e]; w3[3 * e + 1] = w8[8 * e + 1]; w3[3 * e + 2] = w8[8 * e + 2];
}

// bf16 stem, tap-pair layout (include/sat_hip.h): normalised image zero-padded by 3, 4 stored channels
__global__ void normalize_nhwc4_padded_bf16_kernel(const float* __restrict__ img, __bf16* __restrict__ out, int H, int W, long total,
                                                   float m0, float m1, float m2, float s0, float s1, float s2) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;          // over padded pixels
    if (e >= total) return;
    const int Wp = W + 6, Hp = H + 6;
    const unsigned eu = (unsigned)e;                    // total < 2^32 (checked by the launcher)
    const int xp = (int)(eu % Wp); const unsigned t = eu / Wp; const int yp = (int)(t % Hp); const long n = t / Hp;
    const int x = xp - 3, y = yp - 3;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) {
        const long hw = (long)H * W; const float* src = img + n * 3 * hw + (long)y * W + x;
        v = make_float4((src[0] - m0) / s0, (src[hw] - m1) / s1, (src[2 * hw] - m2) / s2, 0.f);
    }
    st4<__bf16>(out, e, v);
}
__global__ void stem_filter_pairs_kernel(const float* __restrict__ w3, __bf16* __restrict__ wp, long n) {      // n = K * 7 * 4 pairs
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int sp = (int)(e % 4); const long kr = e / 4;            // (k, r) row of seven taps
    const float* a = w3 + (kr * 7 + 2 * sp) * 3;
    st4<__bf16>(wp, 2 * e, make_float4(a[0], a[1], a[2], 0.f));
    st4<__bf16>(wp, 2 * e + 1, sp < 3 ? make_float4(a[3], a[4], a[5], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f));
}
__global__ void stem_filter_grad_unpairs_kernel(const float* __restrict__ dwp, float* __restrict__ dw3, long n) {   // n = K * 7 * 7 taps
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int s = (int)(e % 7); const long kr = e / 7;
    const float* a = dwp + ((kr * 4 + (s >> 1)) * 8 + (s & 1) * 4);
    dw3[3 * e] = a[0]; dw3[3 * e + 1] = a[1]; dw3[3 * e + 2] = a[2];
}

// ------------------------------------------------------------------ input: NCHW [0,1] -> normalised NHWC, C padded 3 -> 4
__global__ void normalize_nhwc4_kernel(const float* __restrict__ img, float* __restrict__ out, int H, int W, long total,
                                       float m0, float m1, float m2, float s0, float s1, float s2) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;     // one output pixel (4 channels)
    if (e >= total) return;
    long hw = (long)H * W; long n = e / hw, p = e - n * hw;
    const float* src = img + n * 3 * hw + p;
    float4 v;
    v.x = (src[0] - m0) / s0; v.y = (src[hw] - m1) / s1; v.z = (src[2 * hw] - m2) / s2; v.w = 0.f;
    reinterpret_cast<float4*>(out)[e] = v;
}

// stem filters (K,R,S,3) <-> (K,R,S,4)
__global__ void pad_c3_to_c4_kernel(const float* __restrict__ w3, float* __restrict__ w4, long n) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    reinterpret_cast<float4*>(w4)[e] = make_float4(w3[3 * e], w3[3 * e + 1], w3[3 * e + 2], 0.f);
}
__global__ void unpad_c4_to_c3_kernel(const float* __restrict__ w4, float* __restrict__ w3, long n) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float4 v = reinterpret_cast<const float4*>(w4)[e];
    w3[3 * e] = v.x; w3[3 * e + 1] = v.y; w3[3 * e + 2] = v.z;
}

// ------------------------------------------------------------------ BatchNorm (training mode)
// Column statistics of a (rows x C) activation matrix, streamed at 16 bytes per lane.
//   MODE 0: sum (x - s), sum (x - s)^2 with the shift s = first row (keeps the variance well conditioned)
//   MODE 1: sum g, sum g * xhat          with g = dy * (relu ? y > 0 : 1)
// Block = 256 threads = CV vector columns x RL row lanes; each block owns a row chunk and writes one partial
// per channel.  Accumulation is in double (these kernels are HBM-bound; the fp64 adds are free and the sums
// then match ATen's CPU batch-norm, which accumulates float inputs in double).  Fixed order => deterministic.
template <typename T> struct EPT { static constexpr int n = 16 / sizeof(T); };          // elements per 16-byte vector
template <typename T, int N> __device__ __forceinline__ void ldv(const T* p, long iv, float (&o)[N]);
template <> __device__ __forceinline__ void ldv<float, 4>(const float* p, long iv, float (&o)[4]) {
    float4 v = reinterpret_cast<const float4*>(p)[iv]; o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <> __device__ __forceinline__ void ldv<__bf16, 8>(const __bf16* p, long iv, float (&o)[8]) {
    typedef __bf16 b8 __attribute__((ext_vector_type(8)));
    b8 v = reinterpret_cast<const b8*>(p)[iv];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
}

// 16 packed bytes -> N floats (fp32: 4, bf16: 8)
template <typename T, int N> __device__ __forceinline__ void unpack(const uint4& r, float (&o)[N]);
template <> __device__ __forceinline__ void unpack<float, 4>(const uint4& r, float (&o)[4]) {
    o[0] = __uint_as_float(r.x); o[1] = __uint_as_float(r.y); o[2] = __uint_as_float(r.z); o[3] = __uint_as_float(r.w);
}
template <> __device__ __forceinline__ void unpack<__bf16, 8>(const uint4& r, float (&o)[8]) {
    o[0] = __uint_as_float(r.x << 16); o[1] = __uint_as_float(r.x & 0xFFFF0000u); o[2] = __uint_as_float(r.y << 16); o[3] = __uint_as_float(r.y & 0xFFFF0000u);
    o[4] = __uint_as_float(r.z << 16); o[5] = __uint_as_float(r.z & 0xFFFF0000u); o[6] = __uint_as_float(r.w << 16); o[7] = __uint_as_float(r.w & 0xFFFF0000u);
}
// ReLU sign mask written by the forward BatchNorm+ReLU: bit i of byte b <-> element 8b + i of the (rows x C) activation.
// The backward passes read it instead of the bf16 / fp32 output tensor (1/16 .. 1/32 of the bytes).  N = elements per vector.
template <int N>
__device__ __forceinline__ void mask_to_floats(const unsigned char* __restrict__ m, long iv, float (&o)[N]) {
    const unsigned b = (N == 8) ? m[iv] : (unsigned)(m[iv >> 1] >> ((iv & 1) * 4));
#pragma unroll
    for (int i = 0; i < N; ++i) o[i] = (b >> i) & 1u ? 1.f : 0.f;
}

#ifndef UNROLL_BWD_BF16
#define UNROLL_BWD_BF16 2
#endif
template <int MODE, typename T>
__global__ __launch_bounds__(256) void bn_colstats_kernel(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ y,
                                                          const unsigned char* __restrict__ rmask,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd, int relu,
                                                          long rows, int C, int CV, long rows_per, double* __restrict__ part0, double* __restrict__ part1) {
    constexpr int E = EPT<T>::n;
    constexpr int UN = (MODE == 1 && E == 8) ? UNROLL_BWD_BF16 : 4;          // rows in flight per thread
    __shared__ double sh[2][256][E > 4 ? 4 : E];      // reduced in two halves when E == 8
    const int tid = threadIdx.x, tc = tid % CV, tr = tid / CV, RL = 256 / CV;
    const int cv = blockIdx.x * CV + tc;              // vector column
    const int CVT = C / E;
    double a0[E], a1[E];
#pragma unroll
    for (int i = 0; i < E; ++i) { a0[i] = 0.0; a1[i] = 0.0; }
    const bool act = cv < CVT && tr < RL;
    if (act) {
        float mu[E], is[E];
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < E; ++i) { mu[i] = mean[cv * E + i]; is[i] = invstd[cv * E + i]; }
        } else ldv<T, E>(x, cv, mu);                 // shift = row 0
        // Row groups of UN * RL rows are dealt to the row parts round-robin (part y takes groups y, y + parts, ...): at any moment
        // the whole grid sweeps one contiguous window of the activation, like the apply kernels do, instead of `parts` separate streams
        const long G = (long)UN * RL, ngroups = rows / G;
        const bool interleave = rows_per < 0;
        long r0 = interleave ? 0 : (long)blockIdx.y * rows_per, r1 = interleave ? 0 : r0 + rows_per; if (r1 > rows) r1 = rows;
        long r = r0 + tr;
        const long gstart = interleave ? blockIdx.y : 0, gstride = interleave ? gridDim.y : 1;
        for (long gi = gstart; interleave ? gi < ngroups : r + (UN - 1L) * RL < r1; gi += gstride, r += (long)UN * RL) {
            if (interleave) r = gi * G + tr;     // UN independent 16-byte loads in flight per operand, kept packed until used
            uint4 xr[UN], gr[UN], yr[UN]; unsigned mb[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const long iv = (r + (long)u * RL) * CVT + cv;
                xr[u] = reinterpret_cast<const uint4*>(x)[iv];
                if (MODE == 1) {
                    gr[u] = reinterpret_cast<const uint4*>(dy)[iv];
                    if (relu) {
                        if (rmask) mb[u] = (E == 8) ? rmask[iv] : (unsigned)(rmask[iv >> 1] >> ((iv & 1) * 4));
                        else yr[u] = reinterpret_cast<const uint4*>(y)[iv];
                    }
                }
            }
            // the four rows of a group are summed in fp32 (4 terms: ~1 ulp), the groups in double
            float s0[E], s1[E];
#pragma unroll
            for (int i = 0; i < E; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                float xv[E], gv[E], yv[E];
                unpack<T, E>(xr[u], xv);
                if (MODE == 1) {
                    unpack<T, E>(gr[u], gv);
                    if (relu) {
                        if (rmask) {
#pragma unroll
                            for (int i = 0; i < E; ++i) yv[i] = (mb[u] >> i) & 1u ? 1.f : 0.f;
                        } else unpack<T, E>(yr[u], yv);
                    }
                }
#pragma unroll
                for (int i = 0; i < E; ++i) {
                    if (MODE == 0) { float d = xv[i] - mu[i]; s0[i] += d; s1[i] = fmaf(d, d, s1[i]); }
                    else { float g = (relu && !(yv[i] > 0.f)) ? 0.f : gv[i]; s0[i] += g; s1[i] = fmaf(g, (xv[i] - mu[i]) * is[i], s1[i]); }
                }
            }
#pragma unroll
            for (int i = 0; i < E; ++i) { a0[i] += (double)s0[i]; a1[i] += (double)s1[i]; }
        }
        if (interleave) { r = ngroups * G + tr; r1 = (blockIdx.y == gridDim.y - 1) ? rows : 0; }
        for (; r < r1; r += RL) {
            float xv[E], gv[E], yv[E];
            ldv<T, E>(x, r * CVT + cv, xv);
            if (MODE == 1) {
                ldv<T, E>(dy, r * CVT + cv, gv);
                if (relu) { if (rmask) mask_to_floats<E>(rmask, r * CVT + cv, yv); else ldv<T, E>(y, r * CVT + cv, yv); }
            }
#pragma unroll
            for (int i = 0; i < E; ++i) {
                if (MODE == 0) { float d = xv[i] - mu[i]; a0[i] += d; a1[i] += (double)d * d; }
                else { float g = (relu && !(yv[i] > 0.f)) ? 0.f : gv[i]; a0[i] += g; a1[i] += (double)g * ((xv[i] - mu[i]) * is[i]); }
            }
        }
    }
    // combine the RL row lanes in a fixed order, 4 channels at a time
#pragma unroll
    for (int h = 0; h < E; h += 4) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) { sh[0][tid][i] = a0[h + i]; sh[1][tid][i] = a1[h + i]; }
        __syncthreads();
        if (tr == 0 && cv < CVT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double s0 = 0.0, s1 = 0.0;
                for (int k = 0; k < RL; ++k) { s0 += sh[0][k * CV + tc][i]; s1 += sh[1][k * CV + tc][i]; }
                part0[(long)blockIdx.y * C + cv * E + h + i] = s0;
                part1[(long)blockIdx.y * C + cv * E + h + i] = s1;
            }
        }
    }
}

// Column statistics that the convolution epilogue left per row tile (gemm_bf16_common.h: fp32 sum and sum of squares of the
// stored values per tile and channel) -> the partial layout of bn_colstats_kernel<0>: part0/part1 [nparts][C] in double,
// unshifted.  Block = 32 channels x 8 tile lanes, tiles of a part in a fixed order.
__global__ __launch_bounds__(256) void bn_tile_reduce_kernel(const float* __restrict__ tiles, int ntiles, int C, int tiles_per, double* __restrict__ part0,
                                                             double* __restrict__ part1) {
    __shared__ double sh[2][8][32];
    const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    const int t0 = blockIdx.y * tiles_per, t1 = min(ntiles, t0 + tiles_per);
    double s = 0.0, q = 0.0;
    if (c < C)
        for (int t = t0 + pl; t < t1; t += 8) { const float2 v = reinterpret_cast<const float2*>(tiles)[(long)t * C + c]; s += (double)v.x; q += (double)v.y; }
    sh[0][pl][cl] = s; sh[1][pl][cl] = q;
    __syncthreads();
    if (pl == 0 && c < C) {
        double ss = 0.0, qq = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { ss += sh[0][k][cl]; qq += sh[1][k][cl]; }
        part0[(long)blockIdx.y * C + c] = ss; part1[(long)blockIdx.y * C + c] = qq;
    }
}

// Finalise: one 64-lane wave per channel sums the partials (lane-strided, then a fixed xor tree).
// forward : mean / biased variance -> invstd, running stats (momentum, unbiased variance) as nn.BatchNorm2d
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <typename T>
__global__ void bn_fwd_finalize_kernel(const double* __restrict__ part0, const double* __restrict__ part1, const T* __restrict__ x,
                                       int nparts, int C, long rows, float eps, float momentum, float* __restrict__ mean,
                                       float* __restrict__ invstd, float* __restrict__ running_mean, float* __restrict__ running_var) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int p = lane; p < nparts; p += 64) { s += part0[(long)p * C + c]; q += part1[(long)p * C + c]; }
    s = wave_sum_d(s); q = wave_sum_d(q);
    if (lane == 0) {
        const double n = (double)rows, shift = x ? (double)(float)x[c] : 0.0;      // x == NULL: unshifted sums (conv-epilogue statistics)
        double var = (q - s * s / n) / n; if (var < 0.0) var = 0.0;
        const double mu = shift + s / n;
        mean[c] = (float)mu;
        invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            double unb = rows > 1 ? var * n / (n - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
        }
    }
}
// Few row tiles (the 16x16 and 8x8 stages): tile reduce and finalize in ONE launch - block = 32 channels x 8 tile lanes, the
// lanes' double sums combined in a fixed order, then the channel's thread writes mean / invstd / running statistics.
__global__ __launch_bounds__(256) void bn_tile_finalize_kernel(const float* __restrict__ tiles, int ntiles, int C, long rows, float eps, float momentum,
                                                               float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ running_mean,
                                                               float* __restrict__ running_var) {
    __shared__ double sh[2][8][32];
    const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    double s = 0.0, q = 0.0;
    if (c < C)
        for (int t0 = pl; t0 < ntiles; t0 += 32) {          // four tiles per lane in flight
            float2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int t = t0 + 8 * u; v[u] = t < ntiles ? reinterpret_cast<const float2*>(tiles)[(long)t * C + c] : make_float2(0.f, 0.f); }
#pragma unroll
            for (int u = 0; u < 4; ++u) { s += (double)v[u].x; q += (double)v[u].y; }
        }
    sh[0][pl][cl] = s; sh[1][pl][cl] = q;
    __syncthreads();
    if (pl == 0 && c < C) {
        double ss = 0.0, qq = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { ss += sh[0][k][cl]; qq += sh[1][k][cl]; }
        const double n = (double)rows;
        double var = (qq - ss * ss / n) / n; if (var < 0.0) var = 0.0;
        const double mu = ss / n;
        mean[c] = (float)mu;
        invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unb = rows > 1 ? var * n / (n - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
        }
    }
}
// Tile statistics -> final per-channel values in ONE launch at any tile count (default; SAT_BN_ONEPASS=0 restores the reduce + finalize pair).
// A block owns CPB channels (one CPB * 8-byte run of the [tile][channel](sum, sq) array) and 256 / CPB tile lanes with U loads in flight each,
// so even the 56x56 stage (1568 tiles) is two round trips long; lanes are combined by a fixed xor tree, the waves through LDS in wave
// order - the result does not depend on timing.  Blocks stay at 256 threads: a 1024-thread block waits for a whole free CU, and with the
// weight-gradient GEMMs running beside it on the side stream that wait was 20 - 100 us (measured).
// mode 0: mean / invstd (+ running statistics), mode 1: dbeta = sum g, dgamma = sum g xhat.
template <int CPB, int U>
__global__ __launch_bounds__(256) void bn_tile_finish_kernel(const float* __restrict__ tiles, int ntiles, int C, int mode, long rows, float eps, float momentum,
                                                             float* __restrict__ out0, float* __restrict__ out1, float* __restrict__ running_mean,
                                                             float* __restrict__ running_var) {
    constexpr int NT = 256, TL = NT / CPB, NW = NT / 64;
    __shared__ double sh[2][NW][CPB];
    const int cl = threadIdx.x % CPB, tl = threadIdx.x / CPB, c = blockIdx.x * CPB + cl, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s = 0.0, q = 0.0;
    if (c < C)
        for (int t0 = tl; t0 < ntiles; t0 += TL * U) {
            float2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int t = t0 + TL * u; v[u] = t < ntiles ? reinterpret_cast<const float2*>(tiles)[(long)t * C + c] : make_float2(0.f, 0.f); }
#pragma unroll
            for (int u = 0; u < U; ++u) { s += (double)v[u].x; q += (double)v[u].y; }
        }
#pragma unroll
    for (int o = CPB; o < 64; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    if (lane < CPB) { sh[0][wave][lane] = s; sh[1][wave][lane] = q; }
    __syncthreads();
    if (threadIdx.x >= CPB || c >= C) return;
    double ss = 0.0, qq = 0.0;
#pragma unroll
    for (int k = 0; k < NW; ++k) { ss += sh[0][k][cl]; qq += sh[1][k][cl]; }
    if (mode == 1) { out0[c] = (float)ss; out1[c] = (float)qq; return; }
    const double n = (double)rows;
    double var = (qq - ss * ss / n) / n; if (var < 0.0) var = 0.0;
    const double mu = ss / n;
    out0[c] = (float)mu;
    out1[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unb = rows > 1 ? var * n / (n - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}
// measured per launch inside the C2 step: <8, 8> 5.7 us (fused kernel before: 9.1), <4, 16> 9.3 us (reduce + finalize pair: 10.8 forward, 16.4
// backward); above 2048 tiles a 2-channel form took 20 us against the pair's 11 - 16, so those keep the pair.
static inline bool bn_tile_finish_ok(int ntiles) { return ntiles <= 2048; }
static inline void launch_bn_tile_finish(const float* tiles, int ntiles, int C, int mode, long rows, float eps, float momentum, float* out0, float* out1,
                                         float* running_mean, float* running_var, hipStream_t st) {
    if (ntiles > 256)
        hipLaunchKernelGGL((bn_tile_finish_kernel<4, 16>), dim3(cdiv(C, 4)), dim3(256), 0, st, tiles, ntiles, C, mode, rows, eps, momentum, out0, out1, running_mean, running_var);
    else
        hipLaunchKernelGGL((bn_tile_finish_kernel<8, 8>), dim3(cdiv(C, 8)), dim3(256), 0, st, tiles, ntiles, C, mode, rows, eps, momentum, out0, out1, running_mean, running_var);
}
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ part0, const double* __restrict__ part1, int nparts, int C,
                                       float* __restrict__ dbeta, float* __restrict__ dgamma) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int p0 = lane; p0 < nparts; p0 += 256) {          // four partials per lane in flight: the kernel is one load latency long, not twelve
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + 64 * u; const bool ok = p < nparts;
            a[u] = ok ? part0[(long)p * C + c] : 0.0; b[u] = ok ? part1[(long)p * C + c] : 0.0;
        }
        s += (a[0] + a[1]) + (a[2] + a[3]); q += (b[0] + b[1]) + (b[2] + b[3]);
    }
    s = wave_sum_d(s); q = wave_sum_d(q);
    if (lane == 0) { dbeta[c] = (float)s; dgamma[c] = (float)q; }
}

// y = (x - mean) * invstd * gamma + beta (+ residual) (ReLU), 16 bytes per lane (E = 4 fp32 / 8 bf16 elements).
// eval mode (var_eps >= 0) passes the running variance in `invstd` and the kernel takes 1/sqrt(var + eps) itself.
template <typename T, int N> __device__ __forceinline__ uint4 pack(const float (&v)[N]);
template <> __device__ __forceinline__ uint4 pack<float, 4>(const float (&v)[4]) {
    return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
}
template <> __device__ __forceinline__ uint4 pack<__bf16, 8>(const float (&v)[8]) {
    typedef __bf16 b8 __attribute__((ext_vector_type(8)));
    b8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (__bf16)v[i];
    return *reinterpret_cast<uint4*>(&o);
}
// rbn (train mode): the residual is itself the INPUT of a BatchNorm (the projection shortcut) whose mean / invstd / gamma / beta follow in rbn -
// its normalised value is formed here, rounded to the storage type as if it had been written out, and never touches HBM.
struct ResBn { const float* mean; const float* invstd; const float* gamma; const float* beta; };
template <typename T, bool EVAL, bool RBN = false, int VPT = 1>
__global__ void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ invstd,
                                const float* __restrict__ gamma, const float* __restrict__ beta, const T* __restrict__ res, int relu,
                                T* __restrict__ y, unsigned char* __restrict__ rmask, long totalv, int CV, float var_eps, ResBn rbn = ResBn{nullptr, nullptr, nullptr, nullptr},
                                long part = 0) {
    constexpr int E = EPT<T>::n;
    if (VPT == 1) part = totalv;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= part) return;
    const int c0 = (totalv <= 0xffffffffL ? (int)((unsigned)e % (unsigned)CV) : (int)(e % CV)) * E;      // a 64-bit modulo costs ~100 instructions per lane
    // VPT vectors per thread, `part` vectors apart (a multiple of CV: same channels), all loads issued before the first use
    uint4 xr[VPT], rr[VPT];
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const long idx = e + k * part;
        if (idx < totalv) { xr[k] = reinterpret_cast<const uint4*>(x)[idx]; if (res) rr[k] = reinterpret_cast<const uint4*>(res)[idx]; }
    }
    float isd[E], gm[E], mu[E], bt[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        float is = invstd[c0 + i];
        if (EVAL) is = 1.f / sqrtf(is + var_eps);      // eval mode only: the training pass gets invstd from the finalize kernel
        isd[i] = is; gm[i] = gamma[c0 + i]; mu[i] = mean[c0 + i]; bt[i] = beta[c0 + i];
    }
    float rmu[RBN ? E : 1], ris[RBN ? E : 1], rgm[RBN ? E : 1], rbt[RBN ? E : 1];
    if (RBN) {          // (its own instantiation: as a run-time branch it cost the plain kernel 4x - 157 instead of 36 us per launch)
#pragma unroll
        for (int i = 0; i < E; ++i) { rmu[i] = rbn.mean[c0 + i]; ris[i] = rbn.invstd[c0 + i]; rgm[i] = rbn.gamma[c0 + i]; rbt[i] = rbn.beta[c0 + i]; }
    }
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const long idx = e + k * part;
        if (idx >= totalv) continue;
        float xv[E], rv[E], o[E];
        unpack<T, E>(xr[k], xv);
        if (res) unpack<T, E>(rr[k], rv);
        if (RBN) {
#pragma unroll
            for (int i = 0; i < E; ++i) rv[i] = (float)(T)((rv[i] - rmu[i]) * ris[i] * rgm[i] + rbt[i]);
        }
        unsigned bits = 0;
        if (relu == 2) {          // ReLU6 (mobilenet_v2), a wave-uniform branch: clamp to [0, 6]; the mask bit = "the gradient passes" = 0 < v < 6, hardtanh's rule
#pragma unroll
            for (int i = 0; i < E; ++i) {
                float v = (xv[i] - mu[i]) * isd[i] * gm[i] + bt[i];
                if (res) v += rv[i];
                bits |= ((v > 0.f && v < 6.f) ? 1u : 0u) << i;
                o[i] = v < 0.f ? 0.f : (v > 6.f ? 6.f : v);
            }
        } else {
#pragma unroll
            for (int i = 0; i < E; ++i) {
                float v = (xv[i] - mu[i]) * isd[i] * gm[i] + bt[i];          // this association everywhere (the stem-tail kernel must match bit for bit)
                if (res) v += rv[i];
                bits |= (v > 0.f ? 1u : 0u) << i;
                o[i] = relu ? (v < 0.f ? 0.f : v) : v;            // like torch's ReLU a NaN stays a NaN (fmaxf would turn it into 0 and hide it)
            }
        }
        if (rmask) {
            if (E == 8) rmask[idx] = (unsigned char)bits;
            else {               // fp32: two lanes make a byte (totalv and part are even: the mask needs C % 8 == 0)
                const unsigned other = __shfl_xor(bits, 1, 64);
                if (!(threadIdx.x & 1)) rmask[idx >> 1] = (unsigned char)(bits | (other << 4));
            }
        }
        reinterpret_cast<uint4*>(y)[idx] = pack<T, E>(o);
    }
}

// dx = gamma * invstd * (g - dbeta/M - xhat * dgamma/M);  dres (optional) receives g (the masked upstream gradient)
// VPT vectors per thread, `part` vectors apart (a multiple of CV: the same channels, so the per-channel factors are formed once), every load
// issued before the first use.
template <typename T, int VPT>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ y,
                                    const unsigned char* __restrict__ rmask, const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ gamma, const float* __restrict__ dbeta, const float* __restrict__ dgamma, int relu,
                                    float inv_rows, T* __restrict__ dx, T* __restrict__ dres, int dres_accumulate, long totalv, int CV, long part) {
    constexpr int E = EPT<T>::n;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= part) return;
    const int c0 = (totalv <= 0xffffffffL ? (int)((unsigned)e % (unsigned)CV) : (int)(e % CV)) * E;      // a 64-bit modulo costs ~100 instructions per lane
    uint4 xr[VPT], gr[VPT], yr[VPT], rr[VPT]; unsigned mb[VPT];
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const long idx = e + k * part;
        if (idx < totalv) {
            xr[k] = reinterpret_cast<const uint4*>(x)[idx]; gr[k] = reinterpret_cast<const uint4*>(dy)[idx];
            if (relu) {
                if (rmask) mb[k] = (E == 8) ? rmask[idx] : (unsigned)(rmask[idx >> 1] >> ((idx & 1) * 4));
                else yr[k] = reinterpret_cast<const uint4*>(y)[idx];
            }
            if (dres && dres_accumulate) rr[k] = reinterpret_cast<const uint4*>(dres)[idx];
        }
    }
    float fa[E], fb[E], fc[E], mu[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const float is = invstd[c0 + i];
        fa[i] = gamma[c0 + i] * is; fb[i] = dbeta[c0 + i] * inv_rows; fc[i] = is * dgamma[c0 + i] * inv_rows; mu[i] = mean[c0 + i];
    }
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const long idx = e + k * part;
        if (idx >= totalv) continue;
        float xv[E], g[E], o[E];
        unpack<T, E>(xr[k], xv); unpack<T, E>(gr[k], g);
        if (relu) {
            if (rmask) {
#pragma unroll
                for (int i = 0; i < E; ++i) g[i] = ((mb[k] >> i) & 1u) ? g[i] : 0.f;
            } else {
                float yv[E]; unpack<T, E>(yr[k], yv);
#pragma unroll
                for (int i = 0; i < E; ++i) g[i] = yv[i] > 0.f ? g[i] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < E; ++i) o[i] = fa[i] * (g[i] - fb[i] - (xv[i] - mu[i]) * fc[i]);
        reinterpret_cast<uint4*>(dx)[idx] = pack<T, E>(o);
        if (dres) {
            if (dres_accumulate) {
                float r[E];
                unpack<T, E>(rr[k], r);
#pragma unroll
                for (int i = 0; i < E; ++i) g[i] += r[i];
            }
            reinterpret_cast<uint4*>(dres)[idx] = pack<T, E>(g);
        }
    }
}
template <typename T>
static inline void launch_bn_bwd_apply(const T* x, const T* dy, const T* y, const uint8_t* rmask, const float* mean, const float* invstd, const float* gamma,
                                       const float* dbeta, const float* dgamma, int relu, float inv_rows, T* dx, T* dres, int dres_accumulate, long rows, int CV,
                                       hipStream_t st) {
    const long totalv = rows * CV;
    const int vpt = dev_switch(SW_BN_VPT);
    if (vpt >= 4 && rows >= 16384) {
        const long part = ((rows + 3) / 4) * CV;
        hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 4>), dim3(cdiv(part, 256)), dim3(256), 0, st, x, dy, y, rmask, mean, invstd, gamma, dbeta, dgamma, relu, inv_rows, dx, dres,
                           dres_accumulate, totalv, CV, part);
    } else if (vpt >= 2 && rows >= 4096) {
        const long part = ((rows + 1) / 2) * CV;
        hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 2>), dim3(cdiv(part, 256)), dim3(256), 0, st, x, dy, y, rmask, mean, invstd, gamma, dbeta, dgamma, relu, inv_rows, dx, dres,
                           dres_accumulate, totalv, CV, part);
    } else {
        hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 1>), dim3(cdiv(totalv, 256)), dim3(256), 0, st, x, dy, y, rmask, mean, invstd, gamma, dbeta, dgamma, relu, inv_rows, dx, dres,
                           dres_accumulate, totalv, CV, totalv);
    }
}

// ------------------------------------------------------------------ max pool k=3 s=2 p=1 (torchvision ResNet stem)
// argmax keeps the FIRST maximum in (kh, kw) scan order, like torch's max_pool2d (strict >).
template <typename T>
__global__ void maxpool3x3s2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, unsigned char* __restrict__ amax,
                                        int H, int W, int C4, int P, int Q, long total4) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total4) return;
    int c4 = (int)(e % C4); long t = e / C4; int q = (int)(t % Q); t /= Q; int p = (int)(t % P); long n = t / P;
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int bx = 0, by = 0, bz = 0, bw = 0;
    for (int kh = 0; kh < 3; ++kh) {
        int h = p * 2 - 1 + kh; if ((unsigned)h >= (unsigned)H) continue;
        for (int kw = 0; kw < 3; ++kw) {
            int w = q * 2 - 1 + kw; if ((unsigned)w >= (unsigned)W) continue;
            float4 v = ld4<T>(x, ((n * H + h) * W + w) * C4 + c4);
            int k = kh * 3 + kw;
            if (v.x > best.x || v.x != v.x) { best.x = v.x; bx = k; }
            if (v.y > best.y || v.y != v.y) { best.y = v.y; by = k; }
            if (v.z > best.z || v.z != v.z) { best.z = v.z; bz = k; }
            if (v.w > best.w || v.w != v.w) { best.w = v.w; bw = k; }
        }
    }
    st4<T>(y, e, best);
    reinterpret_cast<uchar4*>(amax)[e] = make_uchar4(bx, by, bz, bw);
}
// gather form (deterministic): every input position sums the outputs whose argmax points at it
template <typename T>
__global__ void maxpool3x3s2_bwd_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ amax, T* __restrict__ dx,
                                        int H, int W, int C4, int P, int Q, long total4) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total4) return;
    int c4 = (int)(e % C4); long t = e / C4; int w = (int)(t % W); t /= W; int h = (int)(t % H); long n = t / H;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int kh = 0; kh < 3; ++kh) {
        int ph = h + 1 - kh; if (ph < 0 || (ph & 1)) continue; int p = ph >> 1; if (p >= P) continue;
        for (int kw = 0; kw < 3; ++kw) {
            int pw = w + 1 - kw; if (pw < 0 || (pw & 1)) continue; int q = pw >> 1; if (q >= Q) continue;
            long o = ((n * P + p) * Q + q) * C4 + c4;
            uchar4 a = reinterpret_cast<const uchar4*>(amax)[o];
            float4 g = ld4<T>(dy, o);
            int k = kh * 3 + kw;
            if (a.x == k) acc.x += g.x; if (a.y == k) acc.y += g.y; if (a.z == k) acc.z += g.z; if (a.w == k) acc.w += g.w;
        }
    }
    st4<T>(dx, e, acc);
}

// 16 bytes per lane (E = 4 fp32 / 8 bf16 channels): the forms the ResNet stem takes (C = 64).  Same arithmetic as above.
template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_fwd_v_kernel(const T* __restrict__ x, T* __restrict__ y, unsigned char* __restrict__ amax,
                                                                 int H, int W, int CV, int P, int Q, long totalv) {
    constexpr int E = EPT<T>::n;
    const unsigned e = blockIdx.x * 256u + threadIdx.x;             // 32-bit index math (the launcher checks totalv < 2^31): 64-bit divisions cost more than the loads
    if (e >= totalv) return;
    const unsigned cv = e % CV; unsigned t = e / CV; const int q = (int)(t % Q); t /= Q; const int p = (int)(t % P); const long n = t / P;
    float best[E]; unsigned bk[E];
#pragma unroll
    for (int i = 0; i < E; ++i) { best[i] = -INFINITY; bk[i] = 0; }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int h = p * 2 - 1 + kh; if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int w = q * 2 - 1 + kw; if ((unsigned)w >= (unsigned)W) continue;
            float v[E];
            unpack<T, E>(reinterpret_cast<const uint4*>(x)[((n * H + h) * W + w) * CV + cv], v);
            const unsigned k = kh * 3 + kw;
#pragma unroll
            for (int i = 0; i < E; ++i) if (v[i] > best[i] || v[i] != v[i]) { best[i] = v[i]; bk[i] = k; }
        }
    }
    reinterpret_cast<uint4*>(y)[e] = pack<T, E>(best);
    if (E == 8) reinterpret_cast<uint2*>(amax)[e] = make_uint2(bk[0] | (bk[1] << 8) | (bk[2] << 16) | (bk[3] << 24), bk[4 % E] | (bk[5 % E] << 8) | (bk[6 % E] << 16) | (bk[7 % E] << 24));
    else reinterpret_cast<unsigned*>(amax)[e] = bk[0] | (bk[1] << 8) | (bk[2] << 16) | (bk[3] << 24);
}
template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_v_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ amax, T* __restrict__ dx,
                                                                 int H, int W, int CV, int P, int Q, long totalv) {
    constexpr int E = EPT<T>::n;
    const unsigned e = blockIdx.x * 256u + threadIdx.x;
    if (e >= totalv) return;
    const unsigned cv = e % CV; unsigned t = e / CV; const int w = (int)(t % W); t /= W; const int h = (int)(t % H); const long n = t / H;
    float acc[E];
#pragma unroll
    for (int i = 0; i < E; ++i) acc[i] = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int ph = h + 1 - kh; if (ph < 0 || (ph & 1)) continue; const int p = ph >> 1; if (p >= P) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int pw = w + 1 - kw; if (pw < 0 || (pw & 1)) continue; const int q = pw >> 1; if (q >= Q) continue;
            const long o = ((n * P + p) * Q + q) * CV + cv;
            unsigned a0, a1 = 0;
            if (E == 8) { const uint2 a = reinterpret_cast<const uint2*>(amax)[o]; a0 = a.x; a1 = a.y; }
            else a0 = reinterpret_cast<const unsigned*>(amax)[o];
            float g[E];
            unpack<T, E>(reinterpret_cast<const uint4*>(dy)[o], g);
            const unsigned k = kh * 3 + kw;
#pragma unroll
            for (int i = 0; i < E; ++i) { const unsigned a = ((i < 4 ? a0 : a1) >> ((i & 3) * 8)) & 0xffu; if (a == k) acc[i] += g[i]; }
        }
    }
    reinterpret_cast<uint4*>(dx)[e] = pack<T, E>(acc);
}

// ------------------------------------------------------------------ ResNet stem tail in one pass: BatchNorm + ReLU + max pool
// The stem's BatchNorm output (N x 128 x 128 x 64 at 256 px: the largest activation of the network) is only ever read by the
// max pool.  Forward: normalise + ReLU the 3x3 window on the fly and keep the pooled map and the argmax; the full-size
// activation and its sign mask are never written.  Backward: the pooled gradient is gathered through the argmax and the ReLU
// sign recomputed from the convolution output (v > 0 for v = (x - mean) * invstd * gamma + beta, the forward's own test), so
// the dense pre-pool gradient is never materialised either.  Values and the first-maximum rule are those of bn_apply_kernel
// followed by maxpool3x3s2_fwd_kernel (the comparison runs on the values rounded to the storage type, as there).
template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<__bf16>(float v) { return (float)(__bf16)v; }

template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y,
                                                                  unsigned char* __restrict__ amax, int H, int W, int CV, int P, int Q, long totalv) {
    constexpr int E = EPT<T>::n;
    const unsigned e = blockIdx.x * 256u + threadIdx.x;
    if (e >= totalv) return;
    const unsigned cv = e % CV; unsigned t = e / CV; const int q = (int)(t % Q); t /= Q; const int p = (int)(t % P); const long n = t / P;
    float mu[E], is[E], ga[E], be[E], best[E]; unsigned bk[E];
#pragma unroll
    for (int i = 0; i < E; ++i) { mu[i] = mean[cv * E + i]; is[i] = invstd[cv * E + i]; ga[i] = gamma[cv * E + i]; be[i] = beta[cv * E + i]; best[i] = -INFINITY; bk[i] = 0; }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int h = p * 2 - 1 + kh; if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int w = q * 2 - 1 + kw; if ((unsigned)w >= (unsigned)W) continue;
            float xv[E];
            unpack<T, E>(reinterpret_cast<const uint4*>(x)[((n * H + h) * W + w) * CV + cv], xv);
            const unsigned k = kh * 3 + kw;
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const float pre = (xv[i] - mu[i]) * is[i] * ga[i] + be[i];
                const float v = round_to<T>(pre < 0.f ? 0.f : pre);       // NaN stays NaN, and wins the window like in torch's max pool
                if (v > best[i] || v != v) { best[i] = v; bk[i] = k; }
            }
        }
    }
    reinterpret_cast<uint4*>(y)[e] = pack<T, E>(best);
    if (E == 8) reinterpret_cast<uint2*>(amax)[e] = make_uint2(bk[0] | (bk[1] << 8) | (bk[2] << 16) | (bk[3] << 24), bk[4 % E] | (bk[5 % E] << 8) | (bk[6 % E] << 16) | (bk[7 % E] << 24));
    else reinterpret_cast<unsigned*>(amax)[e] = bk[0] | (bk[1] << 8) | (bk[2] << 16) | (bk[3] << 24);
}

// Backward, by 2x2 blocks of pre-pool positions: block (p, q) = rows 2p, 2p+1 x columns 2q, 2q+1 is reached by exactly the four
// windows (p, q), (p, q+1), (p+1, q), (p+1, q+1) - eight loads for four positions, no lane divergence (a per-position gather
// walks nine window candidates under divergent parity tests).  The contributions are added in maxpool3x3s2_bwd's order, the sum
// is rounded to the storage type where the separate path stores it, and the ReLU sign is recomputed from x.
template <typename T, int E>
struct StemWin { float d[E]; unsigned a0, a1; };
template <typename T, int E>
__device__ __forceinline__ void stem_load_win(const T* __restrict__ dy, const unsigned char* __restrict__ amax, long n, int p, int q, int cv, int CV, int P, int Q,
                                              StemWin<T, E>& w) {
    if (p < P && q < Q) {
        const long o = ((n * P + p) * Q + q) * CV + cv;
        if (E == 8) { const uint2 a = reinterpret_cast<const uint2*>(amax)[o]; w.a0 = a.x; w.a1 = a.y; }
        else { w.a0 = reinterpret_cast<const unsigned*>(amax)[o]; w.a1 = 0; }
        unpack<T, E>(reinterpret_cast<const uint4*>(dy)[o], w.d);
    } else {
        w.a0 = w.a1 = 0xffffffffu;                   // no window position is 255: contributes nothing
#pragma unroll
        for (int i = 0; i < E; ++i) w.d[i] = 0.f;
    }
}
template <typename T, int E>
__device__ __forceinline__ float stem_pick(const StemWin<T, E>& w, int i, unsigned k) {
    const unsigned a = ((i < 4 ? w.a0 : w.a1) >> ((i & 3) * 8)) & 0xffu;
    return a == k ? w.d[i] : 0.f;
}
// g of the block's position (dh, dw) in {0,1}^2; w00 = window (p, q), w01 = (p, q+1), w10 = (p+1, q), w11 = (p+1, q+1)
template <typename T, int E>
__device__ __forceinline__ void stem_block_grad(int dh, int dw, const StemWin<T, E>& w00, const StemWin<T, E>& w01, const StemWin<T, E>& w10, const StemWin<T, E>& w11,
                                                const float (&xv)[E], const float (&mu)[E], const float (&is)[E], const float (&ga)[E], const float (&be)[E], float (&g)[E]) {
#pragma unroll
    for (int i = 0; i < E; ++i) {
        float s = 0.f;
        if (!dh && !dw) s += stem_pick<T, E>(w00, i, 4);
        else if (!dh) { s += stem_pick<T, E>(w01, i, 3); s += stem_pick<T, E>(w00, i, 5); }
        else if (!dw) { s += stem_pick<T, E>(w10, i, 1); s += stem_pick<T, E>(w00, i, 7); }
        else { s += stem_pick<T, E>(w11, i, 0); s += stem_pick<T, E>(w10, i, 2); s += stem_pick<T, E>(w01, i, 6); s += stem_pick<T, E>(w00, i, 8); }
        s = round_to<T>(s);
        g[i] = ((xv[i] - mu[i]) * is[i] * ga[i] + be[i] > 0.f) ? s : 0.f;
    }
}

// per-channel sums of g and g * xhat over a slice of the 2x2 blocks -> the partial layout of bn_colstats_kernel<1>
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_bwd_stats_kernel(const T* __restrict__ x, const T* __restrict__ dy, const unsigned char* __restrict__ amax,
                                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                        const float* __restrict__ gamma, const float* __restrict__ beta, int H, int W, int P, int Q,
                                                                        long blocks, int C, int CV, long blocks_per, double* __restrict__ part0, double* __restrict__ part1) {
    constexpr int E = EPT<T>::n;
    __shared__ double sh[2][256][E > 4 ? 4 : E];
    const int tid = threadIdx.x, tc = tid % CV, tr = tid / CV, RL = 256 / CV;
    const int cv = blockIdx.x * CV + tc, CVT = C / E;
    double a0[E], a1[E];
#pragma unroll
    for (int i = 0; i < E; ++i) { a0[i] = 0.0; a1[i] = 0.0; }
    if (cv < CVT && tr < RL) {
        float mu[E], is[E], ga[E], be[E];
#pragma unroll
        for (int i = 0; i < E; ++i) { mu[i] = mean[cv * E + i]; is[i] = invstd[cv * E + i]; ga[i] = gamma[cv * E + i]; be[i] = beta[cv * E + i]; }
        long r0 = (long)blockIdx.y * blocks_per, r1 = r0 + blocks_per; if (r1 > blocks) r1 = blocks;
        for (long r = r0 + tr; r < r1; r += RL) {
            const unsigned ru = (unsigned)r; const int q = (int)(ru % Q); const unsigned t = ru / Q; const int p = (int)(t % P); const long n = t / P;
            StemWin<T, E> w00, w01, w10, w11;
            stem_load_win<T, E>(dy, amax, n, p, q, cv, CVT, P, Q, w00); stem_load_win<T, E>(dy, amax, n, p, q + 1, cv, CVT, P, Q, w01);
            stem_load_win<T, E>(dy, amax, n, p + 1, q, cv, CVT, P, Q, w10); stem_load_win<T, E>(dy, amax, n, p + 1, q + 1, cv, CVT, P, Q, w11);
            float s0[E], s1[E];
#pragma unroll
            for (int i = 0; i < E; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int dw = 0; dw < 2; ++dw) {
                    const int h = 2 * p + dh, w = 2 * q + dw;
                    if (h >= H || w >= W) continue;
                    float xv[E], g[E];
                    unpack<T, E>(reinterpret_cast<const uint4*>(x)[((n * H + h) * W + w) * CVT + cv], xv);
                    stem_block_grad<T, E>(dh, dw, w00, w01, w10, w11, xv, mu, is, ga, be, g);
#pragma unroll
                    for (int i = 0; i < E; ++i) { s0[i] += g[i]; s1[i] = fmaf(g[i], (xv[i] - mu[i]) * is[i], s1[i]); }       // four terms in fp32, the blocks in double
                }
#pragma unroll
            for (int i = 0; i < E; ++i) { a0[i] += (double)s0[i]; a1[i] += (double)s1[i]; }
        }
    }
#pragma unroll
    for (int hh = 0; hh < E; hh += 4) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) { sh[0][tid][i] = a0[hh + i]; sh[1][tid][i] = a1[hh + i]; }
        __syncthreads();
        if (tr == 0 && cv < CVT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double s0 = 0.0, s1 = 0.0;
                for (int k = 0; k < RL; ++k) { s0 += sh[0][k * CV + tc][i]; s1 += sh[1][k * CV + tc][i]; }
                part0[(long)blockIdx.y * C + cv * E + hh + i] = s0;
                part1[(long)blockIdx.y * C + cv * E + hh + i] = s1;
            }
        }
    }
}
// dx = gamma * invstd * (g - dbeta/M - xhat * dgamma/M), one thread per 2x2 block and vector column
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, const unsigned char* __restrict__ amax,
                                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                        const float* __restrict__ dbeta, const float* __restrict__ dgamma, float inv_rows,
                                                                        T* __restrict__ dx, int H, int W, int P, int Q, long totalb, int CV) {
    constexpr int E = EPT<T>::n;
    const unsigned e = blockIdx.x * 256u + threadIdx.x;
    if (e >= totalb) return;
    const int cv = (int)(e % CV); unsigned t = e / CV;
    const int q = (int)(t % Q); t /= Q; const int p = (int)(t % P); const long n = t / P;
    float mu[E], is[E], ga[E], be[E], db[E], dg[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        mu[i] = mean[cv * E + i]; is[i] = invstd[cv * E + i]; ga[i] = gamma[cv * E + i]; be[i] = beta[cv * E + i];
        db[i] = dbeta[cv * E + i] * inv_rows; dg[i] = dgamma[cv * E + i] * inv_rows;
    }
    StemWin<T, E> w00, w01, w10, w11;
    stem_load_win<T, E>(dy, amax, n, p, q, cv, CV, P, Q, w00); stem_load_win<T, E>(dy, amax, n, p, q + 1, cv, CV, P, Q, w01);
    stem_load_win<T, E>(dy, amax, n, p + 1, q, cv, CV, P, Q, w10); stem_load_win<T, E>(dy, amax, n, p + 1, q + 1, cv, CV, P, Q, w11);
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int dw = 0; dw < 2; ++dw) {
            const int h = 2 * p + dh, w = 2 * q + dw;
            if (h >= H || w >= W) continue;
            const long iv = ((n * H + h) * W + w) * CV + cv;
            float xv[E], g[E], o[E];
            unpack<T, E>(reinterpret_cast<const uint4*>(x)[iv], xv);
            stem_block_grad<T, E>(dh, dw, w00, w01, w10, w11, xv, mu, is, ga, be, g);
#pragma unroll
            for (int i = 0; i < E; ++i) o[i] = ga[i] * is[i] * (g[i] - db[i] - (xv[i] - mu[i]) * is[i] * dg[i]);
            reinterpret_cast<uint4*>(dx)[iv] = pack<T, E>(o);
        }
}

// ------------------------------------------------------------------ encoder_size resize on the final map (readme.md:118-121)
// adaptive average pool H x W -> P x Q (torch bin edges floor(i*H/P) .. ceil((i+1)*H/P))
// V = 4: four channels (16 bytes) per thread when C % 4 == 0; the backward form visits only the <= 3 x 3 bins that can contain (h, w)
// instead of testing all P x Q (49 trips with two integer divisions each at 8 x 8 -> 7 x 7); same bins, same order, same arithmetic.
template <int V>
__global__ void adaptive_avgpool_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C, int P, int Q, long total, int backward) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int CV = C / V;
    float s[V];
#pragma unroll
    for (int i = 0; i < V; ++i) s[i] = 0.f;
    auto add = [&](long idx, float scale) {
        if (V == 4) { const float4 v = reinterpret_cast<const float4*>(x)[idx]; s[0] += v.x * scale; s[1] += v.y * scale; s[2] += v.z * scale; s[3] += v.w * scale; }
        else s[0] += x[idx] * scale;
    };
    if (!backward) {      // e over outputs (n,p,q,c)
        const unsigned eu = (unsigned)e; int c = (int)(eu % CV); unsigned t = eu / CV; int q = (int)(t % Q); t /= Q; int p = (int)(t % P); long n = t / P;
        int h0 = (p * H) / P, h1 = ((p + 1) * H + P - 1) / P, w0 = (q * W) / Q, w1 = ((q + 1) * W + Q - 1) / Q;
        for (int h = h0; h < h1; ++h) for (int w = w0; w < w1; ++w) add(((n * H + h) * W + w) * CV + c, 1.f);
        const float cnt = (float)((h1 - h0) * (w1 - w0));
#pragma unroll
        for (int i = 0; i < V; ++i) s[i] = s[i] / cnt;
    } else {              // e over inputs (n,h,w,c): x = dy (n,P,Q,c), y = dx
        const unsigned eu = (unsigned)e; int c = (int)(eu % CV); unsigned t = eu / CV; int w = (int)(t % W); t /= W; int h = (int)(t % H); long n = t / H;
        const int pa = max(0, (h * P) / H - 1), pb = min(P - 1, ((h + 1) * P + H - 1) / H), qa = max(0, (w * Q) / W - 1), qb = min(Q - 1, ((w + 1) * Q + W - 1) / W);
        for (int p = pa; p <= pb; ++p) {
            int h0 = (p * H) / P, h1 = ((p + 1) * H + P - 1) / P; if (h < h0 || h >= h1) continue;
            for (int q = qa; q <= qb; ++q) {
                int w0 = (q * W) / Q, w1 = ((q + 1) * W + Q - 1) / Q; if (w < w0 || w >= w1) continue;
                const float cnt = (float)((h1 - h0) * (w1 - w0));
                const long idx = ((n * P + p) * Q + q) * CV + c;
                if (V == 4) { const float4 v = reinterpret_cast<const float4*>(x)[idx]; s[0] += v.x / cnt; s[1] += v.y / cnt; s[2] += v.z / cnt; s[3] += v.w / cnt; }
                else s[0] += x[idx] / cnt;
            }
        }
    }
    if (V == 4) reinterpret_cast<float4*>(y)[e] = make_float4(s[0], s[1], s[2], s[3]);
    else y[e] = s[0];
}
// bilinear resize, align_corners=False (nn.Upsample): src = (dst + 0.5) * scale - 0.5, clamped at 0
__device__ __forceinline__ void bilinear_src(int o, int in, int out, int& i0, int& i1, float& l1) {
    float s = ((float)o + 0.5f) * ((float)in / (float)out) - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s; i1 = i0 + (i0 < in - 1 ? 1 : 0); l1 = s - (float)i0;
}
__global__ void bilinear_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C, int P, int Q, long total) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const unsigned eu = (unsigned)e; int c = (int)(eu % C); unsigned t = eu / C; int q = (int)(t % Q); t /= Q; int p = (int)(t % P); long n = t / P;
    int h0, h1, w0, w1; float lh, lw;
    bilinear_src(p, H, P, h0, h1, lh); bilinear_src(q, W, Q, w0, w1, lw);
    const float* b = x + n * H * W * C + c;
    float v00 = b[((long)h0 * W + w0) * C], v01 = b[((long)h0 * W + w1) * C], v10 = b[((long)h1 * W + w0) * C], v11 = b[((long)h1 * W + w1) * C];
    y[e] = (1.f - lh) * ((1.f - lw) * v00 + lw * v01) + lh * ((1.f - lw) * v10 + lw * v11);
}
__global__ void bilinear_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W, int C, int P, int Q, long total) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;     // over inputs (n,h,w,c); gather over all outputs touching it
    if (e >= total) return;
    const unsigned eu = (unsigned)e; int c = (int)(eu % C); unsigned t = eu / C; int w = (int)(t % W); t /= W; int h = (int)(t % H); long n = t / H;
    float s = 0.f;
    for (int p = 0; p < P; ++p) {
        int h0, h1; float lh; bilinear_src(p, H, P, h0, h1, lh);
        float wh = (h0 == h ? 1.f - lh : 0.f) + (h1 == h ? lh : 0.f);
        if (wh == 0.f) continue;
        for (int q = 0; q < Q; ++q) {
            int w0, w1; float lw; bilinear_src(q, W, Q, w0, w1, lw);
            float ww = (w0 == w ? 1.f - lw : 0.f) + (w1 == w ? lw : 0.f);
            if (ww != 0.f) s += wh * ww * dy[((n * P + p) * Q + q) * C + c];
        }
    }
    dx[e] = s;
}

// ---- grouped 3x3 convolutions (resnext: model.py:28 builds torchvision's resnext50_32x4d / resnext101_32x8d in the ResNet branch)
// The grouped filter (K, R, S, C / groups) becomes the dense block-diagonal filter (K, R, S, C) the implicit-GEMM kernels read; the gradient of the
// grouped filter is the diagonal blocks of the dense filter's gradient.  Pure data movement.
template <typename T>
__global__ void grouped_filter_expand_kernel(const T* __restrict__ wg, T* __restrict__ wd, int K, int C, int RS, int groups) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)K * RS * C;
    if (e >= total) return;
    const int c = (int)(e % C); const long t = e / C; const int rs = (int)(t % RS), k = (int)(t / RS);
    const int cg = C / groups, kg = K / groups;
    const int g = k / kg;
    const int cl = c - g * cg;
    wd[e] = (cl >= 0 && cl < cg) ? wg[((long)k * RS + rs) * cg + cl] : (T)0.f;
}
__global__ void grouped_filter_grad_extract_kernel(const float* __restrict__ dwd, float* __restrict__ dwg, int K, int C, int RS, int groups) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cg = C / groups, kg = K / groups;
    const long total = (long)K * RS * cg;
    if (e >= total) return;
    const int cl = (int)(e % cg); const long t = e / cg; const int rs = (int)(t % RS), k = (int)(t / RS);
    dwg[e] = dwd[((long)k * RS + rs) * C + (k / kg) * cg + cl];
}

}  // namespace sat

using namespace sat;

static int conv_geom(const sat_conv_geom* g, ConvGeom& o, int vec = 4) {
    SAT_REQUIRE(g, "conv: null geometry");
    o.N = g->N; o.H = g->H; o.W = g->W; o.C = g->C; o.K = g->K; o.R = g->R; o.S = g->S; o.stride = g->stride; o.pad = g->pad; o.sw = (g->stride_w > 0 && g->stride_w != g->stride) ? g->stride_w : 0;
    SAT_REQUIRE(o.N > 0 && o.H > 0 && o.W > 0 && o.C > 0 && o.K > 0 && o.R > 0 && o.S > 0 && o.stride > 0 && o.pad >= 0, "conv: bad geometry");
    o.P = (o.H + 2 * o.pad - o.R) / o.stride + 1; o.Q = (o.W + 2 * o.pad - o.S) / (o.sw ? o.sw : o.stride) + 1;
    SAT_REQUIRE(o.P > 0 && o.Q > 0, "conv: empty output");
    SAT_REQUIRE(o.C % vec == 0 && o.K % vec == 0, "conv: C=%d and K=%d must be multiples of %d (pad the stem channels)", o.C, o.K, vec);
    return SAT_OK;
}

extern "C" {

static int conv_fwd_any(const void* x, const void* w, const float* bias, void* y, const sat_conv_geom* geom, int bf16, void* stream,
                        float* tile_stats = nullptr, int* tile_rows = nullptr) {
    ConvGeom g; SAT_TRY(conv_geom(geom, g, bf16 ? 8 : 4));
    if (!x || !w || !y) return fail(SAT_EINVAL, "conv2d_fwd: null pointer");
    GemmArgs a; a.a_bf16 = a.b_bf16 = a.c_bf16 = a.bf16_mfma = bf16;
    a.M = g.N * g.P * g.Q; a.N = g.K; a.K = g.R * g.S * g.C;
    a.B = w; a.ldb = a.K; a.bmode = B_ROW; a.C = y; a.ldc = g.K; a.g = g;
    if (g.R == 1 && g.S == 1 && g.stride == 1 && g.pad == 0) { a.A = x; a.lda = g.C; a.amode = A_ROW; }   // 1x1: a plain GEMM over pixels
    else { a.A = x; a.amode = A_CONV_FWD; }
    if (bias) { a.epi = EPI_BIAS; a.bias = bias; }
    a.tile_stats = tile_stats; a.tile_rows = tile_rows;
    return launch_gemm(a, (hipStream_t)stream);
}
int sat_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, const sat_conv_geom* geom, void* stream) {
    return conv_fwd_any(x, w, bias, y, geom, 0, stream);
}
int sat_conv2d_fwd_bf16(const void* x, const void* w, const float* bias, void* y, const sat_conv_geom* geom, void* stream) {
    return conv_fwd_any(x, w, bias, y, geom, 1, stream);
}
size_t sat_conv2d_fwd_stats_bytes(const sat_conv_geom* geom) {
    ConvGeom g; if (conv_geom(geom, g, 8) != SAT_OK) return 0;
    return (size_t)cdiv((long)g.N * g.P * g.Q, 64L) * g.K * 2 * sizeof(float);
}
int sat_conv2d_fwd_bf16_stats(const void* x, const void* w, void* y, const sat_conv_geom* geom, float* tile_stats, int32_t* tile_rows, void* stream) {
    if (!tile_stats || !tile_rows) return fail(SAT_EINVAL, "conv2d_fwd_bf16_stats: null statistics buffer");
    int tr = 0;
    SAT_TRY(conv_fwd_any(x, w, nullptr, y, geom, 1, stream, tile_stats, &tr));
    *tile_rows = tr;
    return SAT_OK;
}

struct BnBwdStats { const void* x; const unsigned char* mask; const float* mean; const float* invstd; float* tile_stats; int* tile_rows; };
static int conv_dgrad_any(const void* dy, const void* w, void* dx, const sat_conv_geom* geom, int accumulate, int bf16, void* stream,
                          const BnBwdStats* bs = nullptr, const void* add_src = nullptr, const uint8_t* add_mask = nullptr) {
    ConvGeom g; SAT_TRY(conv_geom(geom, g, bf16 ? 8 : 4));
    if (!dy || !w || !dx) return fail(SAT_EINVAL, "conv2d_dgrad: null pointer");
    GemmArgs a; a.a_bf16 = a.b_bf16 = a.c_bf16 = a.bf16_mfma = bf16;
    a.accumulate = accumulate; a.C = dx; a.ldc = g.C; a.g = g;
    if (add_src) {
        SAT_REQUIRE(bf16 && g.stride == 1, "conv2d_dgrad: add_src needs bf16 storage and stride 1");
        a.accumulate = 1; a.add_src = add_src; a.add_mask = add_mask;
    }
    if (bs) {
        *bs->tile_rows = 0;
        if (bf16 && !(g.stride == 2)) {          // (the stride-2 parity classes write interleaved rows: no tile statistics there)
            a.bn_x = bs->x; a.bn_mask = bs->mask; a.bn_mean = bs->mean; a.bn_invstd = bs->invstd; a.tile_stats = bs->tile_stats; a.tile_rows = bs->tile_rows;
        }
    }
    if (g.R == 1 && g.S == 1 && g.stride == 1 && g.pad == 0) {
        a.M = g.N * g.H * g.W; a.N = g.C; a.K = g.K; a.A = dy; a.lda = g.K; a.amode = A_ROW; a.B = w; a.ldb = g.C; a.bmode = B_KMAJOR;
    } else if (bf16 && g.stride == 2) {
        // stride 2: an input pixel only sees the taps whose parity matches its own, so run one implicit GEMM per
        // (h % 2, w % 2) class over exactly those taps (9 tap-products for a 3x3 instead of 36; 1 instead of 4 for a 1x1)
        a.A = dy; a.amode = A_CONV_DGRAD; a.B = w; a.bmode = B_CONV_DGRAD_W; a.N = g.C;
        for (int ph = 0; ph < 2; ++ph)
            for (int pw = 0; pw < 2; ++pw) {
                ConvGeom c = g; c.cls = 1; c.ph = ph; c.pw = pw;
                c.Hc = (g.H - ph + 1) / 2; c.Wc = (g.W - pw + 1) / 2;
                c.r0 = (ph + g.pad) & 1; c.s0 = (pw + g.pad) & 1;
                c.rc = c.r0 < g.R ? (g.R - c.r0 + 1) / 2 : 0; c.sc = c.s0 < g.S ? (g.S - c.s0 + 1) / 2 : 0;
                if (c.Hc <= 0 || c.Wc <= 0) continue;
                a.g = c; a.M = g.N * c.Hc * c.Wc; a.K = c.rc * c.sc * g.K;
                if (a.K == 0 && accumulate) continue;                    // nothing to add for this class
                if (c.sc == 0) a.g.sc = 1;                               // K == 0: the class only receives zeros
                SAT_TRY(launch_gemm(a, (hipStream_t)stream));
            }
        return SAT_OK;
    } else {
        a.M = g.N * g.H * g.W; a.N = g.C; a.K = g.R * g.S * g.K; a.A = dy; a.amode = A_CONV_DGRAD; a.B = w; a.bmode = B_CONV_DGRAD_W;
    }
    return launch_gemm(a, (hipStream_t)stream);
}
int sat_conv2d_dgrad(const float* dy, const float* w, float* dx, const sat_conv_geom* geom, int accumulate, void* stream) {
    return conv_dgrad_any(dy, w, dx, geom, accumulate, 0, stream);
}
int sat_conv2d_dgrad_bf16(const void* dy, const void* w, void* dx, const sat_conv_geom* geom, int accumulate, void* stream) {
    return conv_dgrad_any(dy, w, dx, geom, accumulate, 1, stream);
}
int sat_conv2d_dgrad_bf16_fused(const void* dy, const void* w, void* dx, const sat_conv_geom* geom, const void* add_src, const uint8_t* add_mask, const void* bn_x,
                                const uint8_t* bn_relu_mask, const float* bn_mean, const float* bn_invstd, float* tile_stats, int32_t* tile_rows, void* stream) {
    if (!add_src) return fail(SAT_EINVAL, "conv2d_dgrad_bf16_fused: null add_src (use sat_conv2d_dgrad_bf16[_bnstats])");
    if (!bn_x) return conv_dgrad_any(dy, w, dx, geom, 1, 1, stream, nullptr, add_src, add_mask);
    if (!bn_mean || !bn_invstd || !tile_stats || !tile_rows) return fail(SAT_EINVAL, "conv2d_dgrad_bf16_fused: null statistics pointer");
    int tr = 0;
    BnBwdStats bs{bn_x, bn_relu_mask, bn_mean, bn_invstd, tile_stats, &tr};
    SAT_TRY(conv_dgrad_any(dy, w, dx, geom, 1, 1, stream, &bs, add_src, add_mask));
    *tile_rows = tr;
    return SAT_OK;
}
size_t sat_conv2d_dgrad_stats_bytes(const sat_conv_geom* geom) {
    ConvGeom g; if (conv_geom(geom, g, 8) != SAT_OK) return 0;
    return (size_t)cdiv((long)g.N * g.H * g.W, 64L) * g.C * 2 * sizeof(float);
}
int sat_conv2d_dgrad_bf16_bnstats(const void* dy, const void* w, void* dx, const sat_conv_geom* geom, int accumulate, const void* bn_x, const uint8_t* bn_relu_mask,
                                  const float* bn_mean, const float* bn_invstd, float* tile_stats, int32_t* tile_rows, void* stream) {
    if (!bn_x || !bn_mean || !bn_invstd || !tile_stats || !tile_rows) return fail(SAT_EINVAL, "conv2d_dgrad_bf16_bnstats: null pointer");
    int tr = 0;
    BnBwdStats bs{bn_x, bn_relu_mask, bn_mean, bn_invstd, tile_stats, &tr};
    SAT_TRY(conv_dgrad_any(dy, w, dx, geom, accumulate, 1, stream, &bs));
    *tile_rows = tr;
    return SAT_OK;
}

static int conv_wgrad_any(const void* dy, const void* x, float* dw, const sat_conv_geom* geom, float* slab, int64_t slab_elems, int bf16, void* stream) {
    ConvGeom g; SAT_TRY(conv_geom(geom, g, bf16 ? 8 : 4));
    if (!dy || !x || !dw) return fail(SAT_EINVAL, "conv2d_wgrad: null pointer");
    if (bf16 && wgrad3x3_eligible(g)) return launch_wgrad3x3(dy, x, dw, g, slab, (long)slab_elems, (hipStream_t)stream);      // all nine taps per workgroup (wgrad3x3.hip)
    GemmArgs a; a.a_bf16 = a.b_bf16 = a.bf16_mfma = bf16; a.c_bf16 = 0;
    a.M = g.K; a.N = g.R * g.S * g.C; a.K = g.N * g.P * g.Q;
    a.A = dy; a.lda = g.K; a.amode = A_KMAJOR; a.C = dw; a.ldc = a.N; a.g = g; a.slab = slab; a.slab_elems = slab_elems;
    if (g.R == 1 && g.S == 1 && g.stride == 1 && g.pad == 0) { a.B = x; a.ldb = g.C; a.bmode = B_KMAJOR; }
    else { a.B = x; a.bmode = B_CONV_WGRAD; }
    return launch_gemm(a, (hipStream_t)stream);
}
int sat_conv2d_wgrad(const float* dy, const float* x, float* dw, const sat_conv_geom* geom, float* slab, int64_t slab_elems, void* stream) {
    return conv_wgrad_any(dy, x, dw, geom, slab, slab_elems, 0, stream);
}
int sat_conv2d_wgrad_bf16(const void* dy, const void* x, float* dw, const sat_conv_geom* geom, float* slab, int64_t slab_elems, void* stream) {
    return conv_wgrad_any(dy, x, dw, geom, slab, slab_elems, 1, stream);
}

size_t sat_conv2d_wgrad_slab_bytes(const sat_conv_geom* geom) {
    ConvGeom g; if (conv_geom(geom, g) != SAT_OK) return 0;
    return gemm_slab_bytes(g.K, g.R * g.S * g.C, g.N * g.P * g.Q);
}

int sat_image_normalize_nhwc4(const float* img_nchw, float* out_nhwc4, int32_t N, int32_t H, int32_t W, const float* mean3_host, const float* std3_host, void* stream) {
    if (!img_nchw || !out_nhwc4 || !mean3_host || !std3_host) return fail(SAT_EINVAL, "image_normalize: null pointer");
    long total = (long)N * H * W;
    hipLaunchKernelGGL(normalize_nhwc4_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, img_nchw, out_nhwc4, H, W, total,
                       mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2]);
    return launch_ok("normalize_nhwc4");
}

int sat_pad_channels_3to4(const float* src, float* dst, int64_t pixels, int32_t inverse, void* stream) {
    if (!src || !dst) return fail(SAT_EINVAL, "pad_channels: null pointer");
    if (inverse) hipLaunchKernelGGL(unpad_c4_to_c3_kernel, dim3(cdiv(pixels, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, (long)pixels);
    else hipLaunchKernelGGL(pad_c3_to_c4_kernel, dim3(cdiv(pixels, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, (long)pixels);
    return launch_ok("pad_channels");
}

}  // extern "C" (typed implementations follow)

static void bn_grid(long rows, int C, int E, int& CV, long& rows_per, int& nparts, int backward = 0) {
    int CVT = C / E;
    CV = CVT < 256 ? CVT : 256;
    while (256 % CV) --CV;                      // CV must divide 256
    int RL = 256 / CV;
    int colblocks = cdiv(CVT, CV);
    static const long target_f = getenv("SAT_BN_BLOCKS") ? atol(getenv("SAT_BN_BLOCKS")) : 512;
    static const long target_b = getenv("SAT_BN_BLOCKS_BWD") ? atol(getenv("SAT_BN_BLOCKS_BWD")) : 768;      // the backward pass keeps 2 rows x 2 operands in flight: 4 blocks per CU fit
    const long target = backward ? target_b : target_f;
    long want = target / colblocks; if (want < 1) want = 1;         // ~2 blocks per CU: measured best on the C2 step (fewer, longer blocks; fewer partials to finalise)
    rows_per = cdiv(rows, want);
    long minrows = (long)RL * 8; if (rows_per < minrows) rows_per = minrows;
    nparts = cdiv(rows, rows_per);
}

extern "C" size_t sat_bn_scratch_bytes(int64_t rows, int32_t C) {
    if (rows <= 0 || C <= 0 || C % 4) return 0;
    int np = 0;
    for (int bwd = 0; bwd < 2; ++bwd) {            // the largest partial count of the forward / backward grids at either vector width
        int CV, nparts; long rp; bn_grid(rows, C, 4, CV, rp, nparts, bwd);
        if (nparts > np) np = nparts;
        if (C % 8 == 0) { bn_grid(rows, C, 8, CV, rp, nparts, bwd); if (nparts > np) np = nparts; }
    }
    return (size_t)np * C * 2 * sizeof(double) + 64 + ((size_t)cdiv(C, 32) * sizeof(int) + 63) / 64 * 64;      // partials + one ticket per 32 channels
}

template <typename T>
static inline void launch_bn_apply_train(const T* x, const float* mean, const float* invstd, const float* gamma, const float* beta, const T* residual, int relu, T* y,
                                         uint8_t* relu_mask, long totalv, int CV, ResBn rbn, hipStream_t st) {
    const long rows = totalv / CV;
    const int vpt = dev_switch(SW_BN_VPT);
    if (rbn.mean && vpt >= 2 && rows >= 4096) {
        const long part = ((rows + 1) / 2) * CV;
        hipLaunchKernelGGL((bn_apply_kernel<T, false, true, 2>), dim3(cdiv(part, 256)), dim3(256), 0, st, x, mean, invstd, gamma, beta, residual, relu, y, relu_mask, totalv, CV, -1.0f, rbn, part);
    } else if (rbn.mean)
        hipLaunchKernelGGL((bn_apply_kernel<T, false, true>), dim3(cdiv(totalv, 256)), dim3(256), 0, st, x, mean, invstd, gamma, beta, residual, relu, y, relu_mask, totalv, CV, -1.0f, rbn, totalv);
    else if (vpt >= 4 && rows >= 16384) {
        const long part = ((rows + 3) / 4) * CV;
        hipLaunchKernelGGL((bn_apply_kernel<T, false, false, 4>), dim3(cdiv(part, 256)), dim3(256), 0, st, x, mean, invstd, gamma, beta, residual, relu, y, relu_mask, totalv, CV, -1.0f, rbn, part);
    } else if (vpt >= 2 && rows >= 4096) {
        const long part = ((rows + 1) / 2) * CV;
        hipLaunchKernelGGL((bn_apply_kernel<T, false, false, 2>), dim3(cdiv(part, 256)), dim3(256), 0, st, x, mean, invstd, gamma, beta, residual, relu, y, relu_mask, totalv, CV, -1.0f, rbn, part);
    } else
        hipLaunchKernelGGL((bn_apply_kernel<T, false, false>), dim3(cdiv(totalv, 256)), dim3(256), 0, st, x, mean, invstd, gamma, beta, residual, relu, y, relu_mask, totalv, CV, -1.0f, rbn, totalv);
}

template <typename T>
static int bn_train_fwd_t(const T* x, int64_t rows, int32_t C, const float* gamma, const float* beta, float eps, float momentum,
                          float* running_mean, float* running_var, float* save_mean, float* save_invstd, const T* residual, int32_t relu,
                          T* y, uint8_t* relu_mask, float* scratch, hipStream_t st, const float* tile_stats = nullptr, int tile_rows = 0,
                          ResBn rbn = ResBn{nullptr, nullptr, nullptr, nullptr}) {
    if (!x || !gamma || !beta || !save_mean || !save_invstd || !scratch) return fail(SAT_EINVAL, "bn_train_fwd: null pointer");
    SAT_REQUIRE(!rbn.mean || (residual && rbn.invstd && rbn.gamma && rbn.beta), "bn_train_fwd: the residual's BatchNorm needs the residual and all four vectors");
    SAT_REQUIRE(y || (!residual && !relu_mask), "bn_train_fwd: statistics only (y = NULL) takes no residual / mask");
    SAT_REQUIRE(rows > 0 && C > 0 && C % 4 == 0, "bn_train_fwd: rows=%ld C=%d (C must be a multiple of 4)", (long)rows, C);
    constexpr int E = EPT<T>::n;
    SAT_REQUIRE(C % E == 0, "bn_train_fwd: C=%d must be a multiple of %d for this storage type", C, E);
    SAT_REQUIRE(!relu_mask || (relu && C % 8 == 0), "bn_train_fwd: the ReLU sign mask needs relu and C %% 8 == 0 (C=%d)", C);
    int CV, nparts; long rp; bn_grid(rows, C, E, CV, rp, nparts);
    double* p0 = reinterpret_cast<double*>(scratch); double* p1 = p0 + (long)nparts * C;
    const T* shift_src = x;
    if (tile_stats) {        // statistics came out of the convolution's epilogue: combine the row tiles, no pass over x
        SAT_REQUIRE(tile_rows > 0, "bn_train_fwd: tile_rows=%d", tile_rows);
        const int ntiles = (int)cdiv(rows, (long)tile_rows);
        static const int fuse_upto = getenv("SAT_BN_FUSE_TILES") ? atoi(getenv("SAT_BN_FUSE_TILES")) : 256;
        if (dev_switch(SW_BN_ONEPASS) && bn_tile_finish_ok(ntiles)) {
            launch_bn_tile_finish(tile_stats, ntiles, C, 0, (long)rows, eps, momentum, save_mean, save_invstd, running_mean, running_var, st);
            SAT_TRY(launch_ok("bn_tile_finish"));
            if (!y) return SAT_OK;                   // statistics only: the caller normalises inside its own kernel (stem tail)
            long totalv = rows * (C / E);
            ProfScope prof("bn_apply_fwd", 0.0, (double)rows * C * (sizeof(T) * (residual ? 3 : 2) + (relu_mask ? 0.125 : 0.0)), st);
            launch_bn_apply_train<T>(x, save_mean, save_invstd, gamma, beta, residual, relu, y, relu_mask, totalv, C / E, rbn, st);
            return launch_ok("bn_apply");
        }
        if (ntiles <= fuse_upto) {
            hipLaunchKernelGGL(bn_tile_finalize_kernel, dim3(cdiv(C, 32)), dim3(256), 0, st, tile_stats, ntiles, C, (long)rows, eps, momentum, save_mean, save_invstd,
                               running_mean, running_var);
            SAT_TRY(launch_ok("bn_tile_finalize"));
            if (!y) return SAT_OK;                   // statistics only: the caller normalises inside its own kernel (stem tail)
            long totalv = rows * (C / E);
            ProfScope prof("bn_apply_fwd", 0.0, (double)rows * C * (sizeof(T) * (residual ? 3 : 2) + (relu_mask ? 0.125 : 0.0)), st);
            launch_bn_apply_train<T>(x, save_mean, save_invstd, gamma, beta, residual, relu, y, relu_mask, totalv, C / E, rbn, st);
            return launch_ok("bn_apply");
        }
        int np = cdiv(ntiles, 64); if (np > nparts) np = nparts; if (np < 1) np = 1;          // partials fit the scratch sized for nparts
        const int per = cdiv(ntiles, np); np = cdiv(ntiles, per);
        p1 = p0 + (long)np * C;
        hipLaunchKernelGGL(bn_tile_reduce_kernel, dim3(cdiv(C, 32), np), dim3(256), 0, st, tile_stats, ntiles, C, per, p0, p1);
        SAT_TRY(launch_ok("bn_tile_reduce"));
        nparts = np; shift_src = nullptr;
    } else {
        ProfScope prof("bn_stats_fwd", 0.0, (double)rows * C * sizeof(T), st);
        hipLaunchKernelGGL((bn_colstats_kernel<0, T>), dim3(cdiv(C / E, CV), nparts), dim3(256), 0, st, x, (const T*)nullptr, (const T*)nullptr,
                           (const unsigned char*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, (long)rows, C, CV, rp, p0, p1);
        SAT_TRY(launch_ok("bn_colstats<0>"));
    }
    hipLaunchKernelGGL(bn_fwd_finalize_kernel<T>, dim3(cdiv(C, 4)), dim3(256), 0, st, p0, p1, shift_src, nparts, C, (long)rows, eps, momentum, save_mean, save_invstd, running_mean, running_var);
    SAT_TRY(launch_ok("bn_fwd_finalize"));
    if (!y) return SAT_OK;
    long totalv = rows * (C / E);
    ProfScope prof("bn_apply_fwd", 0.0, (double)rows * C * (sizeof(T) * (residual ? 3 : 2) + (relu_mask ? 0.125 : 0.0)), st);
    launch_bn_apply_train<T>(x, save_mean, save_invstd, gamma, beta, residual, relu, y, relu_mask, totalv, C / E, rbn, st);
    return launch_ok("bn_apply");
}

template <typename T>
static int bn_eval_fwd_t(const T* x, int64_t rows, int32_t C, const float* running_mean, const float* running_var, float eps,
                         const float* gamma, const float* beta, const T* residual, int32_t relu, T* y, hipStream_t st) {
    if (!x || !running_mean || !running_var || !gamma || !beta || !y) return fail(SAT_EINVAL, "bn_eval_fwd: null pointer");
    SAT_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && eps >= 0.f, "bn_eval_fwd: bad shape");
    constexpr int E = EPT<T>::n;
    SAT_REQUIRE(C % E == 0, "bn_eval_fwd: C=%d must be a multiple of %d for this storage type", C, E);
    long totalv = rows * (C / E);
    hipLaunchKernelGGL((bn_apply_kernel<T, true>), dim3(cdiv(totalv, 256)), dim3(256), 0, st, x, running_mean, running_var, gamma, beta, residual, relu, y, (unsigned char*)nullptr, totalv, C / E, eps);
    return launch_ok("bn_apply(eval)");
}

template <typename T>
static int bn_train_bwd_t(const T* dy, const T* x, const T* y, int64_t rows, int32_t C, const float* save_mean, const float* save_invstd,
                          const float* gamma, int32_t relu, T* dx, float* dgamma, float* dbeta, T* dres, int32_t dres_accumulate,
                          const uint8_t* relu_mask, float* scratch, hipStream_t st, const float* tile_stats = nullptr, int tile_rows = 0) {
    if (!dy || !x || !save_mean || !save_invstd || !gamma || !dx || !dgamma || !dbeta || !scratch) return fail(SAT_EINVAL, "bn_train_bwd: null pointer");
    if (relu && !y && !relu_mask) return fail(SAT_EINVAL, "bn_train_bwd: relu needs the forward output or its sign mask");
    if (relu == 2 && !relu_mask) return fail(SAT_EINVAL, "bn_train_bwd: ReLU6 needs the forward's mask (the output alone does not tell 6 from > 6)");
    SAT_REQUIRE(!relu_mask || C % 8 == 0, "bn_train_bwd: the ReLU sign mask needs C %% 8 == 0 (C=%d)", C);
    SAT_REQUIRE(rows > 0 && C > 0 && C % 4 == 0, "bn_train_bwd: bad shape");
    constexpr int E = EPT<T>::n;
    SAT_REQUIRE(C % E == 0, "bn_train_bwd: C=%d must be a multiple of %d for this storage type", C, E);
    int CV, nparts; long rp; bn_grid(rows, C, E, CV, rp, nparts, 1);
    double* p0 = reinterpret_cast<double*>(scratch); double* p1 = p0 + (long)nparts * C;
    bool tiles_done = false;
    if (tile_stats) {        // (sum g, sum g * xhat) per row tile came out of the epilogue of the data-gradient launch that wrote dy: no pass over dy / x
        SAT_REQUIRE(tile_rows > 0, "bn_train_bwd: tile_rows=%d", tile_rows);
        const int ntiles = (int)cdiv(rows, (long)tile_rows);
        int np = cdiv(ntiles, 64); if (np > nparts) np = nparts; if (np < 1) np = 1;
        const int per = cdiv(ntiles, np); np = cdiv(ntiles, per);
        p1 = p0 + (long)np * C;
        if (dev_switch(SW_BN_ONEPASS) && bn_tile_finish_ok(ntiles)) {
            launch_bn_tile_finish(tile_stats, ntiles, C, 1, (long)rows, 0.f, 0.f, dbeta, dgamma, nullptr, nullptr, st);
            SAT_TRY(launch_ok("bn_tile_finish (backward)"));
            tiles_done = true;
        } else {
            hipLaunchKernelGGL(bn_tile_reduce_kernel, dim3(cdiv(C, 32), np), dim3(256), 0, st, tile_stats, ntiles, C, per, p0, p1);
            SAT_TRY(launch_ok("bn_tile_reduce (backward)"));
        }
        nparts = np;
    } else {
        ProfScope prof("bn_stats_bwd", 0.0, (double)rows * C * (sizeof(T) * 2 + (relu ? (relu_mask ? 0.125 : (double)sizeof(T)) : 0.0)), st);
        static const int inter = getenv("SAT_BN_INTERLEAVE") ? atoi(getenv("SAT_BN_INTERLEAVE")) : 1;      // row groups dealt round-robin to the parts (-5 %)
        hipLaunchKernelGGL((bn_colstats_kernel<1, T>), dim3(cdiv(C / E, CV), nparts), dim3(256), 0, st, x, dy, y, relu_mask, save_mean, save_invstd, relu, (long)rows, C, CV, inter ? -1L : rp, p0, p1);
        SAT_TRY(launch_ok("bn_colstats<1>"));
    }
    if (!tiles_done) {
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, p0, p1, nparts, C, dbeta, dgamma);
        SAT_TRY(launch_ok("bn_bwd_finalize"));
    }
    ProfScope prof("bn_apply_bwd", 0.0, (double)rows * C * (sizeof(T) * (3 + (dres ? (dres_accumulate ? 2 : 1) : 0)) + (relu ? (relu_mask ? 0.125 : (double)sizeof(T)) : 0.0)), st);
    launch_bn_bwd_apply<T>(x, dy, y, relu_mask, save_mean, save_invstd, gamma, dbeta, dgamma, relu, 1.0f / (float)rows, dx, dres, dres_accumulate, (long)rows, C / E, st);
    return launch_ok("bn_bwd_apply");
}

template <typename T>
static int stem_tail_fwd_t(const T* x, int32_t N, int32_t H, int32_t W, int32_t C, const float* mean, const float* invstd, const float* gamma, const float* beta,
                           T* y, uint8_t* argmax, hipStream_t st) {
    if (!x || !mean || !invstd || !gamma || !beta || !y || !argmax) return fail(SAT_EINVAL, "stem_tail_fwd: null pointer");
    constexpr int E = EPT<T>::n;
    SAT_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % E == 0, "stem_tail_fwd: N=%d H=%d W=%d C=%d (C must be a multiple of %d)", N, H, W, C, E);
    SAT_REQUIRE((long)N * H * W * (C / E) < (1L << 31), "stem_tail_fwd: more than 2^31 vectors");
    const int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
    const long totalv = (long)N * P * Q * (C / E);
    ProfScope prof("stem_tail_fwd", 0.0, (double)N * H * W * C * sizeof(T) + (double)N * P * Q * C * (sizeof(T) + 1), st);
    hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel<T>, dim3(cdiv(totalv, 256)), dim3(256), 0, st, x, mean, invstd, gamma, beta, y, argmax, H, W, C / E, P, Q, totalv);
    return launch_ok("bn_relu_maxpool_fwd");
}
template <typename T>
static int stem_tail_bwd_t(const T* dy, const uint8_t* argmax, const T* x, int32_t N, int32_t H, int32_t W, int32_t C, const float* mean, const float* invstd,
                           const float* gamma, const float* beta, T* dx, float* dgamma, float* dbeta, float* scratch, hipStream_t st) {
    if (!dy || !argmax || !x || !mean || !invstd || !gamma || !beta || !dx || !dgamma || !dbeta || !scratch) return fail(SAT_EINVAL, "stem_tail_bwd: null pointer");
    constexpr int E = EPT<T>::n;
    SAT_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % E == 0, "stem_tail_bwd: N=%d H=%d W=%d C=%d (C must be a multiple of %d)", N, H, W, C, E);
    SAT_REQUIRE((long)N * H * W * (C / E) < (1L << 31), "stem_tail_bwd: more than 2^31 vectors");
    const int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
    const long rows = (long)N * H * W, blocks = (long)N * P * Q;             // P, Q = ceil(H / 2), ceil(W / 2): the 2x2 blocks tile the map
    int CV, nparts; long bp; bn_grid(blocks, C, E, CV, bp, nparts, 1);
    double* p0 = reinterpret_cast<double*>(scratch); double* p1 = p0 + (long)nparts * C;
    {
        ProfScope prof("stem_tail_bwd_stats", 0.0, (double)rows * C * sizeof(T) + (double)N * P * Q * C * (sizeof(T) + 1), st);
        hipLaunchKernelGGL(bn_relu_maxpool_bwd_stats_kernel<T>, dim3(cdiv(C / E, CV), nparts), dim3(256), 0, st, x, dy, argmax, mean, invstd, gamma, beta, H, W, P, Q,
                           blocks, C, CV, bp, p0, p1);
        SAT_TRY(launch_ok("bn_relu_maxpool_bwd_stats"));
    }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, p0, p1, nparts, C, dbeta, dgamma);
    SAT_TRY(launch_ok("bn_bwd_finalize"));
    const long totalb = blocks * (C / E);
    ProfScope prof("stem_tail_bwd_apply", 0.0, (double)rows * C * sizeof(T) * 2 + (double)N * P * Q * C * (sizeof(T) + 1), st);
    hipLaunchKernelGGL(bn_relu_maxpool_bwd_apply_kernel<T>, dim3(cdiv(totalb, 256)), dim3(256), 0, st, x, dy, argmax, mean, invstd, gamma, beta, dbeta, dgamma,
                       1.0f / (float)rows, dx, H, W, P, Q, totalb, C / E);
    return launch_ok("bn_relu_maxpool_bwd_apply");
}

template <typename T>
static int maxpool_fwd_t(const T* x, T* y, uint8_t* argmax, int32_t N, int32_t H, int32_t W, int32_t C, hipStream_t st) {
    if (!x || !y || !argmax) return fail(SAT_EINVAL, "maxpool_fwd: null pointer");
    SAT_REQUIRE(C % 4 == 0, "maxpool: C must be a multiple of 4");
    int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
    long total4 = (long)N * P * Q * (C / 4);
    constexpr int E = EPT<T>::n;
    if (C % E == 0 && (long)N * H * W * (C / E) < (1L << 31)) {
        const long totalv = total4 * 4 / E;
        hipLaunchKernelGGL(maxpool3x3s2_fwd_v_kernel<T>, dim3(cdiv(totalv, 256)), dim3(256), 0, st, x, y, argmax, H, W, C / E, P, Q, totalv);
        return launch_ok("maxpool_fwd");
    }
    hipLaunchKernelGGL(maxpool3x3s2_fwd_kernel<T>, dim3(cdiv(total4, 256)), dim3(256), 0, st, x, y, argmax, H, W, C / 4, P, Q, total4);
    return launch_ok("maxpool_fwd");
}
template <typename T>
static int maxpool_bwd_t(const T* dy, const uint8_t* argmax, T* dx, int32_t N, int32_t H, int32_t W, int32_t C, hipStream_t st) {
    if (!dy || !dx || !argmax) return fail(SAT_EINVAL, "maxpool_bwd: null pointer");
    SAT_REQUIRE(C % 4 == 0, "maxpool: C must be a multiple of 4");
    int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
    long total4 = (long)N * H * W * (C / 4);
    constexpr int E = EPT<T>::n;
    if (C % E == 0 && total4 * 4 / E < (1L << 31)) {
        const long totalv = total4 * 4 / E;
        hipLaunchKernelGGL(maxpool3x3s2_bwd_v_kernel<T>, dim3(cdiv(totalv, 256)), dim3(256), 0, st, dy, argmax, dx, H, W, C / E, P, Q, totalv);
        return launch_ok("maxpool_bwd");
    }
    hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel<T>, dim3(cdiv(total4, 256)), dim3(256), 0, st, dy, argmax, dx, H, W, C / 4, P, Q, total4);
    return launch_ok("maxpool_bwd");
}

extern "C" {

#define SAT_BY_DTYPE(dtype, CALLF, CALLB)                                                \
    do {                                                                                 \
        if ((dtype) == 0) return CALLF;                                                  \
        if ((dtype) == 1) return CALLB;                                                  \
        return fail(SAT_EINVAL, "dtype %d (0 = fp32, 1 = bf16)", (int)(dtype));           \
    } while (0)
typedef __bf16 bf;

int sat_bn_train_fwd_t(int32_t dtype, const void* x, int64_t rows, int32_t C, const float* gamma, const float* beta, float eps, float momentum,
                       float* running_mean, float* running_var, float* save_mean, float* save_invstd, const void* residual, int32_t relu,
                       void* y, uint8_t* relu_mask, float* scratch, void* stream) {
    SAT_BY_DTYPE(dtype,
        bn_train_fwd_t<float>((const float*)x, rows, C, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, (const float*)residual, relu, (float*)y, relu_mask, scratch, (hipStream_t)stream),
        bn_train_fwd_t<bf>((const bf*)x, rows, C, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, (const bf*)residual, relu, (bf*)y, relu_mask, scratch, (hipStream_t)stream));
}
int sat_bn_train_fwd_tiles_bf16(const void* x, int64_t rows, int32_t C, const float* tile_stats, int32_t tile_rows, const float* gamma, const float* beta,
                                float eps, float momentum, float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                                const void* residual, int32_t relu, void* y, uint8_t* relu_mask, float* scratch, void* stream) {
    if (!tile_stats) return fail(SAT_EINVAL, "bn_train_fwd_tiles: null statistics");
    return bn_train_fwd_t<bf>((const bf*)x, rows, C, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, (const bf*)residual, relu,
                              (bf*)y, relu_mask, scratch, (hipStream_t)stream, tile_stats, tile_rows);
}
int sat_bn_train_fwd_tiles_bf16_resbn(const void* x, int64_t rows, int32_t C, const float* tile_stats, int32_t tile_rows, const float* gamma, const float* beta,
                                      float eps, float momentum, float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                                      const void* residual_raw, const float* res_mean, const float* res_invstd, const float* res_gamma, const float* res_beta,
                                      int32_t relu, void* y, uint8_t* relu_mask, float* scratch, void* stream) {
    if (!tile_stats || !residual_raw || !res_mean || !res_invstd || !res_gamma || !res_beta) return fail(SAT_EINVAL, "bn_train_fwd_tiles_resbn: null pointer");
    return bn_train_fwd_t<bf>((const bf*)x, rows, C, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, (const bf*)residual_raw, relu,
                              (bf*)y, relu_mask, scratch, (hipStream_t)stream, tile_stats, tile_rows, ResBn{res_mean, res_invstd, res_gamma, res_beta});
}
int sat_bn_train_fwd(const float* x, int64_t rows, int32_t C, const float* gamma, const float* beta, float eps, float momentum,
                     float* running_mean, float* running_var, float* save_mean, float* save_invstd, const float* residual, int32_t relu,
                     float* y, float* scratch, void* stream) {
    return sat_bn_train_fwd_t(0, x, rows, C, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, residual, relu, y, nullptr, scratch, stream);
}
int sat_bn_eval_fwd_t(int32_t dtype, const void* x, int64_t rows, int32_t C, const float* running_mean, const float* running_var, float eps,
                      const float* gamma, const float* beta, const void* residual, int32_t relu, void* y, void* stream) {
    SAT_BY_DTYPE(dtype,
        bn_eval_fwd_t<float>((const float*)x, rows, C, running_mean, running_var, eps, gamma, beta, (const float*)residual, relu, (float*)y, (hipStream_t)stream),
        bn_eval_fwd_t<bf>((const bf*)x, rows, C, running_mean, running_var, eps, gamma, beta, (const bf*)residual, relu, (bf*)y, (hipStream_t)stream));
}
int sat_bn_eval_fwd(const float* x, int64_t rows, int32_t C, const float* running_mean, const float* running_var, float eps,
                    const float* gamma, const float* beta, const float* residual, int32_t relu, float* y, void* stream) {
    return sat_bn_eval_fwd_t(0, x, rows, C, running_mean, running_var, eps, gamma, beta, residual, relu, y, stream);
}
int sat_bn_train_bwd_t(int32_t dtype, const void* dy, const void* x, const void* y, int64_t rows, int32_t C, const float* save_mean,
                       const float* save_invstd, const float* gamma, int32_t relu, void* dx, float* dgamma, float* dbeta, void* dres,
                       int32_t dres_accumulate, const uint8_t* relu_mask, float* scratch, void* stream) {
    SAT_BY_DTYPE(dtype,
        bn_train_bwd_t<float>((const float*)dy, (const float*)x, (const float*)y, rows, C, save_mean, save_invstd, gamma, relu, (float*)dx, dgamma, dbeta, (float*)dres, dres_accumulate, relu_mask, scratch, (hipStream_t)stream),
        bn_train_bwd_t<bf>((const bf*)dy, (const bf*)x, (const bf*)y, rows, C, save_mean, save_invstd, gamma, relu, (bf*)dx, dgamma, dbeta, (bf*)dres, dres_accumulate, relu_mask, scratch, (hipStream_t)stream));
}
int sat_bn_train_bwd_tiles_bf16(const void* dy, const void* x, int64_t rows, int32_t C, const float* tile_stats, int32_t tile_rows, const float* save_mean,
                                const float* save_invstd, const float* gamma, int32_t relu, void* dx, float* dgamma, float* dbeta, void* dres,
                                int32_t dres_accumulate, const uint8_t* relu_mask, float* scratch, void* stream) {
    if (!tile_stats || tile_rows <= 0) return fail(SAT_EINVAL, "bn_train_bwd_tiles_bf16: no tile statistics");
    if (relu && !relu_mask) return fail(SAT_EINVAL, "bn_train_bwd_tiles_bf16: relu needs the forward's sign mask");
    return bn_train_bwd_t<bf>((const bf*)dy, (const bf*)x, (const bf*)nullptr, rows, C, save_mean, save_invstd, gamma, relu, (bf*)dx, dgamma, dbeta, (bf*)dres,
                              dres_accumulate, relu_mask, scratch, (hipStream_t)stream, tile_stats, tile_rows);
}
int sat_bn_train_bwd(const float* dy, const float* x, const float* y, int64_t rows, int32_t C, const float* save_mean, const float* save_invstd,
                     const float* gamma, int32_t relu, float* dx, float* dgamma, float* dbeta, float* dres, int32_t dres_accumulate,
                     float* scratch, void* stream) {
    return sat_bn_train_bwd_t(0, dy, x, y, rows, C, save_mean, save_invstd, gamma, relu, dx, dgamma, dbeta, dres, dres_accumulate, nullptr, scratch, stream);
}
int sat_stem_tail_fwd_t(int32_t dtype, const void* x, int32_t N, int32_t H, int32_t W, int32_t C, const float* mean, const float* invstd, const float* gamma,
                        const float* beta, void* y_pool, uint8_t* argmax, void* stream) {
    SAT_BY_DTYPE(dtype, stem_tail_fwd_t<float>((const float*)x, N, H, W, C, mean, invstd, gamma, beta, (float*)y_pool, argmax, (hipStream_t)stream),
                 stem_tail_fwd_t<bf>((const bf*)x, N, H, W, C, mean, invstd, gamma, beta, (bf*)y_pool, argmax, (hipStream_t)stream));
}
int sat_stem_tail_bwd_t(int32_t dtype, const void* dy_pool, const uint8_t* argmax, const void* x, int32_t N, int32_t H, int32_t W, int32_t C, const float* mean,
                        const float* invstd, const float* gamma, const float* beta, void* dx, float* dgamma, float* dbeta, float* scratch, void* stream) {
    SAT_BY_DTYPE(dtype, stem_tail_bwd_t<float>((const float*)dy_pool, argmax, (const float*)x, N, H, W, C, mean, invstd, gamma, beta, (float*)dx, dgamma, dbeta, scratch, (hipStream_t)stream),
                 stem_tail_bwd_t<bf>((const bf*)dy_pool, argmax, (const bf*)x, N, H, W, C, mean, invstd, gamma, beta, (bf*)dx, dgamma, dbeta, scratch, (hipStream_t)stream));
}
int sat_maxpool3x3s2_fwd_t(int32_t dtype, const void* x, void* y, uint8_t* argmax, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    SAT_BY_DTYPE(dtype, maxpool_fwd_t<float>((const float*)x, (float*)y, argmax, N, H, W, C, (hipStream_t)stream),
                 maxpool_fwd_t<bf>((const bf*)x, (bf*)y, argmax, N, H, W, C, (hipStream_t)stream));
}
int sat_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* argmax, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    return sat_maxpool3x3s2_fwd_t(0, x, y, argmax, N, H, W, C, stream);
}
int sat_maxpool3x3s2_bwd_t(int32_t dtype, const void* dy, const uint8_t* argmax, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    SAT_BY_DTYPE(dtype, maxpool_bwd_t<float>((const float*)dy, argmax, (float*)dx, N, H, W, C, (hipStream_t)stream),
                 maxpool_bwd_t<bf>((const bf*)dy, argmax, (bf*)dx, N, H, W, C, (hipStream_t)stream));
}
int sat_maxpool3x3s2_bwd(const float* dy, const uint8_t* argmax, float* dx, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    return sat_maxpool3x3s2_bwd_t(0, dy, argmax, dx, N, H, W, C, stream);
}

/* bf16 plumbing: casts and the 8-channel stem layout */
int sat_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
    if (!src || !dst) return fail(SAT_EINVAL, "cast: null pointer");
    SAT_REQUIRE(n % 4 == 0, "cast: element count must be a multiple of 4");
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, src, (bf*)dst, (long)(n / 4));
    return launch_ok("cast_f32_bf16");
}
int sat_image_normalize_nhwc8_bf16(const float* img_nchw, void* out_nhwc8, int32_t N, int32_t H, int32_t W, const float* mean3_host, const float* std3_host, void* stream) {
    if (!img_nchw || !out_nhwc8 || !mean3_host || !std3_host) return fail(SAT_EINVAL, "image_normalize: null pointer");
    long total = (long)N * H * W;
    hipLaunchKernelGGL(normalize_nhwc8_bf16_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, img_nchw, (bf*)out_nhwc8, H, W, total,
                       mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2]);
    return launch_ok("normalize_nhwc8_bf16");
}
int sat_image_normalize_nhwc4_padded_bf16(const float* img_nchw, void* out, int32_t N, int32_t H, int32_t W, const float* mean3_host, const float* std3_host, void* stream) {
    if (!img_nchw || !out || !mean3_host || !std3_host) return fail(SAT_EINVAL, "image_normalize: null pointer");
    SAT_REQUIRE(N > 0 && H > 0 && W > 0 && W % 2 == 0, "image_normalize (padded): N=%d H=%d W=%d (W must be even)", N, H, W);
    const long total = (long)N * (H + 6) * (W + 6);
    hipLaunchKernelGGL(normalize_nhwc4_padded_bf16_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, img_nchw, (bf*)out, H, W, total,
                       mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2]);
    return launch_ok("normalize_nhwc4_padded_bf16");
}
int sat_grouped_filter_expand(const void* w_grouped, void* w_dense, int32_t K, int32_t C, int32_t RS, int32_t groups, int32_t bf16, void* stream) {
    if (!w_grouped || !w_dense || K <= 0 || C <= 0 || RS <= 0 || groups <= 0 || K % groups || C % groups) return fail(SAT_EINVAL, "grouped_filter_expand: bad argument (K=%d C=%d groups=%d)", K, C, groups);
    const long n = (long)K * RS * C;
    if (bf16) hipLaunchKernelGGL(grouped_filter_expand_kernel<bf>, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const bf*)w_grouped, (bf*)w_dense, K, C, RS, groups);
    else hipLaunchKernelGGL(grouped_filter_expand_kernel<float>, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)w_grouped, (float*)w_dense, K, C, RS, groups);
    return launch_ok("grouped_filter_expand");
}
int sat_grouped_filter_grad_extract(const float* dw_dense, float* dw_grouped, int32_t K, int32_t C, int32_t RS, int32_t groups, void* stream) {
    if (!dw_dense || !dw_grouped || K <= 0 || C <= 0 || RS <= 0 || groups <= 0 || K % groups || C % groups) return fail(SAT_EINVAL, "grouped_filter_grad_extract: bad argument");
    const long n = (long)K * RS * (C / groups);
    hipLaunchKernelGGL(grouped_filter_grad_extract_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dw_dense, dw_grouped, K, C, RS, groups);
    return launch_ok("grouped_filter_grad_extract");
}
int sat_stem_filter_pairs(const float* w3, void* w_pairs_bf16, int32_t K, void* stream) {
    if (!w3 || !w_pairs_bf16 || K <= 0) return fail(SAT_EINVAL, "stem_filter_pairs: bad argument");
    const long n = (long)K * 7 * 4;
    hipLaunchKernelGGL(stem_filter_pairs_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w3, (bf*)w_pairs_bf16, n);
    return launch_ok("stem_filter_pairs");
}
int sat_stem_filter_grad_unpairs(const float* dw_pairs, float* dw3, int32_t K, void* stream) {
    if (!dw_pairs || !dw3 || K <= 0) return fail(SAT_EINVAL, "stem_filter_grad_unpairs: bad argument");
    const long n = (long)K * 7 * 7;
    hipLaunchKernelGGL(stem_filter_grad_unpairs_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dw_pairs, dw3, n);
    return launch_ok("stem_filter_grad_unpairs");
}
int sat_stem_filter_pad(const float* w3, void* w8_bf16, int64_t pixels, void* stream) {
    if (!w3 || !w8_bf16) return fail(SAT_EINVAL, "stem_filter_pad: null pointer");
    hipLaunchKernelGGL(pad_c3_to_c8_bf16_kernel, dim3(cdiv(pixels, 256)), dim3(256), 0, (hipStream_t)stream, w3, (bf*)w8_bf16, (long)pixels);
    return launch_ok("pad_c3_to_c8_bf16");
}
int sat_stem_filter_grad_unpad(const float* dw8, float* dw3, int64_t pixels, void* stream) {
    if (!dw8 || !dw3) return fail(SAT_EINVAL, "stem_filter_grad_unpad: null pointer");
    hipLaunchKernelGGL(unpad_c8_to_c3_kernel, dim3(cdiv(pixels, 256)), dim3(256), 0, (hipStream_t)stream, dw8, dw3, (long)pixels);
    return launch_ok("unpad_c8_to_c3");
}

int sat_resize_fwd(const float* x, float* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t P, int32_t Q, void* stream) {
    if (!x || !y) return fail(SAT_EINVAL, "resize_fwd: null pointer");
    long total = (long)N * P * Q * C;
    SAT_REQUIRE(total < (1L << 32) && (long)N * H * W * C < (1L << 32), "resize: more than 2^32 elements");
    if (P <= H) {
        if (C % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0)
            hipLaunchKernelGGL(adaptive_avgpool_kernel<4>, dim3(cdiv(total / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, C, P, Q, total / 4, 0);
        else hipLaunchKernelGGL(adaptive_avgpool_kernel<1>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, C, P, Q, total, 0);
    }
    else hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, C, P, Q, total);
    return launch_ok("resize_fwd");
}
int sat_resize_bwd(const float* dy, float* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t P, int32_t Q, void* stream) {
    if (!dy || !dx) return fail(SAT_EINVAL, "resize_bwd: null pointer");
    long total = (long)N * H * W * C;
    SAT_REQUIRE(total < (1L << 32) && (long)N * P * Q * C < (1L << 32), "resize: more than 2^32 elements");
    if (P <= H) {
        if (C % 4 == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0 && (reinterpret_cast<uintptr_t>(dx) & 15) == 0)
            hipLaunchKernelGGL(adaptive_avgpool_kernel<4>, dim3(cdiv(total / 4, 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, H, W, C, P, Q, total / 4, 1);
        else hipLaunchKernelGGL(adaptive_avgpool_kernel<1>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, H, W, C, P, Q, total, 1);
    }
    else hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, H, W, C, P, Q, total);
    return launch_ok("resize_bwd");
}

}  // extern "C"
