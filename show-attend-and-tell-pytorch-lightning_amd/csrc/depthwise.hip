// Depthwise 3x3 convolutions and the channel shuffle of ShuffleNetV2 units (the reference's CLI-default encoder is torchvision's
// shufflenet_v2_x0_5: train.py:43, model.py:30-31 keeps everything but the classifier).  NHWC activations (fp32 or bf16), fp32 accumulation.
//
//   depthwise 3x3, pad 1, stride 1 | 2 (InvertedResidual.depthwise_conv): every channel has its own 3x3 filter - 18 FLOP per output element
//     against 2 - 4 bytes of traffic: memory-bound by a wide margin, so these are plain streaming kernels (16 bytes per lane along the
//     channels, neighbours from L1 / L2), not MFMA work;
//   channel_shuffle(cat(a, b), groups = 2): out[c] = (c even ? a : b)[c / 2] - an interleave of the two branches.  The next stride-1 unit
//     splits the result into halves (x.chunk(2, dim = 1)); the interleave therefore writes either the full tensor (stride-2 consumer,
//     conv5) or the two halves as separate dense tensors, so that no consumer ever reads a strided channel slice.
#include "../../include/sat_hip.h"
#include "common.h"

namespace sat {
namespace {

typedef __bf16 bf;

template <typename T> struct Vec;      // V channels per thread = 16 bytes
template <> struct Vec<float> { static constexpr int V = 4; };
template <> struct Vec<bf> { static constexpr int V = 8; };

template <typename T, int V> __device__ __forceinline__ void load_vec(const T* p, float (&o)[V]) {
    if constexpr (sizeof(T) == 4) { const float4 q = *reinterpret_cast<const float4*>(p); o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w; }
    else {
        typedef __bf16 b8 __attribute__((ext_vector_type(8)));
        const b8 q = *reinterpret_cast<const b8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (float)q[i];
    }
}
template <typename T, int V> __device__ __forceinline__ void store_vec(T* p, const float (&o)[V]) {
    if constexpr (sizeof(T) == 4) *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    else {
        typedef __bf16 b8 __attribute__((ext_vector_type(8)));
        b8 q;
#pragma unroll
        for (int i = 0; i < 8; ++i) q[i] = (__bf16)o[i];
        *reinterpret_cast<b8*>(p) = q;
    }
}

// w: the (C, 1, 3, 3) parameter in its own memory order = [C][9] fp32 (master weights).  A thread keeps the 9 x V taps of its V channels in
// registers (V * 9 consecutive floats of w: 16-byte loads) and walks R consecutive rows of one column of the map: per element 9 (overlapping,
// L1-served) 16-byte loads and one store.  (The first version read its taps with one 4-byte load per tap and channel, 72 of them per output
// vector: 0.03 - 0.1 of the HBM peak on the 56 x 56 x 144 maps of a 128-image mobilenet_v2 batch.)
template <int V>
__device__ __forceinline__ void load_taps(const float* __restrict__ w, int c, float (&wr)[9][V]) {
    float flat[V * 9];
    const float4* wp = reinterpret_cast<const float4*>(w + (long)c * 9);          // c % V == 0: (c * 9) floats = a multiple of 16 bytes
#pragma unroll
    for (int j = 0; j < V * 9 / 4; ++j) { const float4 f = wp[j]; flat[4 * j] = f.x; flat[4 * j + 1] = f.y; flat[4 * j + 2] = f.z; flat[4 * j + 3] = f.w; }
#pragma unroll
    for (int i = 0; i < V; ++i)
#pragma unroll
        for (int k = 0; k < 9; ++k) wr[k][i] = flat[i * 9 + k];
}

template <typename T>
__global__ __launch_bounds__(256) void dw3x3_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y, int N, int H, int W, int C,
                                                        int P, int Q, int stride, int R) {
    constexpr int V = Vec<T>::V;
    const int cv = C / V, PB = (P + R - 1) / R;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)N * PB * Q * cv) return;
    const int c = (int)(e % cv) * V; long t = e / cv;
    const int q = (int)(t % Q); t /= Q; const int pb = (int)(t % PB); const int n = (int)(t / PB);
    float wr[9][V];
    load_taps<V>(w, c, wr);
    for (int p = pb * R; p < pb * R + R && p < P; ++p) {
        float acc[V];
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int h = p * stride + r - 1;
            if (h < 0 || h >= H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int ww = q * stride + s - 1;
                if (ww < 0 || ww >= W) continue;
                float xv[V];
                load_vec<T, V>(x + (((long)n * H + h) * W + ww) * C + c, xv);
#pragma unroll
                for (int i = 0; i < V; ++i) acc[i] = fmaf(xv[i], wr[r * 3 + s][i], acc[i]);
            }
        }
        store_vec<T, V>(y + (((long)n * P + p) * Q + q) * C + c, acc);
    }
}

// dx[n, h, w, c] = sum over taps (r, s) with (h + 1 - r) = p * stride, (w + 1 - s) = q * stride of dy[n, p, q, c] * w[c, r, s]
template <typename T>
__global__ __launch_bounds__(256) void dw3x3_dgrad_kernel(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx, int N, int H, int W, int C,
                                                          int P, int Q, int stride, int R) {
    constexpr int V = Vec<T>::V;
    const int cv = C / V, HB = (H + R - 1) / R;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)N * HB * W * cv) return;
    const int c = (int)(e % cv) * V; long t = e / cv;
    const int ww = (int)(t % W); t /= W; const int hb = (int)(t % HB); const int n = (int)(t / HB);
    float wr[9][V];
    load_taps<V>(w, c, wr);
    for (int h = hb * R; h < hb * R + R && h < H; ++h) {
        float acc[V];
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int ph = h + 1 - r;
            if (ph < 0 || ph % stride) continue;
            const int p = ph / stride;
            if (p >= P) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int qw = ww + 1 - s;
                if (qw < 0 || qw % stride) continue;
                const int q = qw / stride;
                if (q >= Q) continue;
                float gv[V];
                load_vec<T, V>(dy + (((long)n * P + p) * Q + q) * C + c, gv);
#pragma unroll
                for (int i = 0; i < V; ++i) acc[i] = fmaf(gv[i], wr[r * 3 + s][i], acc[i]);
            }
        }
        store_vec<T, V>(dx + (((long)n * H + h) * W + ww) * C + c, acc);
    }
}

// Stride 1 (13 of the 16 / 17 depthwise layers of shufflenet_v2 / mobilenet_v2), forward and - with the taps flipped - data gradient: the map
// keeps its size and consecutive rows of one column share two of their three input rows.  The thread keeps a rolling window of 3 x 3 packed
// input vectors: 3 new 16-byte loads per output vector instead of 9.  Out-of-range taps are zero vectors (they add nothing).
template <typename T>
__device__ __forceinline__ void fetch_row(const T* __restrict__ xn, int h, int q, int H, int W, int C, uint4& a, uint4& b, uint4& c) {
    const bool hv = h >= 0 && h < H;
    const uint4* p = reinterpret_cast<const uint4*>(xn + ((long)h * W + q) * C);
    const int cs = C * (int)sizeof(T) / 16;          // one pixel to the right, in 16-byte units
    a = (hv && q > 0) ? p[-cs] : make_uint4(0u, 0u, 0u, 0u);
    b = hv ? p[0] : make_uint4(0u, 0u, 0u, 0u);
    c = (hv && q + 1 < W) ? p[cs] : make_uint4(0u, 0u, 0u, 0u);
}
template <typename T, int V>
__device__ __forceinline__ void tap_fma(const uint4& pk, const float (&wk)[V], float (&acc)[V]) {
    if constexpr (sizeof(T) == 4) {
        acc[0] = fmaf(__uint_as_float(pk.x), wk[0], acc[0]); acc[1] = fmaf(__uint_as_float(pk.y), wk[1], acc[1]);
        acc[2] = fmaf(__uint_as_float(pk.z), wk[2], acc[2]); acc[3] = fmaf(__uint_as_float(pk.w), wk[3], acc[3]);
    } else {
        acc[0] = fmaf(__uint_as_float(pk.x << 16), wk[0], acc[0]); acc[1] = fmaf(__uint_as_float(pk.x & 0xffff0000u), wk[1], acc[1]);
        acc[2] = fmaf(__uint_as_float(pk.y << 16), wk[2], acc[2]); acc[3] = fmaf(__uint_as_float(pk.y & 0xffff0000u), wk[3], acc[3]);
        acc[4] = fmaf(__uint_as_float(pk.z << 16), wk[4], acc[4]); acc[5] = fmaf(__uint_as_float(pk.z & 0xffff0000u), wk[5], acc[5]);
        acc[6] = fmaf(__uint_as_float(pk.w << 16), wk[6], acc[6]); acc[7] = fmaf(__uint_as_float(pk.w & 0xffff0000u), wk[7], acc[7]);
    }
}
template <typename T, bool FLIP>
__global__ __launch_bounds__(256) void dw3x3_s1_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y, int N, int H, int W, int C, int R) {
    constexpr int V = Vec<T>::V;
    const int cv = C / V, HB = (H + R - 1) / R;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)N * HB * W * cv) return;
    const int c = (int)(e % cv) * V; long t = e / cv;
    const int q = (int)(t % W); t /= W; const int hb = (int)(t % HB); const int n = (int)(t / HB);
    float wr[9][V];
    load_taps<V>(w, c, wr);
    const T* xn = x + (long)n * H * W * C + c;
    const int h0 = hb * R, h1 = (h0 + R < H) ? h0 + R : H;
    uint4 a0, a1, a2, b0, b1, b2, c0, c1, c2;          // rows h - 1, h, h + 1; columns q - 1, q, q + 1  (a fourth, prefetched row costs the third wave per SIMD: 176 VGPRs)
    fetch_row<T>(xn, h0 - 1, q, H, W, C, a0, a1, a2); fetch_row<T>(xn, h0, q, H, W, C, b0, b1, b2);
    constexpr int K0 = FLIP ? 8 : 0, KS = FLIP ? -1 : 1;          // tap of window position (row a, column b): a * 3 + b, or its mirror image
    for (int h = h0; h < h1; ++h) {
        fetch_row<T>(xn, h + 1, q, H, W, C, c0, c1, c2);
        float acc[V];
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] = 0.f;
        tap_fma<T, V>(a0, wr[K0 + KS * 0], acc); tap_fma<T, V>(a1, wr[K0 + KS * 1], acc); tap_fma<T, V>(a2, wr[K0 + KS * 2], acc);
        tap_fma<T, V>(b0, wr[K0 + KS * 3], acc); tap_fma<T, V>(b1, wr[K0 + KS * 4], acc); tap_fma<T, V>(b2, wr[K0 + KS * 5], acc);
        tap_fma<T, V>(c0, wr[K0 + KS * 6], acc); tap_fma<T, V>(c1, wr[K0 + KS * 7], acc); tap_fma<T, V>(c2, wr[K0 + KS * 8], acc);
        store_vec<T, V>(y + (((long)n * H + h) * W + q) * C + c, acc);
        a0 = b0; a1 = b1; a2 = b2; b0 = c0; b1 = c1; b2 = c2;
    }
}

// rows per thread: 8 when that still leaves >= ~4 waves per SIMD of the chip, fewer on small maps
static inline int dw_rows(long vectors) {
    const long r = vectors / 262144;
    return (int)(r < 1 ? 1 : (r > 8 ? 8 : r));
}

// dw[c][r][s] = sum over output pixels of dy * x(tap): block (channel vector group, pixel chunk) -> partial [chunk][9][C], then a fixed-order finish
// output pixels per block: ~1024 blocks per launch (the maps of a 32-image batch have 1.5 k .. 100 k pixels: with a fixed 2048-pixel chunk the
// stage-2 launch was 12 blocks on 256 CUs and took 185 us), at least 64 pixels, at most 2048
static inline int dw_chunk(long npix) {
    long c = (npix + 1023) / 1024;
    c = (c + 63) / 64 * 64;
    return (int)(c < 64 ? 64 : (c > 2048 ? 2048 : c));
}
template <typename T>
__global__ __launch_bounds__(256) void dw3x3_wgrad_part_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ part, int N, int H, int W, int C,
                                                               int P, int Q, int stride, int chunk, int cvb) {
    constexpr int V = Vec<T>::V;
    const int cv = C / V;
    // block = (pixel chunk blockIdx.x, group blockIdx.y of <= cvb channel vectors); thread = channel vector tid % cvb of the group, pixel lane
    // tid / cvb: 256 / cvb pixels of the chunk in flight (16 at least - with all of a 960-channel map's 120 vectors in one block it was 2)
    const int pix_par = blockDim.x / cvb;
    const int vl = threadIdx.x % cvb, pl = threadIdx.x / cvb;
    const int vi = blockIdx.y * cvb + vl;
    const bool live = pl < pix_par && vi < cv;
    const int c = vi * V;
    float acc[9][V];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int i = 0; i < V; ++i) acc[k][i] = 0.f;
    const long npix = (long)N * P * Q;
    const long p0 = (long)blockIdx.x * chunk, p1 = p0 + chunk < npix ? p0 + chunk : npix;
    if (live) {
        for (long pix = p0 + pl; pix < p1; pix += pix_par) {
            const int q = (int)(pix % Q); long t = pix / Q; const int p = (int)(t % P); const int n = (int)(t / P);
            float gv[V];
            load_vec<T, V>(dy + pix * C + c, gv);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int h = p * stride + r - 1;
                if (h < 0 || h >= H) continue;
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int ww = q * stride + s - 1;
                    if (ww < 0 || ww >= W) continue;
                    float xv[V];
                    load_vec<T, V>(x + (((long)n * H + h) * W + ww) * C + c, xv);
#pragma unroll
                    for (int i = 0; i < V; ++i) acc[r * 3 + s][i] = fmaf(gv[i], xv[i], acc[r * 3 + s][i]);
                }
            }
        }
    }
    // combine the block's pixel lanes in a fixed order through LDS, one tap at a time: [pixel lane][cvb * V]
    extern __shared__ float sm[];
    const int CB = cvb * V;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        if (pl < pix_par)
#pragma unroll
            for (int i = 0; i < V; ++i) sm[pl * CB + vl * V + i] = acc[k][i];
        __syncthreads();
        for (int o = threadIdx.x; o < CB; o += blockDim.x) {
            const int ch = blockIdx.y * CB + o;
            if (ch < C) {
                float t = 0.f;
                for (int l = 0; l < pix_par; ++l) t += sm[l * CB + o];
                part[((long)blockIdx.x * 9 + k) * C + ch] = t;
            }
        }
        __syncthreads();
    }
}
// Stride 1: the filter gradient with the rolling window of dw3x3_s1_kernel.  A thread owns one channel vector of one column of the map over R rows
// (3 new input vectors + 1 gradient vector per pixel instead of 9 + 1); a block = `cvb` channel vectors x 256 / cvb columns, combined through
// LDS in a fixed order; one partial [9][C] slice per block.
template <typename T>
__global__ __launch_bounds__(256) void dw3x3_wgrad_s1_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ part, int N, int H, int W, int C,
                                                             int R, int cvb) {
    constexpr int V = Vec<T>::V;
    const int cv = C / V, HB = (H + R - 1) / R;
    const int pix_par = blockDim.x / cvb;
    const int vl = threadIdx.x % cvb, pl = threadIdx.x / cvb;
    const int vi = blockIdx.y * cvb + vl;
    const bool live = pl < pix_par && vi < cv;
    const int c = vi * V;
    float acc[9][V];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int i = 0; i < V; ++i) acc[k][i] = 0.f;
    const long ncols = (long)N * HB * W;
    for (long col = (long)blockIdx.x * pix_par + pl; live && col < ncols; col += (long)gridDim.x * pix_par) {          // column = (n, hb, q)
        const int q = (int)(col % W); long t = col / W; const int hb = (int)(t % HB); const int n = (int)(t / HB);
        const T* xn = x + (long)n * H * W * C + c;
        const T* gn = dy + (long)n * H * W * C + c;
        const int h0 = hb * R, h1 = (h0 + R < H) ? h0 + R : H;
        uint4 a0, a1, a2, b0, b1, b2, c0, c1, c2;
        fetch_row<T>(xn, h0 - 1, q, H, W, C, a0, a1, a2); fetch_row<T>(xn, h0, q, H, W, C, b0, b1, b2);
        for (int h = h0; h < h1; ++h) {
            fetch_row<T>(xn, h + 1, q, H, W, C, c0, c1, c2);
            float gv[V];
            load_vec<T, V>(gn + ((long)h * W + q) * C, gv);
            tap_fma<T, V>(a0, gv, acc[0]); tap_fma<T, V>(a1, gv, acc[1]); tap_fma<T, V>(a2, gv, acc[2]);
            tap_fma<T, V>(b0, gv, acc[3]); tap_fma<T, V>(b1, gv, acc[4]); tap_fma<T, V>(b2, gv, acc[5]);
            tap_fma<T, V>(c0, gv, acc[6]); tap_fma<T, V>(c1, gv, acc[7]); tap_fma<T, V>(c2, gv, acc[8]);
            a0 = b0; a1 = b1; a2 = b2; b0 = c0; b1 = c1; b2 = c2;
        }
    }
    extern __shared__ float sm[];
    const int CB = cvb * V;
    const long slice = (long)blockIdx.x;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        if (pl < pix_par)
#pragma unroll
            for (int i = 0; i < V; ++i) sm[pl * CB + vl * V + i] = acc[k][i];
        __syncthreads();
        for (int o = threadIdx.x; o < CB; o += blockDim.x) {
            const int ch = blockIdx.y * CB + o;
            if (ch < C) {
                float tsum = 0.f;
                for (int l = 0; l < pix_par; ++l) tsum += sm[l * CB + o];
                part[(slice * 9 + k) * C + ch] = tsum;
            }
        }
        __syncthreads();
    }
}
// one wave per filter element: lane l adds partials l, l + 64, ... (double), then a fixed-order butterfly
__global__ __launch_bounds__(64) void dw3x3_wgrad_finish_kernel(const float* __restrict__ part, int nparts, int C, float* __restrict__ dw) {
    const int o = blockIdx.x;                                     // o = k * C + c in the partials; dw is [C][9]
    double s = 0.0;
    for (int b = threadIdx.x; b < nparts; b += 64) s += (double)part[(long)b * 9 * C + o];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if (threadIdx.x == 0) {
        const int k = o / C, c = o - k * C;
        dw[c * 9 + k] = (float)s;
    }
}

// channel_shuffle(cat(a, b), 2): full[row][c] = (c odd ? b : a)[row][c / 2]; halves: x1 = full[:, :Ch], x2 = full[:, Ch:].
// Ch = the real branch width, Chp >= Ch the width the branch tensors have in memory (channels past Ch are zero padding, kept zero here:
// shufflenet_v2 x1_0 / x2_0 have 58- / 122-channel branches held in 64 / 128); the whole tensor has Fp >= 2 Ch channels in memory.
template <typename T>
__global__ void shuffle_join_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ full, T* __restrict__ x1, T* __restrict__ x2, long rows, int Ch,
                                    int Chp, int Fp) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int wide = full ? Fp : 2 * Chp;
    if (e >= rows * wide) return;
    const long row = e / wide; const int c = (int)(e - row * wide);
    const T zero = (T)0.f;
    if (full) {
        full[e] = c < 2 * Ch ? ((c & 1) ? b[row * Chp + (c >> 1)] : a[row * Chp + (c >> 1)]) : zero;
        return;
    }
    const int half = c >= Chp, cc = c - half * Chp;          // channel cc of x1 / x2 = channel half * Ch + cc of the whole tensor
    const int g = half * Ch + cc;
    const T v = cc < Ch ? ((g & 1) ? b[row * Chp + (g >> 1)] : a[row * Chp + (g >> 1)]) : zero;
    (half ? x2 : x1)[row * Chp + cc] = v;
}
// its backward: da[row][i] = d full[row][2 i], db[row][i] = d full[row][2 i + 1] (d full given whole, or as its two halves)
template <typename T>
__global__ void shuffle_split_kernel(const T* __restrict__ dfull, const T* __restrict__ dx1, const T* __restrict__ dx2, T* __restrict__ da, T* __restrict__ db, long rows,
                                     int Ch, int Chp, int Fp) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * 2 * Chp) return;
    const long row = e / (2 * Chp); const int c = (int)(e - row * 2 * Chp);
    const int which = c >= Chp, i = c - which * Chp;          // which = 1: db
    T v = (T)0.f;
    if (i < Ch) {
        const int g = 2 * i + which;
        v = dfull ? dfull[row * Fp + g] : (g < Ch ? dx1[row * Chp + g] : dx2[row * Chp + g - Ch]);
    }
    (which ? db : da)[row * Chp + i] = v;
}

__global__ void cast_bf16_f32_kernel(const bf* __restrict__ src, float* __restrict__ dst, long n8) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n8) return;
    float v[8];
    load_vec<bf, 8>(src + e * 8, v);
    *reinterpret_cast<float4*>(dst + e * 8) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(dst + e * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

}  // namespace
}  // namespace sat

using namespace sat;

// launch shape of dw3x3_wgrad_s1_kernel: rows per thread R (<= 1: use the generic kernel), channel vectors per block, number of blocks along x
static int dw_wgrad_s1_blocks(int N, int H, int W, int C, int V, int& R, int& cvb) {
    const int cv = C / V;
    cvb = cv < 32 ? cv : 32;          // the largest divisor of cv up to 32 (no idle lanes; >= 8 columns in flight per block)
    while (cv % cvb) --cvb;
    R = dw_rows((long)N * H * W * cv * 4);          // the reduction wants long columns: 8 rows from 64 k vectors on
    if (R <= 1) return 0;
    const int nb = cdiv((long)N * cdiv(H, R) * W, (long)(256 / cvb));
    return nb < 1024 ? nb : 1024;          // more columns than that: the blocks take several (fewer partials for the finish to add)
}

extern "C" {

static int dw_check(const void* a, const void* b, const void* c, int N, int H, int W, int C, int stride, int dtype, const char* what) {
    if (!a || !b || !c) return fail(SAT_EINVAL, "%s: null pointer", what);
    const int V = dtype ? 8 : 4;
    SAT_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % V == 0 && (stride == 1 || stride == 2), "%s: bad shape (N=%d H=%d W=%d C=%d stride=%d; C %% %d == 0)", what, N, H, W, C,
                stride, V);
    return SAT_OK;
}

int sat_dwconv3x3_fwd_t(int32_t dtype, const void* x, const float* w, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t stride, void* stream) {
    SAT_TRY(dw_check(x, w, y, N, H, W, C, stride, dtype, "dwconv3x3_fwd"));
    const int P = (H + 2 - 3) / stride + 1, Q = (W + 2 - 3) / stride + 1;
    const int cv = C / (dtype ? 8 : 4), R = dw_rows((long)N * P * Q * cv);
    const long total = (long)N * cdiv(P, R) * Q * cv;
    if (stride == 1 && R > 1) {          // (one row per thread: nothing to roll, the plain kernel is 2 us faster on the small maps)
        if (dtype) hipLaunchKernelGGL((dw3x3_s1_kernel<bf, false>), dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf*)x, w, (bf*)y, N, H, W, C, R);
        else hipLaunchKernelGGL((dw3x3_s1_kernel<float, false>), dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)x, w, (float*)y, N, H, W, C, R);
        return launch_ok("dwconv3x3_fwd (stride 1)");
    }
    if (dtype) hipLaunchKernelGGL(dw3x3_fwd_kernel<bf>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf*)x, w, (bf*)y, N, H, W, C, P, Q, stride, R);
    else hipLaunchKernelGGL(dw3x3_fwd_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)x, w, (float*)y, N, H, W, C, P, Q, stride, R);
    return launch_ok("dwconv3x3_fwd");
}

int sat_dwconv3x3_dgrad_t(int32_t dtype, const void* dy, const float* w, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t stride, void* stream) {
    SAT_TRY(dw_check(dy, w, dx, N, H, W, C, stride, dtype, "dwconv3x3_dgrad"));
    const int P = (H + 2 - 3) / stride + 1, Q = (W + 2 - 3) / stride + 1;
    const int cv = C / (dtype ? 8 : 4), R = dw_rows((long)N * H * W * cv);
    const long total = (long)N * cdiv(H, R) * W * cv;
    if (stride == 1 && R > 1) {
        if (dtype) hipLaunchKernelGGL((dw3x3_s1_kernel<bf, true>), dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf*)dy, w, (bf*)dx, N, H, W, C, R);
        else hipLaunchKernelGGL((dw3x3_s1_kernel<float, true>), dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)dy, w, (float*)dx, N, H, W, C, R);
        return launch_ok("dwconv3x3_dgrad (stride 1)");
    }
    if (dtype) hipLaunchKernelGGL(dw3x3_dgrad_kernel<bf>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf*)dy, w, (bf*)dx, N, H, W, C, P, Q, stride, R);
    else hipLaunchKernelGGL(dw3x3_dgrad_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)dy, w, (float*)dx, N, H, W, C, P, Q, stride, R);
    return launch_ok("dwconv3x3_dgrad");
}

size_t sat_dwconv3x3_wgrad_scratch_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t stride) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (stride != 1 && stride != 2)) return 0;
    const int P = (H + 2 - 3) / stride + 1, Q = (W + 2 - 3) / stride + 1;
    size_t parts = (size_t)cdiv((long)N * P * Q, (long)dw_chunk((long)N * P * Q));
    if (stride == 1)          // the rolling form writes one partial per block; whichever storage type the call will have
        for (int V = 4; V <= 8; V += 4) {
            int R, cvb;
            const size_t nb = C % V ? 0 : (size_t)dw_wgrad_s1_blocks(N, H, W, C, V, R, cvb);
            if (nb > parts) parts = nb;
        }
    return parts * 9 * C * sizeof(float);
}

int sat_dwconv3x3_wgrad_t(int32_t dtype, const void* dy, const void* x, float* dw, int32_t N, int32_t H, int32_t W, int32_t C, int32_t stride, float* scratch,
                          void* stream) {
    SAT_TRY(dw_check(dy, x, dw, N, H, W, C, stride, dtype, "dwconv3x3_wgrad"));
    if (!scratch) return fail(SAT_EINVAL, "dwconv3x3_wgrad: null scratch");
    const int V = dtype ? 8 : 4, cv = C / V;
    const int P = (H + 2 - 3) / stride + 1, Q = (W + 2 - 3) / stride + 1;
    const int chunk = dw_chunk((long)N * P * Q);
    const int nparts = cdiv((long)N * P * Q, (long)chunk);
    int R, cvb;
    int nblk = dw_wgrad_s1_blocks(N, H, W, C, V, R, cvb);          // (also sets cvb for the generic form)
    if (stride != 1) nblk = 0;
    const int pix_par = 256 / cvb;
    const size_t lds = (size_t)pix_par * cvb * V * sizeof(float);
    if (nblk > 0) {          // stride 1, long enough columns: the rolling-window form
        const dim3 g1(nblk, cdiv(cv, cvb));
        if (dtype) hipLaunchKernelGGL(dw3x3_wgrad_s1_kernel<bf>, g1, dim3(256), lds, (hipStream_t)stream, (const bf*)dy, (const bf*)x, scratch, N, H, W, C, R, cvb);
        else hipLaunchKernelGGL(dw3x3_wgrad_s1_kernel<float>, g1, dim3(256), lds, (hipStream_t)stream, (const float*)dy, (const float*)x, scratch, N, H, W, C, R, cvb);
        SAT_TRY(launch_ok("dwconv3x3_wgrad (stride 1, partials)"));
        hipLaunchKernelGGL(dw3x3_wgrad_finish_kernel, dim3(9 * C), dim3(64), 0, (hipStream_t)stream, scratch, nblk, C, dw);
        return launch_ok("dwconv3x3_wgrad (finish)");
    }
    const dim3 grid(nparts, cdiv(cv, cvb));
    if (dtype) hipLaunchKernelGGL(dw3x3_wgrad_part_kernel<bf>, grid, dim3(256), lds, (hipStream_t)stream, (const bf*)dy, (const bf*)x, scratch, N, H, W, C, P, Q, stride, chunk, cvb);
    else hipLaunchKernelGGL(dw3x3_wgrad_part_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, (const float*)dy, (const float*)x, scratch, N, H, W, C, P, Q, stride, chunk, cvb);
    SAT_TRY(launch_ok("dwconv3x3_wgrad (partials)"));
    hipLaunchKernelGGL(dw3x3_wgrad_finish_kernel, dim3(9 * C), dim3(64), 0, (hipStream_t)stream, scratch, nparts, C, dw);
    return launch_ok("dwconv3x3_wgrad (finish)");
}

static int shuffle_check(int64_t rows, int32_t Ch, int32_t Chp, const char* what) {
    SAT_REQUIRE(rows > 0 && Ch > 0 && Chp >= Ch && Chp % 4 == 0, "%s: rows=%ld Ch=%d Chp=%d (Chp >= Ch, a multiple of 4)", what, (long)rows, Ch, Chp);
    return SAT_OK;
}
static inline int full_width(int Ch) { return (2 * Ch + 7) / 8 * 8; }

int sat_shuffle_join_t(int32_t dtype, const void* a, const void* b, void* full, void* x1, void* x2, int64_t rows, int32_t Ch, int32_t Chp, void* stream) {
    if (!a || !b || (!full && !(x1 && x2))) return fail(SAT_EINVAL, "shuffle_join: null pointer");
    SAT_TRY(shuffle_check(rows, Ch, Chp, "shuffle_join"));
    const int Fp = full_width(Ch);
    const long total = rows * (full ? Fp : 2 * Chp);
    if (dtype) hipLaunchKernelGGL(shuffle_join_kernel<bf>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf*)a, (const bf*)b, (bf*)full, (bf*)x1, (bf*)x2, (long)rows, Ch, Chp, Fp);
    else hipLaunchKernelGGL(shuffle_join_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)a, (const float*)b, (float*)full, (float*)x1, (float*)x2, (long)rows, Ch, Chp, Fp);
    return launch_ok("shuffle_join");
}

int sat_shuffle_split_t(int32_t dtype, const void* dfull, const void* dx1, const void* dx2, void* da, void* db, int64_t rows, int32_t Ch, int32_t Chp, void* stream) {
    if ((!dfull && !(dx1 && dx2)) || !da || !db) return fail(SAT_EINVAL, "shuffle_split: null pointer");
    SAT_TRY(shuffle_check(rows, Ch, Chp, "shuffle_split"));
    const int Fp = full_width(Ch);
    const long total = rows * 2 * Chp;
    if (dtype) hipLaunchKernelGGL(shuffle_split_kernel<bf>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf*)dfull, (const bf*)dx1, (const bf*)dx2, (bf*)da, (bf*)db, (long)rows, Ch, Chp, Fp);
    else hipLaunchKernelGGL(shuffle_split_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)dfull, (const float*)dx1, (const float*)dx2, (float*)da, (float*)db, (long)rows, Ch, Chp, Fp);
    return launch_ok("shuffle_split");
}

int sat_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream) {
    if (!src || !dst) return fail(SAT_EINVAL, "cast: null pointer");
    SAT_REQUIRE(n > 0 && n % 8 == 0, "cast: element count must be a positive multiple of 8");
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(cdiv(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf*)src, dst, (long)(n / 8));
    return launch_ok("cast_bf16_f32");
}

}
