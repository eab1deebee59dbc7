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

// w: the (C, 1, 3, 3) parameter in its own memory order = [C][9] fp32 (master weights; the filter is tiny: read through the caches)
template <typename T>
__global__ __launch_bounds__(256) void dw3x3_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y, int N, int H, int W, int C,
                                                        int P, int Q, int stride) {
    constexpr int V = Vec<T>::V;
    const int cv = C / V;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)N * P * Q * cv;
    if (e >= total) return;
    const int c = (int)(e % cv) * V; long t = e / cv;
    const int q = (int)(t % Q); t /= Q; const int p = (int)(t % P); const int n = (int)(t / P);
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
            for (int i = 0; i < V; ++i) acc[i] = fmaf(xv[i], w[(c + i) * 9 + r * 3 + s], acc[i]);
        }
    }
    store_vec<T, V>(y + e * V, acc);
}

// dx[n, h, w, c] = sum over taps (r, s) with (h + 1 - r) = p * stride, (w + 1 - s) = q * stride of dy[n, p, q, c] * w[c, r, s]
template <typename T>
__global__ __launch_bounds__(256) void dw3x3_dgrad_kernel(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx, int N, int H, int W, int C,
                                                          int P, int Q, int stride) {
    constexpr int V = Vec<T>::V;
    const int cv = C / V;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)N * H * W * cv;
    if (e >= total) return;
    const int c = (int)(e % cv) * V; long t = e / cv;
    const int ww = (int)(t % W); t /= W; const int h = (int)(t % H); const int n = (int)(t / H);
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
            for (int i = 0; i < V; ++i) acc[i] = fmaf(gv[i], w[(c + i) * 9 + r * 3 + s], acc[i]);
        }
    }
    store_vec<T, V>(dx + e * V, acc);
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
                                                               int P, int Q, int stride, int chunk) {
    constexpr int V = Vec<T>::V;
    const int cv = C / V;
    // thread: channel vector tid % cv (all threads of a block walk different pixels of the chunk for their vector); blockDim is a multiple of cv's divisor
    const int lanes_per_pix = cv;                      // threads needed for one pixel
    const int pix_par = blockDim.x / lanes_per_pix;    // pixels in flight per block
    const int vi = threadIdx.x % lanes_per_pix, pl = threadIdx.x / lanes_per_pix;
    const int c = vi * V;
    float acc[9][V];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int i = 0; i < V; ++i) acc[k][i] = 0.f;
    const long npix = (long)N * P * Q;
    const long p0 = (long)blockIdx.x * chunk, p1 = p0 + chunk < npix ? p0 + chunk : npix;
    if (pl < pix_par) {
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
    // combine the block's pixel lanes in a fixed order through LDS, one tap at a time: [pix lane][C]
    extern __shared__ float sm[];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        if (pl < pix_par)
#pragma unroll
            for (int i = 0; i < V; ++i) sm[pl * C + c + i] = acc[k][i];
        __syncthreads();
        for (int o = threadIdx.x; o < C; o += blockDim.x) {
            float t = 0.f;
            for (int l = 0; l < pix_par; ++l) t += sm[l * C + o];
            part[((long)blockIdx.x * 9 + k) * C + o] = t;
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

// channel_shuffle(cat(a, b), 2): full[row][c] = (c odd ? b : a)[row][c / 2]; halves: x1 = full[:, :Ch], x2 = full[:, Ch:]
template <typename T>
__global__ void shuffle_join_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ full, T* __restrict__ x1, T* __restrict__ x2, long rows, int Ch) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * 2 * Ch) return;
    const long row = e / (2 * Ch); const int c = (int)(e - row * 2 * Ch);
    const T v = (c & 1) ? b[row * Ch + (c >> 1)] : a[row * Ch + (c >> 1)];
    if (full) full[e] = v;
    else if (c < Ch) x1[row * Ch + c] = v;
    else x2[row * Ch + c - Ch] = v;
}
// its backward: da[row][i] = d full[row][2 i], db[row][i] = d full[row][2 i + 1] (d full given whole, or as its two halves)
template <typename T>
__global__ void shuffle_split_kernel(const T* __restrict__ dfull, const T* __restrict__ dx1, const T* __restrict__ dx2, T* __restrict__ da, T* __restrict__ db, long rows,
                                     int Ch) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * 2 * Ch) return;
    const long row = e / (2 * Ch); const int c = (int)(e - row * 2 * Ch);
    const T v = dfull ? dfull[e] : (c < Ch ? dx1[row * Ch + c] : dx2[row * Ch + c - Ch]);
    if (c & 1) db[row * Ch + (c >> 1)] = v; else da[row * Ch + (c >> 1)] = v;
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
    const long total = (long)N * P * Q * (C / (dtype ? 8 : 4));
    if (dtype) hipLaunchKernelGGL(dw3x3_fwd_kernel<bf>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf*)x, w, (bf*)y, N, H, W, C, P, Q, stride);
    else hipLaunchKernelGGL(dw3x3_fwd_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)x, w, (float*)y, N, H, W, C, P, Q, stride);
    return launch_ok("dwconv3x3_fwd");
}

int sat_dwconv3x3_dgrad_t(int32_t dtype, const void* dy, const float* w, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t stride, void* stream) {
    SAT_TRY(dw_check(dy, w, dx, N, H, W, C, stride, dtype, "dwconv3x3_dgrad"));
    const int P = (H + 2 - 3) / stride + 1, Q = (W + 2 - 3) / stride + 1;
    const long total = (long)N * H * W * (C / (dtype ? 8 : 4));
    if (dtype) hipLaunchKernelGGL(dw3x3_dgrad_kernel<bf>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf*)dy, w, (bf*)dx, N, H, W, C, P, Q, stride);
    else hipLaunchKernelGGL(dw3x3_dgrad_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)dy, w, (float*)dx, N, H, W, C, P, Q, stride);
    return launch_ok("dwconv3x3_dgrad");
}

size_t sat_dwconv3x3_wgrad_scratch_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t stride) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (stride != 1 && stride != 2)) return 0;
    const int P = (H + 2 - 3) / stride + 1, Q = (W + 2 - 3) / stride + 1;
    return (size_t)cdiv((long)N * P * Q, (long)dw_chunk((long)N * P * Q)) * 9 * C * sizeof(float);
}

int sat_dwconv3x3_wgrad_t(int32_t dtype, const void* dy, const void* x, float* dw, int32_t N, int32_t H, int32_t W, int32_t C, int32_t stride, float* scratch,
                          void* stream) {
    SAT_TRY(dw_check(dy, x, dw, N, H, W, C, stride, dtype, "dwconv3x3_wgrad"));
    if (!scratch) return fail(SAT_EINVAL, "dwconv3x3_wgrad: null scratch");
    const int V = dtype ? 8 : 4, cv = C / V;
    SAT_REQUIRE(cv <= 256, "dwconv3x3_wgrad: C=%d channels exceed %d", C, 256 * V);
    const int P = (H + 2 - 3) / stride + 1, Q = (W + 2 - 3) / stride + 1;
    const int chunk = dw_chunk((long)N * P * Q);
    const int nparts = cdiv((long)N * P * Q, (long)chunk);
    const int pix_par = 256 / cv;
    const size_t lds = (size_t)pix_par * C * sizeof(float);
    SAT_REQUIRE(lds <= 64 * 1024, "dwconv3x3_wgrad: C=%d needs %zu bytes of LDS", C, lds);
    if (dtype) hipLaunchKernelGGL(dw3x3_wgrad_part_kernel<bf>, dim3(nparts), dim3(256), lds, (hipStream_t)stream, (const bf*)dy, (const bf*)x, scratch, N, H, W, C, P, Q, stride, chunk);
    else hipLaunchKernelGGL(dw3x3_wgrad_part_kernel<float>, dim3(nparts), dim3(256), lds, (hipStream_t)stream, (const float*)dy, (const float*)x, scratch, N, H, W, C, P, Q, stride, chunk);
    SAT_TRY(launch_ok("dwconv3x3_wgrad (partials)"));
    hipLaunchKernelGGL(dw3x3_wgrad_finish_kernel, dim3(9 * C), dim3(64), 0, (hipStream_t)stream, scratch, nparts, C, dw);
    return launch_ok("dwconv3x3_wgrad (finish)");
}

int sat_shuffle_join_t(int32_t dtype, const void* a, const void* b, void* full, void* x1, void* x2, int64_t rows, int32_t Ch, void* stream) {
    if (!a || !b || (!full && !(x1 && x2)) || rows <= 0 || Ch <= 0) return fail(SAT_EINVAL, "shuffle_join: bad argument");
    const long total = rows * 2 * Ch;
    if (dtype) hipLaunchKernelGGL(shuffle_join_kernel<bf>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf*)a, (const bf*)b, (bf*)full, (bf*)x1, (bf*)x2, (long)rows, Ch);
    else hipLaunchKernelGGL(shuffle_join_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)a, (const float*)b, (float*)full, (float*)x1, (float*)x2, (long)rows, Ch);
    return launch_ok("shuffle_join");
}

int sat_shuffle_split_t(int32_t dtype, const void* dfull, const void* dx1, const void* dx2, void* da, void* db, int64_t rows, int32_t Ch, void* stream) {
    if ((!dfull && !(dx1 && dx2)) || !da || !db || rows <= 0 || Ch <= 0) return fail(SAT_EINVAL, "shuffle_split: bad argument");
    const long total = rows * 2 * Ch;
    if (dtype) hipLaunchKernelGGL(shuffle_split_kernel<bf>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf*)dfull, (const bf*)dx1, (const bf*)dx2, (bf*)da, (bf*)db, (long)rows, Ch);
    else hipLaunchKernelGGL(shuffle_split_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)dfull, (const float*)dx1, (const float*)dx2, (float*)da, (float*)db, (long)rows, Ch);
    return launch_ok("shuffle_split");
}

int sat_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream) {
    if (!src || !dst) return fail(SAT_EINVAL, "cast: null pointer");
    SAT_REQUIRE(n > 0 && n % 8 == 0, "cast: element count must be a positive multiple of 8");
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(cdiv(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf*)src, dst, (long)(n / 8));
    return launch_ok("cast_bf16_f32");
}

}
