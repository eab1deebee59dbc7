// Input pipeline on device: a ragged batch of decoded uint8 RGB pictures -> the (B, 3, S, S) fp32 batch the
// reference's DataLoader hands to SAT.train_batch.
// Reference: train.py:208-233 composes, per picture, T.RandomResizedCrop | T.Resize + T.CenterCrop, T.RandomHorizontalFlip,
// T.ToTensor and util.py:121-130 AddGaussianNoise.  On PIL images torchvision's resize is Pillow's Image.resize(BILINEAR): a
// separable, antialiased (support scaled by the shrink factor) triangle filter evaluated in 8-bit fixed point, horizontal pass
// first, each pass rounded and clipped to bytes.  The kernels below compute exactly those bytes (22-bit coefficients from
// double-precision weights, the same rounding), then byte/255 in fp32 and + noise * std.
// Byte work, HBM/L2 bound: one read of the source box, a 4-byte-per-pixel intermediate, one fp32 write.
#include "../../include/sat_hip.h"
#include "common.h"

// Every floating-point operation in this file must round on its own, as the host code it reproduces does: the Makefile
// compiles it with -ffp-contract=off (hipcc's default lets the backend fuse a product with a following sum whatever the
// source says - hip's __fmul_rn / __fadd_rn are plain inline operators and fuse too).  Plain operators below.
#pragma clang fp contract(off)

namespace sat {

constexpr int RS_BITS = 32 - 8 - 2;          // Pillow's PRECISION_BITS for 8-bit pixels
constexpr int RS_MAX_TAPS = 129;             // shrink factors up to 64

__host__ __device__ inline int taps_for(int in_size, int out_size) {
    double fs = (double)in_size / (double)out_size;
    if (fs < 1.0) fs = 1.0;
    return (int)ceil(fs) * 2 + 1;
}

// coefficient tables of one axis of one picture: output index o of the window -> first tap, tap count, integer weights.
// Every double operation is a single correctly rounded IEEE operation in the order Pillow's C evaluates it.
__global__ __launch_bounds__(64) void resample_coeffs_kernel(const sat_image_desc* __restrict__ desc, int out_h, int out_w, int omax, int KT,
                                                             int* __restrict__ bounds, int* __restrict__ coeffs) {
    const int img = blockIdx.z, axis = blockIdx.y;
    const int o = blockIdx.x * 64 + threadIdx.x;
    const sat_image_desc d = desc[img];
    const int n_out = axis ? out_h : out_w;
    if (o >= n_out) return;
    const int in_size = axis ? d.crop_h : d.crop_w;
    const int rs_size = axis ? d.resized_h : d.resized_w;
    const int xx = o + (axis ? d.out_top : d.out_left);
    const double scale = __ddiv_rn((double)in_size, (double)rs_size);
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = filterscale;                 // bilinear support 1.0 * filterscale
    const double ss = __ddiv_rn(1.0, filterscale);
    const double center = __dadd_rn(0.0, __dmul_rn(__dadd_rn((double)xx, 0.5), scale));
    int xmin = (int)__dadd_rn(__dsub_rn(center, support), 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)__dadd_rn(__dadd_rn(center, support), 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    if (xmax > KT) xmax = KT;                          // cannot happen (KT from the same rule); keeps the table in bounds
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
        double a = __dmul_rn(__dadd_rn(__dsub_rn((double)(x + xmin), center), 0.5), ss);
        a = a < 0.0 ? -a : a;
        ww = __dadd_rn(ww, a < 1.0 ? __dsub_rn(1.0, a) : 0.0);
    }
    int* k = coeffs + (((long)img * 2 + axis) * omax + o) * KT;
    for (int x = 0; x < xmax; ++x) {
        double a = __dmul_rn(__dadd_rn(__dsub_rn((double)(x + xmin), center), 0.5), ss);
        a = a < 0.0 ? -a : a;
        double w = a < 1.0 ? __dsub_rn(1.0, a) : 0.0;
        if (ww != 0.0) w = __ddiv_rn(w, ww);
        k[x] = (int)__dadd_rn(0.5, __dmul_rn(w, (double)(1 << RS_BITS)));
    }
    for (int x = xmax; x < KT; ++x) k[x] = 0;
    int* b = bounds + (((long)img * 2 + axis) * omax + o) * 2;
    b[0] = xmin; b[1] = xmax;
}

__device__ inline int clip8(int v) { v >>= RS_BITS; return v < 0 ? 0 : (v > 255 ? 255 : v); }

// horizontal pass over every row of the crop box: tmp[img][row][x] = RGBX bytes
__global__ __launch_bounds__(256) void resample_rows_kernel(const uint8_t* __restrict__ pixels, const sat_image_desc* __restrict__ desc, int out_w, int omax,
                                                            int KT, int hmax, const int* __restrict__ bounds, const int* __restrict__ coeffs,
                                                            uchar4* __restrict__ tmp) {
    const int img = blockIdx.z;
    const sat_image_desc d = desc[img];
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= out_w || row >= d.crop_h) return;
    const int* b = bounds + (((long)img * 2 + 0) * omax + x) * 2;
    const int* k = coeffs + (((long)img * 2 + 0) * omax + x) * KT;
    const int xmin = b[0], n = b[1];
    const uint8_t* src = pixels + d.offset + ((long)(d.crop_top + row) * d.width + d.crop_left + xmin) * 3;
    int s0 = 1 << (RS_BITS - 1), s1 = s0, s2 = s0;
    for (int t = 0; t < n; ++t) {
        const int kv = k[t];
        s0 += src[t * 3 + 0] * kv; s1 += src[t * 3 + 1] * kv; s2 += src[t * 3 + 2] * kv;
    }
    tmp[((long)img * hmax + row) * out_w + x] = make_uchar4((unsigned char)clip8(s0), (unsigned char)clip8(s1), (unsigned char)clip8(s2), 0);
}

// vertical pass + flip + ToTensor + noise; one thread per output pixel, the three colour planes written coalesced along x
__global__ __launch_bounds__(256) void resample_cols_finish_kernel(const sat_image_desc* __restrict__ desc, int out_h, int out_w, int omax, int KT, int hmax,
                                                                   const int* __restrict__ bounds, const int* __restrict__ coeffs,
                                                                   const uchar4* __restrict__ tmp, const float* __restrict__ noise, float noise_std,
                                                                   float* __restrict__ out, uint8_t* __restrict__ out_u8) {
    const int img = blockIdx.z;
    const sat_image_desc d = desc[img];
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= out_w || y >= out_h) return;
    const int* b = bounds + (((long)img * 2 + 1) * omax + y) * 2;
    const int* k = coeffs + (((long)img * 2 + 1) * omax + y) * KT;
    const int ymin = b[0], n = b[1];
    const uchar4* src = tmp + ((long)img * hmax + ymin) * out_w + x;
    int s0 = 1 << (RS_BITS - 1), s1 = s0, s2 = s0;
    for (int t = 0; t < n; ++t) {
        const int kv = k[t];
        const uchar4 p = src[(long)t * out_w];
        s0 += p.x * kv; s1 += p.y * kv; s2 += p.z * kv;
    }
    const int v[3] = {clip8(s0), clip8(s1), clip8(s2)};
    const int xo = d.flip ? out_w - 1 - x : x;
    const long plane = (long)out_h * out_w;
    const long o = (long)img * 3 * plane + (long)y * out_w + xo;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (out) {
            float f = __fdiv_rn((float)v[c], 255.0f);
            if (noise) f = __fadd_rn(f, __fmul_rn(noise[o + c * plane], noise_std));
            out[o + c * plane] = f;
        }
        if (out_u8) out_u8[(((long)img * out_h + y) * out_w + xo) * 3 + c] = (uint8_t)v[c];
    }
}

struct ImagePlan { int KT, hmax, omax; size_t bounds_off, coeffs_off, tmp_off, total; };

static int image_plan(const sat_image_desc* d, int n, int64_t pixels_bytes, int out_h, int out_w, ImagePlan* p) {
    SAT_REQUIRE(n > 0 && out_h > 0 && out_w > 0, "image_batch: n=%d out=%dx%d", n, out_h, out_w);
    int KT = 3, hmax = 1;
    for (int i = 0; i < n; ++i) {
        const sat_image_desc& e = d[i];
        SAT_REQUIRE(e.height > 0 && e.width > 0 && e.offset >= 0 && (pixels_bytes < 0 || e.offset + (int64_t)e.height * e.width * 3 <= pixels_bytes),
                    "image_batch: picture %d (%dx%d at byte %lld) lies outside the pixel buffer", i, e.height, e.width, (long long)e.offset);
        SAT_REQUIRE(e.crop_h > 0 && e.crop_w > 0 && e.crop_top >= 0 && e.crop_left >= 0 && e.crop_top + e.crop_h <= e.height && e.crop_left + e.crop_w <= e.width,
                    "image_batch: picture %d crop box (%d,%d,%d,%d) outside %dx%d", i, e.crop_top, e.crop_left, e.crop_h, e.crop_w, e.height, e.width);
        SAT_REQUIRE(e.resized_h > 0 && e.resized_w > 0 && e.out_top >= 0 && e.out_left >= 0 && e.out_top + out_h <= e.resized_h && e.out_left + out_w <= e.resized_w,
                    "image_batch: picture %d output window (%d,%d)+%dx%d outside the resampled %dx%d", i, e.out_top, e.out_left, out_h, out_w, e.resized_h, e.resized_w);
        const int k = taps_for(e.crop_h, e.resized_h) > taps_for(e.crop_w, e.resized_w) ? taps_for(e.crop_h, e.resized_h) : taps_for(e.crop_w, e.resized_w);
        SAT_REQUIRE(k <= RS_MAX_TAPS, "image_batch: picture %d shrinks by more than 64x", i);
        if (k > KT) KT = k;
        if (e.crop_h > hmax) hmax = e.crop_h;
    }
    p->KT = KT; p->hmax = hmax; p->omax = out_h > out_w ? out_h : out_w;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
    p->bounds_off = take((size_t)n * 2 * p->omax * 2 * sizeof(int));
    p->coeffs_off = take((size_t)n * 2 * p->omax * KT * sizeof(int));
    p->tmp_off = take((size_t)n * hmax * out_w * sizeof(uchar4));
    p->total = off;
    return SAT_OK;
}

}  // namespace sat
using namespace sat;

extern "C" {

size_t sat_image_batch_workspace_bytes(const sat_image_desc* desc_host, int32_t n, int32_t out_h, int32_t out_w) {
    if (!desc_host) { fail(SAT_EINVAL, "image_batch: null descriptors"); return 0; }
    ImagePlan p;
    if (image_plan(desc_host, n, -1, out_h, out_w, &p) != SAT_OK) return 0;
    return p.total;
}

int sat_image_batch_transform(const uint8_t* pixels, int64_t pixels_bytes, const sat_image_desc* desc_host, const sat_image_desc* desc_dev, int32_t n,
                              int32_t out_h, int32_t out_w, const float* noise, float noise_std, float* out_nchw, uint8_t* out_u8,
                              void* workspace, size_t workspace_bytes, void* stream) {
    if (!pixels || !desc_host || !desc_dev || !workspace || (!out_nchw && !out_u8)) return fail(SAT_EINVAL, "image_batch: null pointer");
    ImagePlan p;
    SAT_TRY(image_plan(desc_host, n, pixels_bytes, out_h, out_w, &p));
    SAT_REQUIRE(workspace_bytes >= p.total, "image_batch: workspace %zu < %zu bytes", workspace_bytes, p.total);
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    int* bounds = (int*)(ws + p.bounds_off);
    int* coeffs = (int*)(ws + p.coeffs_off);
    uchar4* tmp = (uchar4*)(ws + p.tmp_off);
    hipLaunchKernelGGL(resample_coeffs_kernel, dim3((p.omax + 63) / 64, 2, n), dim3(64), 0, st, desc_dev, out_h, out_w, p.omax, p.KT, bounds, coeffs);
    SAT_TRY(launch_ok("resample_coeffs"));
    hipLaunchKernelGGL(resample_rows_kernel, dim3((out_w + 63) / 64, (p.hmax + 3) / 4, n), dim3(256), 0, st, pixels, desc_dev, out_w, p.omax, p.KT, p.hmax,
                       bounds, coeffs, tmp);
    SAT_TRY(launch_ok("resample_rows"));
    hipLaunchKernelGGL(resample_cols_finish_kernel, dim3((out_w + 63) / 64, (out_h + 3) / 4, n), dim3(256), 0, st, desc_dev, out_h, out_w, p.omax, p.KT, p.hmax,
                       bounds, coeffs, tmp, noise, noise_std, out_nchw, out_u8);
    return launch_ok("resample_cols_finish");
}

}
