// Multi-tensor optimizer step on device: every parameter tensor of the model in ONE launch.
// Reference: SAT.configure_optimizers (model.py:723-757) builds torch.optim.SGD / Adam / AdamW over per-module
// parameter groups (lr, weight_decay per group); Lightning clips gradients by value or by global norm before the
// step (train.py:93-96, 273-274).  The arithmetic below is torch's single-tensor formulation of those optimizers
// (the reference's defaults: amsgrad off, maximize off, dampening 0), applied per element.
#include "../../include/sat_hip.h"
#include "common.h"

namespace sat {

constexpr int OPT_CHUNK = 1 << 16;      // elements per workgroup

// sum of squares of every gradient, fixed order: one partial per chunk, then one block over the partials
__global__ __launch_bounds__(256) void grad_sqsum_part_kernel(const sat_opt_tensor* __restrict__ tensors, const sat_opt_chunk* __restrict__ chunks,
                                                              double* __restrict__ part) {
    const sat_opt_chunk c = chunks[blockIdx.x];
    const sat_opt_tensor t = tensors[c.tensor];
    const long end = (c.start + OPT_CHUNK < t.n) ? c.start + OPT_CHUNK : t.n;
    double s = 0.0;
    for (long i = c.start + threadIdx.x; i < end; i += 256) { const float g = t.g[i]; s += (double)g * g; }
    __shared__ double sh[256];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}
// coef[0] = min(1, max_norm / (norm + 1e-6))  (torch.nn.utils.clip_grad_norm_), coef[1] = norm
__global__ __launch_bounds__(256) void grad_norm_finish_kernel(const double* __restrict__ part, int n, float max_norm, float* __restrict__ coef) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(sh[0]);
        const float c = max_norm / (norm + 1e-6f);
        coef[0] = c < 1.f ? c : 1.f; coef[1] = norm;
    }
}

__device__ __forceinline__ void opt_update(const sat_opt_hyper& h, float lr, float wd, float coef, float& p, float g, float& m, float& v) {
    g *= coef;
    if (h.clip_value > 0.f) g = fminf(fmaxf(g, -h.clip_value), h.clip_value);        // clip_grad_value_
    if (h.kind == SAT_OPT_SGD) {
        if (wd != 0.f) g = fmaf(wd, p, g);
        if (h.momentum != 0.f) {
            m = h.first_step ? g : fmaf(h.momentum, m, g);                            // torch: buf = grad on the first step
            g = h.nesterov ? fmaf(h.momentum, m, g) : m;
        }
        p -= lr * g;
    } else {
        if (h.kind == SAT_OPT_ADAMW) p *= 1.f - lr * wd;                              // decoupled decay
        else if (wd != 0.f) g = fmaf(wd, p, g);                                       // Adam: L2 term joins the gradient
        m = m + (g - m) * (1.f - h.beta1);                                            // exp_avg.lerp_(grad, 1 - beta1)
        v = h.beta2 * v + (1.f - h.beta2) * g * g;
        const float denom = sqrtf(v) / h.bias_correction2_sqrt + h.eps;
        p -= (lr / h.bias_correction1) * (m / denom);
    }
}

__global__ __launch_bounds__(256) void optimizer_step_kernel(const sat_opt_tensor* __restrict__ tensors, const sat_opt_chunk* __restrict__ chunks,
                                                             sat_opt_hyper h, const sat_opt_hyper* __restrict__ h_dev, const float* __restrict__ clip_coef) {
    if (h_dev) h = *h_dev;          // hyper-parameters read from device memory (sat_optimizer_step_dev: the launch is replayed from a graph, the step count moves on)
    const sat_opt_chunk c = chunks[blockIdx.x];
    const sat_opt_tensor t = tensors[c.tensor];
    const long end = (c.start + OPT_CHUNK < t.n) ? c.start + OPT_CHUNK : t.n;
    const float coef = clip_coef ? clip_coef[0] : 1.f;
    const float lr = t.lr, wd = t.weight_decay;
    const bool has_m = t.m != nullptr, has_v = t.v != nullptr;
    // 16 bytes per lane where the tensor allows it (chunk starts are multiples of 64 Ki elements; bases come 16-byte aligned
    // from the allocator, checked here), scalar tail
    const bool vec = ((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.g) | reinterpret_cast<uintptr_t>(t.m) |
                       reinterpret_cast<uintptr_t>(t.v)) & 15) == 0;
    long i = c.start;
    if (vec) {
        const long nv = (end - c.start) / 4;
        for (long q = threadIdx.x; q < nv; q += 256) {
            const long e = c.start + q * 4;
            float4 p4 = *reinterpret_cast<const float4*>(t.p + e);
            const float4 g4 = *reinterpret_cast<const float4*>(t.g + e);
            float4 m4 = has_m ? *reinterpret_cast<const float4*>(t.m + e) : make_float4(0, 0, 0, 0);
            float4 v4 = has_v ? *reinterpret_cast<const float4*>(t.v + e) : make_float4(0, 0, 0, 0);
            opt_update(h, lr, wd, coef, p4.x, g4.x, m4.x, v4.x); opt_update(h, lr, wd, coef, p4.y, g4.y, m4.y, v4.y);
            opt_update(h, lr, wd, coef, p4.z, g4.z, m4.z, v4.z); opt_update(h, lr, wd, coef, p4.w, g4.w, m4.w, v4.w);
            *reinterpret_cast<float4*>(t.p + e) = p4;
            if (t.shadow_bf16) {
                typedef __bf16 b4 __attribute__((ext_vector_type(4)));
                b4 o; o[0] = (__bf16)p4.x; o[1] = (__bf16)p4.y; o[2] = (__bf16)p4.z; o[3] = (__bf16)p4.w;
                *reinterpret_cast<b4*>(reinterpret_cast<__bf16*>(t.shadow_bf16) + e) = o;
            }
            if (has_m) *reinterpret_cast<float4*>(t.m + e) = m4;
            if (has_v) *reinterpret_cast<float4*>(t.v + e) = v4;
        }
        i = c.start + nv * 4;
    }
    for (i += threadIdx.x; i < end; i += 256) {
        float p = t.p[i], m = has_m ? t.m[i] : 0.f, v = has_v ? t.v[i] : 0.f;
        opt_update(h, lr, wd, coef, p, t.g[i], m, v);
        t.p[i] = p;
        if (t.shadow_bf16) reinterpret_cast<__bf16*>(t.shadow_bf16)[i] = (__bf16)p;
        if (has_m) t.m[i] = m;
        if (has_v) t.v[i] = v;
    }
}

}  // namespace sat

using namespace sat;

extern "C" {

int32_t sat_optimizer_chunk_elems(void) { return OPT_CHUNK; }

int sat_grad_clip_coef(const sat_opt_tensor* tensors, const sat_opt_chunk* chunks, int32_t n_chunks, float max_norm, double* scratch /* n_chunks */,
                       float* coef /* 2 floats */, void* stream) {
    if (!tensors || !chunks || !scratch || !coef) return fail(SAT_EINVAL, "grad_clip_coef: null pointer");
    SAT_REQUIRE(n_chunks > 0 && max_norm > 0.f, "grad_clip_coef: n_chunks=%d max_norm=%g", n_chunks, max_norm);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(grad_sqsum_part_kernel, dim3(n_chunks), dim3(256), 0, st, tensors, chunks, scratch);
    SAT_TRY(launch_ok("grad_sqsum_part"));
    hipLaunchKernelGGL(grad_norm_finish_kernel, dim3(1), dim3(256), 0, st, scratch, n_chunks, max_norm, coef);
    return launch_ok("grad_norm_finish");
}

int sat_optimizer_step(const sat_opt_tensor* tensors, const sat_opt_chunk* chunks, int32_t n_chunks, const sat_opt_hyper* hyper,
                       const float* clip_coef, void* stream) {
    if (!tensors || !chunks || !hyper) return fail(SAT_EINVAL, "optimizer_step: null pointer");
    if (n_chunks <= 0) return SAT_OK;
    SAT_REQUIRE(hyper->kind == SAT_OPT_SGD || hyper->kind == SAT_OPT_ADAM || hyper->kind == SAT_OPT_ADAMW, "optimizer_step: kind %d", hyper->kind);
    SAT_REQUIRE(hyper->kind == SAT_OPT_SGD || (hyper->bias_correction1 > 0.f && hyper->bias_correction2_sqrt > 0.f), "optimizer_step: bias corrections must be positive");
    hipLaunchKernelGGL(optimizer_step_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, tensors, chunks, *hyper, (const sat_opt_hyper*)nullptr, clip_coef);
    return launch_ok("optimizer_step");
}

int sat_optimizer_step_dev(const sat_opt_tensor* tensors, const sat_opt_chunk* chunks, int32_t n_chunks, const sat_opt_hyper* hyper_dev,
                           const float* clip_coef, void* stream) {
    if (!tensors || !chunks || !hyper_dev) return fail(SAT_EINVAL, "optimizer_step_dev: null pointer");
    if (n_chunks <= 0) return SAT_OK;
    sat_opt_hyper none = {};
    hipLaunchKernelGGL(optimizer_step_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, tensors, chunks, none, hyper_dev, clip_coef);
    return launch_ok("optimizer_step_dev");
}

}
