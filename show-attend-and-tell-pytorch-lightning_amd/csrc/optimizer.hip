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

__global__ __launch_bounds__(256) void optimizer_step_kernel(const sat_opt_tensor* __restrict__ tensors, const sat_opt_chunk* __restrict__ chunks,
                                                             sat_opt_hyper h, const float* __restrict__ clip_coef) {
    const sat_opt_chunk c = chunks[blockIdx.x];
    const sat_opt_tensor t = tensors[c.tensor];
    const long end = (c.start + OPT_CHUNK < t.n) ? c.start + OPT_CHUNK : t.n;
    const float coef = clip_coef ? clip_coef[0] : 1.f;
    const float lr = t.lr, wd = t.weight_decay;
    for (long i = c.start + threadIdx.x; i < end; i += 256) {
        float g = t.g[i] * coef;
        if (h.clip_value > 0.f) g = fminf(fmaxf(g, -h.clip_value), h.clip_value);        // clip_grad_value_
        float p = t.p[i];
        if (h.kind == SAT_OPT_SGD) {
            if (wd != 0.f) g = fmaf(wd, p, g);
            if (h.momentum != 0.f) {
                float b = h.first_step ? g : fmaf(h.momentum, t.m[i], g);                 // torch: buf = grad on the first step
                t.m[i] = b;
                g = h.nesterov ? fmaf(h.momentum, b, g) : b;
            }
            t.p[i] = p - lr * g;
        } else {
            if (h.kind == SAT_OPT_ADAMW) p *= 1.f - lr * wd;                              // decoupled decay
            else if (wd != 0.f) g = fmaf(wd, p, g);                                       // Adam: L2 term joins the gradient
            const float m = t.m[i] + (g - t.m[i]) * (1.f - h.beta1);                       // exp_avg.lerp_(grad, 1 - beta1)
            const float v = h.beta2 * t.v[i] + (1.f - h.beta2) * g * g;
            t.m[i] = m; t.v[i] = v;
            const float denom = sqrtf(v) / h.bias_correction2_sqrt + h.eps;
            t.p[i] = p - (lr / h.bias_correction1) * (m / denom);
        }
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
    hipLaunchKernelGGL(optimizer_step_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, tensors, chunks, *hyper, clip_coef);
    return launch_ok("optimizer_step");
}

}
