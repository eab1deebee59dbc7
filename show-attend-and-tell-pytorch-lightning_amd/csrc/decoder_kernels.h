// Device kernels of the SAT attention-LSTM decoder (forward and backward).
// Reference semantics: model.py:76-81 (InitLSTM), 94-109 (SoftAttention),
// 125-131 (DeepOutput), 187-192 (beta), 510-548 (train_batch loop),
// util.py:105-112 (LabelSmoothing), model.py:594 (doubly stochastic term).
//
// Layout (all fp32, row-major):
//   ann    (B, L, D)   annotations, one row per location -- NHWC encoder output viewed flat.
//                      The R captions of an image share it (repeat_interleave is never materialised:
//                      caption row i belongs to image i / R).
//   U      (B, L, A)   att_enc = ann * W_e^T, hoisted out of the time loop.
//   HC[t]  (N, A+D+4n) per step: [ q = h W_d^T | beta = sigmoid(h W_b^T + b) | LSTM gates ]
//   time-major padded buffers (T1, N, .) for everything saved for backward.
#pragma once
#include "common.h"

namespace sat {

constexpr int ATT_RMAX = 8;      // caption rows of one image handled per pass
constexpr int ATT_THREADS = 256;

// ------------------------------------------------------------------ small utilities
__global__ void fill_kernel(float* p, long n, float v) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// fp32 -> bf16 copy, 4 elements per thread (n4 = elements / 4)
__global__ void cast_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long n4) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n4) return;
    const float4 v = reinterpret_cast<const float4*>(src)[e];
    typedef __bf16 b4 __attribute__((ext_vector_type(4)));
    b4 o; o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    reinterpret_cast<b4*>(dst)[e] = o;
}

// rows x cols fp32 block with row stride ld -> compact bf16 (cols % 4 == 0)
__global__ void cast_block_bf16_kernel(const float* __restrict__ src, long ld, __bf16* __restrict__ dst, int rows, int cols) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4 = cols / 4;
    if (e >= (long)rows * c4) return;
    const long r = e / c4; const int c = (int)(e - r * c4) * 4;
    const float4 v = *reinterpret_cast<const float4*>(src + r * ld + c);
    typedef __bf16 b4 __attribute__((ext_vector_type(4)));
    b4 o; o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    *reinterpret_cast<b4*>(dst + r * cols + c) = o;
}
__global__ void set4_int_kernel(int* p, int a, int b, int c, int d) { p[0] = a; p[1] = b; p[2] = c; p[3] = d; }
// plain copies / clears as kernels: the batched beam search enqueues nothing but kernel launches, so that a captured stream
// (hipGraph) is one linear chain of kernel nodes (memset / memcpy nodes were seen to run out of order inside a replayed graph)
__global__ void copy_words_kernel(unsigned* __restrict__ dst, const unsigned* __restrict__ src, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
__global__ void fill_int_kernel(int* p, long n, int v) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---- dropout (nn.Dropout in training mode: model.py:74/78, 117/130, 164/526).  The mask is a counter-based hash of
// (seed, stream, element index): no state, recomputed wherever it is needed (forward, backward), reproducible.
// streams: 0 = InitLSTM mean, 1 = embedding, 2 = DeepOutput.  keep-probability 1-p, kept values scaled by 1/(1-p).
__device__ __forceinline__ float dropout_scale(unsigned long long seed, unsigned stream, unsigned long long idx, float p) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(stream + 1) + idx * 0xD1342543DE82EF95ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);      // 24 bits -> [0,1)
    return (u >= p) ? 1.0f / (1.0f - p) : 0.f;
}

// dst[r, :] = table[tok[r], :] (embedding gather; tok < 0 -> zeros), with embedding_dropout when p > 0
__global__ void gather_rows_kernel(const float* __restrict__ table, const int* __restrict__ tok, float* __restrict__ out, int rows, int width,
                                   float p, unsigned long long seed, long row0) {
    int r = blockIdx.x;
    if (r >= rows) return;
    int t = tok[r];
    for (int c = threadIdx.x; c < width; c += blockDim.x) {
        float v = (t >= 0) ? table[(long)t * width + c] : 0.f;
        if (p > 0.f) v *= dropout_scale(seed, 1, (unsigned long long)(row0 + r) * width + c, p);
        out[(long)r * width + c] = v;
    }
}
// y[r, c] = x[r, c] * mask(stream, (row0 + r) * width + c)   (in place allowed)
__global__ void dropout_rows_kernel(const float* __restrict__ x, float* __restrict__ y, long total, int width, float p,
                                    unsigned long long seed, unsigned stream, long row0) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < total) y[e] = x[e] * dropout_scale(seed, stream, (unsigned long long)row0 * width + e, p);
}
// InitLSTM with dropout: the mean of the REPEATED annotations is dropped per caption row (model.py:78 after 487)
__global__ void init_mean_rows_kernel(const float* __restrict__ mean, float* __restrict__ out, int N, int R, int D, float p, unsigned long long seed) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)N * D) return;
    long i = e / D; int d = (int)(e - i * D);
    out[e] = mean[(i / R) * D + d] * dropout_scale(seed, 0, (unsigned long long)e, p);
}
// dmean[b, d] = sum_r dmean_rows[b*R + r, d] * mask
__global__ void init_mean_rows_bwd_kernel(const float* __restrict__ drows, float* __restrict__ dmean, int B, int R, int D, float p, unsigned long long seed) {
    long o = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= (long)B * D) return;
    long b = o / D; int d = (int)(o - b * D);
    float s = 0.f;
    for (int r = 0; r < R; ++r) { long e = (b * R + r) * D + d; s += drows[e] * dropout_scale(seed, 0, (unsigned long long)e, p); }
    dmean[o] = s;
}

// Packed per-step parameters in one launch: Wcat = [W_d ; W_beta ; W_hh] (A + D + 4n rows of n), its bf16 copy when the per-step GEMMs read
// bf16 operands, and bcat = [0 ; b_beta ; b_ih + b_hh].
__global__ void pack_wcat_kernel(const float* __restrict__ att_dec, const float* __restrict__ beta_w, const float* __restrict__ w_hh,
                                 const float* __restrict__ beta_b, const float* __restrict__ b_ih, const float* __restrict__ b_hh, float* __restrict__ Wcat,
                                 __bf16* __restrict__ Wcat_b, float* __restrict__ bcat, int A, int D, int n) {
    const long HCW = (long)A + D + 4L * n, nw = HCW * n, e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nw) {
        const long r = e / n;
        const float v = r < A ? att_dec[e] : (r < A + D ? beta_w[e - (long)A * n] : w_hh[e - (long)(A + D) * n]);
        Wcat[e] = v;
        if (Wcat_b) Wcat_b[e] = (__bf16)v;
    } else if (e < nw + HCW) {
        const long r = e - nw;
        bcat[r] = r < A ? 0.f : (r < A + D ? beta_b[r - A] : b_ih[r - A - D] + b_hh[r - A - D]);
    }
}

// token ids of step `step` for every caption row (teacher forcing): tok[i] = caps[i*T + step]
// (grid.y > 1: steps step .. step + grid.y - 1 in one launch, tok advancing by N per step)
__global__ void teacher_tokens_kernel(const int* __restrict__ caps, const int* __restrict__ lengths, int* __restrict__ tok, int N, int T, int step) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    step += blockIdx.y; tok += (long)blockIdx.y * N;
    if (i < N) tok[i] = (lengths[i] > step) ? caps[(long)i * T + step] : -1;     // finished captions feed nothing
}

// argmax feedback (model.py:523): tok[i] = argmax_v logits_packed[prow_prev[i], v] (first maximum), -1 for dead rows
__global__ void argmax_tokens_kernel(const float* __restrict__ logits, const int* __restrict__ prow_prev, const int* __restrict__ lengths,
                                     int* __restrict__ tok, int V, int step) {
    int i = blockIdx.x;
    __shared__ float sv[4];
    __shared__ int si[4];
    if (lengths[i] <= step) { if (threadIdx.x == 0) tok[i] = -1; return; }
    const float* row = logits + (long)prow_prev[i] * V;
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
        float x = row[v];
        if (x > best || (x == best && v < bi)) { best = x; bi = v; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        float ob = __shfl_xor(best, o, 64); int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sv[w] = best; si[w] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < (int)(blockDim.x >> 6); ++k)
            if (sv[k] > best || (sv[k] == best && si[k] < bi)) { best = sv[k]; bi = si[k]; }
        tok[i] = bi;
    }
}

// the same decision and the embedding row of the chosen token in ONE launch (scheduled sampling without max-norm renormalisation: model.py:521-526):
// token of caption i at `step` = first maximum of its previous logits row, y[i, :] = embedding[token] (zeros for finished captions)
__global__ __launch_bounds__(256) void argmax_gather_kernel(const float* __restrict__ logits, const int* __restrict__ prow_prev, const int* __restrict__ lengths,
                                                            int* __restrict__ tok, int V, int step, const float* __restrict__ table, float* __restrict__ out, int width,
                                                            float p, unsigned long long seed, long row0) {
    const int i = blockIdx.x;
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ int s_tok;
    const bool live = lengths[i] > step;          // uniform over the block
    int bi = 0x7fffffff;
    if (live) {
        const float* row = logits + (long)prow_prev[i] * V;
        float best = -INFINITY;
        for (int v = threadIdx.x; v < V; v += 256) {
            const float x = row[v];
            if (x > best || (x == best && v < bi)) { best = x; bi = v; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { sv[w] = best; si[w] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int k = 1; k < 4; ++k)
                if (sv[k] > best || (sv[k] == best && si[k] < bi)) { best = sv[k]; bi = si[k]; }
            s_tok = bi; tok[i] = bi;
        }
        __syncthreads();
    } else if (threadIdx.x == 0) tok[i] = -1;
    const int t = live ? s_tok : -1;
    for (int c = threadIdx.x; c < width; c += 256) {
        float v = (t >= 0) ? table[(long)t * width + c] : 0.f;
        if (p > 0.f) v *= dropout_scale(seed, 1, (unsigned long long)(row0 + i) * width + c, p);
        out[(long)i * width + c] = v;
    }
}

// mean over L of the annotations: mean[b, d]  (model.py:78)
__global__ void ann_mean_kernel(const float* __restrict__ ann, float* __restrict__ mean, int L, int D) {
    int b = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float s = 0.f;
        for (int l = 0; l < L; ++l) s += ann[((long)b * L + l) * D + d];
        mean[(long)b * D + d] = s / (float)L;
    }
}

// InitLSTM raw reshape (model.py:79-80): the (N, 2*layers*n) buffer whose row j is init[j / R]
// is reinterpreted as (2*layers, N, n): slabs [0, layers) are h0, the rest c0.  flat element e -> row e / (2*layers*n).
// Layer slab l of h0 / c0 is written at h0 + l * lstride (the time-major state buffers keep one run per layer).
__global__ void init_expand_kernel(const float* __restrict__ init_img, float* __restrict__ h0, float* __restrict__ c0, int N, int R, int n,
                                   int layers, long lstride) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long slab_sz = (long)N * n, w2 = 2L * layers * n;
    if (e >= 2 * layers * slab_sz) return;
    long row = e / w2, col = e % w2;
    float v = init_img[(row / R) * w2 + col];
    long slab = e / slab_sz, within = e - slab * slab_sz;
    if (slab < layers) h0[slab * lstride + within] = v; else c0[(slab - layers) * lstride + within] = v;
}
// backward of the above: dinit_img[b, col] = sum_r dflat[(b*R + r), col]
__global__ void init_expand_bwd_kernel(const float* __restrict__ dh0, const float* __restrict__ dc0, float* __restrict__ dinit_img, int B, int R, int n) {
    long o = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int N = B * R; long half = (long)N * n;
    if (o >= (long)B * 2 * n) return;
    long b = o / (2 * n), col = o % (2 * n);
    float s = 0.f;
    for (int r = 0; r < R; ++r) {
        long e = (b * R + r) * (2 * n) + col;
        s += (e < half) ? dh0[e] : dc0[e - half];
    }
    dinit_img[o] = s;
}

// ------------------------------------------------------------------ attention forward (one decode step)
// grid = (B, D-chunks); block = 256.  Every block recomputes the (cheap) scores + softmax of its image's
// R caption rows, then accumulates its D-chunk of the context while reading the annotation tile ONCE
// for all R rows.  dyn LDS: [RMAX*L scores/alpha][RMAX*A q][A w][part: 4*RMAX*chunk]
constexpr int ATTF_WAVES = 16;                 // forward: 16 waves share the tanh-heavy score phase of one image
constexpr int ATTF_THREADS = ATTF_WAVES * 64;
// RN = caption rows handled per pass (compile-time: the row loops unroll without per-row guards -- a guard per FMA serialises
// every FMA behind its operand read; dead rows run on zeros instead and are masked at the stores).
template <int VW, int RN>
__global__ __launch_bounds__(ATTF_THREADS) void attention_fwd_kernel(
    const float* __restrict__ ann, const float* __restrict__ U, const float* __restrict__ hc, int hc_ld,
    const float* __restrict__ wf, const int* __restrict__ lengths, int step,
    float* __restrict__ alphas, int T1, float* __restrict__ Z, float* __restrict__ XZ,
    int R, int L, int D, int A, int dchunk) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_sc = sm;                        // [RMAX][L]
    float* s_q = s_sc + ATT_RMAX * L;        // [RMAX][A]
    float* s_w = s_q + ATT_RMAX * A;         // [A]
    float* s_part = s_w + A;                 // [4 groups][RMAX][dchunk]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d0 = blockIdx.y * dchunk, dn = min(dchunk, D - d0);
    const float scale = 1.0f / sqrtf((float)L);
    for (int k = tid; k < A; k += ATTF_THREADS) s_w[k] = wf[k];

    for (int r0 = 0; r0 < R; r0 += RN) {
        const int rn = min(RN, R - r0);
        const int i0 = b * R + r0;
        unsigned lmask = 0;
        for (int r = 0; r < rn; ++r) if (lengths[i0 + r] > step) lmask |= 1u << r;
        const bool any = lmask != 0;
        __syncthreads();
        if (any) {
            for (int e = tid; e < RN * A; e += ATTF_THREADS) { int r = e / A, k = e - r * A; s_q[r * A + k] = (r < rn) ? hc[(long)(i0 + r) * hc_ld + k] : 0.f; }
            __syncthreads();
            // ---- scores: wave per location, lanes over the attention dim
            for (int l = wave; l < L; l += ATTF_WAVES) {
                float part[RN];
#pragma unroll
                for (int r = 0; r < RN; ++r) part[r] = 0.f;
                const float* u = U + ((long)b * L + l) * A;
                for (int k = lane; k < A; k += 64) {
                    float uv = u[k], w = s_w[k];
#pragma unroll
                    for (int r = 0; r < RN; ++r) part[r] = fmaf(w, fast_tanh(uv + s_q[r * A + k]), part[r]);
                }
#pragma unroll
                for (int r = 0; r < RN; ++r) { float s = wave_sum(part[r]); if (lane == 0) s_sc[r * L + l] = s * scale; }
            }
            __syncthreads();
            // ---- softmax over L: wave per row
            for (int r = wave; r < rn; r += ATTF_WAVES) {
                if (!((lmask >> r) & 1u)) continue;       // wave-uniform
                float mx = -INFINITY;
                for (int l = lane; l < L; l += 64) mx = fmaxf(mx, s_sc[r * L + l]);
                mx = wave_max(mx);
                float sum = 0.f;
                for (int l = lane; l < L; l += 64) { float e = __expf(s_sc[r * L + l] - mx); s_sc[r * L + l] = e; sum += e; }
                sum = wave_sum(sum);
                float inv = 1.0f / sum;
                for (int l = lane; l < L; l += 64) s_sc[r * L + l] *= inv;
            }
            __syncthreads();
            // dead rows (finished captions, rows past R) take zero attention weights: the context loop below runs unguarded
            for (int e = tid; e < RN * L; e += ATTF_THREADS) { int r = e / L; if (!((lmask >> r) & 1u)) s_sc[e] = 0.f; }
            __syncthreads();
        }
        // ---- alphas out (only the first D-chunk block writes); dead rows are written as zeros
        if (blockIdx.y == 0) {
            for (int e = tid; e < rn * L; e += ATTF_THREADS) {
                int r = e / L, l = e - r * L;
                alphas[((long)(i0 + r) * T1 + step) * L + l] = ((lmask >> r) & 1u) ? s_sc[r * L + l] : 0.f;
            }
        }
        // ---- context: z[r][d] = sum_l alpha[r][l] * ann[b,l,d] for d in this chunk
        const int nv = dn / VW;                         // vectors in the chunk
        const int groups = max(1, min(4, ATT_THREADS / max(nv, 1)));
        const int gsz = ATT_THREADS / groups;           // threads per group
        const bool ctx_thread = tid < ATT_THREADS;       // the context phase runs on the first 4 waves; all waves do the scores
        const int g = ctx_thread ? tid / gsz : groups, tv = tid - (ctx_thread ? g : 0) * gsz;
        float acc[RN][VW];
        if (any) {
            for (int v0 = 0; v0 < nv; v0 += gsz) {
                const int v = v0 + tv;
#pragma unroll
                for (int r = 0; r < RN; ++r)
#pragma unroll
                    for (int c = 0; c < VW; ++c) acc[r][c] = 0.f;
                if (v < nv && g < groups) {
                    const float* base = ann + (long)b * L * D + d0 + v * VW;
                    for (int l = g; l < L; l += groups) {
                        float x[VW];
                        if (VW == 4) { float4 t4 = *reinterpret_cast<const float4*>(base + (long)l * D); x[0] = t4.x; x[1 % VW] = t4.y; x[2 % VW] = t4.z; x[3 % VW] = t4.w; }
                        else x[0] = base[(long)l * D];
#pragma unroll
                        for (int r = 0; r < RN; ++r) { const float al = s_sc[r * L + l];
#pragma unroll
                            for (int c = 0; c < VW; ++c) acc[r][c] = fmaf(al, x[c], acc[r][c]); }
                    }
                }
                // combine the l-groups through LDS in a fixed order
                __syncthreads();
                if (v < nv && g < groups)
#pragma unroll
                    for (int r = 0; r < RN; ++r)
#pragma unroll
                        for (int c = 0; c < VW; ++c) s_part[((g * ATT_RMAX + r) * gsz + tv) * VW + c] = acc[r][c];
                __syncthreads();
                if (g == 0 && v < nv) {
#pragma unroll
                    for (int r = 0; r < RN; ++r) {
                        if (r >= rn) continue;
                        const long orow = (long)(i0 + r);
#pragma unroll
                        for (int c = 0; c < VW; ++c) {
                            float z = 0.f;
                            if (((lmask >> r) & 1u)) for (int gg = 0; gg < groups; ++gg) z += s_part[((gg * ATT_RMAX + r) * gsz + tv) * VW + c];
                            const int d = d0 + v * VW + c;
                            const float beta = ((lmask >> r) & 1u) ? hc[orow * hc_ld + A + d] : 0.f;
                            Z[orow * D + d] = z;
                            XZ[orow * D + d] = beta * z;
                        }
                    }
                }
            }
        } else {
            for (int e = tid; e < rn * dn; e += ATTF_THREADS) { int r = e / dn, d = d0 + e - r * dn; Z[(long)(i0 + r) * D + d] = 0.f; XZ[(long)(i0 + r) * D + d] = 0.f; }
        }
    }
}

// ------------------------------------------------------------------ attention forward, split over the chip
// The fused kernel above runs one workgroup per (image, 256-wide slice of D): 256 workgroups, each recomputing the image's
// scores, and a context phase on 4 waves.  The two kernels below spread the step over ~1000 workgroups:
//   scores : grid (B, ceil(L / 16)), wave per location -> raw scaled scores (N, L) in scratch
//   context: grid (B, D / 64): softmax of the image's RN rows (cheap, redundant per slice), then 16 location groups x 16
//            float4 vectors accumulate the slice of the context; fixed combination order.
constexpr int ATTS_WAVES = 16;
template <int RN>
__global__ __launch_bounds__(ATTS_WAVES * 64) void attention_scores_kernel(const float* __restrict__ U, const float* __restrict__ hc, int hc_ld,
                                                                           const float* __restrict__ wf, float* __restrict__ sc, int R, int L, int A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_q = sm;                 // [RN][A]
    float* s_w = s_q + RN * A;       // [A]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l = blockIdx.y * ATTS_WAVES + wave;
    const float scale = 1.0f / sqrtf((float)L);
    for (int k = tid; k < A; k += ATTS_WAVES * 64) s_w[k] = wf[k];
    for (int r0 = 0; r0 < R; r0 += RN) {
        const int rn = min(RN, R - r0), i0 = b * R + r0;
        __syncthreads();
        for (int e = tid; e < RN * A; e += ATTS_WAVES * 64) { int r = e / A, k = e - r * A; s_q[e] = (r < rn) ? hc[(long)(i0 + r) * hc_ld + k] : 0.f; }
        __syncthreads();
        if (l < L) {
            float part[RN];
#pragma unroll
            for (int r = 0; r < RN; ++r) part[r] = 0.f;
            const float* u = U + ((long)b * L + l) * A;
            for (int k = lane; k < A; k += 64) {
                const float uv = u[k], w = s_w[k];
#pragma unroll
                for (int r = 0; r < RN; ++r) part[r] = fmaf(w, fast_tanh(uv + s_q[r * A + k]), part[r]);
            }
#pragma unroll
            for (int r = 0; r < RN; ++r) { const float s = wave_sum(part[r]); if (lane == 0 && r < rn) sc[(long)(i0 + r) * L + l] = s * scale; }
        }
    }
}

constexpr int ATTC_DCH = 64;          // context slice: 64 features = 16 float4 vectors x 16 location groups per 256 threads
// TA = float, or __bf16: the bf16 copy of the annotations that bf16 mode keeps for the two kernels that stream them every time step
// (context here, dalpha in the backward pass): half the bytes of the decoder's largest per-step stream; products and sums stay fp32.
template <typename TA> __device__ __forceinline__ float4 ld_ann4(const TA* p);
template <> __device__ __forceinline__ float4 ld_ann4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 ld_ann4<__bf16>(const __bf16* p) {
    typedef __bf16 b4 __attribute__((ext_vector_type(4)));
    const b4 q = *reinterpret_cast<const b4*>(p);
    return make_float4((float)q[0], (float)q[1], (float)q[2], (float)q[3]);
}
template <int RN, typename TA>
__global__ __launch_bounds__(256) void attention_context_kernel(const TA* __restrict__ ann, const float* __restrict__ sc, const float* __restrict__ hc,
                                                                int hc_ld, const int* __restrict__ lengths, int step, float* __restrict__ alphas, int T1,
                                                                float* __restrict__ Z, float* __restrict__ XZ, int R, int L, int D, int A,
                                                                __bf16* __restrict__ xzb) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_al = sm;                          // [RN][L]
    float4* s_part = reinterpret_cast<float4*>(sm + ((RN * L + 3) & ~3));   // [16 groups][RN][16 vectors]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d0 = blockIdx.y * ATTC_DCH;
    const int v = tid & 15, g = tid >> 4;
    for (int r0 = 0; r0 < R; r0 += RN) {
        const int rn = min(RN, R - r0), i0 = b * R + r0;
        unsigned lmask = 0;
        for (int r = 0; r < rn; ++r) if (lengths[i0 + r] > step) lmask |= 1u << r;
        __syncthreads();
        // softmax over L of the live rows (wave per row); dead rows and rows past R get zero weights
        for (int r = wave; r < RN; r += 4) {
            const bool livr = (lmask >> r) & 1u;
            float mx = -INFINITY;
            if (livr) for (int l = lane; l < L; l += 64) mx = fmaxf(mx, sc[(long)(i0 + r) * L + l]);
            mx = wave_max(mx);
            float sum = 0.f;
            if (livr) for (int l = lane; l < L; l += 64) { const float e = __expf(sc[(long)(i0 + r) * L + l] - mx); s_al[r * L + l] = e; sum += e; }
            sum = wave_sum(sum);
            const float inv = livr ? 1.0f / sum : 0.f;
            for (int l = lane; l < L; l += 64) s_al[r * L + l] = livr ? s_al[r * L + l] * inv : 0.f;
        }
        __syncthreads();
        if (blockIdx.y == 0)
            for (int e = tid; e < rn * L; e += 256) { int r = e / L, l = e - r * L; alphas[((long)(i0 + r) * T1 + step) * L + l] = s_al[r * L + l]; }
        // context slice
        float4 acc[RN];
#pragma unroll
        for (int r = 0; r < RN; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int d = d0 + 4 * v;
        if (d < D && lmask) {
            const TA* base = ann + (long)b * L * D + d;
            for (int l0 = g; l0 < L; l0 += 64) {          // four locations per trip, their loads issued together (L = 196: 4 trips instead of 13)
                float4 x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int l = l0 + 16 * u; x[u] = l < L ? ld_ann4<TA>(base + (long)l * D) : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int l = l0 + 16 * u;
                    if (l >= L) continue;
#pragma unroll
                    for (int r = 0; r < RN; ++r) {
                        const float al = s_al[r * L + l];
                        acc[r].x = fmaf(al, x[u].x, acc[r].x); acc[r].y = fmaf(al, x[u].y, acc[r].y); acc[r].z = fmaf(al, x[u].z, acc[r].z); acc[r].w = fmaf(al, x[u].w, acc[r].w);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RN; ++r) s_part[(g * RN + r) * 16 + v] = acc[r];
        __syncthreads();
        // thread (r = tid / 16 .., v): sums the 16 location groups in order and writes z and beta * z
        for (int e = tid; e < rn * 16; e += 256) {
            const int r = e >> 4, vv = e & 15, dd = d0 + 4 * vv;
            if (dd >= D) continue;
            float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int gg = 0; gg < 16; ++gg) { const float4 t = s_part[(gg * RN + r) * 16 + vv]; z.x += t.x; z.y += t.y; z.z += t.z; z.w += t.w; }
            const long orow = i0 + r;
            float4 be = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((lmask >> r) & 1u) be = *reinterpret_cast<const float4*>(hc + orow * hc_ld + A + dd);
            *reinterpret_cast<float4*>(Z + orow * D + dd) = z;
            *reinterpret_cast<float4*>(XZ + orow * D + dd) = make_float4(be.x * z.x, be.y * z.y, be.z * z.z, be.w * z.w);
            if (xzb) {
                typedef __bf16 b4 __attribute__((ext_vector_type(4)));
                b4 o; o[0] = (__bf16)(be.x * z.x); o[1] = (__bf16)(be.y * z.y); o[2] = (__bf16)(be.z * z.z); o[3] = (__bf16)(be.w * z.w);
                *reinterpret_cast<b4*>(xzb + orow * D + dd) = o;
            }
        }
    }
}

// ------------------------------------------------------------------ LSTM cell (pointwise part)
// gates (N,4n) pre-activation = h W_hh^T + xz W_ih_z^T + biases (already in HC) + GY (embedding part).
// Gate order i,f,g,o.  Activated gates are written back in place for backward.
__global__ void lstm_cell_fwd_kernel(float* __restrict__ gates, int g_ld, const float* __restrict__ gy,
                                     const float* __restrict__ c_prev, const float* __restrict__ h_prev,
                                     float* __restrict__ c_new, float* __restrict__ h_new,
                                     const int* __restrict__ lengths, int step, int N, int n, __bf16* __restrict__ hb_new = nullptr) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * n) return;
    int i = (int)(idx / n), j = (int)(idx - (long)i * n);
    if (lengths && lengths[i] <= step) {       // finished caption: state is carried unchanged (model.py:544 updates live rows only); lengths NULL: every row live
        const float hp = h_prev[idx];
        c_new[idx] = c_prev[idx]; h_new[idx] = hp;
        if (hb_new) hb_new[idx] = (__bf16)hp;
        float* g = gates + (long)i * g_ld;
        g[j] = 0.f; g[n + j] = 0.f; g[2 * n + j] = 0.f; g[3 * n + j] = 0.f;
        return;
    }
    float* g = gates + (long)i * g_ld;
    const float* y = gy ? gy + (long)i * 4 * n : nullptr;
    float gi = fast_sigmoid(g[j] + (y ? y[j] : 0.f));
    float gf = fast_sigmoid(g[n + j] + (y ? y[n + j] : 0.f));
    float gg = fast_tanh(g[2 * n + j] + (y ? y[2 * n + j] : 0.f));
    float go = fast_sigmoid(g[3 * n + j] + (y ? y[3 * n + j] : 0.f));
    float c = gf * c_prev[idx] + gi * gg;
    c_new[idx] = c;
    const float hn = go * fast_tanh(c);
    h_new[idx] = hn;
    if (hb_new) hb_new[idx] = (__bf16)hn;          // bf16 copy: operand of the next step's GEMM (bf16 mode)
    g[j] = gi; g[n + j] = gf; g[2 * n + j] = gg; g[3 * n + j] = go;
}

// backward of the cell.  dh_carry/dc_carry are the running gradients w.r.t. h_t / c_t from later steps;
// dh_out is the gradient arriving from the output layer of this step.  Writes dG (pre-activation gate
// grads) and leaves dh_carry = 0 for live rows (the h_{t-1} gradient is accumulated by the GEMM after).
__global__ void lstm_cell_bwd_kernel(const float* __restrict__ gates, int g_ld, const float* __restrict__ c_prev,
                                     const float* __restrict__ c_new, const float* __restrict__ dh_out,
                                     float* __restrict__ dh_carry, float* __restrict__ dc_carry,
                                     float* __restrict__ dgates, int dg_ld,
                                     const int* __restrict__ lengths, int step, int N, int n, __bf16* __restrict__ dgb = nullptr) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * n) return;
    int i = (int)(idx / n), j = (int)(idx - (long)i * n);
    float* dg = dgates + (long)i * dg_ld;
    __bf16* db = dgb ? dgb + (long)i * dg_ld : nullptr;            // bf16 copy with the same row stride (operand of this step's GEMMs)
    if (lengths && lengths[i] <= step) {
        dg[j] = 0.f; dg[n + j] = 0.f; dg[2 * n + j] = 0.f; dg[3 * n + j] = 0.f;
        if (db) { db[j] = (__bf16)0.f; db[n + j] = (__bf16)0.f; db[2 * n + j] = (__bf16)0.f; db[3 * n + j] = (__bf16)0.f; }
        return;
    }
    const float* g = gates + (long)i * g_ld;
    float gi = g[j], gf = g[n + j], gg = g[2 * n + j], go = g[3 * n + j];
    float dh = dh_carry[idx] + (dh_out ? dh_out[idx] : 0.f);
    float tc = fast_tanh(c_new[idx]);
    float dc = dc_carry[idx] + dh * go * (1.f - tc * tc);
    const float d0 = dc * gg * gi * (1.f - gi), d1 = dc * c_prev[idx] * gf * (1.f - gf), d2 = dc * gi * (1.f - gg * gg), d3 = dh * tc * go * (1.f - go);
    dg[j] = d0; dg[n + j] = d1; dg[2 * n + j] = d2; dg[3 * n + j] = d3;
    if (db) { db[j] = (__bf16)d0; db[n + j] = (__bf16)d1; db[2 * n + j] = (__bf16)d2; db[3 * n + j] = (__bf16)d3; }
    dc_carry[idx] = dc * gf;
    dh_carry[idx] = 0.f;
}

// ------------------------------------------------------------------ attention backward (one decode step)
// grid = B; block = 256.  Per live caption row i of image b:
//   dz = dZ_out + dXZ*beta ; dbeta_pre = dXZ * z * beta(1-beta)
//   dalpha_l = dz . ann[b,l,:] + dalpha_ext ; ds = alpha (dalpha - sum alpha dalpha)
//   dpre_lk = ds_l L^-1/2 w_k (1 - tanh^2(U_lk + q_k)) ; dq_k = sum_l dpre ; dU += sum_r dpre ; dw_k += sum ds L^-1/2 tanh
// dyn LDS: [RMAX*L alpha][RMAX*L dalpha/ds][RMAX*A q][A w][RMAX*D dz][4*RMAX*A dq partial][4*A dw partial]
constexpr int ATTB_WAVES = 16;                 // 1024 threads: one block per image, so hide latency with waves
constexpr int ATTB_THREADS = ATTB_WAVES * 64;
template <int RN>
__global__ __launch_bounds__(ATTB_THREADS) void attention_bwd_kernel(
    const float* __restrict__ ann, const float* __restrict__ U, const float* __restrict__ hc, int hc_ld,
    const float* __restrict__ wf, const int* __restrict__ lengths, int step,
    const float* __restrict__ alphas, const float* __restrict__ dalphas_ext, int T1,
    const float* __restrict__ Zs, const float* __restrict__ dZ_out, const float* __restrict__ dXZ,
    float* __restrict__ DZ, float* __restrict__ dhc, int dhc_ld, float* __restrict__ dU, float* __restrict__ dwf_part,
    int R, int L, int D, int A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_al = sm;                         // [RMAX][L]
    float* s_da = s_al + ATT_RMAX * L;        // [RMAX][L]
    float* s_q = s_da + ATT_RMAX * L;         // [RMAX][A]
    float* s_w = s_q + ATT_RMAX * A;          // [A]
    float* s_dz = s_w + A;                    // [RMAX][D]
    float* s_dq = s_dz + ATT_RMAX * D;        // [waves][RMAX][A]
    float* s_dw = s_dq + ATTB_WAVES * ATT_RMAX * A;    // [waves][A]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float scale = 1.0f / sqrtf((float)L);
    for (int k = tid; k < A; k += ATTB_THREADS) s_w[k] = wf[k];
    for (int e = tid; e < ATTB_WAVES * A; e += ATTB_THREADS) s_dw[e] = 0.f;

    for (int r0 = 0; r0 < R; r0 += RN) {
        const int rn = min(RN, R - r0);
        const int i0 = b * R + r0;
        unsigned lmask = 0;
        for (int r = 0; r < rn; ++r) if (lengths[i0 + r] > step) lmask |= 1u << r;
        const bool any = lmask != 0;
        __syncthreads();
        // gate backward + dz for every row of the pass (dead rows and rows past R: zeros, so the loops below run unguarded)
        for (int e = tid; e < RN * D; e += ATTB_THREADS) {
            int r = e / D, d = e - r * D; long row = i0 + r;
            float dz = 0.f, dbp = 0.f;
            if (((lmask >> r) & 1u)) {
                float beta = hc[row * hc_ld + A + d], z = Zs[row * D + d], dx = dXZ[row * D + d];
                dz = dZ_out[row * D + d] + dx * beta;
                dbp = dx * z * beta * (1.f - beta);
            }
            s_dz[r * D + d] = dz;
            if (r < rn) { DZ[row * D + d] = dz; dhc[row * dhc_ld + A + d] = dbp; }
        }
        if (!any) {
            for (int e = tid; e < rn * A; e += ATTB_THREADS) { int r = e / A, k = e - r * A; dhc[(long)(i0 + r) * dhc_ld + k] = 0.f; }
            continue;
        }
        for (int e = tid; e < RN * A; e += ATTB_THREADS) { int r = e / A, k = e - r * A; s_q[r * A + k] = (r < rn) ? hc[(long)(i0 + r) * hc_ld + k] : 0.f; }
        for (int e = tid; e < RN * L; e += ATTB_THREADS) { int r = e / L, l = e - r * L; s_al[r * L + l] = ((lmask >> r) & 1u) ? alphas[((long)(i0 + r) * T1 + step) * L + l] : 0.f; }
        __syncthreads();
        // ---- dalpha[r][l] = dz[r] . ann[b,l,:]  (wave per location)
        for (int l = wave; l < L; l += ATTB_WAVES) {
            float part[RN];
#pragma unroll
            for (int r = 0; r < RN; ++r) part[r] = 0.f;
            const float* a = ann + ((long)b * L + l) * D;
            for (int d = lane; d < D; d += 64) {
                float av = a[d];
#pragma unroll
                for (int r = 0; r < RN; ++r) part[r] = fmaf(av, s_dz[r * D + d], part[r]);
            }
#pragma unroll
            for (int r = 0; r < RN; ++r) {
                float s = wave_sum(part[r]);
                if (lane == 0) s_da[r * L + l] = ((lmask >> r) & 1u) ? s + (dalphas_ext ? dalphas_ext[((long)(i0 + r) * T1 + step) * L + l] : 0.f) : 0.f;
            }
        }
        __syncthreads();
        // ---- softmax backward, wave per row: ds = alpha * (dalpha - sum alpha*dalpha)
        for (int r = wave; r < rn; r += ATTB_WAVES) {
            if (!((lmask >> r) & 1u)) continue;
            float dot = 0.f;
            for (int l = lane; l < L; l += 64) dot += s_al[r * L + l] * s_da[r * L + l];
            dot = wave_sum(dot);
            for (int l = lane; l < L; l += 64) s_da[r * L + l] = s_al[r * L + l] * (s_da[r * L + l] - dot) * scale;
        }
        __syncthreads();
        // ---- through tanh: wave per location, lanes over the attention dim (2 per lane at A=128)
        for (int k0 = 0; k0 < A; k0 += 64) {
            const int k = k0 + lane;
            float dq[RN]; float dw = 0.f;
#pragma unroll
            for (int r = 0; r < RN; ++r) dq[r] = 0.f;
            if (k < A) {
                const float w = s_w[k];
                for (int l = wave; l < L; l += ATTB_WAVES) {
                    const long uo = ((long)b * L + l) * A + k;
                    const float uv = U[uo];
                    float du = 0.f;
#pragma unroll
                    for (int r = 0; r < RN; ++r) {             // dead rows: ds = 0
                        float th = fast_tanh(uv + s_q[r * A + k]);
                        float ds = s_da[r * L + l];
                        float dp = ds * w * (1.f - th * th);
                        dq[r] += dp; du += dp; dw = fmaf(ds, th, dw);
                    }
                    dU[uo] += du;                   // this block owns image b: plain read-modify-write, fixed order
                }
#pragma unroll
                for (int r = 0; r < RN; ++r) s_dq[(wave * ATT_RMAX + r) * A + k] = dq[r];
                s_dw[wave * A + k] += dw;
            }
        }
        __syncthreads();
        for (int e = tid; e < rn * A; e += ATTB_THREADS) {
            int r = e / A, k = e - r * A;
            float s = 0.f;
            if (((lmask >> r) & 1u)) for (int w = 0; w < ATTB_WAVES; ++w) s += s_dq[(w * ATT_RMAX + r) * A + k];
            dhc[(long)(i0 + r) * dhc_ld + k] = s;
        }
    }
    __syncthreads();
    for (int k = tid; k < A; k += ATTB_THREADS) { float sw = 0.f; for (int w = 0; w < ATTB_WAVES; ++w) sw += s_dw[w * A + k]; dwf_part[(long)b * A + k] += sw; }
}

// ------------------------------------------------------------------ attention backward, split over the chip (see the forward split)
//   dalpha : grid (B, ceil(L / 16)): dz rows of the image in LDS (block 0 of an image also writes DZ and the gate gradient),
//            wave per location: dalpha[r][l] = dz[r] . ann[b, l, :] (+ external gradient) -> scratch (N, L)
//   tanh   : grid (B, ceil(A / 32)): softmax backward of the image's rows (redundant per slice), then 32 attention units x 32
//            location lanes go through the tanh; the block owns dU[b, :, its units], dq and dw of its units.
template <int RN, typename TA>
__global__ __launch_bounds__(1024) void attention_bwd_dalpha_kernel(const TA* __restrict__ ann, const float* __restrict__ hc, int hc_ld,
        const int* __restrict__ lengths, int step, const float* __restrict__ dalphas_ext, int T1, const float* __restrict__ Zs,
        const float* __restrict__ dZ_out, const float* __restrict__ dXZ, float* __restrict__ DZ, float* __restrict__ dhc, int dhc_ld,
        float* __restrict__ da, int R, int L, int D, int A, __bf16* __restrict__ dhcb) {
    extern __shared__ __attribute__((aligned(16))) float s_dz[];          // [RN][D]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l = blockIdx.y * 16 + wave;
    for (int r0 = 0; r0 < R; r0 += RN) {
        const int rn = min(RN, R - r0), i0 = b * R + r0;
        unsigned lmask = 0;
        for (int r = 0; r < rn; ++r) if (lengths[i0 + r] > step) lmask |= 1u << r;
        __syncthreads();
        for (int e = tid; e < RN * D; e += 1024) {
            const int r = e / D, d = e - r * D; const long row = i0 + r;
            float dz = 0.f, dbp = 0.f;
            if ((lmask >> r) & 1u) {
                const float beta = hc[row * hc_ld + A + d], z = Zs[row * D + d], dx = dXZ[row * D + d];
                dz = dZ_out[row * D + d] + dx * beta;
                dbp = dx * z * beta * (1.f - beta);
            }
            s_dz[e] = dz;
            if (blockIdx.y == 0 && r < rn) { DZ[row * D + d] = dz; dhc[row * dhc_ld + A + d] = dbp; if (dhcb) dhcb[row * dhc_ld + A + d] = (__bf16)dbp; }
        }
        __syncthreads();
        if (l < L) {
            float part[RN];
#pragma unroll
            for (int r = 0; r < RN; ++r) part[r] = 0.f;
            const TA* a = ann + ((long)b * L + l) * D;
            // 16 (fp32) / 8 (bf16) bytes per lane, two loads in flight (D % 4 == 0: the split kernels' launch condition); the 4-byte form
            // walked D / 64 dependent trips: 27.7 us for C4's 65 MB
            for (int d0 = lane * 4; d0 < D; d0 += 512) {
                const int d1 = d0 + 256;
                const float4 av0 = ld_ann4<TA>(a + d0);
                const float4 av1 = d1 < D ? ld_ann4<TA>(a + d1) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int r = 0; r < RN; ++r) {
                    const float4 z0 = *reinterpret_cast<const float4*>(s_dz + r * D + d0);
                    float p = part[r];
                    p = fmaf(av0.x, z0.x, p); p = fmaf(av0.y, z0.y, p); p = fmaf(av0.z, z0.z, p); p = fmaf(av0.w, z0.w, p);
                    if (d1 < D) {
                        const float4 z1 = *reinterpret_cast<const float4*>(s_dz + r * D + d1);
                        p = fmaf(av1.x, z1.x, p); p = fmaf(av1.y, z1.y, p); p = fmaf(av1.z, z1.z, p); p = fmaf(av1.w, z1.w, p);
                    }
                    part[r] = p;
                }
            }
#pragma unroll
            for (int r = 0; r < RN; ++r) {
                const float sdot = wave_sum(part[r]);
                if (lane == 0 && r < rn)
                    da[(long)(i0 + r) * L + l] = ((lmask >> r) & 1u) ? sdot + (dalphas_ext ? dalphas_ext[((long)(i0 + r) * T1 + step) * L + l] : 0.f) : 0.f;
            }
        }
    }
}

constexpr int ATTB_KCH = 32;
template <int RN>
__global__ __launch_bounds__(1024) void attention_bwd_tanh_kernel(const float* __restrict__ U, const float* __restrict__ hc, int hc_ld, const float* __restrict__ wf,
        const int* __restrict__ lengths, int step, const float* __restrict__ alphas, int T1, const float* __restrict__ da, float* __restrict__ dhc,
        int dhc_ld, float* __restrict__ dU, float* __restrict__ dwf_part, int R, int L, int A, __bf16* __restrict__ dhcb) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_ds = sm;                               // [RN][L]
    float* s_q = s_ds + RN * L;                     // [RN][KCH]
    float* s_red = s_q + RN * ATTB_KCH;             // [32 lanes][RN + 1][KCH]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kk = tid & 31, lg = tid >> 5, k = blockIdx.y * ATTB_KCH + kk;
    const float scale = 1.0f / sqrtf((float)L);
    const float w = k < A ? wf[k] : 0.f;
    float dw_total = 0.f;
    for (int r0 = 0; r0 < R; r0 += RN) {
        const int rn = min(RN, R - r0), i0 = b * R + r0;
        unsigned lmask = 0;
        for (int r = 0; r < rn; ++r) if (lengths[i0 + r] > step) lmask |= 1u << r;
        __syncthreads();
        // softmax backward: ds = alpha * (dalpha - sum alpha * dalpha) * L^-1/2, wave per row; dead rows: 0
        for (int r = wave; r < RN; r += 16) {
            const bool livr = (lmask >> r) & 1u;
            float dot = 0.f;
            if (livr) for (int l = lane; l < L; l += 64) dot += alphas[((long)(i0 + r) * T1 + step) * L + l] * da[(long)(i0 + r) * L + l];
            dot = wave_sum(dot);
            for (int l = lane; l < L; l += 64)
                s_ds[r * L + l] = livr ? alphas[((long)(i0 + r) * T1 + step) * L + l] * (da[(long)(i0 + r) * L + l] - dot) * scale : 0.f;
        }
        for (int e = tid; e < RN * ATTB_KCH; e += 1024) {
            const int r = e / ATTB_KCH, k2 = blockIdx.y * ATTB_KCH + (e - r * ATTB_KCH);
            s_q[e] = (r < rn && k2 < A) ? hc[(long)(i0 + r) * hc_ld + k2] : 0.f;
        }
        __syncthreads();
        float dq[RN], dw = 0.f;
#pragma unroll
        for (int r = 0; r < RN; ++r) dq[r] = 0.f;
        if (k < A && lmask) {
            for (int l = lg; l < L; l += 32) {
                const long uo = ((long)b * L + l) * A + k;
                const float uv = U[uo];
                float du = 0.f;
#pragma unroll
                for (int r = 0; r < RN; ++r) {
                    const float th = fast_tanh(uv + s_q[r * ATTB_KCH + kk]);
                    const float ds = s_ds[r * L + l];
                    const float dp = ds * w * (1.f - th * th);
                    dq[r] += dp; du += dp; dw = fmaf(ds, th, dw);
                }
                dU[uo] += du;                        // (b, k) belongs to this block alone: plain read-modify-write, fixed order
            }
        }
#pragma unroll
        for (int r = 0; r < RN; ++r) s_red[(lg * (RN + 1) + r) * ATTB_KCH + kk] = dq[r];
        s_red[(lg * (RN + 1) + RN) * ATTB_KCH + kk] = dw;
        __syncthreads();
        for (int e = tid; e < (RN + 1) * ATTB_KCH; e += 1024) {
            const int r = e / ATTB_KCH, k2 = e - r * ATTB_KCH, kg = blockIdx.y * ATTB_KCH + k2;
            float sacc = 0.f;
            for (int g2 = 0; g2 < 32; ++g2) sacc += s_red[(g2 * (RN + 1) + r) * ATTB_KCH + k2];
            if (kg < A) {
                if (r < RN) { if (r < rn) { dhc[(long)(i0 + r) * dhc_ld + kg] = sacc; if (dhcb) dhcb[(long)(i0 + r) * dhc_ld + kg] = (__bf16)sacc; } }
                else s_q[k2] = sacc;                 // dw of this pass (s_q is free until the next pass reloads it)
            }
        }
        __syncthreads();
        if (tid < ATTB_KCH) dw_total += s_q[tid];
    }
    if (tid < ATTB_KCH && k < A) dwf_part[(long)b * A + k] += dw_total;
}

// d ann[b, l, :] += sum over the image's captions r and steps t of alpha[i, t, l] * DZ[t, i, :]  (i = b*R + r):
// the context path of the attention backward, summed over time.  One block per (image, 256-wide slice of D).  The
// image's alphas are staged in LDS once, rows zero-padded to a multiple of 4 floats so that a location slab is NQ
// 16-byte LDS reads with a compile-time trip count (a guard per location serialised every FMA behind its own LDS
// read: 390 us); the (r, t) loop has no barrier and keeps 8 DZ loads in flight.  Fixed order: deterministic.
template <int NQ>
__global__ __launch_bounds__(256) void dann_from_context_kernel(const float* __restrict__ alphas, const float* __restrict__ DZ, const int* __restrict__ lengths,
                                         float* __restrict__ dann, int accumulate, int R, int N, int T1, int L, int D) {
    extern __shared__ float4 s_a4[];  // [R*T1][Lq] float4: alphas of this image, rows padded with zeros
    const int b = blockIdx.x, d = blockIdx.y * blockDim.x + threadIdx.x;
    const int Lq = ((L + 3) / 4 + NQ - 1) / NQ * NQ, Lp = Lq * 4;          // padded row: whole slabs
    float* s_a = reinterpret_cast<float*>(s_a4);
    for (int idx = threadIdx.x; idx < R * T1 * Lp; idx += blockDim.x) {
        const int row = idx / Lp, l = idx - row * Lp;
        s_a[idx] = l < L ? alphas[((long)b * R * T1 + row) * L + l] : 0.f;
    }
    __syncthreads();
    if (d >= D) return;
    for (int q0 = 0; q0 < Lq; q0 += NQ) {
        float acc[NQ * 4];
#pragma unroll
        for (int j = 0; j < NQ * 4; ++j) acc[j] = 0.f;
        for (int r = 0; r < R; ++r) {
            const int i = b * R + r; const int len = min(lengths[i], T1);
            const float4* sa = s_a4 + (long)r * T1 * Lq + q0;
            for (int t0 = 0; t0 < len; t0 += 8) {             // 8 independent DZ loads in flight, then 8 x NQ*4 FMAs
                float dz[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) dz[u] = (t0 + u < len) ? DZ[((long)(t0 + u) * N + i) * D + d] : 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int tt = (t0 + u < T1) ? t0 + u : T1 - 1;          // stay inside the staged rows; dz is 0 past len
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const float4 a4 = sa[tt * Lq + q];
                        acc[4 * q] = fmaf(a4.x, dz[u], acc[4 * q]); acc[4 * q + 1] = fmaf(a4.y, dz[u], acc[4 * q + 1]);
                        acc[4 * q + 2] = fmaf(a4.z, dz[u], acc[4 * q + 2]); acc[4 * q + 3] = fmaf(a4.w, dz[u], acc[4 * q + 3]);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NQ * 4; ++j) {
            const int l = q0 * 4 + j;
            if (l < L) {
                float* p = dann + ((long)b * L + l) * D + d;
                *p = accumulate ? (*p + acc[j]) : acc[j];
            }
        }
    }
}

// dann[b,l,d] += dmean[b,d] / L  (backward of the spatial mean in InitLSTM)
__global__ void dann_add_mean_kernel(float* __restrict__ dann, const float* __restrict__ dmean, int L, int D, long total) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    long b = e / ((long)L * D); int d = (int)(e % D);
    dann[e] += dmean[b * D + d] / (float)L;
}

// ------------------------------------------------------------------ losses
// Label-smoothed cross entropy over packed rows (util.py:105-112).  One block per packed token row:
// online log-sum-exp, row argmax (accuracy, model.py:596-597) and the row loss; ce_finish_kernel reduces
// loss_rows in a fixed order.  The gradient is a second kernel so that autograd's grad_output scales it.
// VEC = 4: 16-byte loads, four of them in flight per thread (V % 4 == 0, i.e. rows stay 16-byte aligned); VEC = 1: any V.
// (The first version read 4 bytes per lane per trip round the online-softmax recurrence: 144 us for the 54 MB of C2's logits.)
template <int VEC>
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, const int* __restrict__ target, int V,
                                                      float smoothing, float* __restrict__ lse_rows,
                                                      float* __restrict__ loss_rows, int* __restrict__ correct_rows) {
    const int p = blockIdx.x, tid = threadIdx.x;
    const float* x = logits + (long)p * V;
    __shared__ float s_m[4], s_s[4], s_t[4], s_b[4]; __shared__ int s_i[4];
    float mx = -INFINITY, sum = 0.f, tot = 0.f; int bi = 0x7fffffff; float bv = -INFINITY;
    if (VEC == 4) {
        constexpr int U = 4;
        const int V4 = V >> 2;
        for (int v0 = tid; v0 < V4; v0 += 256 * U) {
            float4 q[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int v = v0 + 256 * u;
                q[u] = v < V4 ? reinterpret_cast<const float4*>(x)[v] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int v = v0 + 256 * u;
                if (v >= V4) continue;
                const float e[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
                float m4 = fmaxf(fmaxf(e[0], e[1]), fmaxf(e[2], e[3]));
#pragma unroll
                for (int i = 0; i < 4; ++i) {           // ascending index: the first maximum wins, as torch.argmax
                    tot += e[i];
                    if (e[i] > bv) { bv = e[i]; bi = 4 * v + i; }
                }
                if (m4 > mx) { sum *= __expf(mx - m4); mx = m4; }          // exp(-inf) = 0 on the first trip
                if (mx != -INFINITY) sum += (__expf(e[0] - mx) + __expf(e[1] - mx)) + (__expf(e[2] - mx) + __expf(e[3] - mx));
            }
        }
    } else {
        for (int v = tid; v < V; v += 256) {
            float xv = x[v];
            tot += xv;
            if (xv > bv || (xv == bv && v < bi)) { bv = xv; bi = v; }
            if (xv > mx) { sum = sum * __expf(mx - xv) + 1.f; mx = xv; } else sum += __expf(xv - mx);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        float om = __shfl_xor(mx, o, 64), os = __shfl_xor(sum, o, 64);
        float nm = fmaxf(mx, om);
        float fa = (mx == nm) ? 1.f : __expf(mx - nm), fb = (om == nm) ? 1.f : __expf(om - nm);   // (-inf) - (-inf) guard
        sum = sum * fa + os * fb; mx = nm;
        tot += __shfl_xor(tot, o, 64);
        float ob = __shfl_xor(bv, o, 64); int oi = __shfl_xor(bi, o, 64);
        if (ob > bv || (ob == bv && oi < bi)) { bv = ob; bi = oi; }
    }
    const int w = tid >> 6;
    if ((tid & 63) == 0) { s_m[w] = mx; s_s[w] = sum; s_t[w] = tot; s_b[w] = bv; s_i[w] = bi; }
    __syncthreads();
    if (tid == 0) {
        float M = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
        float S = 0.f, Tt = 0.f, Bv = -INFINITY; int Bi = 0x7fffffff;
        for (int k = 0; k < 4; ++k) {
            S += (s_m[k] == M) ? s_s[k] : s_s[k] * __expf(s_m[k] - M); Tt += s_t[k];
            if (s_b[k] > Bv || (s_b[k] == Bv && s_i[k] < Bi)) { Bv = s_b[k]; Bi = s_i[k]; }
        }
        const float lse = M + __logf(S);
        const int t = target[p];
        float nll = lse - x[t];
        float smooth = lse - Tt / (float)V;
        lse_rows[p] = lse;
        loss_rows[p] = (1.f - smoothing) * nll + smoothing * smooth;
        correct_rows[p] = (Bi == t) ? 1 : 0;
    }
}
// dlogits[p, v] = g/P * (softmax - smoothing/V - (1-smoothing)[v == target])
template <int VEC>
__global__ __launch_bounds__(256) void ce_grad_kernel(const float* __restrict__ logits, const int* __restrict__ target, const float* __restrict__ lse_rows,
                                                      int V, float smoothing, float inv_rows, const float* __restrict__ gscale, float* __restrict__ dlogits) {
    const int p = blockIdx.x;
    const float* x = logits + (long)p * V; float* g = dlogits + (long)p * V;
    const float lse = lse_rows[p], sv = smoothing / (float)V, conf = 1.f - smoothing;
    const float sc = inv_rows * (gscale ? gscale[0] : 1.f);
    const int t = target[p];
    if (VEC == 4) {
        constexpr int U = 4;
        const int V4 = V >> 2;
        for (int v0 = threadIdx.x; v0 < V4; v0 += 256 * U) {
            float4 q[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int v = v0 + 256 * u; if (v < V4) q[u] = reinterpret_cast<const float4*>(x)[v]; }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int v = v0 + 256 * u;
                if (v >= V4) continue;
                float4 o;
                o.x = (__expf(q[u].x - lse) - sv - ((4 * v == t) ? conf : 0.f)) * sc;
                o.y = (__expf(q[u].y - lse) - sv - ((4 * v + 1 == t) ? conf : 0.f)) * sc;
                o.z = (__expf(q[u].z - lse) - sv - ((4 * v + 2 == t) ? conf : 0.f)) * sc;
                o.w = (__expf(q[u].w - lse) - sv - ((4 * v + 3 == t) ? conf : 0.f)) * sc;
                reinterpret_cast<float4*>(g)[v] = o;
            }
        }
    } else {
        for (int v = threadIdx.x; v < V; v += 256) g[v] = (__expf(x[v] - lse) - sv - ((v == t) ? conf : 0.f)) * sc;
    }
}

// out[0] = mean(loss_rows), out[1] = #correct / P   (single block, fixed-order tree)
__global__ void ce_finish_kernel(const float* __restrict__ loss_rows, const int* __restrict__ correct_rows, int P, float* __restrict__ out) {
    __shared__ double s_l[256]; __shared__ int s_c[256];
    double s = 0.0; int c = 0;
    for (int p = threadIdx.x; p < P; p += 256) { s += (double)loss_rows[p]; c += correct_rows[p]; }
    s_l[threadIdx.x] = s; s_c[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) { s_l[threadIdx.x] += s_l[threadIdx.x + o]; s_c[threadIdx.x] += s_c[threadIdx.x + o]; } __syncthreads(); }
    if (threadIdx.x == 0) { out[0] = (float)(s_l[0] / (double)P); out[1] = (float)s_c[0] / (float)P; }
}

// Doubly-stochastic term (model.py:594).  asum[i,l] = sum_t alphas[i,t,l]; part[block] = sum (1-asum)^2
__global__ void ds_rows_kernel(const float* __restrict__ alphas, float* __restrict__ asum, float* __restrict__ part, int N, int T1, int L) {
    __shared__ float s_p[256];
    long e = (long)blockIdx.x * 256 + threadIdx.x;
    float v = 0.f;
    if (e < (long)N * L) {
        long i = e / L; int l = (int)(e % L);
        float s = 0.f;
        for (int t = 0; t < T1; ++t) s += alphas[(i * T1 + t) * L + l];
        asum[e] = s;
        v = (1.f - s) * (1.f - s);
    }
    s_p[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) s_p[threadIdx.x] += s_p[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) part[blockIdx.x] = s_p[0];
}
__global__ void ds_finish_kernel(const float* __restrict__ part, int nparts, float gamma, long count, float* __restrict__ out) {
    __shared__ double s_p[256];
    double s = 0.0;
    for (int p = threadIdx.x; p < nparts; p += 256) s += (double)part[p];
    s_p[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) s_p[threadIdx.x] += s_p[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = gamma * (float)(s_p[0] / (double)count);
}
// dalphas[i,t,l] = gscale[0] * gamma * 2 (asum[i,l] - 1) / (N L)   for every t
__global__ void ds_grad_kernel(const float* __restrict__ asum, const float* __restrict__ gscale, float gamma, float* __restrict__ dalphas, int N, int T1, int L) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)N * T1 * L) return;
    long i = e / ((long)T1 * L); int l = (int)(e % L);
    float g = gscale ? gscale[0] : 1.f;
    dalphas[e] = g * gamma * 2.f * (asum[i * L + l] - 1.f) / (float)((long)N * L);
}

// ------------------------------------------------------------------ reductions for bias grads
// out[c] (+)= sum_r x[r*ld + c], rows split over grid.y into fixed chunks, partials reduced in order.
__global__ void colsum_part_kernel(const float* __restrict__ x, long ld, int rows, int cols, int rows_per, float* __restrict__ part) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    int r0 = blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += x[(long)r * ld + c];
    part[(long)blockIdx.y * cols + c] = s;
}
// the same, four columns per thread (16-byte loads; x 16-byte aligned, ld % 4 == 0, cols % 4 == 0) and four independent row
// accumulators so that four loads are in flight per thread; combined in a fixed order
__global__ void colsum_part4_kernel(const float* __restrict__ x, long ld, int rows, int cols4, int rows_per, float* __restrict__ part) {
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c4 >= cols4) return;
    const int r0 = blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
    float4 a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    int r = r0;
    for (; r + 3 < r1; r += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 v = *reinterpret_cast<const float4*>(x + (long)(r + u) * ld + 4 * c4);
            a[u].x += v.x; a[u].y += v.y; a[u].z += v.z; a[u].w += v.w;
        }
    }
    for (; r < r1; ++r) { const float4 v = *reinterpret_cast<const float4*>(x + (long)r * ld + 4 * c4); a[0].x += v.x; a[0].y += v.y; a[0].z += v.z; a[0].w += v.w; }
    float4 s;
    s.x = (a[0].x + a[1].x) + (a[2].x + a[3].x); s.y = (a[0].y + a[1].y) + (a[2].y + a[3].y);
    s.z = (a[0].z + a[1].z) + (a[2].z + a[3].z); s.w = (a[0].w + a[1].w) + (a[2].w + a[3].w);
    *reinterpret_cast<float4*>(part + (long)blockIdx.y * cols4 * 4 + 4 * c4) = s;
}
__global__ void colsum_finish_kernel(const float* __restrict__ part, int nparts, int cols, float* __restrict__ out, int accumulate, float scale) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += part[(long)p * cols + c];
    s *= scale;
    out[c] = accumulate ? out[c] + s : s;
}

// embedding gradient, deterministic (no floating-point atomics: the result does not depend on scheduling).  The padding row
// and rows without a token (-1) get zeros.
constexpr int EMB_LIST = 1024;
// Four small launches:
// count tokens per vocabulary row (integer atomics: the counts do not depend on the order), exclusive scan, scatter the token
// positions into per-row segments (unordered), then one block per row SORTS its segment and adds the dY rows in increasing
// position -- the summation order is again fixed.  Rows with more than EMB_LIST tokens take the scanning kernel's path.
__global__ void embedding_count_kernel(const int* __restrict__ tok, int rows, int V, int padding_idx, int* __restrict__ count) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    int v = tok[r];
    if (v >= 0 && v < V && v != padding_idx) atomicAdd(&count[v], 1);
}
__global__ __launch_bounds__(1024) void embedding_scan_kernel(const int* __restrict__ count, int V, int* __restrict__ offset, int* __restrict__ cursor) {
    __shared__ int s_part[1024];
    const int tid = threadIdx.x, per = (V + 1023) / 1024, b0 = tid * per, b1 = min(V, b0 + per);
    int s = 0;
    for (int v = b0; v < b1; ++v) s += count[v];
    s_part[tid] = s;
    __syncthreads();
    if (tid == 0) { int run = 0; for (int i = 0; i < 1024; ++i) { int t = s_part[i]; s_part[i] = run; run += t; } offset[V] = run; }
    __syncthreads();
    int run = s_part[tid];
    for (int v = b0; v < b1; ++v) { offset[v] = run; cursor[v] = 0; run += count[v]; }
}
__global__ void embedding_place_kernel(const int* __restrict__ tok, int rows, int V, int padding_idx, const int* __restrict__ offset, int* __restrict__ cursor,
                                       int* __restrict__ list) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    int v = tok[r];
    if (v >= 0 && v < V && v != padding_idx) list[offset[v] + atomicAdd(&cursor[v], 1)] = r;
}
__global__ __launch_bounds__(256) void embedding_sum_kernel(const float* __restrict__ dY, const int* __restrict__ tok, const int* __restrict__ offset,
                                                            const int* __restrict__ list, float* __restrict__ dE, int rows, int width, int padding_idx) {
    __shared__ int s_list[EMB_LIST];
    const int v = blockIdx.x, tid = threadIdx.x;
    float* out = dE + (long)v * width;
    const int o0 = offset[v], n = offset[v + 1] - o0;
    if (n == 0 || v == padding_idx) { for (int c = tid; c < width; c += 256) out[c] = 0.f; return; }
    if (n > EMB_LIST) {          // very frequent token: ordered compaction by scanning the token array, flushed whenever the
                                 // next 256-token chunk might not fit the list any more
        __shared__ int s_wcnt[4]; __shared__ int s_total;
        const int lane = tid & 63, wave = tid >> 6;
        for (int c = tid; c < width; c += 256) out[c] = 0.f;
        if (tid == 0) s_total = 0;
        __syncthreads();
        for (int base = 0;; base += 256) {           // one extra pass after the last chunk flushes what is left
            const bool last = base >= rows;
            if (last || s_total + 256 > EMB_LIST) {
                const int cnt = s_total;
                for (int c = tid; c < width; c += 256) { float acc = out[c]; for (int i = 0; i < cnt; ++i) acc += dY[(long)s_list[i] * width + c]; out[c] = acc; }
                __syncthreads();
                if (tid == 0) s_total = 0;
                __syncthreads();
                if (last) break;
            }
            const int r = base + tid;
            const bool hit = (r < rows) && (tok[r] == v);
            const unsigned long long mk = __ballot(hit);
            if (lane == 0) s_wcnt[wave] = __popcll(mk);
            __syncthreads();
            int off = s_total;
            for (int w = 0; w < wave; ++w) off += s_wcnt[w];
            const int pos = off + __popcll(mk & ((1ull << lane) - 1ull));
            if (hit) s_list[pos] = r;
            __syncthreads();
            if (tid == 0) s_total += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
            __syncthreads();
        }
        return;
    }
    // bitonic sort of the segment (padded with INT_MAX to a power of two) in LDS
    int np2 = 1; while (np2 < n) np2 <<= 1;
    for (int i = tid; i < np2; i += 256) s_list[i] = i < n ? list[o0 + i] : 0x7fffffff;
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const int a = s_list[i], b = s_list[ixj];
                    if (((i & k) == 0) ? (a > b) : (a < b)) { s_list[i] = b; s_list[ixj] = a; }
                }
            }
            __syncthreads();
        }
    for (int c = tid; c < width; c += 256) {
        float acc = 0.f;
        for (int i = 0; i < n; ++i) acc += dY[(long)s_list[i] * width + c];
        out[c] = acc;
    }
}

// nn.Embedding(max_norm=...) (model.py:161): rows that are looked up and whose L2 norm exceeds max_norm are rescaled
// in place by max_norm / (norm + 1e-7), as torch.embedding_renorm_ does.  flags marks the looked-up rows first so
// that a row used by several tokens is rescaled exactly once.
__global__ void embedding_mark_kernel(const int* __restrict__ tok, int n, int* __restrict__ flags, int V) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { int t = tok[i]; if (t >= 0 && t < V) flags[t] = 1; }
}
__global__ __launch_bounds__(64) void embedding_renorm_kernel(float* __restrict__ table, int* __restrict__ flags, int width, float max_norm) {
    const int v = blockIdx.x, lane = threadIdx.x;
    if (!flags[v]) return;
    float* row = table + (long)v * width;
    float s = 0.f;
    for (int c = lane; c < width; c += 64) s += row[c] * row[c];
    s = wave_sum(s);
    const float norm = sqrtf(s);
    if (norm > max_norm) { const float sc = max_norm / (norm + 1e-7f); for (int c = lane; c < width; c += 64) row[c] *= sc; }
    if (lane == 0) flags[v] = 0;
}

// out[i] = a[i] + b[i]
__global__ void sigmoid_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dpre, long n) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) { const float v = y[e]; dpre[e] = dy[e] * v * (1.f - v); }
}
// g[e] *= 1 - u[e]^2 : backward of tanh from its output
__global__ void mul_dtanh_kernel(float* __restrict__ g, const float* __restrict__ u, long n) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) { const float t = u[e]; g[e] *= 1.f - t * t; }
}
__global__ void add_kernel(float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ b, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}
// y[r, :] (+)= x[map[r], :]   (map < 0 -> zero contribution)
__global__ void scatter_rows_kernel(const float* __restrict__ src, const int* __restrict__ prow, float* __restrict__ dst, int rows, int width) {
    int r = blockIdx.x;
    if (r >= rows) return;
    int p = prow[r];
    for (int c = threadIdx.x; c < width; c += blockDim.x) dst[(long)r * width + c] = (p >= 0) ? src[(long)p * width + c] : 0.f;
}


// ------------------------------------------------------------------ inference (beam search, model.py:329-343, 351-359)
// scores[r, v] = log_softmax(logits[r, :] / T)[v] (+ parent[r]); masked token ids become -inf.  One block per beam row.
__global__ __launch_bounds__(256) void beam_scores_kernel(const float* __restrict__ logits, int V, float inv_temp,
                                                          const int* __restrict__ masked, int n_masked,
                                                          const float* __restrict__ parent, float* __restrict__ scores) {
    const int r = blockIdx.x, tid = threadIdx.x;
    const float* x = logits + (long)r * V;
    __shared__ float s_m[4], s_s[4];
    float mx = -INFINITY, sum = 0.f;
    for (int v = tid; v < V; v += 256) {
        float xv = x[v] * inv_temp;
        if (xv > mx) { sum = sum * __expf(mx - xv) + 1.f; mx = xv; } else sum += __expf(xv - mx);
    }
    for (int o = 32; o > 0; o >>= 1) {
        float om = __shfl_xor(mx, o, 64), os = __shfl_xor(sum, o, 64);
        float nm = fmaxf(mx, om);
        float fa = (mx == nm) ? 1.f : __expf(mx - nm), fb = (om == nm) ? 1.f : __expf(om - nm);
        sum = sum * fa + os * fb; mx = nm;
    }
    if ((tid & 63) == 0) { s_m[tid >> 6] = mx; s_s[tid >> 6] = sum; }
    __syncthreads();
    float M = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3])), S = 0.f;
    for (int k = 0; k < 4; ++k) S += (s_m[k] == M) ? s_s[k] : s_s[k] * __expf(s_m[k] - M);
    const float lse = M + __logf(S), add = parent ? parent[r] : 0.f;
    float* o = scores + (long)r * V;
    for (int v = tid; v < V; v += 256) o[v] = x[v] * inv_temp - lse + add;
    __syncthreads();
    for (int k = tid; k < n_masked; k += 256) { int id = masked[k]; if (id >= 0 && id < V) o[id] = -INFINITY; }
}

// top-k of a flat array, descending, ties to the lowest index (torch.topk on the flattened beam scores, model.py:359).
// Single block; `work` is a scratch copy of x that gets the winners knocked out.
__global__ __launch_bounds__(1024) void topk_kernel(const float* __restrict__ x, float* __restrict__ work, long n, int k,
                                                    float* __restrict__ values, int* __restrict__ indices) {
    __shared__ float s_v[16]; __shared__ long s_i[16];
    const int tid = threadIdx.x;
    for (long i = tid; i < n; i += 1024) work[i] = x[i];
    __syncthreads();
    for (int it = 0; it < k; ++it) {
        float bv = -INFINITY; long bi = 0x7fffffffffffffffL;
        for (long i = tid; i < n; i += 1024) { float v = work[i]; if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; } }
        for (int o = 32; o > 0; o >>= 1) {
            float ov = __shfl_xor(bv, o, 64); long oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if ((tid & 63) == 0) { s_v[tid >> 6] = bv; s_i[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 16; ++w) if (s_v[w] > bv || (s_v[w] == bv && s_i[w] < bi)) { bv = s_v[w]; bi = s_i[w]; }
            values[it] = bv; indices[it] = (int)bi;
            if (bi < n) work[bi] = -INFINITY;
        }
        __syncthreads();
    }
}


// ------------------------------------------------------------------ batched beam search (all images of a batch at once)
// Rows are laid out (B, K): image b owns rows b*K .. b*K + K - 1, its klive[b] live hypotheses first.
// InitLSTM over the K expanded copies of ONE image's annotation (model.py:265-269), for every image: the raw reshape of
// F3 acts on each image's own (K, 2*layers*n) buffer.  h, c: (layers, B*K, n).
__global__ void init_expand_images_kernel(const float* __restrict__ init_img, float* __restrict__ h0, float* __restrict__ c0, int B, int K, int n,
                                          int layers) {
    const long per = 2L * layers * K * n;                      // flat elements of one image
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= per * B) return;
    const int b = (int)(e / per); const long f = e - (long)b * per;
    const long w2 = 2L * layers * n, slab_sz = (long)K * n;
    const float v = init_img[(long)b * w2 + f % w2];           // every one of the K rows is the image's init vector
    const long slab = f / slab_sz, within = f - slab * slab_sz;
    const long row = within / n, col = within - row * n;
    const long N = (long)B * K;
    if (slab < layers) h0[(slab * N + (long)b * K + row) * n + col] = v;
    else c0[((slab - layers) * N + (long)b * K + row) * n + col] = v;
}
// live[i] = 1 for the first klive[b] rows of image b (the kernels' `lengths[i] > step` predicate with step = 0)
__global__ void beam_live_kernel(const int* __restrict__ klive, int* __restrict__ live, int B, int K) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * K) live[i] = (i % K) < klive[i / K] ? 1 : 0;
}
// torch.topk of the flattened live scores of every image (model.py:343 at step 0: row 0 only; model.py:359 afterwards),
// descending, ties to the lowest index.  One block per image; `work` (B, K*V) is scratch.
__global__ __launch_bounds__(1024) void beam_topk_kernel(const float* __restrict__ scores, float* __restrict__ work, const int* __restrict__ klive,
                                                         int K, int V, int first_step, float* __restrict__ values, int* __restrict__ indices) {
    __shared__ float s_v[16]; __shared__ long s_i[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int k = klive[b];
    if (k <= 0) return;
    const long n = first_step ? V : (long)k * V;
    const float* x = scores + (long)b * K * V; float* wk = work + (long)b * K * V;
    for (long i = tid; i < n; i += 1024) wk[i] = x[i];
    __syncthreads();
    for (int it = 0; it < k; ++it) {
        float bv = -INFINITY; long bi = 0x7fffffffffffffffL;
        for (long i = tid; i < n; i += 1024) { float v = wk[i]; if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; } }
        for (int o = 32; o > 0; o >>= 1) {
            float ov = __shfl_xor(bv, o, 64); long oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if ((tid & 63) == 0) { s_v[tid >> 6] = bv; s_i[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 16; ++w) if (s_v[w] > bv || (s_v[w] == bv && s_i[w] < bi)) { bv = s_v[w]; bi = s_i[w]; }
            values[b * K + it] = bv; indices[b * K + it] = (int)bi;
            if (bi < n) wk[bi] = -INFINITY;
        }
        __syncthreads();
    }
}
// ---- sampled continuation of the beams (model.py:360-379) for all images at once.
// torch.multinomial(p, k) without replacement draws an ordered sample from the Plackett-Luce distribution of p; so does
// "top-k of log p + Gumbel noise" (Gumbel-top-k), which needs no sequential renormalisation and reuses the per-image top-k
// kernel.  The draws are therefore not torch's: they come from a counter-based hash of (seed, step, row, candidate), or from a
// caller-supplied table of Gumbel variates (tests).
__device__ __forceinline__ float hash_uniform(unsigned long long seed, unsigned long long stream, unsigned long long idx) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (stream + 1) + idx * 0xD1342543DE82EF95ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return ((float)(z >> 40) + 0.5f) * (1.0f / 16777216.0f);       // 24 bits -> (0, 1)
}
__device__ __forceinline__ float gumbel_of(float u) { return -__logf(-__logf(u)); }
// keys[i, :] for live row i = (b, j):  method 1 ("multinomial", model.py:360-364): log softmax_v(20 * s / step) + G
//                                      method 2 ("topk", model.py:365-379): s / step + G for the row's sample_topk best
//                                      candidates (torch.topk order; the Gumbel variate belongs to the candidate's RANK t), else -inf
// gumbel: optional table (B*K rows, gstride floats per row) of this step: method 1 reads [v], method 2 reads [t].
__global__ __launch_bounds__(256) void beam_sample_keys_kernel(const float* __restrict__ scores, const int* __restrict__ klive, int K, int V, int method,
                                                               int sample_topk, float step, unsigned long long seed, unsigned long long stream,
                                                               const float* __restrict__ gumbel, int gstride, float* __restrict__ keys) {
    const int i = blockIdx.x, b = i / K, j = i - b * K, tid = threadIdx.x;
    if (j >= klive[b]) return;
    const float* s = scores + (long)i * V; float* key = keys + (long)i * V;
    __shared__ float s_v[4]; __shared__ int s_i[4]; __shared__ float s_b; __shared__ int s_bi;
    if (method == 1) {
        const float sc = 20.0f / step;
        float mx = -INFINITY;
        for (int v = tid; v < V; v += 256) mx = fmaxf(mx, s[v] * sc);
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if ((tid & 63) == 0) s_v[tid >> 6] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(s_v[0], s_v[1]), fmaxf(s_v[2], s_v[3]));
        __syncthreads();
        float sum = 0.f;
        for (int v = tid; v < V; v += 256) sum += __expf(s[v] * sc - mx);
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if ((tid & 63) == 0) s_v[tid >> 6] = sum;
        __syncthreads();
        const float lse = mx + __logf(s_v[0] + s_v[1] + s_v[2] + s_v[3]);
        for (int v = tid; v < V; v += 256) {
            const float g = gumbel ? gumbel[(long)i * gstride + v] : gumbel_of(hash_uniform(seed, stream, (unsigned long long)i * V + v));
            key[v] = s[v] * sc - lse + g;                 // -inf scores (masked tokens) stay -inf: probability zero
        }
    } else {
        for (int v = tid; v < V; v += 256) key[v] = -INFINITY;
        __syncthreads();
        // the row's best sample_topk candidates, one per pass: descending value, ties to the lowest index
        float lastv = INFINITY; int lasti = -1;
        for (int t = 0; t < sample_topk; ++t) {
            float bv = -INFINITY; int bi = 0x7fffffff;
            for (int v = tid; v < V; v += 256) {
                const float x = s[v];
                const bool after = x < lastv || (x == lastv && v > lasti);      // strictly after the previous pick in (value desc, index asc) order
                if (after && (x > bv || (x == bv && v < bi))) { bv = x; bi = v; }
            }
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if ((tid & 63) == 0) { s_v[tid >> 6] = bv; s_i[tid >> 6] = bi; }
            __syncthreads();
            if (tid == 0) {
                for (int w = 1; w < 4; ++w) if (s_v[w] > bv || (s_v[w] == bv && s_i[w] < bi)) { bv = s_v[w]; bi = s_i[w]; }
                s_b = bv; s_bi = bi;
                if (bi < V) {
                    const float g = gumbel ? gumbel[(long)i * gstride + t] : gumbel_of(hash_uniform(seed, stream, (unsigned long long)i * V + t));
                    key[bi] = bv / step + g;
                }
            }
            __syncthreads();
            lastv = s_b; lasti = s_bi;
            if (lasti >= V) break;                        // fewer than sample_topk finite candidates
        }
    }
}
// after the top-k over the keys: the kept hypotheses carry their SCORES (model.py:382 top_scores = seq_scores.reshape(-1)[pred_idx])
__global__ void beam_take_scores_kernel(const float* __restrict__ scores, const int* __restrict__ inds, const int* __restrict__ klive, int B, int K, int V,
                                        float* __restrict__ values) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * K) return;
    const int b = i / K, j = i - b * K;
    if (j < klive[b]) { const int ind = inds[i]; if (ind >= 0 && ind < K * V) values[i] = scores[(long)b * K * V + ind]; }
}
// decoder noise (model.py:322-324): out = N(0, 1) * scale for every layer and live row (0 for dead rows); normals from the table
// or the hash (Box-Muller).  The noise joins h after attention and the gate were taken from the clean state, so it reaches the
// step through the recurrent products only: the caller adds out * W_hh^T to the gate pre-activations.
__global__ void beam_state_noise_kernel(float* __restrict__ out, const int* __restrict__ live, long N, int n, int layers, float scale,
                                        unsigned long long seed, unsigned long long stream, const float* __restrict__ normals) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)layers * N * n) return;
    const long i = (e / n) % N;
    float z = 0.f;
    if (live[i]) {
        if (normals) z = normals[e];
        else {
            const float u1 = hash_uniform(seed, stream, 2ull * e), u2 = hash_uniform(seed, stream, 2ull * e + 1);
            z = sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530718f * u2);
        }
    }
    out[e] = z * scale;
}
// Beam bookkeeping of one step for every image (model.py:343-447), one thread per image, in the reference's order:
// extend each kept hypothesis (parent row, word), record the completed ones (word == END) in index order, drop them from the
// beam, and at the last step record whatever is left.  Writes the next step's inputs: token, parent row (for the back-trace
// and the state gather), parent score; the finished list keeps (step, parent row, raw score, mean of the beam's scores at
// that moment -- what the BAR rescoring needs).
__global__ void beam_update_kernel(const float* __restrict__ values, const int* __restrict__ indices, int* __restrict__ klive, int B, int K, int V,
                                   int step, int last_step, int end_id, int* __restrict__ tok_next, int* __restrict__ prev_next,
                                   float* __restrict__ top_scores, int* __restrict__ gmap, int* __restrict__ fin_count, int* __restrict__ fin_step,
                                   int* __restrict__ fin_row, float* __restrict__ fin_score, float* __restrict__ fin_mean) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int k = klive[b];
    int nk = 0;
    if (k > 0) {
        float mean_all = 0.f;
        for (int j = 0; j < k; ++j) mean_all += values[b * K + j];
        mean_all /= (float)k;
        int fc = fin_count[b];
        // completed hypotheses first, in index order (torch.nonzero(complete))
        for (int j = 0; j < k; ++j) {
            const int ind = indices[b * K + j];
            const int parent = step == 0 ? j : ind / V, tok = step == 0 ? ind : ind - (ind / V) * V;
            if (tok == end_id) {
                fin_step[b * K + fc] = step; fin_row[b * K + fc] = parent; fin_score[b * K + fc] = values[b * K + j]; fin_mean[b * K + fc] = mean_all; ++fc;
            }
        }
        // the incomplete ones continue, order kept
        float rest = 0.f;
        for (int j = 0; j < k; ++j) {
            const int ind = indices[b * K + j];
            const int parent = step == 0 ? j : ind / V, tok = step == 0 ? ind : ind - (ind / V) * V;
            if (tok != end_id) {
                tok_next[b * K + nk] = tok; prev_next[b * K + nk] = parent; top_scores[b * K + nk] = values[b * K + j]; gmap[b * K + nk] = parent;
                rest += values[b * K + j]; ++nk;
            }
        }
        if (step >= last_step && nk > 0) {           // model.py:438-447: cut at max_gen_length
            const float mean_rest = rest / (float)nk;
            for (int j = 0; j < nk; ++j) {
                fin_step[b * K + fc] = step; fin_row[b * K + fc] = prev_next[b * K + j]; fin_score[b * K + fc] = top_scores[b * K + j]; fin_mean[b * K + fc] = mean_rest; ++fc;
            }
            nk = 0;
        }
        fin_count[b] = fc;
    }
    klive[b] = nk;
}
// state of the next step's row j' = this step's new state of its parent row (model.py:366: h, c = h[:, keep], c[:, keep])
__global__ void beam_gather_state_kernel(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ gmap, const int* __restrict__ klive,
                                         int B, int K, int n, int layers) {
    const long N = (long)B * K;
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)layers * N * n) return;
    const long l = e / (N * n); const long rem = e - l * N * n; const long i = rem / n; const int col = (int)(rem - i * n);
    const int b = (int)(i / K), j = (int)(i - (long)b * K);
    if (j < klive[b]) dst[e] = src[(l * N + (long)b * K + gmap[i]) * n + col];
}

}  // namespace sat
