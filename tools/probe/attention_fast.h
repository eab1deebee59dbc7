// Attention step kernels, latency-shaped forms (round 3) for the shapes the BASELINE configs run (D % 8 == 0, A % 64 == 0, A <= 256, L <= 256).
//
// One decode step touches little data per workgroup (C4: 25 KB of annotations per (image, 64-feature slice); C2: 6 KB), so the kernels of
// decoder_kernels.h - the generic forms, kept for every other shape - ran at 0.17 - 0.29 of the HBM peak at C4: each workgroup walked a CHAIN of
// dependent memory round trips (lengths -> scores for the maximum -> scores again for the exponentials -> annotations -> gates), ~1 - 2 us
// each under load, and the backward recomputed the image's dz rows (80 KB of reads) in every one of its 13 location blocks.
// Here every global load a workgroup needs is issued at kernel entry, before the first dependent use: annotation vectors for all the
// locations a thread owns (16 bytes per lane, whole 128-byte row segments per 8 lanes), the score row, the gate row; the softmax runs on
// registers while the annotation loads are in flight.  The backward streams several consecutive locations per wave (C4: 4 blocks per
// image instead of 13: a quarter of the redundant dz prologue, 8 loads in flight per lane).
// Reference arithmetic: SoftAttention.forward, model.py:94-109 (and the gate, model.py:538) - unchanged; only summation orders differ
// from the generic forms (fixed orders: results stay bit-reproducible from run to run).
#pragma once
#include "decoder_kernels.h"

namespace sat {

constexpr int ATTX_LMAX = 256;       // locations: 32 location groups x 8 register slots (context), 4 score values per lane (softmax)

// 8 consecutive features of an annotation row as floats (16 bytes of bf16, or two 16-byte fp32 loads)
template <typename TA> struct Ann8;
template <> struct Ann8<__bf16> {
    typedef __bf16 raw_t __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ raw_t load(const __bf16* p) { return *reinterpret_cast<const raw_t*>(p); }
    static __device__ __forceinline__ raw_t zero() { raw_t z; for (int i = 0; i < 8; ++i) z[i] = (__bf16)0.f; return z; }
    static __device__ __forceinline__ float get(const raw_t& r, int i) { return (float)r[i]; }
};
template <> struct Ann8<float> {
    struct raw_t { float4 a, b; };
    static __device__ __forceinline__ raw_t load(const float* p) { raw_t r; r.a = *reinterpret_cast<const float4*>(p); r.b = *reinterpret_cast<const float4*>(p + 4); return r; }
    static __device__ __forceinline__ raw_t zero() { raw_t r; r.a = make_float4(0.f, 0.f, 0.f, 0.f); r.b = r.a; return r; }
    static __device__ __forceinline__ float get(const raw_t& r, int i) {
        switch (i) { case 0: return r.a.x; case 1: return r.a.y; case 2: return r.a.z; case 3: return r.a.w; case 4: return r.b.x; case 5: return r.b.y; case 6: return r.b.z; default: return r.b.w; }
    }
};

// ------------------------------------------------------------------ scores: grid (B, ceil(L / 16)), wave per location
// U values of the wave's location are in flight before the query rows are staged (NK = A / 64 values per lane).
template <int RN, int NK>
__global__ __launch_bounds__(ATTS_WAVES * 64) void attention_scores_fast_kernel(const float* __restrict__ U, const float* __restrict__ hc, int hc_ld,
                                                                                const float* __restrict__ wf, float* __restrict__ sc, int R, int L, int A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_q = sm;                 // [RN][A]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l = blockIdx.y * ATTS_WAVES + wave;
    const float scale = 1.0f / sqrtf((float)L);
    float uv[NK], w[NK];
    const float* u = U + ((long)b * L + (l < L ? l : L - 1)) * A;
#pragma unroll
    for (int j = 0; j < NK; ++j) { uv[j] = u[lane + 64 * j]; w[j] = wf[lane + 64 * j]; }
    for (int r0 = 0; r0 < R; r0 += RN) {
        const int rn = min(RN, R - r0), i0 = b * R + r0;
        __syncthreads();
        for (int e = tid; e < RN * A; e += ATTS_WAVES * 64) { const int r = e / A, k = e - r * A; s_q[e] = (r < rn) ? hc[(long)(i0 + r) * hc_ld + k] : 0.f; }
        __syncthreads();
        if (l < L) {
            float part[RN];
#pragma unroll
            for (int r = 0; r < RN; ++r) part[r] = 0.f;
#pragma unroll
            for (int j = 0; j < NK; ++j)
#pragma unroll
                for (int r = 0; r < RN; ++r) part[r] = fmaf(w[j], fast_tanh(uv[j] + s_q[r * A + lane + 64 * j]), part[r]);
#pragma unroll
            for (int r = 0; r < RN; ++r) { const float s = wave_sum(part[r]); if (lane == 0 && r < rn) sc[(long)(i0 + r) * L + l] = s * scale; }
        }
    }
}

// ------------------------------------------------------------------ context: grid (B, D / 64), 256 threads = 32 location groups x 8 feature vectors
// LDS: alphas [RN][L] | partial sums [32 groups][RN][64 features]
template <int RN, typename TA>
__global__ __launch_bounds__(256) void attention_context_fast_kernel(const TA* __restrict__ ann, const float* __restrict__ sc, const float* __restrict__ hc,
                                                                     int hc_ld, const int* __restrict__ lengths, int step, float* __restrict__ alphas, int T1,
                                                                     float* __restrict__ Z, float* __restrict__ XZ, int R, int L, int D, int A,
                                                                     __bf16* __restrict__ xzb) {
    typedef Ann8<TA> AV;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_al = sm;                                       // [RN][L]
    float* s_part = sm + ((RN * L + 3) & ~3);               // [32][RN][64]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d0 = blockIdx.y * ATTC_DCH;
    const int v = tid & 7, g = tid >> 3;
    const int d = d0 + 8 * v;
    // every annotation vector this thread owns, issued now (L <= 256: at most 8 locations per thread)
    typename AV::raw_t x[8];
    {
        const TA* base = ann + (long)b * L * D + (d < D ? d : 0);
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int l = g + 32 * u; x[u] = (l < L && d < D) ? AV::load(base + (long)l * D) : AV::zero(); }
    }
    for (int r0 = 0; r0 < R; r0 += RN) {
        const int rn = min(RN, R - r0), i0 = b * R + r0;
        // score rows (wave w: rows w and w + 4), lengths and the gate row of the output phase: all independent, all issued before use
        float sv[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = wave + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int l = lane + 64 * q; sv[h][q] = (r < rn && l < L) ? sc[(long)(i0 + r) * L + l] : -INFINITY; }
        }
        unsigned lmask = 0;
        for (int r = 0; r < rn; ++r) if (lengths[i0 + r] > step) lmask |= 1u << r;
        const int er = tid >> 4, ev = tid & 15, edd = d0 + 4 * ev;            // output phase: thread (row er, float4 ev)
        float4 be = make_float4(0.f, 0.f, 0.f, 0.f);
        if (er < rn && edd < D) be = *reinterpret_cast<const float4*>(hc + (long)(i0 + er) * hc_ld + A + edd);
        __syncthreads();                                    // (second pass: the previous pass is done with s_al / s_part)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = wave + 4 * h;
            if (r >= RN) continue;
            const bool livr = r < rn && ((lmask >> r) & 1u);
            float mx = fmaxf(fmaxf(sv[h][0], sv[h][1]), fmaxf(sv[h][2], sv[h][3]));
            mx = wave_max(mx);
            float e[4], sum = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) { e[q] = (lane + 64 * q < L && livr) ? __expf(sv[h][q] - mx) : 0.f; sum += e[q]; }
            sum = wave_sum(sum);
            const float inv = livr ? 1.0f / sum : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int l = lane + 64 * q; if (l < L) s_al[r * L + l] = e[q] * inv; }
        }
        __syncthreads();
        if (blockIdx.y == 0)
            for (int e = tid; e < rn * L; e += 256) { const int r = e / L, l = e - r * L; alphas[((long)(i0 + r) * T1 + step) * L + l] = s_al[r * L + l]; }
        float acc[RN][8];
#pragma unroll
        for (int r = 0; r < RN; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[r][k] = 0.f;
        if (lmask) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int l = g + 32 * u;
                if (l < L) {
#pragma unroll
                    for (int r = 0; r < RN; ++r) {
                        const float al = s_al[r * L + l];
#pragma unroll
                        for (int k = 0; k < 8; ++k) acc[r][k] = fmaf(al, AV::get(x[u], k), acc[r][k]);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RN; ++r) {
            float4* q = reinterpret_cast<float4*>(s_part + ((g * RN + r) * 64 + 8 * v));
            q[0] = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
            q[1] = make_float4(acc[r][4], acc[r][5], acc[r][6], acc[r][7]);
        }
        __syncthreads();
        if (er < rn && edd < D) {          // 16 threads per row: the 32 location groups in order, then z and beta * z
            float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
            for (int gg = 0; gg < 32; ++gg) { const float4 t = *reinterpret_cast<const float4*>(s_part + ((gg * RN + er) * 64 + 4 * ev)); z.x += t.x; z.y += t.y; z.z += t.z; z.w += t.w; }
            const long orow = i0 + er;
            if (!((lmask >> er) & 1u)) be = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(Z + orow * D + edd) = z;
            const float4 xz = make_float4(be.x * z.x, be.y * z.y, be.z * z.z, be.w * z.w);
            *reinterpret_cast<float4*>(XZ + orow * D + edd) = xz;
            if (xzb) {
                typedef __bf16 b4 __attribute__((ext_vector_type(4)));
                b4 o; o[0] = (__bf16)xz.x; o[1] = (__bf16)xz.y; o[2] = (__bf16)xz.z; o[3] = (__bf16)xz.w;
                *reinterpret_cast<b4*>(xzb + orow * D + edd) = o;
            }
        }
    }
}

// ------------------------------------------------------------------ backward, dalpha: grid (B, ceil(L / (16 * LW))), 1024 threads
// wave w of block y streams LW consecutive locations (all their vectors in flight at entry); the image's dz rows go to LDS once per block.
// NC = D / 512 register chunks per location (8 features per lane and chunk).
template <int RN, typename TA, int LW, int NC>
__global__ __launch_bounds__(1024) void attention_bwd_dalpha_fast_kernel(const TA* __restrict__ ann, const float* __restrict__ hc, int hc_ld,
        const int* __restrict__ lengths, int step, const float* __restrict__ dalphas_ext, int T1, const float* __restrict__ Zs,
        const float* __restrict__ dZ_out, const float* __restrict__ dXZ, float* __restrict__ DZ, float* __restrict__ dhc, int dhc_ld,
        float* __restrict__ da, int R, int L, int D, int A, __bf16* __restrict__ dhcb) {
    typedef Ann8<TA> AV;
    extern __shared__ __attribute__((aligned(16))) float s_dz[];          // [RN][D]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l0 = (blockIdx.y * 16 + wave) * LW;
    typename AV::raw_t x[LW][NC];
#pragma unroll
    for (int j = 0; j < LW; ++j) {
        const int l = l0 + j;
        const TA* a = ann + ((long)b * L + (l < L ? l : 0)) * D;
#pragma unroll
        for (int c = 0; c < NC; ++c) { const int dd = c * 512 + lane * 8; x[j][c] = (l < L && dd < D) ? AV::load(a + dd) : AV::zero(); }
    }
    const int nvec = RN * D / 4;
    for (int r0 = 0; r0 < R; r0 += RN) {
        const int rn = min(RN, R - r0), i0 = b * R + r0;
        unsigned lmask = 0;
        for (int r = 0; r < rn; ++r) if (lengths[i0 + r] > step) lmask |= 1u << r;
        __syncthreads();
        // dz rows: two float4 stages per thread and trip, their twelve loads issued together
        for (int base = 0; base < nvec; base += 2048) {
            float4 be[2], zz[2], dx[2], dzo[2];
            int rr[2], dd[2]; bool ok[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int e = base + s * 1024 + tid;
                rr[s] = (e * 4) / D; dd[s] = e * 4 - rr[s] * D;
                ok[s] = e < nvec && rr[s] < rn;
                const long row = i0 + (ok[s] ? rr[s] : 0);
                const int dq = ok[s] ? dd[s] : 0;
                be[s] = *reinterpret_cast<const float4*>(hc + row * hc_ld + A + dq);
                zz[s] = *reinterpret_cast<const float4*>(Zs + row * D + dq);
                dx[s] = *reinterpret_cast<const float4*>(dXZ + row * D + dq);
                dzo[s] = *reinterpret_cast<const float4*>(dZ_out + row * D + dq);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int e = base + s * 1024 + tid;
                if (e >= nvec) continue;
                float4 dz = make_float4(0.f, 0.f, 0.f, 0.f), dbp = dz;
                if (ok[s] && ((lmask >> rr[s]) & 1u)) {
                    dz = make_float4(dzo[s].x + dx[s].x * be[s].x, dzo[s].y + dx[s].y * be[s].y, dzo[s].z + dx[s].z * be[s].z, dzo[s].w + dx[s].w * be[s].w);
                    dbp = make_float4(dx[s].x * zz[s].x * be[s].x * (1.f - be[s].x), dx[s].y * zz[s].y * be[s].y * (1.f - be[s].y),
                                      dx[s].z * zz[s].z * be[s].z * (1.f - be[s].z), dx[s].w * zz[s].w * be[s].w * (1.f - be[s].w));
                }
                *reinterpret_cast<float4*>(s_dz + (long)e * 4) = dz;
                if (blockIdx.y == 0 && ok[s]) {
                    const long row = i0 + rr[s];
                    *reinterpret_cast<float4*>(DZ + row * D + dd[s]) = dz;
                    *reinterpret_cast<float4*>(dhc + row * dhc_ld + A + dd[s]) = dbp;
                    if (dhcb) {
                        typedef __bf16 b4 __attribute__((ext_vector_type(4)));
                        b4 o; o[0] = (__bf16)dbp.x; o[1] = (__bf16)dbp.y; o[2] = (__bf16)dbp.z; o[3] = (__bf16)dbp.w;
                        *reinterpret_cast<b4*>(dhcb + row * dhc_ld + A + dd[s]) = o;
                    }
                }
            }
        }
        __syncthreads();
        float part[LW][RN];
#pragma unroll
        for (int j = 0; j < LW; ++j)
#pragma unroll
            for (int r = 0; r < RN; ++r) part[j][r] = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int dd = c * 512 + lane * 8;
            if (dd < D) {
#pragma unroll
                for (int r = 0; r < RN; ++r) {
                    const float4 z0 = *reinterpret_cast<const float4*>(s_dz + r * D + dd), z1 = *reinterpret_cast<const float4*>(s_dz + r * D + dd + 4);
#pragma unroll
                    for (int j = 0; j < LW; ++j) {
                        float p = part[j][r];
                        p = fmaf(AV::get(x[j][c], 0), z0.x, p); p = fmaf(AV::get(x[j][c], 1), z0.y, p); p = fmaf(AV::get(x[j][c], 2), z0.z, p); p = fmaf(AV::get(x[j][c], 3), z0.w, p);
                        p = fmaf(AV::get(x[j][c], 4), z1.x, p); p = fmaf(AV::get(x[j][c], 5), z1.y, p); p = fmaf(AV::get(x[j][c], 6), z1.z, p); p = fmaf(AV::get(x[j][c], 7), z1.w, p);
                        part[j][r] = p;
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < LW; ++j) {
            const int l = l0 + j;
#pragma unroll
            for (int r = 0; r < RN; ++r) {
                const float sdot = wave_sum(part[j][r]);
                if (lane == 0 && r < rn && l < L)
                    da[(long)(i0 + r) * L + l] = ((lmask >> r) & 1u) ? sdot + (dalphas_ext ? dalphas_ext[((long)(i0 + r) * T1 + step) * L + l] : 0.f) : 0.f;
            }
        }
    }
}

// ------------------------------------------------------------------ backward, tanh: grid (B, A / 32), 1024 threads = 32 units x 32 location groups
// the softmax backward of a row runs on registers (one pass over alphas / d(alpha)); the thread's U and dU values (L <= 256: 8 of each) are in
// flight from the start.
template <int RN>
__global__ __launch_bounds__(1024) void attention_bwd_tanh_fast_kernel(const float* __restrict__ U, const float* __restrict__ hc, int hc_ld, const float* __restrict__ wf,
        const int* __restrict__ lengths, int step, const float* __restrict__ alphas, int T1, const float* __restrict__ da, float* __restrict__ dhc,
        int dhc_ld, float* __restrict__ dU, float* __restrict__ dwf_part, int R, int L, int A, __bf16* __restrict__ dhcb) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_ds = sm;                               // [RN][L]
    float* s_q = s_ds + RN * L;                     // [RN][KCH]
    float* s_red = s_q + RN * ATTB_KCH;             // [32 groups][RN + 1][KCH]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kk = tid & 31, lg = tid >> 5, k = blockIdx.y * ATTB_KCH + kk;
    const float scale = 1.0f / sqrtf((float)L);
    const float w = k < A ? wf[k] : 0.f;
    float uvals[8], duold[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int l = lg + 32 * i;
        const long uo = ((long)b * L + (l < L ? l : 0)) * A + (k < A ? k : 0);
        uvals[i] = U[uo]; duold[i] = dU[uo];
    }
    float dw_total = 0.f;
    for (int r0 = 0; r0 < R; r0 += RN) {
        const int rn = min(RN, R - r0), i0 = b * R + r0;
        unsigned lmask = 0;
        for (int r = 0; r < rn; ++r) if (lengths[i0 + r] > step) lmask |= 1u << r;
        __syncthreads();
        // softmax backward: ds = alpha * (dalpha - sum alpha * dalpha) * L^-1/2, wave per row; dead rows: 0
        if (wave < RN) {
            const int r = wave;
            const bool livr = r < rn && ((lmask >> r) & 1u);
            float al[4], dv[4], dot = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int l = lane + 64 * q; const bool in = livr && l < L;
                al[q] = in ? alphas[((long)(i0 + r) * T1 + step) * L + l] : 0.f;
                dv[q] = in ? da[(long)(i0 + r) * L + l] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) dot += al[q] * dv[q];
            dot = wave_sum(dot);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int l = lane + 64 * q; if (l < L) s_ds[r * L + l] = livr ? al[q] * (dv[q] - dot) * scale : 0.f; }
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
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int l = lg + 32 * i;
                if (l < L) {
                    float du = 0.f;
#pragma unroll
                    for (int r = 0; r < RN; ++r) {
                        const float th = fast_tanh(uvals[i] + s_q[r * ATTB_KCH + kk]);
                        const float ds = s_ds[r * L + l];
                        const float dp = ds * w * (1.f - th * th);
                        dq[r] += dp; du += dp; dw = fmaf(ds, th, dw);
                    }
                    duold[i] += du;                  // (b, k) belongs to this block alone; written once, after the last pass
                }
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
    if (k < A) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int l = lg + 32 * i; if (l < L) dU[((long)b * L + l) * A + k] = duold[i]; }
    }
    if (tid < ATTB_KCH && k < A) dwf_part[(long)b * A + k] += dw_total;
}

}  // namespace sat
