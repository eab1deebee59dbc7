// Host-side orchestration of the SAT decoder on one HIP stream: the whole train_batch time loop
// (model.py:487-548) and its backward through time, as a fixed sequence of kernel launches.
// No allocation, no synchronisation: everything lives in the caller's workspace.
#include "../../include/sat_hip.h"
#include <stdlib.h>
#include "decoder.h"
#include "profile.h"
#include "decoder_kernels.h"
#include "gemm.h"

namespace sat {

// ------------------------------------------------------------------ workspace layout
struct Ws {
    size_t total = 0;
    // saved by forward
    float *U, *mean, *f, *init_img, *H_all, *C_all, *HC, *Z, *XZ, *Y, *GY, *Uact, *Wcat, *bcat;
    float *Udrop, *mean_rows, *f_rows, *init_rows, *df_rows;     // only carved when dropout > 0
    float *GU, *DGU, *bup;                                       // stacked LSTM layers (layers > 1): gates, gate grads, bias sums
    __bf16 *Wb_out, *Ub;                                         // bf16 mode: bf16 copies of output.output.weight and of the packed deep-output rows
    __bf16 *annb;                                                // bf16 mode: annotations as the attention context / dalpha kernels stream them
    __bf16 *Hb, *XZb, *Wcat_b, *Wz_b, *DHCb;                            // bf16 mode, one LSTM layer: operands of the per-step GEMMs (state, gated context, weights)
    int* Tok; int* flags;
    int *emb_count, *emb_offset, *emb_cursor, *emb_list;         // embedding gradient: per-row token segments
    float *SC, *DA;                                              // attention: raw scores / score gradients of one step (N, L)
    // backward scratch
    float *dA, *dHout, *dZout, *DZ, *DHC, *dXZ, *dHc, *dCc, *dU, *dwf_part, *dY, *colpart, *dinit_img, *df, *dmean, *slab;
    char *zero_begin, *zero_end;                                 // [dHout .. dwf_part]: what decoder_bwd zeroes in one memset
    long slab_elems;
};

Ws layout(const sat_decoder_dims& d, char* base) {
    Ws w; size_t off = 0;
    const long N = (long)d.B * d.R, T1 = d.T - 1, HCW = d.A + d.D + 4L * d.n;
    auto take = [&](size_t elems, size_t esz = 4) { size_t o = off; off += (elems * esz + 255) & ~(size_t)255; return base ? base + o : (char*)nullptr; };
    w.U = (float*)take((size_t)d.B * d.L * d.A);
    w.mean = (float*)take((size_t)d.B * d.D);
    w.f = (float*)take((size_t)d.B * d.m);
    const int NL = d.layers;
    w.init_img = (float*)take((size_t)d.B * 2 * d.n * NL);
    // hidden / cell state of every layer and step, one time-major run per layer: [layer][T1 + 1][N][n]
    w.H_all = (float*)take((size_t)NL * (T1 + 1) * N * d.n);
    w.C_all = (float*)take((size_t)NL * (T1 + 1) * N * d.n);
    w.GU = w.DGU = w.bup = nullptr;
    if (NL > 1) {
        w.GU = (float*)take((size_t)(NL - 1) * T1 * N * 4 * d.n); w.DGU = (float*)take((size_t)(NL - 1) * T1 * N * 4 * d.n);
        w.bup = (float*)take((size_t)(NL - 1) * 4 * d.n);
    }
    w.HC = (float*)take((size_t)T1 * N * HCW);
    w.Z = (float*)take((size_t)T1 * N * d.D);
    w.XZ = (float*)take((size_t)T1 * N * d.D);
    w.Y = (float*)take((size_t)T1 * N * d.m);
    w.GY = (float*)take((size_t)T1 * N * 4 * d.n);
    w.Uact = (float*)take((size_t)(d.P > 0 ? d.P : 1) * d.m);
    w.Wcat = (float*)take((size_t)HCW * d.n);
    w.bcat = (float*)take((size_t)HCW);
    w.Tok = (int*)take((size_t)T1 * N);
    w.Udrop = w.mean_rows = w.f_rows = w.init_rows = w.df_rows = nullptr;
    if (d.dropout > 0.f) {
        w.Udrop = (float*)take((size_t)(d.P > 0 ? d.P : 1) * d.m); w.mean_rows = (float*)take((size_t)N * d.D); w.f_rows = (float*)take((size_t)N * d.m);
        w.init_rows = (float*)take((size_t)N * 2 * d.n * NL); w.df_rows = (float*)take((size_t)N * d.m);
    }
    w.Wb_out = w.Ub = nullptr;
    if (d.precision && d.m % 64 == 0 && d.V % 8 == 0) {           // operands of the vocabulary projection for the direct-to-LDS GEMM
        w.Wb_out = (__bf16*)take((size_t)d.V * d.m, 2); w.Ub = (__bf16*)take((size_t)(d.P > 0 ? d.P : 1) * d.m, 2);
    }
    w.Hb = w.XZb = w.Wcat_b = w.Wz_b = w.DHCb = nullptr;
    // bf16 mode: a bf16 copy of the annotations for the two kernels that stream them every time step (attention context / dalpha)
    w.annb = (d.precision && d.D % 4 == 0) ? (__bf16*)take((size_t)d.B * d.L * d.D, 2) : nullptr;
    if (d.precision && d.layers == 1 && d.n % 64 == 0 && d.D % 64 == 0 && d.A % 4 == 0 && d.m % 4 == 0) {
        w.Hb = (__bf16*)take((size_t)(T1 + 1) * N * d.n, 2); w.XZb = (__bf16*)take((size_t)N * d.D, 2);
        w.Wcat_b = (__bf16*)take((size_t)HCW * d.n, 2); w.Wz_b = (__bf16*)take((size_t)4 * d.n * d.D, 2);
        w.DHCb = (__bf16*)take((size_t)N * HCW, 2);
    }
    w.flags = (int*)take((size_t)d.V);
    w.emb_count = (int*)take((size_t)d.V); w.emb_offset = (int*)take((size_t)d.V + 1); w.emb_cursor = (int*)take((size_t)d.V);
    w.emb_list = (int*)take((size_t)T1 * N);
    w.SC = (float*)take((size_t)N * d.L); w.DA = (float*)take((size_t)N * d.L);
    w.dA = (float*)take((size_t)(d.P > 0 ? d.P : 1) * d.m);
    w.DZ = (float*)take((size_t)T1 * N * d.D);
    w.DHC = (float*)take((size_t)T1 * N * HCW);
    w.dXZ = (float*)take((size_t)N * d.D);
    // the accumulators the backward pass starts from zero, adjacent: one memset (zero_begin .. zero_end)
    w.dHout = (float*)take((size_t)T1 * N * d.n); w.zero_begin = (char*)w.dHout;
    w.dZout = (float*)take((size_t)T1 * N * d.D);
    w.dHc = (float*)take((size_t)NL * N * d.n);
    w.dCc = (float*)take((size_t)NL * N * d.n);
    w.dU = (float*)take((size_t)d.B * d.L * d.A);
    w.dwf_part = (float*)take((size_t)d.B * d.A);
    w.zero_end = base ? base + off : (char*)nullptr;
    w.dY = (float*)take((size_t)T1 * N * d.m);
    long maxrows = d.P > T1 * N ? d.P : T1 * N; if (maxrows < d.B * (long)d.L) maxrows = d.B * (long)d.L;
    long maxcols = d.V > HCW ? d.V : HCW;
    w.colpart = (float*)take((size_t)cdiv(maxrows, 256) * maxcols);
    w.dinit_img = (float*)take((size_t)d.B * 2 * d.n * NL);
    w.df = (float*)take((size_t)d.B * d.m);
    w.dmean = (float*)take((size_t)d.B * d.D);
    w.slab_elems = 8L << 20;   // 32 MiB of split-K partials
    w.slab = (float*)take((size_t)w.slab_elems);
    w.total = off;
    return w;
}

int check_dims(const sat_decoder_dims* d) {
    SAT_REQUIRE(d, "decoder: null dims");
    SAT_REQUIRE(d->B > 0 && d->R > 0 && d->T >= 2 && d->L > 0 && d->D > 0 && d->A > 0 && d->m > 0 && d->n > 0 && d->V > 0,
                "decoder: non-positive dimension (B=%d R=%d T=%d L=%d D=%d A=%d m=%d n=%d V=%d)", d->B, d->R, d->T, d->L, d->D, d->A, d->m, d->n, d->V);
    SAT_REQUIRE(d->P >= 0 && (long)d->P <= (long)d->B * d->R * (d->T - 1), "decoder: packed token count %d out of range", d->P);
    SAT_REQUIRE(d->layers >= 1 && d->layers <= SAT_MAX_LSTM_LAYERS, "decoder: layers=%d outside 1..%d", d->layers, SAT_MAX_LSTM_LAYERS);
    SAT_REQUIRE(d->dropout >= 0.f && d->dropout < 1.f && d->embedding_dropout >= 0.f && d->embedding_dropout < 1.f, "decoder: dropout must lie in [0,1)");
    return SAT_OK;
}

// ------------------------------------------------------------------ small launch helpers
static thread_local int t_bf16_mfma = 0;      // set per call from sat_decoder_dims.precision
static int gemm(hipStream_t st, int amode, int bmode, const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                int M, int N, int K, int acc = 0, int epi = EPI_NONE, const float* bias = nullptr, const int* a_rows = nullptr,
                const int* c_rows = nullptr, const float* e0 = nullptr, long lde0 = 0, int c0 = 0, int c1 = 0, float* slab = nullptr, long slab_elems = 0) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.amode = amode; g.bmode = bmode; g.accumulate = acc; g.epi = epi; g.bias = bias; g.a_rows = a_rows; g.c_rows = c_rows;
    g.e0 = e0; g.lde0 = lde0; g.c0 = c0; g.c1 = c1; g.slab = slab; g.slab_elems = slab_elems;
    g.bf16_mfma = t_bf16_mfma;
    return launch_gemm(g, st);
}

// fp32 -> bf16 copy of n elements (n % 4 == 0)
static int cast_bf16(hipStream_t st, const float* src, __bf16* dst, long n) {
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, st, src, dst, n / 4);
    return launch_ok("cast_bf16");
}
// C (fp32) = A (bf16, rows x K) * B^T (bf16, N x K): both operands bf16 in HBM -> gemm_glds.hip
static int gemm_bb_nt(hipStream_t st, const __bf16* A, long lda, const __bf16* B, long ldb, float* C, long ldc, int M, int N, int K, int epi, const float* bias,
                      int acc = 0, int c0 = 0, int c1 = 0) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.amode = A_ROW; g.bmode = B_ROW; g.epi = epi; g.bias = bias; g.a_bf16 = g.b_bf16 = 1; g.c_bf16 = 0; g.bf16_mfma = 1;
    g.accumulate = acc; g.c0 = c0; g.c1 = c1;
    return launch_gemm(g, st);
}

// C (fp32) (+)= A (bf16, rows x K, row stride lda) * B (bf16, K x N k-major): data gradients of the per-step products
static int gemm_bb_nn(hipStream_t st, const __bf16* A, long lda, const __bf16* B, long ldb, float* C, long ldc, int M, int N, int K, int acc, float* slab, long slab_elems) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.amode = A_ROW; g.bmode = B_KMAJOR; g.a_bf16 = g.b_bf16 = 1; g.c_bf16 = 0; g.bf16_mfma = 1; g.accumulate = acc; g.slab = slab; g.slab_elems = slab_elems;
    return launch_gemm(g, st);
}

static int colsum(hipStream_t st, const Ws& w, const float* x, long ld, int rows, int cols, float* out, int accumulate = 0, float scale = 1.f) {
    if (cols <= 0) return SAT_OK;
    if (rows <= 0) { if (!accumulate) SAT_TRY(dev_fill_bytes(st, out, 0, (size_t)cols * 4)); return SAT_OK; }
    int nparts = cdiv(rows, 256);
    if (cols % 4 == 0 && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
        hipLaunchKernelGGL(colsum_part4_kernel, dim3(cdiv(cols / 4, 128), nparts), dim3(128), 0, st, x, ld, rows, cols / 4, 256, w.colpart);
    else
        hipLaunchKernelGGL(colsum_part_kernel, dim3(cdiv(cols, 256), nparts), dim3(256), 0, st, x, ld, rows, cols, 256, w.colpart);
    SAT_TRY(launch_ok("colsum_part"));
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, st, w.colpart, nparts, cols, out, accumulate, scale);
    return launch_ok("colsum_finish");
}

static int live_steps(const sat_decoder_dims& d, const sat_decoder_batch& b) {
    const int T1 = d.T - 1; int ts = 0;
    for (int t = 0; t < T1; ++t) if (b.step_offsets_host[t + 1] > b.step_offsets_host[t]) ts = t + 1;
    return ts;
}

static size_t att_fwd_lds(int L, int A, int vw) { return (size_t)(ATT_RMAX * L + ATT_RMAX * A + A + ATT_RMAX * ATT_THREADS * vw) * 4; }
static size_t att_bwd_lds(int L, int A, int D) { return (size_t)(2 * ATT_RMAX * L + ATT_RMAX * A + A + ATT_RMAX * D + ATTB_WAVES * ATT_RMAX * A + ATTB_WAVES * A) * 4; }

static bool ann_bf16_enabled() { static const bool off = getenv("SAT_ANN_BF16") && !atoi(getenv("SAT_ANN_BF16")); return !off; }      // dev switch
static bool attention_split_enabled() {
    static const int no_split = getenv("SAT_ATT_FUSED") ? atoi(getenv("SAT_ATT_FUSED")) : 0;
    return !no_split;
}
template <int RN, typename TA>
static int attention_fwd_split(hipStream_t st, const TA* ann, const float* U, const float* hc, int hc_ld, const float* wf, const int* lengths, int step,
                               float* alphas, int T1, float* Z, float* XZ, int B, int R, int L, int D, int A, float* sc, __bf16* xzb) {
    const size_t lds_s = (size_t)(RN * A + A) * 4, lds_c = (size_t)((RN * L + 3) & ~3) * 4 + (size_t)16 * RN * 16 * 16;
    SAT_REQUIRE(lds_s <= 160 * 1024 && lds_c <= 160 * 1024, "attention_fwd: L=%d A=%d do not fit the LDS", L, A);
    SAT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_scores_kernel<RN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
    SAT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_context_kernel<RN, TA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c));
    // algorithmic bytes (per image-step): scores read att_enc U (L x A fp32) and the R query rows, write R x L raw scores; context streams the
    // image's annotations (L x D) ONCE for its R captions, reads scores + gates, writes alphas, z and the gated context
    const double N = (double)B * R;
    {
        ProfScope prof("attention_scores", 2.0 * N * L * A, 4.0 * ((double)B * L * A + N * (A + L)), st);
        hipLaunchKernelGGL(attention_scores_kernel<RN>, dim3(B, cdiv(L, ATTS_WAVES)), dim3(ATTS_WAVES * 64), lds_s, st, U, hc, hc_ld, wf, sc, R, L, A);
        SAT_TRY(launch_ok("attention_scores"));
    }
    ProfScope prof("attention_context", 2.0 * N * L * D, (double)sizeof(TA) * B * L * D + 4.0 * N * (2.0 * L + 3.0 * D), st);
    hipLaunchKernelGGL((attention_context_kernel<RN, TA>), dim3(B, cdiv(D, ATTC_DCH)), dim3(256), lds_c, st, ann, sc, hc, hc_ld, lengths, step, alphas, T1, Z, XZ, R, L, D, A, xzb);
    return launch_ok("attention_context");
}

int launch_attention_fwd(hipStream_t st, const float* ann, const float* U, const float* hc, int hc_ld, const float* wf,
                                const int* lengths, int step, float* alphas, int T1, float* Z, float* XZ, int B, int R, int L, int D, int A, float* sc, void* xzb_v,
                                const void* annb_v) {
    __bf16* xzb = reinterpret_cast<__bf16*>(xzb_v);
    const __bf16* annb = reinterpret_cast<const __bf16*>(annb_v);          // bf16 copy of ann for the context stream (split kernels only), or NULL
    static const int no_split = getenv("SAT_ATT_FUSED") ? atoi(getenv("SAT_ATT_FUSED")) : 0;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (sc && !no_split && D % 4 == 0 && A % 4 == 0 && hc_ld % 4 == 0 && al16(ann) && al16(hc) && al16(Z) && al16(XZ)) {
        // scores and context as two chip-wide launches (scratch: raw scores (B*R, L))
        switch (R < ATT_RMAX ? R : ATT_RMAX) {
            case 1: return annb ? attention_fwd_split<1, __bf16>(st, annb, U, hc, hc_ld, wf, lengths, step, alphas, T1, Z, XZ, B, R, L, D, A, sc, xzb)
                                : attention_fwd_split<1, float>(st, ann, U, hc, hc_ld, wf, lengths, step, alphas, T1, Z, XZ, B, R, L, D, A, sc, xzb);
            case 2: return annb ? attention_fwd_split<2, __bf16>(st, annb, U, hc, hc_ld, wf, lengths, step, alphas, T1, Z, XZ, B, R, L, D, A, sc, xzb)
                                : attention_fwd_split<2, float>(st, ann, U, hc, hc_ld, wf, lengths, step, alphas, T1, Z, XZ, B, R, L, D, A, sc, xzb);
            case 3: return annb ? attention_fwd_split<3, __bf16>(st, annb, U, hc, hc_ld, wf, lengths, step, alphas, T1, Z, XZ, B, R, L, D, A, sc, xzb)
                                : attention_fwd_split<3, float>(st, ann, U, hc, hc_ld, wf, lengths, step, alphas, T1, Z, XZ, B, R, L, D, A, sc, xzb);
            case 4: return annb ? attention_fwd_split<4, __bf16>(st, annb, U, hc, hc_ld, wf, lengths, step, alphas, T1, Z, XZ, B, R, L, D, A, sc, xzb)
                                : attention_fwd_split<4, float>(st, ann, U, hc, hc_ld, wf, lengths, step, alphas, T1, Z, XZ, B, R, L, D, A, sc, xzb);
            case 5: return annb ? attention_fwd_split<5, __bf16>(st, annb, U, hc, hc_ld, wf, lengths, step, alphas, T1, Z, XZ, B, R, L, D, A, sc, xzb)
                                : attention_fwd_split<5, float>(st, ann, U, hc, hc_ld, wf, lengths, step, alphas, T1, Z, XZ, B, R, L, D, A, sc, xzb);
            case 6: return annb ? attention_fwd_split<6, __bf16>(st, annb, U, hc, hc_ld, wf, lengths, step, alphas, T1, Z, XZ, B, R, L, D, A, sc, xzb)
                                : attention_fwd_split<6, float>(st, ann, U, hc, hc_ld, wf, lengths, step, alphas, T1, Z, XZ, B, R, L, D, A, sc, xzb);
            case 7: return annb ? attention_fwd_split<7, __bf16>(st, annb, U, hc, hc_ld, wf, lengths, step, alphas, T1, Z, XZ, B, R, L, D, A, sc, xzb)
                                : attention_fwd_split<7, float>(st, ann, U, hc, hc_ld, wf, lengths, step, alphas, T1, Z, XZ, B, R, L, D, A, sc, xzb);
            default: return annb ? attention_fwd_split<8, __bf16>(st, annb, U, hc, hc_ld, wf, lengths, step, alphas, T1, Z, XZ, B, R, L, D, A, sc, xzb)
                                 : attention_fwd_split<8, float>(st, ann, U, hc, hc_ld, wf, lengths, step, alphas, T1, Z, XZ, B, R, L, D, A, sc, xzb);
        }
    }
    SAT_REQUIRE(!xzb && !annb, "attention_fwd: the bf16 copies need the split kernels (aligned shapes)");
    const bool vec = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(ann) & 15) == 0);
    const int vw = vec ? 4 : 1;
    int dchunk = vec ? 256 : 64;
    if (dchunk > D) dchunk = D;
    size_t lds = att_fwd_lds(L, A, vw);
    SAT_REQUIRE(lds <= 160 * 1024, "attention_fwd: L=%d A=%d need %zu B of LDS (> 160 KiB)", L, A, lds);
    dim3 grid(B, cdiv(D, dchunk));
    const int RN = R < ATT_RMAX ? R : ATT_RMAX;            // rows per pass, compile-time in the kernel
#define SAT_ATTF(VWV, RNV)                                                                                                                  \
    case RNV:                                                                                                                               \
        SAT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_fwd_kernel<VWV, RNV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((attention_fwd_kernel<VWV, RNV>), grid, dim3(ATTF_THREADS), lds, st, ann, U, hc, hc_ld, wf, lengths, step, alphas, T1, Z, XZ, R, L, D, A, dchunk); \
        break;
    if (vec) {
        switch (RN) { SAT_ATTF(4, 1) SAT_ATTF(4, 2) SAT_ATTF(4, 3) SAT_ATTF(4, 4) SAT_ATTF(4, 5) SAT_ATTF(4, 6) SAT_ATTF(4, 7) SAT_ATTF(4, 8) default: break; }
    } else {
        switch (RN) { SAT_ATTF(1, 1) SAT_ATTF(1, 2) SAT_ATTF(1, 3) SAT_ATTF(1, 4) SAT_ATTF(1, 5) SAT_ATTF(1, 6) SAT_ATTF(1, 7) SAT_ATTF(1, 8) default: break; }
    }
#undef SAT_ATTF
    return launch_ok("attention_fwd");
}

template <int RN, typename TA>
static int attention_bwd_split_t(hipStream_t st, const TA* ann, const float* U, const float* hc, int hc_ld, const float* wf, const int* lengths, int step,
                                 const float* alphas, const float* dalphas, int T1, const float* Zs, const float* dZ_out, const float* dXZ, float* DZ, float* dhc,
                                 int dhc_ld, float* dU, float* dwf_part, float* da, int B, int R, int L, int D, int A, __bf16* dhcb) {
    const size_t lds_a = (size_t)RN * D * 4, lds_t = (size_t)(RN * L + RN * ATTB_KCH + 32 * (RN + 1) * ATTB_KCH) * 4;
    SAT_REQUIRE(lds_a <= 160 * 1024 && lds_t <= 160 * 1024, "attention_bwd: L=%d D=%d do not fit the LDS", L, D);
    SAT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_dalpha_kernel<RN, TA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
    SAT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_tanh_kernel<RN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
    // algorithmic bytes: dalpha streams the annotations once per image, reads the R gradient rows of z / gated z / the gate (3 x D) and z, writes dz and
    // d(alpha); tanh reads att_enc and d(alpha), alphas, the queries, and adds into dU (read + write)
    const double N = (double)B * R;
    {
        ProfScope prof("attention_bwd_dalpha", 2.0 * N * L * D, (double)sizeof(TA) * B * L * D + 4.0 * N * (5.0 * D + 2.0 * L), st);
        hipLaunchKernelGGL((attention_bwd_dalpha_kernel<RN, TA>), dim3(B, cdiv(L, 16)), dim3(1024), lds_a, st, ann, hc, hc_ld, lengths, step, dalphas, T1, Zs, dZ_out, dXZ, DZ, dhc,
                           dhc_ld, da, R, L, D, A, dhcb);
        SAT_TRY(launch_ok("attention_bwd_dalpha"));
    }
    ProfScope prof("attention_bwd_tanh", 4.0 * N * L * A, 4.0 * (3.0 * B * L * A + N * (2.0 * A + 2.0 * L)), st);
    hipLaunchKernelGGL(attention_bwd_tanh_kernel<RN>, dim3(B, cdiv(A, ATTB_KCH)), dim3(1024), lds_t, st, U, hc, hc_ld, wf, lengths, step, alphas, T1, da, dhc, dhc_ld, dU,
                       dwf_part, R, L, A, dhcb);
    return launch_ok("attention_bwd_tanh");
}
static int attention_bwd_split(int RN, hipStream_t st, const float* ann, const float* U, const float* hc, int hc_ld, const float* wf, const int* lengths, int step,
                               const float* alphas, const float* dalphas, int T1, const float* Zs, const float* dZ_out, const float* dXZ, float* DZ, float* dhc,
                               int dhc_ld, float* dU, float* dwf_part, float* da, int B, int R, int L, int D, int A, __bf16* dhcb, const __bf16* annb = nullptr) {
#define SAT_ATTB(RNV) case RNV: return annb ? attention_bwd_split_t<RNV, __bf16>(st, annb, U, hc, hc_ld, wf, lengths, step, alphas, dalphas, T1, Zs, dZ_out, dXZ, DZ, dhc, dhc_ld, dU, dwf_part, da, B, R, L, D, A, dhcb) \
                                            : attention_bwd_split_t<RNV, float>(st, ann, U, hc, hc_ld, wf, lengths, step, alphas, dalphas, T1, Zs, dZ_out, dXZ, DZ, dhc, dhc_ld, dU, dwf_part, da, B, R, L, D, A, dhcb);
    switch (RN) { SAT_ATTB(1) SAT_ATTB(2) SAT_ATTB(3) SAT_ATTB(4) SAT_ATTB(5) SAT_ATTB(6) SAT_ATTB(7) default: SAT_ATTB(8) }
#undef SAT_ATTB
}

// ------------------------------------------------------------------ output stage for packed rows [p0, p1)
static int flush_outputs(hipStream_t st, const sat_decoder_dims& d, const sat_decoder_params& p, const sat_decoder_batch& b,
                         const Ws& w, float* logits, int p0, int p1) {
    const int rows = p1 - p0;
    if (rows <= 0) return SAT_OK;
    const long N = (long)d.B * d.R;
    const float* H1 = w.H_all + ((long)(d.layers - 1) * d.T + 1) * N * d.n;   // top layer's hidden state AFTER each step (F8: the new h)
    float* u = w.Uact + (long)p0 * d.m;
    const int* rows_map = b.src_row + p0;
    if (d.deep_output) {
        SAT_TRY(gemm(st, A_ROW, B_ROW, H1, d.n, p.out_hidden, d.n, u, d.m, rows, d.m, d.n, 0, EPI_NONE, nullptr, rows_map));
        SAT_TRY(gemm(st, A_ROW, B_ROW, w.Z, d.D, p.out_context, d.D, u, d.m, rows, d.m, d.D, 1, EPI_ADD_TANH, nullptr, rows_map, nullptr, w.Y, d.m));
    } else {
        SAT_TRY(gemm(st, A_ROW, B_ROW, H1, d.n, p.out_hidden, d.n, u, d.m, rows, d.m, d.n, 0, EPI_NONE, nullptr, rows_map));
    }
    const float* uin = u;
    if (d.dropout > 0.f) {                       // DeepOutput.dropout (model.py:130) on the packed rows
        float* ud = w.Udrop + (long)p0 * d.m;
        hipLaunchKernelGGL(dropout_rows_kernel, dim3(cdiv((long)rows * d.m, 256)), dim3(256), 0, st, u, ud, (long)rows * d.m, d.m, d.dropout,
                           (unsigned long long)d.dropout_seed, 2u, (long)p0);
        SAT_TRY(launch_ok("output dropout"));
        uin = ud;
    }
    if (w.Wb_out) {          // bf16 mode: vocabulary projection on bf16 copies of both operands.  The weight copy is made at the first flush of a
                             // forward pass (p0 == 0) and again at every later one only under weight tying with max-norm renormalisation: the
                             // embedding table IS the output weight then and is edited during the loop (scheduled sampling flushes every step)
        if (p0 == 0 || (p.out_w == p.embedding && d.embed_max_norm > 0.f)) SAT_TRY(cast_bf16(st, p.out_w, w.Wb_out, (long)d.V * d.m));
        SAT_TRY(cast_bf16(st, uin, w.Ub + (long)p0 * d.m, (long)rows * d.m));
        return gemm_bb_nt(st, w.Ub + (long)p0 * d.m, d.m, w.Wb_out, d.m, logits + (long)p0 * d.V, d.V, rows, d.V, d.m, p.out_b ? EPI_BIAS : EPI_NONE, p.out_b);
    }
    return gemm(st, A_ROW, B_ROW, uin, d.m, p.out_w, d.m, logits + (long)p0 * d.V, d.V, rows, d.V, d.m, 0,
                p.out_b ? EPI_BIAS : EPI_NONE, p.out_b);
}

int decoder_fwd(const sat_decoder_dims& d, const sat_decoder_params& p, const sat_decoder_batch& b, float* logits, float* alphas,
                       char* ws, size_t ws_bytes, hipStream_t st) {
    Ws w = layout(d, ws);
    t_bf16_mfma = d.precision ? 1 : 0;
    SAT_REQUIRE(ws_bytes >= w.total, "decoder_fwd: workspace %zu < %zu bytes", ws_bytes, w.total);
    SAT_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "decoder: workspace must be 256-byte aligned");
    const int N = d.B * d.R, T1 = d.T - 1, HCW = d.A + d.D + 4 * d.n, n = d.n, A = d.A, D = d.D, m = d.m;
    const int NL = d.layers, top = NL - 1;
    const long LS = (long)(T1 + 1) * N * n;                                   // layer stride of H_all / C_all
    auto Hs = [&](int t, int l) { return w.H_all + l * LS + (long)t * N * n; };
    auto Cs = [&](int t, int l) { return w.C_all + l * LS + (long)t * N * n; };
    auto GUs = [&](int t, int l) { return w.GU + ((long)(l - 1) * T1 + t) * N * 4 * n; };
    const int ts = live_steps(d, b);
    SAT_REQUIRE(b.step_offsets_host[T1] == d.P, "decoder: step_offsets[T-1]=%d != P=%d", b.step_offsets_host[T1], d.P);
    for (int t = 0; t < ts; ++t) SAT_REQUIRE(b.teacher_host[t] || t > 0, "decoder: step 0 must be teacher forced");

    // packed parameters: Wcat = [W_d ; W_beta ; W_hh] (HCW, n), bcat = [0 ; b_beta ; b_ih + b_hh]
    const bool use_b = w.Hb != nullptr && attention_split_enabled();
    hipLaunchKernelGGL(pack_wcat_kernel, dim3(cdiv((long)HCW * n + HCW, 256)), dim3(256), 0, st, p.att_dec, p.beta_w, p.w_hh, p.beta_b, p.b_ih, p.b_hh, w.Wcat,
                       use_b ? w.Wcat_b : (__bf16*)nullptr, w.bcat, A, D, n);
    SAT_TRY(launch_ok("pack Wcat"));
    for (int l = 1; l < NL; ++l) {
        hipLaunchKernelGGL(add_kernel, dim3(cdiv(4 * n, 256)), dim3(256), 0, st, w.bup + (long)(l - 1) * 4 * n, p.up_b_ih[l - 1], p.up_b_hh[l - 1], (long)4 * n);
        SAT_TRY(launch_ok("bias add (stacked layer)"));
    }

    // att_enc, hoisted (model.py:100): U = ann * W_e^T once per image
    SAT_TRY(gemm(st, A_ROW, B_ROW, b.ann, D, p.att_enc, D, w.U, A, d.B * d.L, A, D));
    // InitLSTM (model.py:76-81) on the B images, then the raw reshape over the repeated rows (F3)
    hipLaunchKernelGGL(ann_mean_kernel, dim3(d.B), dim3(256), 0, st, b.ann, w.mean, d.L, D);
    SAT_TRY(launch_ok("ann_mean"));
    if (d.dropout > 0.f) {
        // dropout acts on the mean of the repeated annotations: every caption row has its own mask, so run the
        // two Linears on N rows; the (N, 2n) result IS the (2, N, n) state buffer (raw reshape, F3)
        hipLaunchKernelGGL(init_mean_rows_kernel, dim3(cdiv((long)N * D, 256)), dim3(256), 0, st, w.mean, w.mean_rows, N, d.R, D, d.dropout, (unsigned long long)d.dropout_seed);
        SAT_TRY(launch_ok("init_mean_rows"));
        SAT_TRY(gemm(st, A_ROW, B_ROW, w.mean_rows, D, p.init_f_w, D, w.f_rows, m, N, m, D, 0, EPI_BIAS, p.init_f_b));
        SAT_TRY(gemm(st, A_ROW, B_ROW, w.f_rows, m, p.init_i_w, m, w.init_rows, 2 * n * NL, N, 2 * n * NL, m, 0, EPI_BIAS, p.init_i_b));
        for (int l = 0; l < NL; ++l) {
            SAT_TRY(dev_copy_bytes(st, Hs(0, l), w.init_rows + (long)l * N * n, (size_t)N * n * 4));
            SAT_TRY(dev_copy_bytes(st, Cs(0, l), w.init_rows + (long)(NL + l) * N * n, (size_t)N * n * 4));
        }
    } else {
        SAT_TRY(gemm(st, A_ROW, B_ROW, w.mean, D, p.init_f_w, D, w.f, m, d.B, m, D, 0, EPI_BIAS, p.init_f_b));
        SAT_TRY(gemm(st, A_ROW, B_ROW, w.f, m, p.init_i_w, m, w.init_img, 2 * n * NL, d.B, 2 * n * NL, m, 0, EPI_BIAS, p.init_i_b));
        hipLaunchKernelGGL(init_expand_kernel, dim3(cdiv(2L * NL * N * n, 256)), dim3(256), 0, st, w.init_img, w.H_all, w.C_all, N, d.R, n, NL, LS);
        SAT_TRY(launch_ok("init_expand"));
    }

    // bf16 mode, one layer: the per-step GEMMs read bf16 copies (state written by the cell kernel, gated context by the attention
    // kernel, weights cast here once) through the direct-to-LDS kernel instead of rounding fp32 operands in registers every step
    const __bf16* annb = (w.annb && attention_split_enabled() && ann_bf16_enabled()) ? w.annb : nullptr;
    if (annb) SAT_TRY(cast_bf16(st, b.ann, w.annb, (long)d.B * d.L * D));
    if (use_b) {
        hipLaunchKernelGGL(cast_block_bf16_kernel, dim3(cdiv((long)4 * n * (D / 4), 256)), dim3(256), 0, st, p.w_ih + m, (long)(m + D), w.Wz_b, 4 * n, D);
        SAT_TRY(launch_ok("cast W_ih[:, m:]"));
        SAT_TRY(cast_bf16(st, Hs(0, 0), w.Hb, (long)N * n));
    }
    // alphas of steps that never run stay zero (model.py:506)
    if (ts < T1) SAT_TRY(dev_fill_bytes(st, alphas, 0, (size_t)N * T1 * d.L * 4));

    // tokens + embeddings + the embedding half of the LSTM input GEMM for every teacher-forced step, in one batch
    SAT_TRY(dev_fill_bytes(st, w.Tok, 0xFF, (size_t)T1 * N * 4));       // -1: no token
    for (int t = 0; t < ts;) {          // one launch per run of teacher-forced steps (the whole caption when epsilon = 1)
        if (!b.teacher_host[t]) { ++t; continue; }
        int t1 = t + 1;
        while (t1 < ts && b.teacher_host[t1]) ++t1;
        hipLaunchKernelGGL(teacher_tokens_kernel, dim3(cdiv(N, 256), t1 - t), dim3(256), 0, st, b.caps, b.lengths, w.Tok + (long)t * N, N, d.T, t);
        SAT_TRY(launch_ok("teacher_tokens"));
        t = t1;
    }
    auto renorm = [&](const int* tok, int count) -> int {
        if (!(d.embed_max_norm > 0.f)) return SAT_OK;
        hipLaunchKernelGGL(embedding_mark_kernel, dim3(cdiv(count, 256)), dim3(256), 0, st, tok, count, w.flags, d.V);
        SAT_TRY(launch_ok("embedding_mark"));
        hipLaunchKernelGGL(embedding_renorm_kernel, dim3(d.V), dim3(64), 0, st, p.embedding, w.flags, m, d.embed_max_norm);
        return launch_ok("embedding_renorm");
    };
    if (d.embed_max_norm > 0.f) SAT_TRY(dev_fill_bytes(st, w.flags, 0, (size_t)d.V * 4));
    if (ts > 0) {
        SAT_TRY(renorm(w.Tok, ts * N));
        hipLaunchKernelGGL(gather_rows_kernel, dim3(ts * N), dim3(64), 0, st, p.embedding, w.Tok, w.Y, ts * N, m, d.embedding_dropout, (unsigned long long)d.dropout_seed, 0L);
        SAT_TRY(launch_ok("embedding gather"));
        SAT_TRY(gemm(st, A_ROW, B_ROW, w.Y, m, p.w_ih, m + D, w.GY, 4 * n, ts * N, 4 * n, m));
    }

    int pending = 0;
    for (int t = 0; t < ts; ++t) {
        float* hc = w.HC + (long)t * N * HCW;
        if (!b.teacher_host[t]) {
            // scheduled sampling (model.py:521-523): feed argmax of the previous step's logits
            SAT_TRY(flush_outputs(st, d, p, b, w, logits, b.step_offsets_host[pending], b.step_offsets_host[t]));
            pending = t;
            if (d.embed_max_norm > 0.f) {       // the chosen rows are renormalised in place between the decision and the gather (model.py:158-164)
                hipLaunchKernelGGL(argmax_tokens_kernel, dim3(N), dim3(256), 0, st, logits, b.prow + (long)(t - 1) * N, b.lengths, w.Tok + (long)t * N, d.V, t);
                SAT_TRY(launch_ok("argmax_tokens"));
                SAT_TRY(renorm(w.Tok + (long)t * N, N));
                hipLaunchKernelGGL(gather_rows_kernel, dim3(N), dim3(64), 0, st, p.embedding, w.Tok + (long)t * N, w.Y + (long)t * N * m, N, m, d.embedding_dropout, (unsigned long long)d.dropout_seed, (long)t * N);
                SAT_TRY(launch_ok("embedding gather"));
            } else {                            // decision + embedding row in one launch
                hipLaunchKernelGGL(argmax_gather_kernel, dim3(N), dim3(256), 0, st, logits, b.prow + (long)(t - 1) * N, b.lengths, w.Tok + (long)t * N, d.V, t, p.embedding,
                                   w.Y + (long)t * N * m, m, d.embedding_dropout, (unsigned long long)d.dropout_seed, (long)t * N);
                SAT_TRY(launch_ok("argmax + embedding gather"));
            }
            SAT_TRY(gemm(st, A_ROW, B_ROW, w.Y + (long)t * N * m, m, p.w_ih, m + D, w.GY + (long)t * N * 4 * n, 4 * n, N, 4 * n, m));
        }
        // [q | beta | gates_h] = h_{t-1} * Wcat^T + bcat, sigmoid on the beta columns.  Attention and the gate read the TOP
        // layer's state (h[-1], model.py:533,538); the recurrent term of layer 0 reads layer 0's.
        if (use_b) {
            SAT_TRY(gemm_bb_nt(st, w.Hb + (long)t * N * n, n, w.Wcat_b, n, hc, HCW, N, HCW, n, EPI_BIAS_SIGMOID_RANGE, w.bcat, 0, A, A + D));
        } else if (NL == 1) {
            SAT_TRY(gemm(st, A_ROW, B_ROW, Hs(t, 0), n, w.Wcat, n, hc, HCW, N, HCW, n, 0, EPI_BIAS_SIGMOID_RANGE, w.bcat,
                         nullptr, nullptr, nullptr, 0, A, A + D));
        } else {
            SAT_TRY(gemm(st, A_ROW, B_ROW, Hs(t, top), n, w.Wcat, n, hc, HCW, N, A + D, n, 0, EPI_BIAS_SIGMOID_RANGE, w.bcat,
                         nullptr, nullptr, nullptr, 0, A, A + D));
            SAT_TRY(gemm(st, A_ROW, B_ROW, Hs(t, 0), n, w.Wcat + (long)(A + D) * n, n, hc + A + D, HCW, N, 4 * n, n, 0, EPI_BIAS, w.bcat + A + D));
        }
        SAT_TRY(launch_attention_fwd(st, b.ann, w.U, hc, HCW, p.att_f, b.lengths, t, alphas, T1, w.Z + (long)t * N * D, w.XZ + (long)t * N * D,
                                     d.B, d.R, d.L, D, A, w.SC, use_b ? w.XZb : nullptr, annb));
        // gates += (beta*z) * W_ih[:, m:]^T
        if (use_b) SAT_TRY(gemm_bb_nt(st, w.XZb, D, w.Wz_b, D, hc + A + D, HCW, N, 4 * n, D, EPI_NONE, nullptr, 1));
        else SAT_TRY(gemm(st, A_ROW, B_ROW, w.XZ + (long)t * N * D, D, p.w_ih + m, m + D, hc + A + D, HCW, N, 4 * n, D, 1));
        hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(cdiv((long)N * n, 256)), dim3(256), 0, st, hc + A + D, HCW, w.GY + (long)t * N * 4 * n,
                           Cs(t, 0), Hs(t, 0), Cs(t + 1, 0), Hs(t + 1, 0), b.lengths, t, N, n, use_b ? w.Hb + (long)(t + 1) * N * n : (__bf16*)nullptr);
        SAT_TRY(launch_ok("lstm_cell_fwd"));
        for (int l = 1; l < NL; ++l) {          // stacked layers: input = the layer below's new h (nn.LSTM, no inter-layer dropout: model.py:175-180)
            float* gu = GUs(t, l);
            SAT_TRY(gemm(st, A_ROW, B_ROW, Hs(t + 1, l - 1), n, p.up_w_ih[l - 1], n, gu, 4 * n, N, 4 * n, n, 0, EPI_BIAS, w.bup + (long)(l - 1) * 4 * n));
            SAT_TRY(gemm(st, A_ROW, B_ROW, Hs(t, l), n, p.up_w_hh[l - 1], n, gu, 4 * n, N, 4 * n, n, 1));
            hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(cdiv((long)N * n, 256)), dim3(256), 0, st, gu, 4 * n, (const float*)nullptr,
                               Cs(t, l), Hs(t, l), Cs(t + 1, l), Hs(t + 1, l), b.lengths, t, N, n);
            SAT_TRY(launch_ok("lstm_cell_fwd (stacked layer)"));
        }
    }
    return flush_outputs(st, d, p, b, w, logits, b.step_offsets_host[pending], b.step_offsets_host[ts]);
}

int decoder_bwd(const sat_decoder_dims& d, const sat_decoder_params& p, const sat_decoder_batch& b, const float* dlogits,
                       const float* alphas, const float* dalphas, const sat_decoder_params& g, float* dann, char* ws, size_t ws_bytes, hipStream_t st) {
    Ws w = layout(d, ws);
    t_bf16_mfma = d.precision ? 1 : 0;
    SAT_REQUIRE(ws_bytes >= w.total, "decoder_bwd: workspace %zu < %zu bytes", ws_bytes, w.total);
    const int N = d.B * d.R, T1 = d.T - 1, HCW = d.A + d.D + 4 * d.n, n = d.n, A = d.A, D = d.D, m = d.m, V = d.V, P = d.P;
    const int ts = live_steps(d, b);
    const int KR = ts * N;                              // rows of the time-major padded buffers that were touched
    const int NL = d.layers, top = NL - 1;
    const long LS = (long)(T1 + 1) * N * n;
    auto Hs = [&](int t, int l) { return w.H_all + l * LS + (long)t * N * n; };
    auto Cs = [&](int t, int l) { return w.C_all + l * LS + (long)t * N * n; };
    auto GUs = [&](int t, int l) { return w.GU + ((long)(l - 1) * T1 + t) * N * 4 * n; };
    auto DGUs = [&](int t, int l) { return w.DGU + ((long)(l - 1) * T1 + t) * N * 4 * n; };
    auto dHcs = [&](int l) { return w.dHc + (long)l * N * n; };
    auto dCcs = [&](int l) { return w.dCc + (long)l * N * n; };
    float* const slab = w.slab; const long se = w.slab_elems;

    SAT_TRY(dev_fill_bytes(st, w.zero_begin, 0, (size_t)(w.zero_end - w.zero_begin)));       // dHout, dZout, dHc, dCc, dU, dwf_part
    if (NL > 1 && KR < T1 * N) SAT_TRY(dev_fill_bytes(st, w.DGU, 0, (size_t)(NL - 1) * T1 * N * 4 * n * 4));

    // ---- output layer, all packed rows at once (DeepOutput backward)
    if (P > 0) {
        SAT_TRY(gemm(st, A_ROW, B_KMAJOR, dlogits, V, p.out_w, m, w.dA, m, P, m, V, 0, d.deep_output ? EPI_MUL_DTANH : EPI_NONE, nullptr,
                     nullptr, nullptr, w.Uact, m, 0, 0, slab, se));           // few output tiles, long reduction over the vocabulary: split-K
        if (d.dropout > 0.f) {
            hipLaunchKernelGGL(dropout_rows_kernel, dim3(cdiv((long)P * m, 256)), dim3(256), 0, st, w.dA, w.dA, (long)P * m, m, d.dropout,
                               (unsigned long long)d.dropout_seed, 2u, 0L);
            SAT_TRY(launch_ok("output dropout bwd"));
        }
        SAT_TRY(gemm(st, A_KMAJOR, B_KMAJOR, dlogits, V, d.dropout > 0.f ? w.Udrop : w.Uact, m, g.out_w, m, V, m, P, 0, EPI_NONE, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, slab, se));
        if (g.out_b) SAT_TRY(colsum(st, w, dlogits, V, P, V, g.out_b));
        SAT_TRY(gemm(st, A_ROW, B_KMAJOR, w.dA, m, p.out_hidden, n, w.dHout, n, P, n, m, 0, EPI_NONE, nullptr, nullptr, b.src_row));
        if (d.deep_output)
            SAT_TRY(gemm(st, A_ROW, B_KMAJOR, w.dA, m, p.out_context, D, w.dZout, D, P, D, m, 0, EPI_NONE, nullptr, nullptr, b.src_row));
    } else {
        SAT_TRY(dev_fill_bytes(st, g.out_w, 0, (size_t)V * m * 4));
        if (g.out_b) SAT_TRY(dev_fill_bytes(st, g.out_b, 0, (size_t)V * 4));
    }
    // dA in time-major padded rows (zeros for finished captions): operand of the weight-grad GEMMs and of dY
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(T1 * N), dim3(64), 0, st, w.dA, b.prow, w.dY, T1 * N, m);
    SAT_TRY(launch_ok("scatter dA"));
    const float* H1 = Hs(1, top);
    SAT_TRY(gemm(st, A_KMAJOR, B_KMAJOR, w.dY, m, H1, n, g.out_hidden, n, m, n, KR, 0, EPI_NONE, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, slab, se));
    if (d.deep_output)
        SAT_TRY(gemm(st, A_KMAJOR, B_KMAJOR, w.dY, m, w.Z, D, g.out_context, D, m, D, KR, 0, EPI_NONE, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, slab, se));
    else
        SAT_TRY(dev_fill_bytes(st, w.dY, 0, (size_t)T1 * N * m * 4));     // shallow output does not see the embedding

    // ---- back through time
    const size_t lds_b = att_bwd_lds(d.L, A, D);
    SAT_REQUIRE(lds_b <= 160 * 1024, "attention_bwd: L=%d A=%d D=%d need %zu B of LDS (> 160 KiB)", d.L, A, D, lds_b);
    static const int att_fused = getenv("SAT_ATT_FUSED") ? atoi(getenv("SAT_ATT_FUSED")) : 0;
    const bool use_b = w.Hb != nullptr && !att_fused;           // bf16 operand copies (the forward cast the weights into Wcat_b / Wz_b)
    const __bf16* annb = (w.annb && !att_fused && ann_bf16_enabled()) ? w.annb : nullptr;      // the forward left the bf16 annotations there
    typedef void (*attb_fn)(const float*, const float*, const float*, int, const float*, const int*, int, const float*, const float*, int, const float*,
                            const float*, const float*, float*, float*, int, float*, float*, int, int, int, int);
    attb_fn attb = nullptr;
    switch (d.R < ATT_RMAX ? d.R : ATT_RMAX) {           // rows per pass are compile-time in the kernel
        case 1: attb = attention_bwd_kernel<1>; break; case 2: attb = attention_bwd_kernel<2>; break; case 3: attb = attention_bwd_kernel<3>; break;
        case 4: attb = attention_bwd_kernel<4>; break; case 5: attb = attention_bwd_kernel<5>; break; case 6: attb = attention_bwd_kernel<6>; break;
        case 7: attb = attention_bwd_kernel<7>; break; default: attb = attention_bwd_kernel<8>; break;
    }
    SAT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attb), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
    for (int t = ts - 1; t >= 0; --t) {
        const float* hc = w.HC + (long)t * N * HCW;
        float* dhc = w.DHC + (long)t * N * HCW;
        for (int l = top; l >= 1; --l) {        // stacked layers, top down: gate grads, then into the layer below and the own past
            float* dgu = DGUs(t, l);
            hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(cdiv((long)N * n, 256)), dim3(256), 0, st, GUs(t, l), 4 * n, Cs(t, l), Cs(t + 1, l),
                               l == top ? w.dHout + (long)t * N * n : (const float*)nullptr, dHcs(l), dCcs(l), dgu, 4 * n, b.lengths, t, N, n);
            SAT_TRY(launch_ok("lstm_cell_bwd (stacked layer)"));
            SAT_TRY(gemm(st, A_ROW, B_KMAJOR, dgu, 4 * n, p.up_w_ih[l - 1], n, dHcs(l - 1), n, N, n, 4 * n, 1, EPI_NONE, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, slab, se));
            SAT_TRY(gemm(st, A_ROW, B_KMAJOR, dgu, 4 * n, p.up_w_hh[l - 1], n, dHcs(l), n, N, n, 4 * n, 1, EPI_NONE, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, slab, se));
        }
        hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(cdiv((long)N * n, 256)), dim3(256), 0, st, hc + A + D, HCW, Cs(t, 0),
                           Cs(t + 1, 0), top == 0 ? w.dHout + (long)t * N * n : (const float*)nullptr, dHcs(0), dCcs(0), dhc + A + D, HCW, b.lengths, t, N, n,
                           use_b ? w.DHCb + A + D : (__bf16*)nullptr);
        SAT_TRY(launch_ok("lstm_cell_bwd"));
        // d(beta*z) = dG * W_ih[:, m:]
        if (use_b) SAT_TRY(gemm_bb_nn(st, w.DHCb + A + D, HCW, w.Wz_b, D, w.dXZ, D, N, D, 4 * n, 0, slab, se));
        else SAT_TRY(gemm(st, A_ROW, B_KMAJOR, dhc + A + D, HCW, p.w_ih + m, m + D, w.dXZ, D, N, D, 4 * n, 0, EPI_NONE, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, slab, se));
        if (!att_fused) {
            SAT_TRY(attention_bwd_split(d.R < ATT_RMAX ? d.R : ATT_RMAX, st, b.ann, w.U, hc, HCW, p.att_f, b.lengths, t, alphas, dalphas, T1, w.Z + (long)t * N * D,
                                        w.dZout + (long)t * N * D, w.dXZ, w.DZ + (long)t * N * D, dhc, HCW, w.dU, w.dwf_part, w.DA, d.B, d.R, d.L, D, A,
                                        use_b ? w.DHCb : nullptr, annb));
        } else
        hipLaunchKernelGGL(attb, dim3(d.B), dim3(ATTB_THREADS), lds_b, st, b.ann, w.U, hc, HCW, p.att_f, b.lengths, t, alphas,
                           dalphas, T1, w.Z + (long)t * N * D, w.dZout + (long)t * N * D, w.dXZ, w.DZ + (long)t * N * D, dhc, HCW, w.dU, w.dwf_part,
                           d.R, d.L, D, A);
        SAT_TRY(launch_ok("attention_bwd"));
        // dh_{t-1} += [dq | dbeta_pre | dG] * Wcat
        if (use_b) {
            SAT_TRY(gemm_bb_nn(st, w.DHCb, HCW, w.Wcat_b, n, w.dHc, n, N, n, HCW, 1, slab, se));
        } else if (NL == 1) {
            SAT_TRY(gemm(st, A_ROW, B_KMAJOR, dhc, HCW, w.Wcat, n, w.dHc, n, N, n, HCW, 1, EPI_NONE, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, slab, se));
        } else {
            SAT_TRY(gemm(st, A_ROW, B_KMAJOR, dhc, HCW, w.Wcat, n, dHcs(top), n, N, n, A + D, 1, EPI_NONE, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, slab, se));
            SAT_TRY(gemm(st, A_ROW, B_KMAJOR, dhc + A + D, HCW, w.Wcat + (long)(A + D) * n, n, dHcs(0), n, N, n, 4 * n, 1, EPI_NONE, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, slab, se));
        }
    }
    // now dHc = dL/dh0 and dCc = dL/dc0

    // ---- weight gradients, batched over every executed step (reduction length KR = ts*N)
    auto wgrad = [&](const float* dy, long ldy, const float* x, long ldx, float* out, long ldo, int M, int Nn) {
        return gemm(st, A_KMAJOR, B_KMAJOR, dy, ldy, x, ldx, out, ldo, M, Nn, KR, 0, EPI_NONE, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, slab, se);
    };
    const float* dG = w.DHC + A + D;
    SAT_TRY(wgrad(w.DHC, HCW, Hs(0, top), n, g.att_dec, n, A, n));
    SAT_TRY(wgrad(w.DHC + A, HCW, Hs(0, top), n, g.beta_w, n, D, n));
    SAT_TRY(wgrad(dG, HCW, Hs(0, 0), n, g.w_hh, n, 4 * n, n));
    for (int l = 1; l < NL; ++l) {
        const float* dgu = DGUs(0, l);
        SAT_TRY(wgrad(dgu, 4 * n, Hs(1, l - 1), n, g.up_w_ih[l - 1], n, 4 * n, n));
        SAT_TRY(wgrad(dgu, 4 * n, Hs(0, l), n, g.up_w_hh[l - 1], n, 4 * n, n));
        SAT_TRY(colsum(st, w, dgu, 4 * n, KR, 4 * n, g.up_b_ih[l - 1]));
        SAT_TRY(dev_copy_bytes(st, g.up_b_hh[l - 1], g.up_b_ih[l - 1], (size_t)4 * n * 4));
    }
    SAT_TRY(colsum(st, w, w.DHC + A, HCW, KR, D, g.beta_b));
    SAT_TRY(colsum(st, w, dG, HCW, KR, 4 * n, g.b_ih));
    SAT_TRY(dev_copy_bytes(st, g.b_hh, g.b_ih, (size_t)4 * n * 4));
    SAT_TRY(wgrad(dG, HCW, w.Y, m, g.w_ih, m + D, 4 * n, m));
    SAT_TRY(wgrad(dG, HCW, w.XZ, D, g.w_ih + m, m + D, 4 * n, D));
    // embedding: dY = dA (deep output) + dG * W_ih[:, :m], scattered into the table (padding row skipped)
    SAT_TRY(gemm(st, A_ROW, B_KMAJOR, dG, HCW, p.w_ih, m + D, w.dY, m, KR, m, 4 * n, 1));
    if (d.embedding_dropout > 0.f && KR > 0) {
        hipLaunchKernelGGL(dropout_rows_kernel, dim3(cdiv((long)KR * m, 256)), dim3(256), 0, st, w.dY, w.dY, (long)KR * m, m, d.embedding_dropout,
                           (unsigned long long)d.dropout_seed, 1u, 0L);
        SAT_TRY(launch_ok("embedding dropout bwd"));
    }
    if (KR > 0) {
        SAT_TRY(dev_fill_bytes(st, w.emb_count, 0, (size_t)V * 4));
        hipLaunchKernelGGL(embedding_count_kernel, dim3(cdiv(KR, 256)), dim3(256), 0, st, w.Tok, KR, V, d.padding_idx, w.emb_count);
        hipLaunchKernelGGL(embedding_scan_kernel, dim3(1), dim3(1024), 0, st, w.emb_count, V, w.emb_offset, w.emb_cursor);
        hipLaunchKernelGGL(embedding_place_kernel, dim3(cdiv(KR, 256)), dim3(256), 0, st, w.Tok, KR, V, d.padding_idx, w.emb_offset, w.emb_cursor, w.emb_list);
        hipLaunchKernelGGL(embedding_sum_kernel, dim3(V), dim3(256), 0, st, w.dY, w.Tok, w.emb_offset, w.emb_list, g.embedding, KR, m, d.padding_idx);
        SAT_TRY(launch_ok("embedding gradient"));
    } else SAT_TRY(dev_fill_bytes(st, g.embedding, 0, (size_t)V * m * 4));
    // attention parameters and the annotation gradient
    SAT_TRY(colsum(st, w, w.dwf_part, A, d.B, A, g.att_f));
    SAT_TRY(gemm(st, A_KMAJOR, B_KMAJOR, w.dU, A, b.ann, D, g.att_enc, D, A, D, d.B * d.L, 0, EPI_NONE, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, slab, se));
    SAT_TRY(gemm(st, A_ROW, B_KMAJOR, w.dU, A, p.att_enc, D, dann, D, d.B * d.L, D, A));
    {
        // location slab = NQ float4 per LDS row: 13 covers L <= 52 (7x7 maps) in one pass, 16 the 8x8 / 14x14 maps
        const int lq4 = (d.L + 3) / 4;
        const int NQ = (lq4 <= 13) ? 13 : 16;
        const int Lq = (lq4 + NQ - 1) / NQ * NQ;
        const size_t lds_dann = (size_t)d.R * T1 * Lq * 16;
        SAT_REQUIRE(lds_dann <= 160 * 1024, "dann_from_context: R*(T-1)*L = %d floats of alphas do not fit the LDS", d.R * T1 * d.L);
        if (NQ == 13) {
            SAT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dann_from_context_kernel<13>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dann));
            hipLaunchKernelGGL(dann_from_context_kernel<13>, dim3(d.B, cdiv(D, 256)), dim3(256), lds_dann, st, alphas, w.DZ, b.lengths, dann, 1, d.R, N, T1, d.L, D);
        } else {
            SAT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dann_from_context_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dann));
            hipLaunchKernelGGL(dann_from_context_kernel<16>, dim3(d.B, cdiv(D, 256)), dim3(256), lds_dann, st, alphas, w.DZ, b.lengths, dann, 1, d.R, N, T1, d.L, D);
        }
    }
    SAT_TRY(launch_ok("dann_from_context"));
    // InitLSTM backward (the raw reshape is a reinterpretation: gradients of the repeated rows add up per image)
    if (d.dropout > 0.f) {                 // per-caption-row path (see decoder_fwd)
        const int n2 = 2 * n * NL;
        SAT_TRY(dev_copy_bytes(st, w.init_rows, w.dHc, (size_t)NL * N * n * 4));
        SAT_TRY(dev_copy_bytes(st, w.init_rows + (long)NL * N * n, w.dCc, (size_t)NL * N * n * 4));
        SAT_TRY(gemm(st, A_KMAJOR, B_KMAJOR, w.init_rows, n2, w.f_rows, m, g.init_i_w, m, n2, m, N));
        SAT_TRY(colsum(st, w, w.init_rows, n2, N, n2, g.init_i_b));
        SAT_TRY(gemm(st, A_ROW, B_KMAJOR, w.init_rows, n2, p.init_i_w, m, w.df_rows, m, N, m, n2));
        SAT_TRY(gemm(st, A_KMAJOR, B_KMAJOR, w.df_rows, m, w.mean_rows, D, g.init_f_w, D, m, D, N));
        SAT_TRY(colsum(st, w, w.df_rows, m, N, m, g.init_f_b));
        SAT_TRY(gemm(st, A_ROW, B_KMAJOR, w.df_rows, m, p.init_f_w, D, w.mean_rows, D, N, D, m));          // d(mean rows), reuses the buffer
        hipLaunchKernelGGL(init_mean_rows_bwd_kernel, dim3(cdiv((long)d.B * D, 256)), dim3(256), 0, st, w.mean_rows, w.dmean, d.B, d.R, D, d.dropout,
                           (unsigned long long)d.dropout_seed);
        SAT_TRY(launch_ok("init_mean_rows_bwd"));
    } else {
    const int n2 = 2 * n * NL;
    hipLaunchKernelGGL(init_expand_bwd_kernel, dim3(cdiv((long)d.B * n2, 256)), dim3(256), 0, st, w.dHc, w.dCc, w.dinit_img, d.B, d.R, n * NL);
    SAT_TRY(launch_ok("init_expand_bwd"));
    SAT_TRY(gemm(st, A_KMAJOR, B_KMAJOR, w.dinit_img, n2, w.f, m, g.init_i_w, m, n2, m, d.B));
    SAT_TRY(colsum(st, w, w.dinit_img, n2, d.B, n2, g.init_i_b));
    SAT_TRY(gemm(st, A_ROW, B_KMAJOR, w.dinit_img, n2, p.init_i_w, m, w.df, m, d.B, m, n2));
    SAT_TRY(gemm(st, A_KMAJOR, B_KMAJOR, w.df, m, w.mean, D, g.init_f_w, D, m, D, d.B));
    SAT_TRY(colsum(st, w, w.df, m, d.B, m, g.init_f_b));
    SAT_TRY(gemm(st, A_ROW, B_KMAJOR, w.df, m, p.init_f_w, D, w.dmean, D, d.B, D, m));
    }
    const long tot = (long)d.B * d.L * D;
    hipLaunchKernelGGL(dann_add_mean_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, st, dann, w.dmean, d.L, D, tot);
    return launch_ok("dann_add_mean");
}


int colsum_public(const float* x, long ld, long rows, int cols, float* out, float* scratch, hipStream_t st) {
    Ws w; w.colpart = scratch;
    return colsum(st, w, x, ld, (int)rows, cols, out);
}

size_t decoder_workspace_bytes(const sat_decoder_dims& d) { return layout(d, nullptr).total; }

// ------------------------------------------------------------------ inference: one image, K live beams
struct InferWs { size_t total; float *U, *Wcat, *bcat, *mean, *f, *init_img, *hc, *Z, *XZ, *Y, *u, *gu, *bup, *sc; int* ones; };
static InferWs infer_layout(const sat_decoder_dims& d, int Kmax, char* base) {
    InferWs w; size_t off = 0;
    const long HCW = d.A + d.D + 4L * d.n;
    auto take = [&](size_t elems) { size_t o = off; off += (elems * 4 + 255) & ~(size_t)255; return base ? base + o : (char*)nullptr; };
    w.U = (float*)take((size_t)d.L * d.A); w.Wcat = (float*)take((size_t)HCW * d.n); w.bcat = (float*)take((size_t)HCW);
    w.mean = (float*)take(d.D); w.f = (float*)take(d.m); w.init_img = (float*)take(2 * (size_t)d.n * d.layers);
    w.gu = (float*)take((size_t)Kmax * 4 * d.n); w.bup = (float*)take((size_t)d.layers * 4 * d.n);
    w.hc = (float*)take((size_t)Kmax * HCW); w.Z = (float*)take((size_t)Kmax * d.D); w.XZ = (float*)take((size_t)Kmax * d.D);
    w.Y = (float*)take((size_t)Kmax * d.m); w.u = (float*)take((size_t)Kmax * d.m); w.ones = (int*)take(Kmax); w.sc = (float*)take((size_t)Kmax * d.L);
    w.total = off;
    return w;
}
size_t decoder_infer_workspace_bytes(const sat_decoder_dims& d, int Kmax) { return infer_layout(d, Kmax, nullptr).total; }

// model.py:255-281 for one image: att_enc, the stacked step weights, and the initial state of K beams (F3 reshape)
int decoder_infer_begin(const sat_decoder_dims& d, const sat_decoder_params& p, const float* ann, int K, int Kmax, float* h, float* c,
                        char* ws, size_t ws_bytes, hipStream_t st) {
    InferWs w = infer_layout(d, Kmax, ws);
    SAT_REQUIRE(ws_bytes >= w.total && K >= 1 && K <= Kmax, "decoder_infer_begin: workspace %zu < %zu or bad beam count %d/%d", ws_bytes, w.total, K, Kmax);
    t_bf16_mfma = d.precision ? 1 : 0;
    const int n = d.n, A = d.A, D = d.D, m = d.m;
    SAT_TRY(dev_copy_bytes(st, w.Wcat, p.att_dec, (size_t)A * n * 4));
    SAT_TRY(dev_copy_bytes(st, w.Wcat + (long)A * n, p.beta_w, (size_t)D * n * 4));
    SAT_TRY(dev_copy_bytes(st, w.Wcat + (long)(A + D) * n, p.w_hh, (size_t)4 * n * n * 4));
    SAT_TRY(dev_fill_bytes(st, w.bcat, 0, (size_t)A * 4));
    SAT_TRY(dev_copy_bytes(st, w.bcat + A, p.beta_b, (size_t)D * 4));
    hipLaunchKernelGGL(add_kernel, dim3(cdiv(4 * n, 256)), dim3(256), 0, st, w.bcat + A + D, p.b_ih, p.b_hh, (long)4 * n);
    SAT_TRY(launch_ok("bias add"));
    for (int l = 1; l < d.layers; ++l) {
        hipLaunchKernelGGL(add_kernel, dim3(cdiv(4 * n, 256)), dim3(256), 0, st, w.bup + (long)(l - 1) * 4 * n, p.up_b_ih[l - 1], p.up_b_hh[l - 1], (long)4 * n);
        SAT_TRY(launch_ok("bias add (stacked layer)"));
    }
    hipLaunchKernelGGL(fill_int_kernel, dim3(cdiv(Kmax, 256)), dim3(256), 0, st, w.ones, (long)Kmax, 1);
    SAT_TRY(launch_ok("fill ones"));
    SAT_TRY(gemm(st, A_ROW, B_ROW, ann, D, p.att_enc, D, w.U, A, d.L, A, D));
    hipLaunchKernelGGL(ann_mean_kernel, dim3(1), dim3(256), 0, st, ann, w.mean, d.L, D);
    SAT_TRY(launch_ok("ann_mean"));
    SAT_TRY(gemm(st, A_ROW, B_ROW, w.mean, D, p.init_f_w, D, w.f, m, 1, m, D, 0, EPI_BIAS, p.init_f_b));
    SAT_TRY(gemm(st, A_ROW, B_ROW, w.f, m, p.init_i_w, m, w.init_img, 2 * n * d.layers, 1, 2 * n * d.layers, m, 0, EPI_BIAS, p.init_i_b));
    hipLaunchKernelGGL(init_expand_kernel, dim3(cdiv(2L * d.layers * K * n, 256)), dim3(256), 0, st, w.init_img, h, c, K, K, n, d.layers,
                       (long)K * n);   // one image, R = K rows
    return launch_ok("init_expand");
}

// model.py:298-327: embedding -> attention -> beta gate -> LSTM -> deep output for the K live beams of one image
int decoder_infer_step(const sat_decoder_dims& d, const sat_decoder_params& p, const float* ann, const int* tokens, int K, int Kmax,
                       float* h, float* c, float* logits, float* alpha, const float* h_noise, char* ws, size_t ws_bytes, hipStream_t st) {
    InferWs w = infer_layout(d, Kmax, ws);
    const int NL = d.layers; const long KS = (long)K * d.n;       // h, c: (layers, K, n)
    float* htop = h + (NL - 1) * KS;
    SAT_REQUIRE(ws_bytes >= w.total && K >= 1 && K <= Kmax, "decoder_infer_step: workspace or beam count");
    t_bf16_mfma = d.precision ? 1 : 0;
    const int n = d.n, A = d.A, D = d.D, m = d.m, HCW = A + D + 4 * n;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(K), dim3(64), 0, st, p.embedding, tokens, w.Y, K, m, 0.f, 0ull, 0L);
    SAT_TRY(launch_ok("embedding gather"));
    if (NL == 1) {
        SAT_TRY(gemm(st, A_ROW, B_ROW, h, n, w.Wcat, n, w.hc, HCW, K, HCW, n, 0, EPI_BIAS_SIGMOID_RANGE, w.bcat, nullptr, nullptr, nullptr, 0, A, A + D));
    } else {
        SAT_TRY(gemm(st, A_ROW, B_ROW, htop, n, w.Wcat, n, w.hc, HCW, K, A + D, n, 0, EPI_BIAS_SIGMOID_RANGE, w.bcat, nullptr, nullptr, nullptr, 0, A, A + D));
        SAT_TRY(gemm(st, A_ROW, B_ROW, h, n, w.Wcat + (long)(A + D) * n, n, w.hc + A + D, HCW, K, 4 * n, n, 0, EPI_BIAS, w.bcat + A + D));
    }
    // decoder_noise (model.py:322-324): the noise joins h after attention and the gate were taken from the clean state, so
    // it reaches the step through the recurrent products only: gates += noise * W_hh^T (the GEMM is linear in h)
    if (h_noise) SAT_TRY(gemm(st, A_ROW, B_ROW, h_noise, n, w.Wcat + (long)(A + D) * n, n, w.hc + A + D, HCW, K, 4 * n, n, 1));
    SAT_TRY(launch_attention_fwd(st, ann, w.U, w.hc, HCW, p.att_f, w.ones, 0, alpha, 1, w.Z, w.XZ, 1, K, d.L, D, A, w.sc));
    SAT_TRY(gemm(st, A_ROW, B_ROW, w.Y, m, p.w_ih, m + D, w.hc + A + D, HCW, K, 4 * n, m, 1));
    SAT_TRY(gemm(st, A_ROW, B_ROW, w.XZ, D, p.w_ih + m, m + D, w.hc + A + D, HCW, K, 4 * n, D, 1));
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(cdiv((long)K * n, 256)), dim3(256), 0, st, w.hc + A + D, HCW, (const float*)nullptr, c, h, c, h, w.ones, 0, K, n);
    SAT_TRY(launch_ok("lstm_cell_fwd"));
    for (int l = 1; l < NL; ++l) {
        float* hl = h + l * KS; float* cl = c + l * KS;
        SAT_TRY(gemm(st, A_ROW, B_ROW, hl - KS, n, p.up_w_ih[l - 1], n, w.gu, 4 * n, K, 4 * n, n, 0, EPI_BIAS, w.bup + (long)(l - 1) * 4 * n));
        SAT_TRY(gemm(st, A_ROW, B_ROW, hl, n, p.up_w_hh[l - 1], n, w.gu, 4 * n, K, 4 * n, n, 1));
        if (h_noise) SAT_TRY(gemm(st, A_ROW, B_ROW, h_noise + l * KS, n, p.up_w_hh[l - 1], n, w.gu, 4 * n, K, 4 * n, n, 1));
        hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(cdiv((long)K * n, 256)), dim3(256), 0, st, w.gu, 4 * n, (const float*)nullptr, cl, hl, cl, hl, w.ones, 0, K, n);
        SAT_TRY(launch_ok("lstm_cell_fwd (stacked layer)"));
    }
    if (d.deep_output) {
        SAT_TRY(gemm(st, A_ROW, B_ROW, htop, n, p.out_hidden, n, w.u, m, K, m, n));
        SAT_TRY(gemm(st, A_ROW, B_ROW, w.Z, D, p.out_context, D, w.u, m, K, m, D, 1, EPI_ADD_TANH, nullptr, nullptr, nullptr, w.Y, m));
    } else {
        SAT_TRY(gemm(st, A_ROW, B_ROW, htop, n, p.out_hidden, n, w.u, m, K, m, n));
    }
    return gemm(st, A_ROW, B_ROW, w.u, m, p.out_w, m, logits, d.V, K, d.V, m, 0, p.out_b ? EPI_BIAS : EPI_NONE, p.out_b);
}

// ------------------------------------------------------------------ batched beam search: every image of the batch at once
// (the reference loops over images, model.py:260).  Rows (B, K); the per-image semantics -- F3 initial
// states, top-k over the live beams' flattened scores, completed hypotheses leaving the beam, cut at max_gen_length -- are
// those of model.py:262-448; the whole loop is enqueued without a host round trip and leaves a back-trace
// (token and parent row of every step, attention maps, finished list) for the host to read once.
struct BeamWs {
    size_t total; float *U, *Wcat, *bcat, *mean, *f, *init_img, *hc, *Z, *XZ, *Y, *u, *gu, *bup, *h, *c, *h2, *c2, *logits, *scores, *work, *keys, *noise, *vals, *top, *sc;
    int *live, *klive, *inds, *gmap, *mask_first, *mask_rest;
};
static BeamWs beam_layout(const sat_decoder_dims& d, int K, char* base) {
    BeamWs w; size_t off = 0;
    const long HCW = d.A + d.D + 4L * d.n, N = (long)d.B * K;
    auto take = [&](size_t elems) { size_t o = off; off += (elems * 4 + 255) & ~(size_t)255; return base ? base + o : (char*)nullptr; };
    w.U = (float*)take((size_t)d.B * d.L * d.A); w.Wcat = (float*)take((size_t)HCW * d.n); w.bcat = (float*)take((size_t)HCW);
    w.mean = (float*)take((size_t)d.B * d.D); w.f = (float*)take((size_t)d.B * d.m); w.init_img = (float*)take((size_t)d.B * 2 * d.n * d.layers);
    w.gu = (float*)take((size_t)N * 4 * d.n); w.bup = (float*)take((size_t)d.layers * 4 * d.n);
    w.hc = (float*)take((size_t)N * HCW); w.Z = (float*)take((size_t)N * d.D); w.XZ = (float*)take((size_t)N * d.D);
    w.Y = (float*)take((size_t)N * d.m); w.u = (float*)take((size_t)N * d.m);
    w.h = (float*)take((size_t)d.layers * N * d.n); w.c = (float*)take((size_t)d.layers * N * d.n);
    w.h2 = (float*)take((size_t)d.layers * N * d.n); w.c2 = (float*)take((size_t)d.layers * N * d.n);
    w.logits = (float*)take((size_t)N * d.V); w.scores = (float*)take((size_t)N * d.V); w.work = (float*)take((size_t)N * d.V); w.keys = (float*)take((size_t)N * d.V); w.noise = (float*)take((size_t)d.layers * N * d.n);
    w.vals = (float*)take(N); w.top = (float*)take(N); w.sc = (float*)take((size_t)N * d.L);
    w.live = (int*)take(N); w.klive = (int*)take(d.B); w.inds = (int*)take(N); w.gmap = (int*)take(N); w.mask_first = (int*)take(4); w.mask_rest = (int*)take(4);
    w.total = off;
    return w;
}
size_t decoder_beam_workspace_bytes(const sat_decoder_dims& d, int K) { return beam_layout(d, K, nullptr).total; }

int decoder_beam_batched(const sat_decoder_dims& d, const sat_decoder_params& p, const float* ann, int K, int max_gen_length, const float* temps_host,
                         int n_temps, const int* special_host /* START, PAD, END, UNK */, int* tok_in, int* prev_row, float* alpha_hist, int* fin_count,
                         int* fin_step, int* fin_row, float* fin_score, float* fin_mean, char* ws, size_t ws_bytes, hipStream_t st, const sat_beam_sampling* smp) {
    BeamWs w = beam_layout(d, K, ws);
    const int method = smp ? smp->method : 0;
    SAT_REQUIRE(method >= 0 && method <= 2 && (method != 2 || (smp->sample_topk >= 1 && smp->sample_topk <= d.V)), "beam_batched: sampling method %d, sample_topk %d",
                method, smp ? smp->sample_topk : 0);
    SAT_REQUIRE(ws_bytes >= w.total && K >= 1 && max_gen_length >= 0 && n_temps >= 1, "beam_batched: workspace %zu < %zu, K=%d, max_gen_length=%d", ws_bytes, w.total, K, max_gen_length);
    t_bf16_mfma = d.precision ? 1 : 0;
    const int B = d.B, N = B * K, n = d.n, A = d.A, D = d.D, m = d.m, V = d.V, NL = d.layers, HCW = A + D + 4 * n;
    const int START = special_host[0], PAD = special_host[1], END = special_host[2], UNK = special_host[3];
    // ---- once per batch: stacked step weights, att_enc, initial states of the K copies of every image
    auto copy_f = [&](float* dst, const float* src, long n_) {
        hipLaunchKernelGGL(copy_words_kernel, dim3(cdiv(n_, 256)), dim3(256), 0, st, reinterpret_cast<unsigned*>(dst), reinterpret_cast<const unsigned*>(src), n_);
    };
    auto zero_w = [&](void* dst, long n_) { hipLaunchKernelGGL(fill_int_kernel, dim3(cdiv(n_, 256)), dim3(256), 0, st, reinterpret_cast<int*>(dst), n_, 0); };
    copy_f(w.Wcat, p.att_dec, (long)A * n);
    copy_f(w.Wcat + (long)A * n, p.beta_w, (long)D * n);
    copy_f(w.Wcat + (long)(A + D) * n, p.w_hh, 4L * n * n);
    zero_w(w.bcat, A);
    copy_f(w.bcat + A, p.beta_b, D);
    SAT_TRY(launch_ok("beam: step weights"));
    hipLaunchKernelGGL(add_kernel, dim3(cdiv(4 * n, 256)), dim3(256), 0, st, w.bcat + A + D, p.b_ih, p.b_hh, (long)4 * n);
    SAT_TRY(launch_ok("bias add"));
    for (int l = 1; l < NL; ++l) {
        hipLaunchKernelGGL(add_kernel, dim3(cdiv(4 * n, 256)), dim3(256), 0, st, w.bup + (long)(l - 1) * 4 * n, p.up_b_ih[l - 1], p.up_b_hh[l - 1], (long)4 * n);
        SAT_TRY(launch_ok("bias add (stacked layer)"));
    }
    hipLaunchKernelGGL(set4_int_kernel, dim3(1), dim3(1), 0, st, w.mask_first, START, PAD, END, UNK);
    hipLaunchKernelGGL(set4_int_kernel, dim3(1), dim3(1), 0, st, w.mask_rest, START, PAD, -1, -1);
    SAT_TRY(launch_ok("beam masks"));
    SAT_TRY(gemm(st, A_ROW, B_ROW, ann, D, p.att_enc, D, w.U, A, B * d.L, A, D));
    hipLaunchKernelGGL(ann_mean_kernel, dim3(B), dim3(256), 0, st, ann, w.mean, d.L, D);
    SAT_TRY(launch_ok("ann_mean"));
    SAT_TRY(gemm(st, A_ROW, B_ROW, w.mean, D, p.init_f_w, D, w.f, m, B, m, D, 0, EPI_BIAS, p.init_f_b));
    SAT_TRY(gemm(st, A_ROW, B_ROW, w.f, m, p.init_i_w, m, w.init_img, 2 * n * NL, B, 2 * n * NL, m, 0, EPI_BIAS, p.init_i_b));
    hipLaunchKernelGGL(init_expand_images_kernel, dim3(cdiv(2L * NL * N * n, 256)), dim3(256), 0, st, w.init_img, w.h, w.c, B, K, n, NL);
    SAT_TRY(launch_ok("init_expand_images"));
    hipLaunchKernelGGL(fill_int_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, w.klive, (long)B, K);
    SAT_TRY(launch_ok("fill klive"));
    zero_w(tok_in, (long)(max_gen_length + 2) * N);       // rows that are not live still index the embedding table
    zero_w(prev_row, (long)(max_gen_length + 2) * N);
    hipLaunchKernelGGL(fill_int_kernel, dim3(cdiv(N, 256)), dim3(256), 0, st, tok_in, (long)N, START);
    SAT_TRY(launch_ok("fill start tokens"));
    zero_w(w.h2, (long)NL * N * n);
    zero_w(w.c2, (long)NL * N * n);
    zero_w(w.top, N);
    zero_w(fin_count, B);
    SAT_TRY(launch_ok("beam: clears"));

    float *h = w.h, *c = w.c, *h2 = w.h2, *c2 = w.c2;
    for (int step = 0; step <= max_gen_length; ++step) {
        const int* tok = tok_in + (long)step * N;
        float* alpha = alpha_hist + (long)step * N * d.L;
        const long KS = (long)N * n;
        float* htop = h + (NL - 1) * KS;
        hipLaunchKernelGGL(beam_live_kernel, dim3(cdiv(N, 256)), dim3(256), 0, st, w.klive, w.live, B, K);
        SAT_TRY(launch_ok("beam_live"));
        // ---- model.py:298-327 for all rows (dead rows: zero attention, state carried, ignored below)
        hipLaunchKernelGGL(gather_rows_kernel, dim3(N), dim3(64), 0, st, p.embedding, tok, w.Y, N, m, 0.f, 0ull, 0L);
        SAT_TRY(launch_ok("embedding gather"));
        const bool noisy = smp && smp->decoder_noise != 0.f;
        if (noisy) {        // model.py:322-324: randn * decoder_noise / (step + 1) joins h of every layer before the LSTM
            const long tot = (long)NL * N * n;
            hipLaunchKernelGGL(beam_state_noise_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, st, w.noise, w.live, (long)N, n, NL, smp->decoder_noise / (float)(step + 1),
                               (unsigned long long)smp->seed, 0x1000ull + step, smp->normals ? smp->normals + (long)step * tot : (const float*)nullptr);
            SAT_TRY(launch_ok("beam_state_noise"));
        }
        if (NL == 1) {
            SAT_TRY(gemm(st, A_ROW, B_ROW, h, n, w.Wcat, n, w.hc, HCW, N, HCW, n, 0, EPI_BIAS_SIGMOID_RANGE, w.bcat, nullptr, nullptr, nullptr, 0, A, A + D));
        } else {
            SAT_TRY(gemm(st, A_ROW, B_ROW, htop, n, w.Wcat, n, w.hc, HCW, N, A + D, n, 0, EPI_BIAS_SIGMOID_RANGE, w.bcat, nullptr, nullptr, nullptr, 0, A, A + D));
            SAT_TRY(gemm(st, A_ROW, B_ROW, h, n, w.Wcat + (long)(A + D) * n, n, w.hc + A + D, HCW, N, 4 * n, n, 0, EPI_BIAS, w.bcat + A + D));
        }
        SAT_TRY(launch_attention_fwd(st, ann, w.U, w.hc, HCW, p.att_f, w.live, 0, alpha, 1, w.Z, w.XZ, B, K, d.L, D, A, w.sc));
        if (noisy) SAT_TRY(gemm(st, A_ROW, B_ROW, w.noise, n, w.Wcat + (long)(A + D) * n, n, w.hc + A + D, HCW, N, 4 * n, n, 1));
        SAT_TRY(gemm(st, A_ROW, B_ROW, w.Y, m, p.w_ih, m + D, w.hc + A + D, HCW, N, 4 * n, m, 1));
        SAT_TRY(gemm(st, A_ROW, B_ROW, w.XZ, D, p.w_ih + m, m + D, w.hc + A + D, HCW, N, 4 * n, D, 1));
        hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(cdiv((long)N * n, 256)), dim3(256), 0, st, w.hc + A + D, HCW, (const float*)nullptr, c, h, c, h, w.live, 0, N, n);
        SAT_TRY(launch_ok("lstm_cell_fwd"));
        for (int l = 1; l < NL; ++l) {
            float* hl = h + l * KS; float* cl = c + l * KS;
            SAT_TRY(gemm(st, A_ROW, B_ROW, hl - KS, n, p.up_w_ih[l - 1], n, w.gu, 4 * n, N, 4 * n, n, 0, EPI_BIAS, w.bup + (long)(l - 1) * 4 * n));
            SAT_TRY(gemm(st, A_ROW, B_ROW, hl, n, p.up_w_hh[l - 1], n, w.gu, 4 * n, N, 4 * n, n, 1));
            if (noisy) SAT_TRY(gemm(st, A_ROW, B_ROW, w.noise + l * KS, n, p.up_w_hh[l - 1], n, w.gu, 4 * n, N, 4 * n, n, 1));
            hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(cdiv((long)N * n, 256)), dim3(256), 0, st, w.gu, 4 * n, (const float*)nullptr, cl, hl, cl, hl, w.live, 0, N, n);
            SAT_TRY(launch_ok("lstm_cell_fwd (stacked layer)"));
        }
        SAT_TRY(gemm(st, A_ROW, B_ROW, htop, n, p.out_hidden, n, w.u, m, N, m, n));
        if (d.deep_output) SAT_TRY(gemm(st, A_ROW, B_ROW, w.Z, D, p.out_context, D, w.u, m, N, m, D, 1, EPI_ADD_TANH, nullptr, nullptr, nullptr, w.Y, m));
        SAT_TRY(gemm(st, A_ROW, B_ROW, w.u, m, p.out_w, m, w.logits, V, N, V, m, 0, p.out_b ? EPI_BIAS : EPI_NONE, p.out_b));
        // ---- model.py:330-359: log-softmax, special-token masks, parent scores, top-k per image
        const float T = temps_host[step % n_temps];
        hipLaunchKernelGGL(beam_scores_kernel, dim3(N), dim3(256), 0, st, w.logits, V, 1.0f / T, step == 0 ? w.mask_first : w.mask_rest, step == 0 ? 4 : 2,
                           step == 0 ? (const float*)nullptr : w.top, w.scores);
        SAT_TRY(launch_ok("beam_scores"));
        if (method == 0 || step == 0) {                // the first step always takes the top beamk words (model.py:343)
            hipLaunchKernelGGL(beam_topk_kernel, dim3(B), dim3(1024), 0, st, w.scores, w.work, w.klive, K, V, step == 0 ? 1 : 0, w.vals, w.inds);
            SAT_TRY(launch_ok("beam_topk"));
        } else {                                       // model.py:360-379: draw the continuing hypotheses (Gumbel-top-k == multinomial without replacement)
            const int gstride = method == 1 ? V : smp->sample_topk;
            hipLaunchKernelGGL(beam_sample_keys_kernel, dim3(N), dim3(256), 0, st, w.scores, w.klive, K, V, method, smp->sample_topk, (float)step,
                               (unsigned long long)smp->seed, (unsigned long long)step, smp->gumbel ? smp->gumbel + (long)step * N * gstride : (const float*)nullptr,
                               gstride, w.keys);
            SAT_TRY(launch_ok("beam_sample_keys"));
            hipLaunchKernelGGL(beam_topk_kernel, dim3(B), dim3(1024), 0, st, w.keys, w.work, w.klive, K, V, 0, w.vals, w.inds);
            SAT_TRY(launch_ok("beam_topk (keys)"));
            hipLaunchKernelGGL(beam_take_scores_kernel, dim3(cdiv(N, 256)), dim3(256), 0, st, w.scores, w.inds, w.klive, B, K, V, w.vals);
            SAT_TRY(launch_ok("beam_take_scores"));
        }
        hipLaunchKernelGGL(beam_update_kernel, dim3(cdiv(B, 64)), dim3(64), 0, st, w.vals, w.inds, w.klive, B, K, V, step, max_gen_length, END,
                           tok_in + (long)(step + 1) * N, prev_row + (long)(step + 1) * N, w.top, w.gmap, fin_count, fin_step, fin_row, fin_score, fin_mean);
        SAT_TRY(launch_ok("beam_update"));
        if (step < max_gen_length) {
            const long tot = (long)NL * N * n;
            hipLaunchKernelGGL(beam_gather_state_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, st, h, h2, w.gmap, w.klive, B, K, n, NL);
            SAT_TRY(launch_ok("beam_gather h"));
            hipLaunchKernelGGL(beam_gather_state_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, st, c, c2, w.gmap, w.klive, B, K, n, NL);
            SAT_TRY(launch_ok("beam_gather c"));
            float* t1 = h; h = h2; h2 = t1; t1 = c; c = c2; c2 = t1;
        }
    }
    return SAT_OK;
}

int beam_scores(const float* logits, int K, int V, float temperature, const int* masked, int n_masked, const float* parent, float* scores, hipStream_t st) {
    SAT_REQUIRE(K > 0 && V > 0 && temperature > 0.f, "beam_scores: bad arguments");
    hipLaunchKernelGGL(beam_scores_kernel, dim3(K), dim3(256), 0, st, logits, V, 1.0f / temperature, masked, n_masked, parent, scores);
    return launch_ok("beam_scores");
}
int topk(const float* x, float* work, long n, int k, float* values, int* indices, hipStream_t st) {
    SAT_REQUIRE(n > 0 && k > 0 && k <= n, "topk: bad arguments (n=%ld k=%d)", n, k);
    hipLaunchKernelGGL(topk_kernel, dim3(1), dim3(1024), 0, st, x, work, n, k, values, indices);
    return launch_ok("topk");
}

// ------------------------------------------------------------------ sub-module step entry points (SURVEY 8b minimum set)
// The reference's sub-modules called on their own (model.py:76-81, 94-109, 125-131, 175-180 / 326, 544): the same kernels and GEMMs
// as the train loop above, for N independent rows, exact fp32 MFMA.  Scratch comes from the caller (sizes in include/sat_hip.h).
static int fill_f(hipStream_t st, float* p, long n, float v) {
    if (n <= 0) return SAT_OK;
    hipLaunchKernelGGL(fill_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, p, n, v);
    return launch_ok("fill");
}

int attention_step_bwd(const float* ann, const float* U, const float* hc, int hc_ld, const float* wf, const int* lengths, int step, const float* alphas,
                       const float* dalphas, int T1, const float* Z, const float* dZ, const float* dXZ, float* DZ, float* dhc, int dhc_ld, float* dU,
                       float* dwf_part, float* da, int B, int R, int L, int D, int A, hipStream_t st) {
    t_bf16_mfma = 0;
    return attention_bwd_split(R < ATT_RMAX ? R : ATT_RMAX, st, ann, U, hc, hc_ld, wf, lengths, step, alphas, dalphas, T1, Z, dZ, dXZ, DZ, dhc, dhc_ld, dU, dwf_part,
                               da, B, R, L, D, A, nullptr);
}

int attention_context_bwd(const float* alphas, const float* DZ, const int* lengths, float* dann, int accumulate, int B, int R, int T1, int L, int D, hipStream_t st) {
    const int lq4 = (L + 3) / 4;
    const int NQ = (lq4 <= 13) ? 13 : 16;
    const int Lq = (lq4 + NQ - 1) / NQ * NQ;
    const size_t lds = (size_t)R * T1 * Lq * 16;
    SAT_REQUIRE(lds <= 160 * 1024, "attention_context_bwd: R*T1*L = %d floats of alphas do not fit the LDS", R * T1 * L);
    if (NQ == 13) {
        SAT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dann_from_context_kernel<13>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(dann_from_context_kernel<13>, dim3(B, cdiv(D, 256)), dim3(256), lds, st, alphas, DZ, lengths, dann, accumulate, R, B * R, T1, L, D);
    } else {
        SAT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dann_from_context_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(dann_from_context_kernel<16>, dim3(B, cdiv(D, 256)), dim3(256), lds, st, alphas, DZ, lengths, dann, accumulate, R, B * R, T1, L, D);
    }
    return launch_ok("dann_from_context");
}

// nn.LSTM, one layer, one time step (model.py:175-180; calls at 326, 544): gates = x W_ih^T + b_ih + h W_hh^T + b_hh, i,f,g,o
int lstm_cell_fwd(const float* x, int in, const float* h_prev, const float* c_prev, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                  float* gates, float* h_new, float* c_new, float* bias_scratch, int N, int n, hipStream_t st) {
    t_bf16_mfma = 0;
    hipLaunchKernelGGL(add_kernel, dim3(cdiv(4 * n, 256)), dim3(256), 0, st, bias_scratch, b_ih, b_hh, (long)4 * n);
    SAT_TRY(launch_ok("bias add"));
    SAT_TRY(gemm(st, A_ROW, B_ROW, x, in, w_ih, in, gates, 4 * n, N, 4 * n, in, 0, EPI_BIAS, bias_scratch));
    SAT_TRY(gemm(st, A_ROW, B_ROW, h_prev, n, w_hh, n, gates, 4 * n, N, 4 * n, n, 1));
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(cdiv((long)N * n, 256)), dim3(256), 0, st, gates, 4 * n, (const float*)nullptr, c_prev, h_prev, c_new, h_new,
                       (const int*)nullptr, 0, N, n, (__bf16*)nullptr);
    return launch_ok("lstm_cell_fwd");
}

int lstm_cell_bwd(const float* x, int in, const float* h_prev, const float* c_prev, const float* c_new, const float* gates, const float* dh_new, const float* dc_new,
                  const float* w_ih, const float* w_hh, float* dx, float* dh_prev, float* dc_prev, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh,
                  float* dgates, float* scratch, int N, int n, hipStream_t st) {
    t_bf16_mfma = 0;
    // the cell kernel adds its `carry` inputs and leaves dc_prev in the cell carry: start them from the incoming cell gradient / zero
    if (dc_new) SAT_TRY(dev_copy_bytes(st, dc_prev, dc_new, (size_t)N * n * 4));
    else SAT_TRY(fill_f(st, dc_prev, (long)N * n, 0.f));
    SAT_TRY(fill_f(st, dh_prev, (long)N * n, 0.f));
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(cdiv((long)N * n, 256)), dim3(256), 0, st, gates, 4 * n, c_prev, c_new, dh_new, dh_prev, dc_prev, dgates, 4 * n,
                       (const int*)nullptr, 0, N, n, (__bf16*)nullptr);
    SAT_TRY(launch_ok("lstm_cell_bwd"));
    SAT_TRY(gemm(st, A_ROW, B_KMAJOR, dgates, 4 * n, w_ih, in, dx, in, N, in, 4 * n));
    SAT_TRY(gemm(st, A_ROW, B_KMAJOR, dgates, 4 * n, w_hh, n, dh_prev, n, N, n, 4 * n));
    SAT_TRY(gemm(st, A_KMAJOR, B_KMAJOR, dgates, 4 * n, x, in, dw_ih, in, 4 * n, in, N));
    SAT_TRY(gemm(st, A_KMAJOR, B_KMAJOR, dgates, 4 * n, h_prev, n, dw_hh, n, 4 * n, n, N));
    Ws w; w.colpart = scratch;
    SAT_TRY(colsum(st, w, dgates, 4 * n, N, 4 * n, db_ih));
    SAT_TRY(dev_copy_bytes(st, db_hh, db_ih, (size_t)4 * n * 4));
    return SAT_OK;
}

// DeepOutput.forward (model.py:125-131)
int deep_output_fwd(const float* prev_embed, const float* hidden, const float* context, const float* w_hidden, const float* w_context, const float* w_out,
                    const float* b_out, float dropout, unsigned long long seed, float* u, float* udrop, float* logits, int N, int m, int n, int D, int V, hipStream_t st) {
    t_bf16_mfma = 0;
    if (context) {
        SAT_TRY(gemm(st, A_ROW, B_ROW, hidden, n, w_hidden, n, u, m, N, m, n));
        SAT_TRY(gemm(st, A_ROW, B_ROW, context, D, w_context, D, u, m, N, m, D, 1, EPI_ADD_TANH, nullptr, nullptr, nullptr, prev_embed, m));
    } else {
        SAT_TRY(gemm(st, A_ROW, B_ROW, hidden, n, w_hidden, n, u, m, N, m, n));
    }
    const float* uin = u;
    if (dropout > 0.f) {
        hipLaunchKernelGGL(dropout_rows_kernel, dim3(cdiv((long)N * m, 256)), dim3(256), 0, st, u, udrop, (long)N * m, m, dropout, seed, 2u, 0L);
        SAT_TRY(launch_ok("output dropout"));
        uin = udrop;
    }
    return gemm(st, A_ROW, B_ROW, uin, m, w_out, m, logits, V, N, V, m, 0, b_out ? EPI_BIAS : EPI_NONE, b_out);
}

int deep_output_bwd(const float* dlogits, const float* hidden, const float* context, const float* u, const float* udrop, const float* w_hidden, const float* w_context,
                    const float* w_out, float dropout, unsigned long long seed, float* d_prev_embed, float* d_hidden, float* d_context, float* dw_hidden,
                    float* dw_context, float* dw_out, float* db_out, float* scratch, int N, int m, int n, int D, int V, hipStream_t st) {
    t_bf16_mfma = 0;
    const bool deep = context != nullptr;
    float* du = d_prev_embed;                 // gradient of the pre-tanh sum = gradient of the embedding input (deep); scratch of the same shape (shallow)
    if (dropout > 0.f) {
        SAT_TRY(gemm(st, A_ROW, B_KMAJOR, dlogits, V, w_out, m, du, m, N, m, V));
        hipLaunchKernelGGL(dropout_rows_kernel, dim3(cdiv((long)N * m, 256)), dim3(256), 0, st, du, du, (long)N * m, m, dropout, seed, 2u, 0L);
        SAT_TRY(launch_ok("output dropout bwd"));
        if (deep) {
            hipLaunchKernelGGL(mul_dtanh_kernel, dim3(cdiv((long)N * m, 256)), dim3(256), 0, st, du, u, (long)N * m);
            SAT_TRY(launch_ok("tanh bwd"));
        }
    } else {
        SAT_TRY(gemm(st, A_ROW, B_KMAJOR, dlogits, V, w_out, m, du, m, N, m, V, 0, deep ? EPI_MUL_DTANH : EPI_NONE, nullptr, nullptr, nullptr, u, m));
    }
    SAT_TRY(gemm(st, A_KMAJOR, B_KMAJOR, dlogits, V, dropout > 0.f ? udrop : u, m, dw_out, m, V, m, N));
    Ws w; w.colpart = scratch;
    if (db_out) SAT_TRY(colsum(st, w, dlogits, V, N, V, db_out));
    SAT_TRY(gemm(st, A_ROW, B_KMAJOR, du, m, w_hidden, n, d_hidden, n, N, n, m));
    SAT_TRY(gemm(st, A_KMAJOR, B_KMAJOR, du, m, hidden, n, dw_hidden, n, m, n, N));
    if (deep) {
        SAT_TRY(gemm(st, A_ROW, B_KMAJOR, du, m, w_context, D, d_context, D, N, D, m));
        SAT_TRY(gemm(st, A_KMAJOR, B_KMAJOR, du, m, context, D, dw_context, D, m, D, N));
    }
    return SAT_OK;
}

// InitLSTM.forward (model.py:76-81) on N rows; `init` (N, n2) row-major IS the (2*layers, N, n) buffer of the raw reshape (F3)
int init_lstm_fwd(const float* ann, const float* w_f, const float* b_f, const float* w_i, const float* b_i, float dropout, unsigned long long seed, float* mean,
                  float* f, float* init, int N, int L, int D, int m, int n2, hipStream_t st) {
    t_bf16_mfma = 0;
    hipLaunchKernelGGL(ann_mean_kernel, dim3(N), dim3(256), 0, st, ann, mean, L, D);
    SAT_TRY(launch_ok("ann_mean"));
    if (dropout > 0.f) {
        hipLaunchKernelGGL(init_mean_rows_kernel, dim3(cdiv((long)N * D, 256)), dim3(256), 0, st, mean, mean, N, 1, D, dropout, seed);
        SAT_TRY(launch_ok("init_mean_rows"));
    }
    SAT_TRY(gemm(st, A_ROW, B_ROW, mean, D, w_f, D, f, m, N, m, D, 0, EPI_BIAS, b_f));
    return gemm(st, A_ROW, B_ROW, f, m, w_i, m, init, n2, N, n2, m, 0, EPI_BIAS, b_i);
}

int init_lstm_bwd(const float* dinit, const float* mean, const float* f, const float* w_f, const float* w_i, float dropout, unsigned long long seed, float* dw_f,
                  float* db_f, float* dw_i, float* db_i, float* dann, float* df, float* dmean, float* scratch, int N, int L, int D, int m, int n2, hipStream_t st) {
    t_bf16_mfma = 0;
    Ws w; w.colpart = scratch;
    SAT_TRY(gemm(st, A_KMAJOR, B_KMAJOR, dinit, n2, f, m, dw_i, m, n2, m, N));
    SAT_TRY(colsum(st, w, dinit, n2, N, n2, db_i));
    SAT_TRY(gemm(st, A_ROW, B_KMAJOR, dinit, n2, w_i, m, df, m, N, m, n2));
    SAT_TRY(gemm(st, A_KMAJOR, B_KMAJOR, df, m, mean, D, dw_f, D, m, D, N));
    SAT_TRY(colsum(st, w, df, m, N, m, db_f));
    SAT_TRY(gemm(st, A_ROW, B_KMAJOR, df, m, w_f, D, dmean, D, N, D, m));
    if (dropout > 0.f) {
        hipLaunchKernelGGL(init_mean_rows_bwd_kernel, dim3(cdiv((long)N * D, 256)), dim3(256), 0, st, dmean, dmean, N, 1, D, dropout, seed);
        SAT_TRY(launch_ok("init_mean_rows_bwd"));
    }
    const long tot = (long)N * L * D;
    SAT_TRY(fill_f(st, dann, tot, 0.f));
    hipLaunchKernelGGL(dann_add_mean_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, st, dann, dmean, L, D, tot);
    return launch_ok("dann_add_mean");
}

// nn.Embedding forward (model.py:158-164, use at 298, 526): optional in-place max-norm renormalisation of the rows used, then the gather
int embedding_fwd(float* table, const int* tokens, float* out, int rows, int V, int m, float max_norm, int* flags, hipStream_t st) {
    if (max_norm > 0.f) {
        hipLaunchKernelGGL(fill_int_kernel, dim3(cdiv(V, 256)), dim3(256), 0, st, flags, (long)V, 0);
        hipLaunchKernelGGL(embedding_mark_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, st, tokens, rows, flags, V);
        SAT_TRY(launch_ok("embedding_mark"));
        hipLaunchKernelGGL(embedding_renorm_kernel, dim3(V), dim3(64), 0, st, table, flags, m, max_norm);
        SAT_TRY(launch_ok("embedding_renorm"));
    }
    hipLaunchKernelGGL(gather_rows_kernel, dim3(rows), dim3(64), 0, st, table, tokens, out, rows, m, 0.f, 0ull, 0L);
    return launch_ok("embedding gather");
}
// its gradient: rows of dY added into the table rows of their tokens in increasing row order (no floating-point atomics); padding row zero
int embedding_bwd(const float* dY, const int* tokens, float* dtable, int rows, int V, int m, int padding_idx, int* scratch, hipStream_t st) {
    int* count = scratch; int* offset = count + V; int* cursor = offset + V + 1; int* list = cursor + V;
    hipLaunchKernelGGL(fill_int_kernel, dim3(cdiv(V, 256)), dim3(256), 0, st, count, (long)V, 0);
    hipLaunchKernelGGL(embedding_count_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, st, tokens, rows, V, padding_idx, count);
    hipLaunchKernelGGL(embedding_scan_kernel, dim3(1), dim3(1024), 0, st, count, V, offset, cursor);
    hipLaunchKernelGGL(embedding_place_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, st, tokens, rows, V, padding_idx, offset, cursor, list);
    hipLaunchKernelGGL(embedding_sum_kernel, dim3(V), dim3(256), 0, st, dY, tokens, offset, list, dtable, rows, m, padding_idx);
    return launch_ok("embedding gradient");
}
// backward of y = sigmoid(pre): dpre = dy * y * (1 - y)   (the beta gate, model.py:187-192)
int sigmoid_bwd(const float* dy, const float* y, float* dpre, long n, hipStream_t st) {
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, dy, y, dpre, n);
    return launch_ok("sigmoid_bwd");
}

// ------------------------------------------------------------------ losses
int ce_fwd(const float* logits, const int* targets, int P, int V, float smoothing, float* lse_rows, float* loss_rows, int* correct_rows, float* out, hipStream_t st) {
    SAT_REQUIRE(P > 0 && V > 0, "ce_fwd: empty input (P=%d V=%d)", P, V);
    SAT_REQUIRE(smoothing >= 0.f && smoothing < 1.f, "ce_fwd: smoothing %g out of range", smoothing);
    if (V % 4 == 0 && (uintptr_t)logits % 16 == 0) hipLaunchKernelGGL(ce_rows_kernel<4>, dim3(P), dim3(256), 0, st, logits, targets, V, smoothing, lse_rows, loss_rows, correct_rows);
    else hipLaunchKernelGGL(ce_rows_kernel<1>, dim3(P), dim3(256), 0, st, logits, targets, V, smoothing, lse_rows, loss_rows, correct_rows);
    SAT_TRY(launch_ok("ce_rows"));
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(256), 0, st, loss_rows, correct_rows, P, out);
    return launch_ok("ce_finish");
}
int ce_bwd(const float* logits, const int* targets, const float* lse_rows, int P, int V, float smoothing, const float* gscale, float* dlogits, hipStream_t st) {
    SAT_REQUIRE(P > 0 && V > 0, "ce_bwd: empty input");
    if (V % 4 == 0 && (uintptr_t)logits % 16 == 0 && (uintptr_t)dlogits % 16 == 0) hipLaunchKernelGGL(ce_grad_kernel<4>, dim3(P), dim3(256), 0, st, logits, targets, lse_rows, V, smoothing, 1.0f / (float)P, gscale, dlogits);
    else hipLaunchKernelGGL(ce_grad_kernel<1>, dim3(P), dim3(256), 0, st, logits, targets, lse_rows, V, smoothing, 1.0f / (float)P, gscale, dlogits);
    return launch_ok("ce_grad");
}
int ds_fwd(const float* alphas, int N, int T1, int L, float gamma, float* asum, float* part, float* out, hipStream_t st) {
    SAT_REQUIRE(N > 0 && T1 > 0 && L > 0, "ds_fwd: empty input");
    const int nb = cdiv((long)N * L, 256);
    hipLaunchKernelGGL(ds_rows_kernel, dim3(nb), dim3(256), 0, st, alphas, asum, part, N, T1, L);
    SAT_TRY(launch_ok("ds_rows"));
    hipLaunchKernelGGL(ds_finish_kernel, dim3(1), dim3(256), 0, st, part, nb, gamma, (long)N * L, out);
    return launch_ok("ds_finish");
}
int ds_bwd(const float* asum, const float* gscale, int N, int T1, int L, float gamma, float* dalphas, hipStream_t st) {
    hipLaunchKernelGGL(ds_grad_kernel, dim3(cdiv((long)N * T1 * L, 256)), dim3(256), 0, st, asum, gscale, gamma, dalphas, N, T1, L);
    return launch_ok("ds_grad");
}

}  // namespace sat
