// extern "C" surface of libsat_hip.so (declared in include/sat_hip.h).
#include <stdarg.h>

#include "../../include/sat_hip.h"
#include "decoder.h"
#include "gemm.h"
#include "gemm_bf16_common.h"

namespace sat {
static thread_local char g_err[512] = {0};
int& dev_switch(int which) {
    static int v[SW_COUNT] = {0 /* (ticket-based BatchNorm reduce: removed in round 3, measured slower) */, getenv("SAT_NO_WGRAD3X3") ? !atoi(getenv("SAT_NO_WGRAD3X3")) : 1,
                              getenv("SAT_REDUCE_Z16") ? atoi(getenv("SAT_REDUCE_Z16")) : 1, getenv("SAT_WIDE_TILES") ? atoi(getenv("SAT_WIDE_TILES")) : 0,
                              getenv("SAT_BN_ONEPASS") ? atoi(getenv("SAT_BN_ONEPASS")) : 1,
                              getenv("SAT_BN_VPT") ? atoi(getenv("SAT_BN_VPT")) : 2,
                              getenv("SAT_ACC_PREFETCH") ? atoi(getenv("SAT_ACC_PREFETCH")) : 0};
    return v[which];
}
int& trace_launches() { static int on = getenv("SAT_TRACE_LAUNCH") ? 1 : 0; return on; }
char* last_error_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return code;
}
}  // namespace sat

using namespace sat;

extern "C" {

int sat_abi_version(void) { return SAT_HIP_ABI_VERSION; }
int sat_debug_trace_launches(int32_t on) { sat::trace_launches() = on ? 1 : 0; return SAT_OK; }
int sat_debug_option(const char* name, int32_t value) {
    if (!name) return fail(SAT_EINVAL, "debug_option: null name");
    if (!strcmp(name, "wgrad3x3")) { dev_switch(SW_WGRAD3X3) = value; return SAT_OK; }
    if (!strcmp(name, "reduce_z16")) { dev_switch(SW_REDUCE_Z16) = value; return SAT_OK; }
    if (!strcmp(name, "acc_prefetch")) { dev_switch(SW_ACC_PREFETCH) = value; return SAT_OK; }
    if (!strcmp(name, "bn_vpt")) { dev_switch(SW_BN_VPT) = value; return SAT_OK; }
    if (!strcmp(name, "bn_onepass")) { dev_switch(SW_BN_ONEPASS) = value; return SAT_OK; }
    if (!strcmp(name, "wide_tiles")) { dev_switch(SW_WIDE_TILES) = value; return SAT_OK; }
    if (!strcmp(name, "glds_tile")) { glds_force_tile() = value; return SAT_OK; }
    if (!strcmp(name, "glds_stages8")) { glds_stages8() = value; return SAT_OK; }
    if (!strcmp(name, "glds_tall_k")) { glds_tall_k() = value; return SAT_OK; }
    if (!strcmp(name, "glds_tall_conv")) { glds_tall_conv() = value; return SAT_OK; }
    if (!strcmp(name, "tile_override")) { gemm_tile_override() = value; return SAT_OK; }
    return fail(SAT_EINVAL, "debug_option: unknown option %s", name);
}
const char* sat_last_error(void) { return last_error_buf(); }

static int gemm_from_desc(const sat_gemm_desc* d, const sat_gemm_types* t, void* stream);
int sat_gemm_f32(const sat_gemm_desc* d, void* stream) { return gemm_from_desc(d, nullptr, stream); }
int sat_gemm_ex(const sat_gemm_desc* d, const sat_gemm_types* t, void* stream) {
    if (!t) return fail(SAT_EINVAL, "sat_gemm_ex: null type descriptor");
    return gemm_from_desc(d, t, stream);
}
static int gemm_from_desc(const sat_gemm_desc* d, const sat_gemm_types* t, void* stream) {
    if (!d) return fail(SAT_EINVAL, "sat_gemm_f32: null descriptor");
    if (d->amode != A_ROW && d->amode != A_KMAJOR) return fail(SAT_EINVAL, "sat_gemm_f32: amode %d (dense modes only)", d->amode);
    if (d->bmode != B_ROW && d->bmode != B_KMAJOR) return fail(SAT_EINVAL, "sat_gemm_f32: bmode %d (dense modes only)", d->bmode);
    GemmArgs g;
    g.A = d->A; g.lda = d->lda; g.a_rows = d->a_rows; g.B = d->B; g.ldb = d->ldb; g.C = d->C; g.ldc = d->ldc; g.c_rows = d->c_rows;
    g.M = d->M; g.N = d->N; g.K = d->K; g.amode = d->amode; g.bmode = d->bmode; g.accumulate = d->accumulate; g.epi = d->epi;
    g.bias = d->bias; g.e0 = d->e0; g.lde0 = d->lde0; g.c0 = d->c0; g.c1 = d->c1; g.slab = d->slab; g.slab_elems = d->slab_elems;
    if (t) { g.a_bf16 = t->a_bf16; g.b_bf16 = t->b_bf16; g.c_bf16 = t->c_bf16; g.bf16_mfma = t->bf16_mfma; }
    return launch_gemm(g, (hipStream_t)stream);
}

size_t sat_decoder_workspace_bytes(const sat_decoder_dims* d) {
    if (check_dims(d) != SAT_OK) return 0;
    return decoder_workspace_bytes(*d);
}

static int check_params(const sat_decoder_dims* d, const sat_decoder_params* w, const char* what) {
    if (!w) return fail(SAT_EINVAL, "%s: null parameter struct", what);
    if (!w->embedding || !w->init_f_w || !w->init_f_b || !w->init_i_w || !w->init_i_b || !w->w_ih || !w->w_hh || !w->b_ih || !w->b_hh ||
        !w->att_enc || !w->att_dec || !w->att_f || !w->beta_w || !w->beta_b || !w->out_hidden || !w->out_w)
        return fail(SAT_EINVAL, "%s: null tensor in parameter struct", what);
    if (d->deep_output && !w->out_context) return fail(SAT_EINVAL, "%s: deep output needs output.context.weight", what);
    for (int l = 1; l < d->layers; ++l)
        if (!w->up_w_ih[l - 1] || !w->up_w_hh[l - 1] || !w->up_b_ih[l - 1] || !w->up_b_hh[l - 1])
            return fail(SAT_EINVAL, "%s: null tensor for LSTM layer %d", what, l);
    return SAT_OK;
}
static int check_batch(const sat_decoder_batch* b) {
    if (!b || !b->ann || !b->caps || !b->lengths || !b->prow || !b->src_row || !b->step_offsets_host || !b->teacher_host)
        return fail(SAT_EINVAL, "decoder: null field in batch struct");
    return SAT_OK;
}

int sat_decoder_train_fwd(const sat_decoder_dims* d, const sat_decoder_params* w, const sat_decoder_batch* b, float* logits_packed,
                          float* alphas, void* workspace, size_t workspace_bytes, void* stream) {
    SAT_TRY(check_dims(d)); SAT_TRY(check_params(d, w, "decoder_train_fwd")); SAT_TRY(check_batch(b));
    if (!alphas || !workspace || (d->P > 0 && !logits_packed)) return fail(SAT_EINVAL, "decoder_train_fwd: null output/workspace");
    return decoder_fwd(*d, *w, *b, logits_packed, alphas, (char*)workspace, workspace_bytes, (hipStream_t)stream);
}

int sat_decoder_train_bwd(const sat_decoder_dims* d, const sat_decoder_params* w, const sat_decoder_batch* b, const float* dlogits_packed,
                          const float* alphas, const float* dalphas, const sat_decoder_params* g, float* dann, void* workspace,
                          size_t workspace_bytes, void* stream) {
    SAT_TRY(check_dims(d)); SAT_TRY(check_params(d, w, "decoder_train_bwd")); SAT_TRY(check_params(d, g, "decoder_train_bwd(grads)")); SAT_TRY(check_batch(b));
    if (!alphas || !dann || !workspace || (d->P > 0 && !dlogits_packed)) return fail(SAT_EINVAL, "decoder_train_bwd: null input/output/workspace");
    return decoder_bwd(*d, *w, *b, dlogits_packed, alphas, dalphas, *g, dann, (char*)workspace, workspace_bytes, (hipStream_t)stream);
}

int sat_ce_label_smooth_fwd(const float* logits, const int32_t* targets, int32_t P, int32_t V, float smoothing, float* lse_rows,
                            float* loss_rows, int32_t* correct_rows, float* out, void* stream) {
    if (!logits || !targets || !lse_rows || !loss_rows || !correct_rows || !out) return fail(SAT_EINVAL, "ce_label_smooth_fwd: null pointer");
    return ce_fwd(logits, targets, P, V, smoothing, lse_rows, loss_rows, correct_rows, out, (hipStream_t)stream);
}
int sat_ce_label_smooth_bwd(const float* logits, const int32_t* targets, const float* lse_rows, int32_t P, int32_t V, float smoothing,
                            const float* gscale, float* dlogits, void* stream) {
    if (!logits || !targets || !lse_rows || !dlogits) return fail(SAT_EINVAL, "ce_label_smooth_bwd: null pointer");
    return ce_bwd(logits, targets, lse_rows, P, V, smoothing, gscale, dlogits, (hipStream_t)stream);
}
int sat_doubly_stochastic_fwd(const float* alphas, int32_t N, int32_t T1, int32_t L, float gamma, float* asum, float* part, float* out, void* stream) {
    if (!alphas || !asum || !part || !out) return fail(SAT_EINVAL, "doubly_stochastic_fwd: null pointer");
    return ds_fwd(alphas, N, T1, L, gamma, asum, part, out, (hipStream_t)stream);
}
int sat_doubly_stochastic_bwd(const float* asum, const float* gscale, int32_t N, int32_t T1, int32_t L, float gamma, float* dalphas, void* stream) {
    if (!asum || !dalphas) return fail(SAT_EINVAL, "doubly_stochastic_bwd: null pointer");
    return ds_bwd(asum, gscale, N, T1, L, gamma, dalphas, (hipStream_t)stream);
}

size_t sat_decoder_infer_workspace_bytes(const sat_decoder_dims* d, int32_t max_beams) {
    if (check_dims(d) != SAT_OK || max_beams < 1) return 0;
    return decoder_infer_workspace_bytes(*d, max_beams);
}
int sat_decoder_infer_begin(const sat_decoder_dims* d, const sat_decoder_params* w, const float* ann, int32_t beams, int32_t max_beams,
                            float* h, float* c, void* workspace, size_t workspace_bytes, void* stream) {
    SAT_TRY(check_dims(d)); SAT_TRY(check_params(d, w, "decoder_infer_begin"));
    if (!ann || !h || !c || !workspace) return fail(SAT_EINVAL, "decoder_infer_begin: null pointer");
    return decoder_infer_begin(*d, *w, ann, beams, max_beams, h, c, (char*)workspace, workspace_bytes, (hipStream_t)stream);
}
int sat_decoder_infer_step(const sat_decoder_dims* d, const sat_decoder_params* w, const float* ann, const int32_t* tokens, int32_t beams,
                           int32_t max_beams, float* h, float* c, float* logits, float* alpha, const float* h_noise, void* workspace,
                           size_t workspace_bytes, void* stream) {
    SAT_TRY(check_dims(d)); SAT_TRY(check_params(d, w, "decoder_infer_step"));
    if (!ann || !tokens || !h || !c || !logits || !alpha || !workspace) return fail(SAT_EINVAL, "decoder_infer_step: null pointer");
    return decoder_infer_step(*d, *w, ann, tokens, beams, max_beams, h, c, logits, alpha, h_noise, (char*)workspace, workspace_bytes, (hipStream_t)stream);
}
size_t sat_beam_search_workspace_bytes(const sat_decoder_dims* d, int32_t beamk) {
    if (check_dims(d) != SAT_OK || beamk < 1) return 0;
    return decoder_beam_workspace_bytes(*d, beamk);
}
int sat_beam_search_batched(const sat_decoder_dims* d, const sat_decoder_params* w, const float* ann, int32_t beamk, int32_t max_gen_length,
                            const float* temperatures_host, int32_t n_temperatures, const int32_t* special_ids_host, int32_t* tok_in, int32_t* prev_row,
                            float* alpha_hist, int32_t* fin_count, int32_t* fin_step, int32_t* fin_row, float* fin_score, float* fin_mean, void* workspace,
                            size_t workspace_bytes, void* stream) {
    SAT_TRY(check_dims(d)); SAT_TRY(check_params(d, w, "beam_search_batched"));
    if (!ann || !temperatures_host || !special_ids_host || !tok_in || !prev_row || !alpha_hist || !fin_count || !fin_step || !fin_row || !fin_score || !fin_mean || !workspace)
        return fail(SAT_EINVAL, "beam_search_batched: null pointer");
    for (int i = 0; i < n_temperatures; ++i) if (!(temperatures_host[i] > 0.f)) return fail(SAT_EINVAL, "beam_search_batched: temperature %g", temperatures_host[i]);
    return decoder_beam_batched(*d, *w, ann, beamk, max_gen_length, temperatures_host, n_temperatures, special_ids_host, tok_in, prev_row, alpha_hist, fin_count,
                                fin_step, fin_row, fin_score, fin_mean, (char*)workspace, workspace_bytes, (hipStream_t)stream, nullptr);
}
int sat_beam_search_sampled(const sat_decoder_dims* d, const sat_decoder_params* w, const float* ann, int32_t beamk, int32_t max_gen_length,
                            const float* temperatures_host, int32_t n_temperatures, const int32_t* special_ids_host, const sat_beam_sampling* sampling,
                            int32_t* tok_in, int32_t* prev_row, float* alpha_hist, int32_t* fin_count, int32_t* fin_step, int32_t* fin_row, float* fin_score,
                            float* fin_mean, void* workspace, size_t workspace_bytes, void* stream) {
    SAT_TRY(check_dims(d)); SAT_TRY(check_params(d, w, "beam_search_sampled"));
    if (!ann || !temperatures_host || !special_ids_host || !tok_in || !prev_row || !alpha_hist || !fin_count || !fin_step || !fin_row || !fin_score || !fin_mean || !workspace)
        return fail(SAT_EINVAL, "beam_search_sampled: null pointer");
    for (int i = 0; i < n_temperatures; ++i) if (!(temperatures_host[i] > 0.f)) return fail(SAT_EINVAL, "beam_search_sampled: temperature %g", temperatures_host[i]);
    return decoder_beam_batched(*d, *w, ann, beamk, max_gen_length, temperatures_host, n_temperatures, special_ids_host, tok_in, prev_row, alpha_hist, fin_count,
                                fin_step, fin_row, fin_score, fin_mean, (char*)workspace, workspace_bytes, (hipStream_t)stream, sampling);
}
int sat_beam_scores(const float* logits, int32_t beams, int32_t V, float temperature, const int32_t* masked_ids, int32_t n_masked,
                    const float* parent_scores, float* scores, void* stream) {
    if (!logits || !scores || (n_masked > 0 && !masked_ids)) return fail(SAT_EINVAL, "beam_scores: null pointer");
    return beam_scores(logits, beams, V, temperature, masked_ids, n_masked, parent_scores, scores, (hipStream_t)stream);
}
int sat_topk(const float* x, float* work, int64_t n, int32_t k, float* values, int32_t* indices, void* stream) {
    if (!x || !work || !values || !indices) return fail(SAT_EINVAL, "topk: null pointer");
    return topk(x, work, n, k, values, indices, (hipStream_t)stream);
}

int sat_colsum(const float* x, int64_t ld, int64_t rows, int32_t cols, float* out, float* scratch, void* stream) {
    if (!x || !out || !scratch) return fail(SAT_EINVAL, "colsum: null pointer");
    if (rows <= 0 || cols <= 0 || ld < cols) return fail(SAT_EINVAL, "colsum: bad shape");
    return colsum_public(x, ld, rows, cols, out, scratch, (hipStream_t)stream);
}

int sat_attention_precompute(const float* ann, const float* att_enc_w, float* U, int32_t B, int32_t L, int32_t D, int32_t A, void* stream) {
    if (!ann || !att_enc_w || !U) return fail(SAT_EINVAL, "attention_precompute: null pointer");
    GemmArgs g; g.A = ann; g.lda = D; g.B = att_enc_w; g.ldb = D; g.C = U; g.ldc = A; g.M = B * L; g.N = A; g.K = D;
    return launch_gemm(g, (hipStream_t)stream);
}
int sat_attention_step_fwd(const float* ann, const float* U, const float* hc, int32_t hc_ld, const float* att_f, const int32_t* lengths,
                           int32_t step, float* alphas, int32_t T1, float* Z, float* XZ, int32_t B, int32_t R, int32_t L, int32_t D, int32_t A, void* stream) {
    if (!ann || !U || !hc || !att_f || !lengths || !alphas || !Z || !XZ) return fail(SAT_EINVAL, "attention_step_fwd: null pointer");
    if (hc_ld < A + D) return fail(SAT_EINVAL, "attention_step_fwd: hc_ld %d < A+D", hc_ld);
    return launch_attention_fwd((hipStream_t)stream, ann, U, hc, hc_ld, att_f, lengths, step, alphas, T1, Z, XZ, B, R, L, D, A);
}

int sat_attention_step_bwd(const float* ann, const float* U, const float* hc, int32_t hc_ld, const float* att_f, const int32_t* lengths, int32_t step,
                           const float* alphas, const float* dalphas, int32_t T1, const float* Z, const float* dZ, const float* dXZ, float* DZ, float* dhc,
                           int32_t dhc_ld, float* dU, float* dwf_part, float* da_scratch, int32_t B, int32_t R, int32_t L, int32_t D, int32_t A, void* stream) {
    if (!ann || !U || !hc || !att_f || !lengths || !alphas || !Z || !dZ || !dXZ || !DZ || !dhc || !dU || !dwf_part || !da_scratch)
        return fail(SAT_EINVAL, "attention_step_bwd: null pointer");
    if (hc_ld < A + D || dhc_ld < A + D) return fail(SAT_EINVAL, "attention_step_bwd: hc_ld %d / dhc_ld %d < A+D", hc_ld, dhc_ld);
    if (B < 1 || R < 1 || L < 1 || D < 1 || A < 1 || T1 < 1 || step < 0 || step >= T1) return fail(SAT_EINVAL, "attention_step_bwd: bad shape");
    return attention_step_bwd(ann, U, hc, hc_ld, att_f, lengths, step, alphas, dalphas, T1, Z, dZ, dXZ, DZ, dhc, dhc_ld, dU, dwf_part, da_scratch, B, R, L, D, A,
                              (hipStream_t)stream);
}
int sat_attention_context_bwd(const float* alphas, const float* DZ, const int32_t* lengths, float* dann, int32_t accumulate, int32_t B, int32_t R, int32_t T1,
                              int32_t L, int32_t D, void* stream) {
    if (!alphas || !DZ || !lengths || !dann) return fail(SAT_EINVAL, "attention_context_bwd: null pointer");
    if (B < 1 || R < 1 || L < 1 || D < 1 || T1 < 1) return fail(SAT_EINVAL, "attention_context_bwd: bad shape");
    return attention_context_bwd(alphas, DZ, lengths, dann, accumulate, B, R, T1, L, D, (hipStream_t)stream);
}
int sat_lstm_cell_fwd(const float* x, int32_t in, const float* h_prev, const float* c_prev, const float* w_ih, const float* w_hh, const float* b_ih,
                      const float* b_hh, float* gates, float* h_new, float* c_new, float* bias_scratch, int32_t N, int32_t n, void* stream) {
    if (!x || !h_prev || !c_prev || !w_ih || !w_hh || !b_ih || !b_hh || !gates || !h_new || !c_new || !bias_scratch) return fail(SAT_EINVAL, "lstm_cell_fwd: null pointer");
    if (N < 1 || n < 1 || in < 1) return fail(SAT_EINVAL, "lstm_cell_fwd: bad shape (N=%d n=%d in=%d)", N, n, in);
    return lstm_cell_fwd(x, in, h_prev, c_prev, w_ih, w_hh, b_ih, b_hh, gates, h_new, c_new, bias_scratch, N, n, (hipStream_t)stream);
}
int sat_lstm_cell_bwd(const float* x, int32_t in, const float* h_prev, const float* c_prev, const float* c_new, const float* gates, const float* dh_new,
                      const float* dc_new, const float* w_ih, const float* w_hh, float* dx, float* dh_prev, float* dc_prev, float* dw_ih, float* dw_hh,
                      float* db_ih, float* db_hh, float* dgates, float* scratch, int32_t N, int32_t n, void* stream) {
    if (!x || !h_prev || !c_prev || !c_new || !gates || !dh_new || !w_ih || !w_hh || !dx || !dh_prev || !dc_prev || !dw_ih || !dw_hh || !db_ih || !db_hh || !dgates || !scratch)
        return fail(SAT_EINVAL, "lstm_cell_bwd: null pointer");
    if (N < 1 || n < 1 || in < 1) return fail(SAT_EINVAL, "lstm_cell_bwd: bad shape (N=%d n=%d in=%d)", N, n, in);
    return lstm_cell_bwd(x, in, h_prev, c_prev, c_new, gates, dh_new, dc_new, w_ih, w_hh, dx, dh_prev, dc_prev, dw_ih, dw_hh, db_ih, db_hh, dgates, scratch, N, n,
                         (hipStream_t)stream);
}
int sat_deep_output_fwd(const float* prev_embed, const float* hidden, const float* context, const float* w_hidden, const float* w_context, const float* w_out,
                        const float* b_out, float dropout, uint64_t seed, float* u, float* udrop, float* logits, int32_t N, int32_t m, int32_t n, int32_t D,
                        int32_t V, void* stream) {
    if (!hidden || !w_hidden || !w_out || !u || !logits) return fail(SAT_EINVAL, "deep_output_fwd: null pointer");
    if (context && (!w_context || !prev_embed)) return fail(SAT_EINVAL, "deep_output_fwd: the deep form needs output.context.weight and the previous embedding");
    if (!(dropout >= 0.f && dropout < 1.f) || (dropout > 0.f && !udrop)) return fail(SAT_EINVAL, "deep_output_fwd: dropout %g needs a udrop buffer", dropout);
    if (N < 1 || m < 1 || n < 1 || V < 1 || (context && D < 1)) return fail(SAT_EINVAL, "deep_output_fwd: bad shape");
    return deep_output_fwd(prev_embed, hidden, context, w_hidden, w_context, w_out, b_out, dropout, seed, u, udrop, logits, N, m, n, D, V, (hipStream_t)stream);
}
int sat_deep_output_bwd(const float* dlogits, const float* hidden, const float* context, const float* u, const float* udrop, const float* w_hidden,
                        const float* w_context, const float* w_out, float dropout, uint64_t seed, float* d_prev_embed, float* d_hidden, float* d_context,
                        float* dw_hidden, float* dw_context, float* dw_out, float* db_out, float* scratch, int32_t N, int32_t m, int32_t n, int32_t D, int32_t V,
                        void* stream) {
    if (!dlogits || !hidden || !u || !w_hidden || !w_out || !d_prev_embed || !d_hidden || !dw_hidden || !dw_out || !scratch) return fail(SAT_EINVAL, "deep_output_bwd: null pointer");
    if (context && (!w_context || !d_context || !dw_context)) return fail(SAT_EINVAL, "deep_output_bwd: the deep form needs the context weight and gradient buffers");
    if (!(dropout >= 0.f && dropout < 1.f) || (dropout > 0.f && !udrop)) return fail(SAT_EINVAL, "deep_output_bwd: dropout %g needs the forward's udrop buffer", dropout);
    if (N < 1 || m < 1 || n < 1 || V < 1 || (context && D < 1)) return fail(SAT_EINVAL, "deep_output_bwd: bad shape");
    return deep_output_bwd(dlogits, hidden, context, u, udrop, w_hidden, w_context, w_out, dropout, seed, d_prev_embed, d_hidden, d_context, dw_hidden, dw_context,
                           dw_out, db_out, scratch, N, m, n, D, V, (hipStream_t)stream);
}
int sat_init_lstm_fwd(const float* ann, const float* w_f, const float* b_f, const float* w_i, const float* b_i, float dropout, uint64_t seed, float* mean, float* f,
                      float* init, int32_t N, int32_t L, int32_t D, int32_t m, int32_t n2, void* stream) {
    if (!ann || !w_f || !b_f || !w_i || !b_i || !mean || !f || !init) return fail(SAT_EINVAL, "init_lstm_fwd: null pointer");
    if (N < 1 || L < 1 || D < 1 || m < 1 || n2 < 2 || !(dropout >= 0.f && dropout < 1.f)) return fail(SAT_EINVAL, "init_lstm_fwd: bad shape / dropout");
    return init_lstm_fwd(ann, w_f, b_f, w_i, b_i, dropout, seed, mean, f, init, N, L, D, m, n2, (hipStream_t)stream);
}
int sat_init_lstm_bwd(const float* dinit, const float* mean, const float* f, const float* w_f, const float* w_i, float dropout, uint64_t seed, float* dw_f,
                      float* db_f, float* dw_i, float* db_i, float* dann, float* df, float* dmean, float* scratch, int32_t N, int32_t L, int32_t D, int32_t m,
                      int32_t n2, void* stream) {
    if (!dinit || !mean || !f || !w_f || !w_i || !dw_f || !db_f || !dw_i || !db_i || !dann || !df || !dmean || !scratch) return fail(SAT_EINVAL, "init_lstm_bwd: null pointer");
    if (N < 1 || L < 1 || D < 1 || m < 1 || n2 < 2 || !(dropout >= 0.f && dropout < 1.f)) return fail(SAT_EINVAL, "init_lstm_bwd: bad shape / dropout");
    return init_lstm_bwd(dinit, mean, f, w_f, w_i, dropout, seed, dw_f, db_f, dw_i, db_i, dann, df, dmean, scratch, N, L, D, m, n2, (hipStream_t)stream);
}

int sat_embedding_fwd(float* table, const int32_t* tokens, float* out, int32_t rows, int32_t V, int32_t m, float max_norm, int32_t* flags_scratch, void* stream) {
    if (!table || !tokens || !out || (max_norm > 0.f && !flags_scratch)) return fail(SAT_EINVAL, "embedding_fwd: null pointer");
    if (rows < 1 || V < 1 || m < 1) return fail(SAT_EINVAL, "embedding_fwd: bad shape");
    return embedding_fwd(table, tokens, out, rows, V, m, max_norm, flags_scratch, (hipStream_t)stream);
}
int sat_embedding_bwd(const float* dY, const int32_t* tokens, float* dtable, int32_t rows, int32_t V, int32_t m, int32_t padding_idx, int32_t* scratch, void* stream) {
    if (!dY || !tokens || !dtable || !scratch) return fail(SAT_EINVAL, "embedding_bwd: null pointer");
    if (rows < 1 || V < 1 || m < 1) return fail(SAT_EINVAL, "embedding_bwd: bad shape");
    return embedding_bwd(dY, tokens, dtable, rows, V, m, padding_idx, scratch, (hipStream_t)stream);
}
int sat_sigmoid_bwd(const float* dy, const float* y, float* dpre, int64_t n, void* stream) {
    if (!dy || !y || !dpre || n < 1) return fail(SAT_EINVAL, "sigmoid_bwd: null pointer / empty");
    return sigmoid_bwd(dy, y, dpre, n, (hipStream_t)stream);
}

}  // extern "C"
