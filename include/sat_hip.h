/* sat_hip.h -- C ABI of libsat_hip.so, the MI355X (gfx950) implementation of the
 * Show-Attend-and-Tell train-step hot path.
 *
 * The reference (Lukeasargen/Show-Attend-and-Tell-Pytorch-Lightning) is pure Python and
 * has no FFI: the boundary it offers is the Python surface of `class SAT`
 * (model.py:134) and its sub-modules.  Each entry point below replaces the stock
 * PyTorch ops behind one of those call sites (cited per function).  INTEGRATION.md
 * shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *   - fp32 tensors, int32 indices, row-major; Linear weights keep torch's (out,in)
 *     layout, conv weights are KRSC (torch channels_last memory), activations NHWC;
 *   - no allocation, no synchronisation, no ownership transfer: the caller provides
 *     outputs and a workspace (size from the *_workspace_bytes query) and a hipStream_t
 *     (passed as void*); calls are re-entrant per stream.  Stream capture (hipGraph): every entry point enqueues KERNEL
 *     launches only (clears and device-to-device copies are kernels too, csrc/devmem.hip), nothing reads host memory at
 *     execution time and nothing synchronises, so a captured stream is a chain of kernel nodes.  `_host` arrays
 *     (step_offsets_host, teacher_host) are read while ENQUEUEING: they shape the launch sequence, so a captured train step
 *     is valid for that packing plan and those teacher-forcing flags (sat_amd/graph.py keys its graphs by them).  Tested under
 *     capture + replay: sat_beam_search_batched / _sampled (tests/test_gpu_inference.py) and the whole train step - encoder,
 *     sat_decoder_train_fwd / _bwd, the losses, sat_optimizer_step_dev - bit-equal to the eager step (tests/test_gpu_graph.py);
 *   - return 0 on success, non-zero otherwise with text in sat_last_error()
 *     (thread-local).  Nothing throws, nothing exits.
 */
#ifndef SAT_HIP_H
#define SAT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAT_HIP_ABI_VERSION 23

int sat_abi_version(void);
/* dev aid: after every kernel launch wait for the device and print the launch's name to stderr (a device fault then points at the
 * launch after the last printed line).  Also switched on by SAT_TRACE_LAUNCH=1.  Not for captured streams. */
int sat_debug_trace_launches(int32_t on);
/* dev aid: tuning switches of the GEMM launcher by name ("glds_tile": force a tile form of csrc/gemm_glds.hip, -1 = automatic;
 * "glds_stages8": LDS ring depth of its 8-wave forms; "tile_override": replace the launcher's own choice).  Not for production use. */
int sat_debug_option(const char* name, int32_t value);
const char* sat_last_error(void);

/* ------------------------------------------------------------------ kernel timing for the roofline report
 * Between start and stop every instrumented launch is bracketed by HIP events on its own stream; stop
 * synchronises them and returns one entry per kernel family (process-wide; measurement runs only). */
typedef struct sat_profile_entry {
    char name[96];
    int64_t launches;
    double total_ms;      /* sum of event-measured durations          */
    double flops;         /* algorithmic FLOPs of those launches       */
    double bytes;         /* algorithmic bytes of those launches       */
} sat_profile_entry;
int sat_profile_start(void);
/* the same, recording only the scopes of one family, or of every family with a common prefix when `name` ends in '*' */
int sat_profile_start_only(const char* name);
/* recording off (1) / on again (0) between start and stop, keeping what was recorded: bench.py instruments every n-th timed step */
int sat_profile_pause(int32_t paused);
int sat_profile_stop(sat_profile_entry* out, int32_t max_entries, int32_t* n_out);

/* ------------------------------------------------------------------ GEMM family
 * C[MxN] = op(A)[MxK] * op(B)[KxN] on v_mfma_f32_32x32x2_f32.  Replaces nn.Linear
 * (model.py:72-73, 90-92, 119-123, 188) and its autograd.  amode/bmode/epi: enum values
 * of csrc/gemm.h (0 = row-major k-contiguous, 1 = k-major). */
typedef struct sat_gemm_desc {
    const float* A; int64_t lda; const int32_t* a_rows; /* optional row gather, -1 = zero row */
    const float* B; int64_t ldb;
    float* C; int64_t ldc;       const int32_t* c_rows; /* optional row scatter, -1 = skip */
    int32_t M, N, K;
    int32_t amode, bmode;
    int32_t accumulate;          /* C += */
    int32_t epi;                 /* 0 none, 1 +bias, 2 +bias & sigmoid on cols [c0,c1), 3 tanh(v+e0), 4 v*(1-e0^2), 5 relu(v+bias) */
    const float* bias; const float* e0; int64_t lde0; int32_t c0, c1;
    float* slab; int64_t slab_elems;   /* optional split-K scratch */
} sat_gemm_desc;
int sat_gemm_f32(const sat_gemm_desc* d, void* stream);
/* Same contraction with explicit storage types and matrix-core choice: *_bf16 = operand stored as bf16 in HBM
 * (pointers in the descriptor are then reinterpreted), bf16_mfma = use v_mfma_f32_32x32x16_bf16 with fp32
 * accumulation (fp32 operands are rounded to bf16 on the way into LDS). */
typedef struct sat_gemm_types { int32_t a_bf16, b_bf16, c_bf16, bf16_mfma; } sat_gemm_types;
int sat_gemm_ex(const sat_gemm_desc* d, const sat_gemm_types* t, void* stream);

/* ------------------------------------------------------------------ decoder (train_batch, model.py:474-557) */
typedef struct sat_decoder_dims {
    int32_t B;            /* images in the batch                                   */
    int32_t R;            /* captions per image (lengths.size(1), model.py:487)    */
    int32_t T;            /* caption tensor width; T-1 decode steps                */
    int32_t L;            /* locations h*w                                         */
    int32_t D;            /* encoder_dim                                           */
    int32_t A;            /* attention_dim                                         */
    int32_t m;            /* embed_dim                                             */
    int32_t n;            /* decoder_dim                                           */
    int32_t V;            /* vocab_size                                            */
    int32_t P;            /* packed tokens = sum(lengths)                          */
    int32_t deep_output;  /* DeepOutput.deep (model.py:116)                        */
    int32_t padding_idx;  /* <PAD> id (model.py:162)                               */
    int32_t precision;    /* 0: exact fp32 MFMA (parity mode); 1: bf16 MFMA, fp32 accumulate/state */
    int32_t layers;       /* nn.LSTM num_layers (model.py:178), 1..SAT_MAX_LSTM_LAYERS; h/c are (layers, N, n) */
    float embed_max_norm; /* nn.Embedding max_norm (model.py:161); <= 0: off.  Renormalises embedding rows IN PLACE */
    float dropout;        /* nn.Dropout p of InitLSTM / DeepOutput (model.py:74,117) in training mode; 0: off     */
    float embedding_dropout; /* p of embedding_dropout (model.py:164); 0: off                                    */
    uint64_t dropout_seed;   /* masks = counter-based hash of (seed, stream, element): same seed for fwd and bwd */
} sat_decoder_dims;

/* state-dict tensors of the decoder; used for weights and, with the same
 * field meaning, for gradient outputs. */
#define SAT_MAX_LSTM_LAYERS 4
typedef struct sat_decoder_params {
    float* embedding;                 /* embedding.weight            (V, m)      */
    float* init_f_w; float* init_f_b; /* init_lstm.factorize         (m, D), (m) */
    float* init_i_w; float* init_i_b; /* init_lstm.init              (2n*layers, m), (2n*layers) */
    float* w_ih; float* w_hh;         /* lstm.weight_ih_l0 (4n, m+D), lstm.weight_hh_l0 (4n, n) */
    float* b_ih; float* b_hh;         /* lstm.bias_ih_l0, lstm.bias_hh_l0 (4n)   */
    float* att_enc;                   /* attention.encoder_att.weight (A, D)     */
    float* att_dec;                   /* attention.decoder_att.weight (A, n)     */
    float* att_f;                     /* attention.f_att.weight       (1, A)     */
    float* beta_w; float* beta_b;     /* beta.0                       (D, n), (D) */
    float* out_hidden;                /* output.hidden.weight         (m, n)     */
    float* out_context;               /* output.context.weight        (m, D), NULL when shallow */
    float* out_w; float* out_b;       /* output.output (V, m), (V); out_b NULL under weight tying */
    /* stacked layers l = 1 .. layers-1 at index l-1: lstm.weight_ih_l{l} (4n, n), weight_hh_l{l} (4n, n), bias_ih_l{l}, bias_hh_l{l} (4n) */
    float* up_w_ih[SAT_MAX_LSTM_LAYERS - 1]; float* up_w_hh[SAT_MAX_LSTM_LAYERS - 1];
    float* up_b_ih[SAT_MAX_LSTM_LAYERS - 1]; float* up_b_hh[SAT_MAX_LSTM_LAYERS - 1];
} sat_decoder_params;

typedef struct sat_decoder_batch {
    const float* ann;            /* (B, L, D) annotations = encoder output, NHWC flattened          */
    const int32_t* caps;         /* (B*R, T) encoded captions                                        */
    const int32_t* lengths;      /* (B*R)                                                            */
    const int32_t* prow;         /* (T-1, B*R) packed row of (step, caption) or -1 when finished    */
    const int32_t* src_row;      /* (P) step*N + caption of packed row p                             */
    const int32_t* step_offsets_host; /* HOST (T) : first packed row of each step, [T-1] = P         */
    const int32_t* teacher_host;      /* HOST (T-1): 1 = feed the caption token, 0 = argmax of the
                                         previous step's logits (scheduled sampling, model.py:518-523) */
} sat_decoder_batch;

size_t sat_decoder_workspace_bytes(const sat_decoder_dims* d);

/* train_batch decoder forward (model.py:487-548): writes logits in packed order
 * (pack_padded_sequence(...).data order, model.py:553) and alphas (N, T-1, L). */
int sat_decoder_train_fwd(const sat_decoder_dims* d, const sat_decoder_params* w, const sat_decoder_batch* b,
                          float* logits_packed, float* alphas, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of the above through time (what autograd does for loss.backward()).
 * dlogits_packed (P, V); dalphas (N, T-1, L) or NULL.  Overwrites every field of `g`
 * and dann (B, L, D).  Must follow a forward on the same workspace. */
int sat_decoder_train_bwd(const sat_decoder_dims* d, const sat_decoder_params* w, const sat_decoder_batch* b,
                          const float* dlogits_packed, const float* alphas /* forward output */, const float* dalphas,
                          const sat_decoder_params* g, float* dann, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ losses
 * LabelSmoothing.forward (util.py:105-112) over packed rows.  out[0] = loss, out[1] = accuracy
 * (model.py:596-597).  scratch: P floats lse + P floats loss + P ints. */
int sat_ce_label_smooth_fwd(const float* logits, const int32_t* targets, int32_t P, int32_t V, float smoothing,
                            float* lse_rows, float* loss_rows, int32_t* correct_rows, float* out, void* stream);
/* dlogits = gscale[0] * d loss / d logits  (gscale device pointer or NULL = 1) */
int sat_ce_label_smooth_bwd(const float* logits, const int32_t* targets, const float* lse_rows, int32_t P, int32_t V,
                            float smoothing, const float* gscale, float* dlogits, void* stream);
/* model.py:594: out[0] = gamma * mean((1 - sum_t alphas)^2); asum (N, L) kept for backward;
 * part: ceil(N*L/256) floats scratch. */
int sat_doubly_stochastic_fwd(const float* alphas, int32_t N, int32_t T1, int32_t L, float gamma,
                              float* asum, float* part, float* out, void* stream);
int sat_doubly_stochastic_bwd(const float* asum, const float* gscale, int32_t N, int32_t T1, int32_t L, float gamma,
                              float* dalphas, void* stream);

/* ------------------------------------------------------------------ inference: SAT.forward / caption (model.py:214-472)
 * The reference decodes one image at a time with the beam as the batch (model.py:260-266).  `begin` computes att_enc
 * and the initial state of `beams` rows (InitLSTM over the expanded annotations, F3 reshape: model.py:265-269);
 * `step` is one pass of model.py:298-327 for the live beams (h, c updated in place; d->B/R/T/P are ignored);
 * `sat_beam_scores` = log_softmax(logit / temperature) with the special tokens masked and the parent scores added
 * (model.py:330-351); `sat_topk` = torch.topk on the flattened scores (model.py:343, 359). */
size_t sat_decoder_infer_workspace_bytes(const sat_decoder_dims* d, int32_t max_beams);
int sat_decoder_infer_begin(const sat_decoder_dims* d, const sat_decoder_params* w, const float* ann /* (L, D) */, int32_t beams,
                            int32_t max_beams, float* h /* (layers, beams, n) */, float* c, void* workspace, size_t workspace_bytes, void* stream);
int sat_decoder_infer_step(const sat_decoder_dims* d, const sat_decoder_params* w, const float* ann, const int32_t* tokens, int32_t beams,
                           int32_t max_beams, float* h, float* c, float* logits /* (beams, V) */, float* alpha /* (beams, L) */,
                           const float* h_noise /* (layers, beams, n) or NULL: decoder_noise (model.py:322-324), added to h for the
                                                   LSTM update only; attention and the beta gate see the clean h[-1] */,
                           void* workspace, size_t workspace_bytes, void* stream);
/* Batched beam search ("beam" sampling, model.py:260-448 for every image of the batch at once; d->B = images, rows are (B, beamk)).
 * Enqueues max_gen_length + 1 decode steps without a host round trip and leaves the back-trace on the device:
 *   tok_in   (max_gen_length + 2, B, beamk)  token fed to each live row of each step ([0] = START)
 *   prev_row (max_gen_length + 2, B, beamk)  row of the previous step the hypothesis came from
 *   alpha_hist (max_gen_length + 1, B, beamk, L)  attention weights of each step's rows
 *   fin_*    (B, beamk)  finished hypotheses in the reference's append order: step at which they ended, the row they came from, raw score,
 *            mean beam score at that moment (for "BAR" rescoring); fin_count (B)
 * temperatures and special ids {START, PAD, END, UNK} are HOST arrays (read while enqueueing). */
size_t sat_beam_search_workspace_bytes(const sat_decoder_dims* d, int32_t beamk);
int sat_beam_search_batched(const sat_decoder_dims* d, const sat_decoder_params* w, const float* ann /* (B, L, D) */, int32_t beamk,
                            int32_t max_gen_length, const float* temperatures_host, int32_t n_temperatures, const int32_t* special_ids_host,
                            int32_t* tok_in, int32_t* prev_row, float* alpha_hist, int32_t* fin_count, int32_t* fin_step, int32_t* fin_row,
                            float* fin_score, float* fin_mean, void* workspace, size_t workspace_bytes, void* stream);
/* The same search with the continuing hypotheses DRAWN instead of taken (SAT.forward sample_method, model.py:360-379) and / or
 * decoder noise (model.py:322-324).  torch.multinomial(p, k) without replacement is an ordered Plackett-Luce sample; it is drawn
 * here as the top k of log p + Gumbel noise.  The variates come from a counter-based hash of (seed, step, row, candidate) or
 * from the caller's tables (tests):
 *   gumbel  : method 1: (max_gen_length + 1, B * beamk, V)            Gumbel(0, 1) variate of candidate word v of a row
 *             method 2: (max_gen_length + 1, B * beamk, sample_topk)  ... of the row's t-th best candidate (torch.topk order)
 *   normals : (max_gen_length + 1, layers, B * beamk, n) standard normals of the state noise
 * rows are the compacted beam rows of that step (image b owns rows b * beamk ...).  sampling = NULL is sat_beam_search_batched. */
#define SAT_SAMPLE_BEAM 0
#define SAT_SAMPLE_MULTINOMIAL 1
#define SAT_SAMPLE_TOPK 2
typedef struct sat_beam_sampling {
    int32_t method;            /* SAT_SAMPLE_*                                                  */
    int32_t sample_topk;       /* candidates per hypothesis, method 2                            */
    uint64_t seed;
    const float* gumbel;       /* device pointer or NULL                                         */
    float decoder_noise;       /* 0 = off; step s adds N(0,1) * decoder_noise / (s + 1) to h     */
    int32_t reserved;
    const float* normals;      /* device pointer or NULL                                         */
} sat_beam_sampling;
int sat_beam_search_sampled(const sat_decoder_dims* d, const sat_decoder_params* w, const float* ann, int32_t beamk, int32_t max_gen_length,
                            const float* temperatures_host, int32_t n_temperatures, const int32_t* special_ids_host, const sat_beam_sampling* sampling,
                            int32_t* tok_in, int32_t* prev_row, float* alpha_hist, int32_t* fin_count, int32_t* fin_step, int32_t* fin_row,
                            float* fin_score, float* fin_mean, void* workspace, size_t workspace_bytes, void* stream);
int sat_beam_scores(const float* logits, int32_t beams, int32_t V, float temperature, const int32_t* masked_ids /* device */,
                    int32_t n_masked, const float* parent_scores /* (beams) or NULL */, float* scores, void* stream);
int sat_topk(const float* x, float* work /* n floats scratch */, int64_t n, int32_t k, float* values, int32_t* indices, void* stream);

/* out[c] = sum_r x[r*ld + c] in a fixed order (bias gradients).  scratch: ceil(rows/256)*cols floats */
int sat_colsum(const float* x, int64_t ld, int64_t rows, int32_t cols, float* out, float* scratch, void* stream);

/* ------------------------------------------------------------------ sub-module entry points
 * (drop-in for attention / init_lstm called on their own, e.g. from caption(), model.py:269,299) */
/* att_enc = ann * W_e^T (model.py:100), hoisted */
int sat_attention_precompute(const float* ann, const float* att_enc_w, float* U, int32_t B, int32_t L, int32_t D, int32_t A, void* stream);
/* one SoftAttention.forward + beta gate for N = B*R rows: hc (N, hc_ld) holds [q | beta | ...] */
int sat_attention_step_fwd(const float* ann, const float* U, const float* hc, int32_t hc_ld, const float* att_f,
                           const int32_t* lengths, int32_t step, float* alphas, int32_t T1, float* Z, float* XZ,
                           int32_t B, int32_t R, int32_t L, int32_t D, int32_t A, void* stream);

/* Backward of sat_attention_step_fwd (what autograd does for SoftAttention.forward + the gate product, model.py:94-109, 541).
 * dZ: gradient of the un-gated context Z (from DeepOutput), dXZ: gradient of the gated context beta * z (from the LSTM input).
 * Outputs: DZ (rows, D) = total gradient of z; dhc[:, 0:A] = gradient of q = h W_d^T, dhc[:, A:A+D] = gradient of the gate's
 * pre-activation; dU (B, L, A) and dwf_part (B, A) are ACCUMULATED (+=: zero them before the first step; attention.f_att.weight's
 * gradient is the column sum of dwf_part); da_scratch: (B*R, L) floats.  dalphas: external gradient of alphas (same layout) or NULL.
 * The annotation gradient is  dann = sat_attention_context_bwd(alphas, DZ) + dU * W_e  (the second term a GEMM). */
int sat_attention_step_bwd(const float* ann, const float* U, const float* hc, int32_t hc_ld, const float* att_f, const int32_t* lengths, int32_t step,
                           const float* alphas, const float* dalphas, int32_t T1, const float* Z, const float* dZ, const float* dXZ, float* DZ, float* dhc,
                           int32_t dhc_ld, float* dU, float* dwf_part, float* da_scratch, int32_t B, int32_t R, int32_t L, int32_t D, int32_t A, void* stream);
/* dann[b, l, :] (+)= sum over the image's R captions and their live steps t < lengths of alphas[b*R + r, t, l] * DZ[t, b*R + r, :]
 * (alphas (B*R, T1, L); DZ time-major (T1, B*R, D)) */
int sat_attention_context_bwd(const float* alphas, const float* DZ, const int32_t* lengths, float* dann, int32_t accumulate, int32_t B, int32_t R, int32_t T1,
                              int32_t L, int32_t D, void* stream);
/* nn.LSTM, one layer, one time step for N rows (model.py:175-180, called with seq_len 1 at model.py:326, 544).  x (N, in), weights in
 * torch's layout (4n, in) / (4n, n), gate order i,f,g,o.  `gates` (N, 4n) receives the ACTIVATED gates, the only tensor backward needs
 * besides the states.  bias_scratch: 4n floats.  Backward: dx, dh_prev, dc_prev and the four parameter gradients are overwritten;
 * dc_new may be NULL (no gradient into the new cell state); dgates (N, 4n) and scratch (ceil(N/256) * 4n floats) are work space. */
int sat_lstm_cell_fwd(const float* x, int32_t in, const float* h_prev, const float* c_prev, const float* w_ih, const float* w_hh, const float* b_ih,
                      const float* b_hh, float* gates, float* h_new, float* c_new, float* bias_scratch, int32_t N, int32_t n, void* stream);
int sat_lstm_cell_bwd(const float* x, int32_t in, const float* h_prev, const float* c_prev, const float* c_new, const float* gates, const float* dh_new,
                      const float* dc_new, const float* w_ih, const float* w_hh, float* dx, float* dh_prev, float* dc_prev, float* dw_ih, float* dw_hh,
                      float* db_ih, float* db_hh, float* dgates, float* scratch, int32_t N, int32_t n, void* stream);
/* DeepOutput.forward (model.py:125-131): logits = (dropout(tanh(prev_embed + hidden W_h^T + context W_c^T))) W_o^T + b_o; context = NULL is the
 * shallow form (x = hidden W_h^T; prev_embed unused).  u (N, m) keeps x for backward, udrop (N, m) the dropped copy (dropout > 0 only; masks =
 * counter-based hash of (seed, element), the same in backward).  Backward overwrites every output; d_prev_embed doubles as the (N, m) work
 * buffer in the shallow form; db_out may be NULL (weight tying); scratch: ceil(N/256) * V floats. */
int sat_deep_output_fwd(const float* prev_embed, const float* hidden, const float* context, const float* w_hidden, const float* w_context, const float* w_out,
                        const float* b_out, float dropout, uint64_t seed, float* u, float* udrop, float* logits, int32_t N, int32_t m, int32_t n, int32_t D,
                        int32_t V, void* stream);
int sat_deep_output_bwd(const float* dlogits, const float* hidden, const float* context, const float* u, const float* udrop, const float* w_hidden,
                        const float* w_context, const float* w_out, float dropout, uint64_t seed, float* d_prev_embed, float* d_hidden, float* d_context,
                        float* dw_hidden, float* dw_context, float* dw_out, float* db_out, float* scratch, int32_t N, int32_t m, int32_t n, int32_t D, int32_t V,
                        void* stream);
/* InitLSTM.forward (model.py:76-81) for N annotation rows (N, L, D): mean over L -> dropout -> factorize -> init.  `init` (N, n2 = 2 n layers)
 * row-major IS the reference's (2 layers, N, n) buffer: its `.reshape` without a permute is a reinterpretation, h0 = the first
 * layers*N*n floats, c0 = the rest.  mean (N, D) (after dropout) and f (N, m) are kept for backward, which overwrites the four parameter
 * gradients and dann (N, L, D); df (N, m), dmean (N, D), scratch (ceil(N/256) * max(n2, m) floats) are work space. */
int sat_init_lstm_fwd(const float* ann, const float* w_f, const float* b_f, const float* w_i, const float* b_i, float dropout, uint64_t seed, float* mean, float* f,
                      float* init, int32_t N, int32_t L, int32_t D, int32_t m, int32_t n2, void* stream);
int sat_init_lstm_bwd(const float* dinit, const float* mean, const float* f, const float* w_f, const float* w_i, float dropout, uint64_t seed, float* dw_f,
                      float* db_f, float* dw_i, float* db_i, float* dann, float* df, float* dmean, float* scratch, int32_t N, int32_t L, int32_t D, int32_t m,
                      int32_t n2, void* stream);

/* nn.Embedding(V, m, max_norm, padding_idx) as the reference calls it (model.py:158-164; uses at 298, 526).  Forward: rows of `table`
 * gathered by token id (-1 = zero row); max_norm > 0 first renormalises, IN PLACE like torch, the table rows that occur in `tokens`
 * (flags_scratch: V ints).  Backward: dtable fully overwritten; rows of dY are added per vocabulary row in increasing row order (no
 * floating-point atomics), the padding row stays zero; scratch: 3 V + 1 + rows ints. */
int sat_embedding_fwd(float* table, const int32_t* tokens, float* out, int32_t rows, int32_t V, int32_t m, float max_norm, int32_t* flags_scratch, void* stream);
int sat_embedding_bwd(const float* dY, const int32_t* tokens, float* dtable, int32_t rows, int32_t V, int32_t m, int32_t padding_idx, int32_t* scratch, void* stream);
/* backward of the beta gate's Sigmoid (model.py:187-192): dpre = dy * y * (1 - y); the Linear around it is sat_gemm_f32 (epi 2 forward) */
int sat_sigmoid_bwd(const float* dy, const float* y, float* dpre, int64_t n, void* stream);

/* ------------------------------------------------------------------ encoder (get_encoder, model.py:16-63)
 * The reference builds the encoder from torchvision ResNet layers (model.py:19-29) + Normalize
 * (model.py:59) + an optional 1x1 Conv2d with bias (model.py:53).  These entry points are the layers
 * of that nn.Sequential on NHWC fp32 activations and KRSC filters. */
typedef struct sat_conv_geom {
    int32_t N, H, W, C;     /* input  (N, H, W, C), C % 4 == 0                    */
    int32_t K, R, S;        /* filter (K, R, S, C), K % 4 == 0                    */
    int32_t stride, pad;    /* output (N, P, Q, K), P = (H + 2 pad - R)/stride + 1 */
    int32_t stride_w;       /* 0: = stride.  Otherwise the horizontal stride (bf16 forward / weight gradient only): the stem's tap-pair view */
} sat_conv_geom;
/* nn.Conv2d forward / autograd (input gradient, weight gradient) as implicit GEMMs on MFMA */
int sat_conv2d_fwd(const float* x, const float* w, const float* bias /* or NULL */, float* y, const sat_conv_geom* g, void* stream);
int sat_conv2d_dgrad(const float* dy, const float* w, float* dx, const sat_conv_geom* g, int32_t accumulate, void* stream);
int sat_conv2d_wgrad(const float* dy, const float* x, float* dw, const sat_conv_geom* g, float* slab, int64_t slab_elems, void* stream);
size_t sat_conv2d_wgrad_slab_bytes(const sat_conv_geom* g);
/* bf16 storage variants (activations and filters bf16 in HBM, C % 8 == 0 and K % 8 == 0): bf16 MFMA, fp32
 * accumulation; the weight gradient is produced in fp32 (master weights stay fp32). */
int sat_conv2d_fwd_bf16(const void* x, const void* w, const float* bias, void* y, const sat_conv_geom* g, void* stream);
/* forward that also leaves BatchNorm statistics of its (bf16-rounded) output: per row tile of *tile_rows output pixels and per
 * filter the fp32 (sum, sum of squares) -> tile_stats[tile][K][2] (sat_conv2d_fwd_stats_bytes).  *tile_rows = 0 when the launch
 * went to a kernel without that epilogue (the caller then takes sat_bn_train_fwd_t).  sat_bn_train_fwd_tiles_bf16 is
 * sat_bn_train_fwd_t(dtype = bf16) without the statistics pass over x. */
size_t sat_conv2d_fwd_stats_bytes(const sat_conv_geom* g);
int sat_conv2d_fwd_bf16_stats(const void* x, const void* w, void* y, const sat_conv_geom* g, float* tile_stats, int32_t* tile_rows, void* stream);
int sat_bn_train_fwd_tiles_bf16(const void* x, int64_t rows, int32_t C, const float* tile_stats, int32_t tile_rows, const float* gamma,
                                const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* save_mean,
                                float* save_invstd, const void* residual, int32_t relu, void* y, uint8_t* relu_mask, float* scratch, void* stream);
/* The same with the residual given RAW: residual_raw is the input of another train-mode BatchNorm (the projection shortcut of a ResNet block,
 * statistics already in res_mean / res_invstd); y = relu(bn(x) + bn_res(residual_raw)) with the shortcut's normalised value rounded to bf16 as if it had
 * been stored, but never written.  Bit-identical to sat_bn_train_fwd_tiles_bf16 on the materialised shortcut. */
int sat_bn_train_fwd_tiles_bf16_resbn(const void* x, int64_t rows, int32_t C, const float* tile_stats, int32_t tile_rows, const float* gamma, const float* beta,
                                      float eps, float momentum, float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                                      const void* residual_raw, const float* res_mean, const float* res_invstd, const float* res_gamma, const float* res_beta,
                                      int32_t relu, void* y, uint8_t* relu_mask, float* scratch, void* stream);
int sat_conv2d_dgrad_bf16(const void* dy, const void* w, void* dx, const sat_conv_geom* g, int32_t accumulate, void* stream);
/* Data gradient that also leaves the BACKWARD statistics of the BatchNorm in front of this convolution: dx is the gradient of that
 * BatchNorm's (ReLU'd) output, so with its input bn_x (same NHWC shape as dx), the forward's ReLU sign mask (or NULL: no ReLU) and its
 * saved mean / invstd, the epilogue adds up per row tile and channel (sum g, sum g * xhat), g = the stored dx masked by the sign bits ->
 * tile_stats[tile][C][2] (sat_conv2d_dgrad_stats_bytes).  sat_bn_train_bwd_tiles_bf16 is sat_bn_train_bwd_t(dtype = bf16) without the
 * statistics pass over dy and x.  *tile_rows = 0 when the launch could not produce them (stride 2, odd shapes): take sat_bn_train_bwd_t.
 * With accumulate = 1 the statistics are those of the accumulated dx. */
size_t sat_conv2d_dgrad_stats_bytes(const sat_conv_geom* g);
int sat_conv2d_dgrad_bf16_bnstats(const void* dy, const void* w, void* dx, const sat_conv_geom* g, int32_t accumulate, const void* bn_x,
                                  const uint8_t* bn_relu_mask, const float* bn_mean, const float* bn_invstd, float* tile_stats, int32_t* tile_rows, void* stream);
/* The same launch joining a residual block's identity path: dx = dgrad(dy, w) + add_src, add_src (bf16, dx's shape: the gradient that arrived at
 * the block's output) gated per element by the bits of add_mask (the block's final ReLU sign mask, or NULL) - the masked gradient is never
 * written out on its own (torchvision Bottleneck / BasicBlock backward: out = relu(bn(...) + identity), model.py:19-29).  stride 1 only.
 * bn_x = NULL: no statistics (the other bn_* / tile_* arguments are ignored). */
int sat_conv2d_dgrad_bf16_fused(const void* dy, const void* w, void* dx, const sat_conv_geom* g, const void* add_src, const uint8_t* add_mask, const void* bn_x,
                                const uint8_t* bn_relu_mask, const float* bn_mean, const float* bn_invstd, float* tile_stats, int32_t* tile_rows, void* stream);
int sat_bn_train_bwd_tiles_bf16(const void* dy, const void* x, int64_t rows, int32_t C, const float* tile_stats, int32_t tile_rows, const float* save_mean,
                                const float* save_invstd, const float* gamma, int32_t relu, void* dx, float* dgamma, float* dbeta, void* dres,
                                int32_t dres_accumulate, const uint8_t* relu_mask, float* scratch, void* stream);
int sat_conv2d_wgrad_bf16(const void* dy, const void* x, float* dw, const sat_conv_geom* g, float* slab, int64_t slab_elems, void* stream);
/* ---- the residual-block loop of the trunk inside the library (training, bf16 storage; csrc/encoder_loop.hip) --------------------------------
 * Replaces, per block, the ~10 forward and ~14 backward per-layer calls above (and the tensor allocations between them) that a host-side
 * driver of torchvision's BasicBlock / Bottleneck (inside `self.encoder(img)`, model.py:19-29, 483, and its autograd) would issue: ONE call
 * for the forward of a run of blocks, ONE for their backward, over one activation arena whose layout is a pure function of the descriptors.
 * The loop calls the same per-layer entry points in the same order: results are bit-identical to driving them one by one.
 * Pointers: filters are the bf16 KRSC copies the convolutions read; BatchNorm index 0..3 = bn1, bn2, bn3 (bottleneck only), downsample's;
 * gradient destinations are fp32 (filters KRSC).  Backward only: dw* / dgamma / dbeta.                                                      */
typedef struct sat_block_desc {
    int32_t kind;                 /* 0 = BasicBlock (3x3, 3x3), 1 = Bottleneck (1x1, 3x3 strided, 1x1)                       */
    int32_t stride, cin, mid, cout, has_ds;
    int32_t N, H, W;              /* the block's input map (N, H, W, cin)                                                     */
    int32_t fwd_res_bn;           /* projection blocks: the shortcut's BatchNorm applied inside the last BatchNorm's kernel   */
    int32_t dgrad_join;           /* the block's input gradient joins identity + conv path inside the data-gradient launch    */
    int32_t bn_bwd_epilogue;      /* BatchNorm-backward statistics out of the data-gradient epilogues                         */
    const void *w1, *w2, *w3, *wd;
    const float* gamma[4]; const float* beta[4]; float* running_mean[4]; float* running_var[4]; float eps[4]; float momentum[4];
    float *dw1, *dw2, *dw3, *dwd; float* dgamma[4]; float* dbeta[4];
} sat_block_desc;
size_t sat_encoder_blocks_arena_bytes(const sat_block_desc* blocks, int32_t nblocks);
/* x: bf16 (N, H, W, cin) input of blocks[0]; *out_ptr = the last block's output inside the arena; bn_scratch: sat_bn_scratch_bytes of the largest map */
int sat_encoder_blocks_fwd(const sat_block_desc* blocks, int32_t nblocks, const void* x, void* arena, size_t arena_bytes, float* bn_scratch, void** out_ptr,
                           void* stream);
/* Backward of blocks [last .. first], descending (the whole trunk, or one ResNet stage at a time so that a gradient exchange can start per stage).
 * dout: bf16 gradient of block `last`'s output; dout_tiles / dout_tile_rows: the BatchNorm-backward statistics that came with it from the previous
 * call (NULL / 0 for the first call).  *dx_ptr = gradient of block `first`'s input inside the arena, *dx_tiles / *dx_tile_rows its statistics.
 * side_stream (or NULL): the weight gradients are launched there (fork / join through `event`, a hipEvent_t of the caller; joined before the call
 * returns), with their own split-K scratch.                                                                                                   */
int sat_encoder_blocks_bwd(const sat_block_desc* blocks, int32_t nblocks, int32_t first, int32_t last, const void* x, void* arena, size_t arena_bytes, const void* dout,
                           const float* dout_tiles, int32_t dout_tile_rows, float* bn_scratch, float* slab_main, int64_t slab_main_elems, float* slab_side,
                           int64_t slab_side_elems, void* side_stream, void* event, void** dx_ptr, float** dx_tiles, int32_t* dx_tile_rows, void* stream);

/* ResNet stem tail in one pass (model.py:19-29 keeps torchvision's bn1 -> relu -> maxpool): BatchNorm(train statistics already in
 * mean / invstd: sat_bn_train_fwd_t with y = NULL computes them and updates the running statistics) + ReLU + MaxPool2d(3, 2, 1)
 * of the NHWC convolution output x (N, H, W, C) -> y_pool (N, P, Q, C), argmax (same shape, bytes: window position of the first
 * maximum).  The backward takes the pooled gradient and returns dx, dgamma, dbeta; the full-size BatchNorm output, its sign
 * mask and the dense pre-pool gradient are never materialised.  scratch: sat_bn_scratch_bytes(N * H * W, C). */
int sat_stem_tail_fwd_t(int32_t dtype, const void* x, int32_t N, int32_t H, int32_t W, int32_t C, const float* mean, const float* invstd, const float* gamma,
                        const float* beta, void* y_pool, uint8_t* argmax, void* stream);
int sat_stem_tail_bwd_t(int32_t dtype, const void* dy_pool, const uint8_t* argmax, const void* x, int32_t N, int32_t H, int32_t W, int32_t C, const float* mean,
                        const float* invstd, const float* gamma, const float* beta, void* dx, float* dgamma, float* dbeta, float* scratch, void* stream);
/* torchvision Normalize(mean, std) (model.py:59) fused with NCHW -> NHWC and 3 -> 4 channel padding */
int sat_image_normalize_nhwc4(const float* img_nchw, float* out_nhwc4, int32_t N, int32_t H, int32_t W,
                              const float* mean3_host, const float* std3_host, void* stream);
/* bf16 stem as a 7 x 4 convolution over PAIRS of pixels (model.py:19-29 keeps torchvision's conv1: 7x7, stride 2, pad 3, 3 channels).
 * With 4 stored channels a pixel is 8 bytes, so the 16 bytes one lane gathers hold two horizontally adjacent pixels = two filter
 * taps.  The normalised image is written zero-padded by 3 on every side, (N, H + 6, W + 6, 4) bf16; viewed as (N, H + 6, (W + 6) / 2, 8)
 * the stem is R = 7, S = 4, C = 8, stride 2 vertically and 1 horizontally (stride_w), no padding: 224 products per output instead
 * of the 392 of an 8-channel layout (5 of 8 channels zero).  W must be even.  filter_pairs: (K, 7, 7, 3) fp32 -> (K, 7, 4, 8) bf16
 * (tap 2s in slots 0-2, tap 2s + 1 in slots 4-6, the eighth tap and slots 3 / 7 zero); grad_unpairs: its fp32 gradient back. */
int sat_image_normalize_nhwc4_padded_bf16(const float* img_nchw, void* out, int32_t N, int32_t H, int32_t W, const float* mean3_host, const float* std3_host, void* stream);
int sat_stem_filter_pairs(const float* w3, void* w_pairs_bf16, int32_t K, void* stream);
/* resnext archs (model.py:28 keeps torchvision's resnext50_32x4d / resnext101_32x8d in the ResNet branch): their 3x3 convolutions have `groups`
 * groups.  The grouped filter (K, R, S, C / groups; fp32 or bf16) is expanded into the dense block-diagonal filter (K, R, S, C) that
 * sat_conv2d_* read, and the gradient of the grouped filter is gathered from the diagonal blocks of the dense filter's fp32 gradient.       */
int sat_grouped_filter_expand(const void* w_grouped, void* w_dense, int32_t K, int32_t C, int32_t RS, int32_t groups, int32_t bf16, void* stream);
int sat_grouped_filter_grad_extract(const float* dw_dense, float* dw_grouped, int32_t K, int32_t C, int32_t RS, int32_t groups, void* stream);
int sat_stem_filter_grad_unpairs(const float* dw_pairs, float* dw3, int32_t K, void* stream);
/* ShuffleNetV2 trunk (the reference's CLI default --encoder_arch shufflenet_v2_x0_5, train.py:43; model.py:30-31 keeps torchvision's
 * conv1, maxpool, stage2-4, conv5).  dtype: 0 = fp32, 1 = bf16 activations; NHWC; C a multiple of 4 (fp32) / 8 (bf16).
 *   depthwise 3x3, pad 1, stride 1 | 2 (InvertedResidual.depthwise_conv): w = the (C, 1, 3, 3) fp32 parameter as it lies in memory ([C][9],
 *     master weights in both modes); fp32 accumulation.  wgrad: fp32 dw [C][9] from per-chunk partial sums added in a fixed order
 *     (scratch: sat_dwconv3x3_wgrad_scratch_bytes);
 *   channel_shuffle(cat(a, b), groups = 2) of two (rows, Ch) branches: element c of the result = (c odd ? b : a)[c / 2].  join writes
 *     either `full` (rows, 2 Ch), or - full == NULL - its two halves x1 = [:, :Ch], x2 = [:, Ch:] as separate dense tensors (what the
 *     next stride-1 unit's x.chunk(2, dim = 1) reads); split is the backward: (d full | its halves) -> (da, db).
 *     Ch = the real branch width, Chp >= Ch (a multiple of 4) the width of the branch tensors and of the halves in memory: channels past Ch
 *     are zero padding and stay zero (x1_0 / x2_0: 58- / 122-channel branches held in 64 / 128); `full` has (2 Ch rounded up to 8) channels. */
int sat_dwconv3x3_fwd_t(int32_t dtype, const void* x, const float* w, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t stride, void* stream);
int sat_dwconv3x3_dgrad_t(int32_t dtype, const void* dy, const float* w, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t stride, void* stream);
size_t sat_dwconv3x3_wgrad_scratch_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t stride);
int sat_dwconv3x3_wgrad_t(int32_t dtype, const void* dy, const void* x, float* dw, int32_t N, int32_t H, int32_t W, int32_t C, int32_t stride,
                          float* scratch, void* stream);
int sat_shuffle_join_t(int32_t dtype, const void* a, const void* b, void* full, void* x1, void* x2, int64_t rows, int32_t Ch, int32_t Chp, void* stream);
int sat_shuffle_split_t(int32_t dtype, const void* dfull, const void* dx1, const void* dx2, void* da, void* db, int64_t rows, int32_t Ch, int32_t Chp, void* stream);
/* bf16 -> fp32 copy (n % 8 == 0): the annotations of a bf16 trunk without the 1x1 projection (encoder_dim == trunk width, model.py:56-57) */
int sat_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream);
/* (pixels, 3) <-> (pixels, 4) zero padded; used for the stem filters */
int sat_pad_channels_3to4(const float* src, float* dst, int64_t pixels, int32_t inverse, void* stream);
/* nn.BatchNorm2d in training mode over a (rows, C) NHWC view, fused with the residual add and ReLU of the
 * ResNet blocks.  scratch: sat_bn_scratch_bytes(rows, C).  Updates running stats like PyTorch. */
size_t sat_bn_scratch_bytes(int64_t rows, int32_t C);
int sat_bn_train_fwd(const float* x, int64_t rows, int32_t C, const float* gamma, const float* beta, float eps, float momentum,
                     float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                     const float* residual /* or NULL */, int32_t relu, float* y, float* scratch, void* stream);
/* eval-mode BatchNorm: y = (x - running_mean) / sqrt(running_var + eps) * gamma + beta (+ residual)(ReLU) */
int sat_bn_eval_fwd(const float* x, int64_t rows, int32_t C, const float* running_mean, const float* running_var, float eps,
                    const float* gamma, const float* beta, const float* residual, int32_t relu, float* y, void* stream);
/* backward: dy is the gradient of the fused output y; dres (optional) receives the gradient of the residual input */
int sat_bn_train_bwd(const float* dy, const float* x, const float* y, int64_t rows, int32_t C, const float* save_mean,
                     const float* save_invstd, const float* gamma, int32_t relu, float* dx, float* dgamma, float* dbeta,
                     float* dres, int32_t dres_accumulate, float* scratch, void* stream);
/* Storage-typed variants (dtype 0 = fp32, 1 = bf16 activations; statistics, affine parameters and their
 * gradients stay fp32, reductions accumulate in double) */
int sat_bn_train_fwd_t(int32_t dtype, const void* x, int64_t rows, int32_t C, const float* gamma, const float* beta, float eps, float momentum,
                       float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                       const void* residual, int32_t relu, void* y,
                       uint8_t* relu_mask /* or NULL; rows*C/8 bytes, C % 8 == 0: bit i of byte b = (element 8b+i of y > 0); relu == 2 (ReLU6,
                                             mobilenet_v2: y clamped to [0, 6]): bit = (0 < pre-clamp value < 6), i.e. "the gradient passes" */,
                       float* scratch, void* stream);
int sat_bn_eval_fwd_t(int32_t dtype, const void* x, int64_t rows, int32_t C, const float* running_mean, const float* running_var, float eps,
                      const float* gamma, const float* beta, const void* residual, int32_t relu, void* y, void* stream);
int sat_bn_train_bwd_t(int32_t dtype, const void* dy, const void* x, const void* y, int64_t rows, int32_t C, const float* save_mean,
                       const float* save_invstd, const float* gamma, int32_t relu, void* dx, float* dgamma, float* dbeta,
                       void* dres, int32_t dres_accumulate,
                       const uint8_t* relu_mask /* or NULL: the forward's sign mask, read instead of y (which may then be NULL) */,
                       float* scratch, void* stream);
int sat_maxpool3x3s2_fwd_t(int32_t dtype, const void* x, void* y, uint8_t* argmax, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int sat_maxpool3x3s2_bwd_t(int32_t dtype, const void* dy, const uint8_t* argmax, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
/* bf16 plumbing: fp32 -> bf16 copies of filters/gradients (n % 4 == 0); Normalize fused with NCHW -> NHWC and
 * 3 -> 8 channel padding; stem filter (K,7,7,3) fp32 -> (K,7,7,8) bf16 and its fp32 gradient back */
int sat_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
int sat_image_normalize_nhwc8_bf16(const float* img_nchw, void* out_nhwc8, int32_t N, int32_t H, int32_t W,
                                   const float* mean3_host, const float* std3_host, void* stream);
int sat_stem_filter_pad(const float* w3, void* w8_bf16, int64_t pixels, void* stream);
int sat_stem_filter_grad_unpad(const float* dw8, float* dw3, int64_t pixels, void* stream);
/* nn.MaxPool2d(3, 2, 1) of the ResNet stem; argmax (N,P,Q,C) bytes keep the window position */
int sat_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* argmax, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int sat_maxpool3x3s2_bwd(const float* dy, const uint8_t* argmax, float* dx, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
/* encoder_size option (readme.md:118-121): AdaptiveAvgPool2d when P <= H, bilinear Upsample(align_corners=False) otherwise */
int sat_resize_fwd(const float* x, float* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t P, int32_t Q, void* stream);
int sat_resize_bwd(const float* dy, float* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t P, int32_t Q, void* stream);

/* ------------------------------------------------------------------ optimizer (SAT.configure_optimizers, model.py:723-757;
 * gradient clipping train.py:93-96,273-274): every parameter tensor updated by one launch.
 * `tensors` and `chunks` are DEVICE arrays built by the caller: one sat_opt_tensor per parameter (its gradient pointer
 * changes per step), one sat_opt_chunk per sat_optimizer_chunk_elems() elements of a tensor. */
#define SAT_OPT_SGD 0
#define SAT_OPT_ADAM 1
#define SAT_OPT_ADAMW 2
typedef struct sat_opt_tensor {
    float* p; const float* g;
    float* m;             /* Adam exp_avg / SGD momentum_buffer (NULL when unused) */
    float* v;             /* Adam exp_avg_sq                                       */
    int64_t n;
    float lr, weight_decay;    /* of the parameter's group                        */
    void* shadow_bf16;         /* or NULL: the updated parameter is also written here as bf16 (same element order): the
                                  filter copies the bf16 convolutions read, kept current without a cast pass per step */
} sat_opt_tensor;
typedef struct sat_opt_chunk { int32_t tensor; int32_t reserved; int64_t start; } sat_opt_chunk;
typedef struct sat_opt_hyper {
    int32_t kind;              /* SAT_OPT_*                                            */
    int32_t nesterov, first_step;   /* SGD: nesterov momentum; first step (momentum buffer = gradient) */
    float beta1, beta2, eps;
    float bias_correction1;         /* 1 - beta1^step                                  */
    float bias_correction2_sqrt;    /* sqrt(1 - beta2^step)                            */
    float momentum;
    float clip_value;               /* > 0: clamp every gradient element to [-v, v] (clip_grad_value_) */
} sat_opt_hyper;
int32_t sat_optimizer_chunk_elems(void);
/* coef[0] = min(1, max_norm / (||g||_2 + 1e-6)) over ALL tensors (clip_grad_norm_), coef[1] = the norm; fixed-order reduction */
int sat_grad_clip_coef(const sat_opt_tensor* tensors, const sat_opt_chunk* chunks, int32_t n_chunks, float max_norm, double* scratch,
                       float* coef, void* stream);
/* p, m, v updated in place from g * clip_coef[0] (clip_coef NULL = 1) */
int sat_optimizer_step(const sat_opt_tensor* tensors, const sat_opt_chunk* chunks, int32_t n_chunks, const sat_opt_hyper* hyper,
                       const float* clip_coef, void* stream);
/* the same launch with the hyper-parameters in DEVICE memory (one sat_opt_hyper): a captured / replayed launch (hipGraph) then follows the
 * step count (bias corrections) and the clipping settings that the host writes there before every replay (sat_amd/graph.py)            */
int sat_optimizer_step_dev(const sat_opt_tensor* tensors, const sat_opt_chunk* chunks, int32_t n_chunks, const sat_opt_hyper* hyper_dev,
                           const float* clip_coef, void* stream);

/* ---- input pipeline on device ------------------------------------------------------------------------
 * Replaces, per batch, the per-sample PIL / torchvision chain of train.py:208-233: T.RandomResizedCrop | T.Resize +
 * T.CenterCrop (Pillow's antialiased BILINEAR Image.resize, bit exact), T.RandomHorizontalFlip, T.ToTensor, and
 * util.py:121-130 AddGaussianNoise.  Input: the decoded pictures of one batch, uint8 RGB (height, width, 3), concatenated
 * in one device buffer; one descriptor per picture (the random draws are the caller's: torch RNG on the host).
 * Output: (n, 3, out_h, out_w) fp32 in [0, 1] (+ noise * std) - the `img` of the reference's batch (util.py:40-45).      */
typedef struct sat_image_desc {
    int64_t offset;                                    /* byte offset of the picture in `pixels`                        */
    int32_t height, width;
    int32_t crop_top, crop_left, crop_h, crop_w;       /* box cut out first (PIL crop)                                  */
    int32_t resized_h, resized_w;                      /* size the box is resampled to (PIL resize, BILINEAR)           */
    int32_t out_top, out_left;                         /* (out_h, out_w) window of the resampled picture (T.CenterCrop) */
    int32_t flip;                                      /* mirror left-right                                             */
    int32_t reserved;
} sat_image_desc;
size_t sat_image_batch_workspace_bytes(const sat_image_desc* desc_host, int32_t n, int32_t out_h, int32_t out_w);
/* desc_host / desc_dev: the same n descriptors in host memory (validated, sizes the launch) and in device memory (read by
 * the kernels).  noise: (n, 3, out_h, out_w) standard normal draws or NULL.  out_u8 (optional): the (n, out_h, out_w, 3)
 * bytes before ToTensor, i.e. what PIL would hold.  At least one of out_nchw / out_u8.  pixels_bytes bounds the offsets. */
int sat_image_batch_transform(const uint8_t* pixels, int64_t pixels_bytes, const sat_image_desc* desc_host, const sat_image_desc* desc_dev,
                              int32_t n, int32_t out_h, int32_t out_w, const float* noise, float noise_std, float* out_nchw,
                              uint8_t* out_u8, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SAT_HIP_H */
