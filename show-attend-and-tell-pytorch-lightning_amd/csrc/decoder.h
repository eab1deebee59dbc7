// Internal C++ entry points of the decoder (wrapped by the C ABI in api.hip).
#pragma once
#include "../../include/sat_hip.h"
#include "common.h"

namespace sat {
struct Ws;
int check_dims(const sat_decoder_dims* d);
size_t decoder_workspace_bytes(const sat_decoder_dims& d);
int decoder_fwd(const sat_decoder_dims& d, const sat_decoder_params& p, const sat_decoder_batch& b, float* logits, float* alphas,
                char* ws, size_t ws_bytes, hipStream_t st);
int decoder_bwd(const sat_decoder_dims& d, const sat_decoder_params& p, const sat_decoder_batch& b, const float* dlogits,
                const float* alphas, const float* dalphas, const sat_decoder_params& g, float* dann, char* ws, size_t ws_bytes, hipStream_t st);
int launch_attention_fwd(hipStream_t st, const float* ann, const float* U, const float* hc, int hc_ld, const float* wf,
                         const int* lengths, int step, float* alphas, int T1, float* Z, float* XZ, int B, int R, int L, int D, int A, float* sc = nullptr, void* xzb = nullptr,
                         const void* annb = nullptr);
size_t decoder_infer_workspace_bytes(const sat_decoder_dims& d, int Kmax);
int decoder_infer_begin(const sat_decoder_dims& d, const sat_decoder_params& p, const float* ann, int K, int Kmax, float* h, float* c,
                        char* ws, size_t ws_bytes, hipStream_t st);
int decoder_infer_step(const sat_decoder_dims& d, const sat_decoder_params& p, const float* ann, const int* tokens, int K, int Kmax,
                       float* h, float* c, float* logits, float* alpha, const float* h_noise, char* ws, size_t ws_bytes, hipStream_t st);
size_t decoder_beam_workspace_bytes(const sat_decoder_dims& d, int K);
int decoder_beam_batched(const sat_decoder_dims& d, const sat_decoder_params& p, const float* ann, int K, int max_gen_length, const float* temps_host,
                         int n_temps, const int* special_host, int* tok_in, int* prev_row, float* alpha_hist, int* fin_count, int* fin_step, int* fin_row,
                         float* fin_score, float* fin_mean, char* ws, size_t ws_bytes, hipStream_t st, const sat_beam_sampling* sampling);
int attention_step_bwd(const float* ann, const float* U, const float* hc, int hc_ld, const float* wf, const int* lengths, int step, const float* alphas,
                       const float* dalphas, int T1, const float* Z, const float* dZ, const float* dXZ, float* DZ, float* dhc, int dhc_ld, float* dU,
                       float* dwf_part, float* da, int B, int R, int L, int D, int A, hipStream_t st);
int attention_context_bwd(const float* alphas, const float* DZ, const int* lengths, float* dann, int accumulate, int B, int R, int T1, int L, int D, hipStream_t st);
int lstm_cell_fwd(const float* x, int in, const float* h_prev, const float* c_prev, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                  float* gates, float* h_new, float* c_new, float* bias_scratch, int N, int n, hipStream_t st);
int lstm_cell_bwd(const float* x, int in, const float* h_prev, const float* c_prev, const float* c_new, const float* gates, const float* dh_new, const float* dc_new,
                  const float* w_ih, const float* w_hh, float* dx, float* dh_prev, float* dc_prev, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh,
                  float* dgates, float* scratch, int N, int n, hipStream_t st);
int deep_output_fwd(const float* prev_embed, const float* hidden, const float* context, const float* w_hidden, const float* w_context, const float* w_out,
                    const float* b_out, float dropout, unsigned long long seed, float* u, float* udrop, float* logits, int N, int m, int n, int D, int V, hipStream_t st);
int deep_output_bwd(const float* dlogits, const float* hidden, const float* context, const float* u, const float* udrop, const float* w_hidden, const float* w_context,
                    const float* w_out, float dropout, unsigned long long seed, float* d_prev_embed, float* d_hidden, float* d_context, float* dw_hidden,
                    float* dw_context, float* dw_out, float* db_out, float* scratch, int N, int m, int n, int D, int V, hipStream_t st);
int init_lstm_fwd(const float* ann, const float* w_f, const float* b_f, const float* w_i, const float* b_i, float dropout, unsigned long long seed, float* mean,
                  float* f, float* init, int N, int L, int D, int m, int n2, hipStream_t st);
int init_lstm_bwd(const float* dinit, const float* mean, const float* f, const float* w_f, const float* w_i, float dropout, unsigned long long seed, float* dw_f,
                  float* db_f, float* dw_i, float* db_i, float* dann, float* df, float* dmean, float* scratch, int N, int L, int D, int m, int n2, hipStream_t st);
int embedding_fwd(float* table, const int* tokens, float* out, int rows, int V, int m, float max_norm, int* flags, hipStream_t st);
int embedding_bwd(const float* dY, const int* tokens, float* dtable, int rows, int V, int m, int padding_idx, int* scratch, hipStream_t st);
int sigmoid_bwd(const float* dy, const float* y, float* dpre, long n, hipStream_t st);
int beam_scores(const float* logits, int K, int V, float temperature, const int* masked, int n_masked, const float* parent, float* scores, hipStream_t st);
int topk(const float* x, float* work, long n, int k, float* values, int* indices, hipStream_t st);
int colsum_public(const float* x, long ld, long rows, int cols, float* out, float* scratch, hipStream_t st);
int ce_fwd(const float* logits, const int* targets, int P, int V, float smoothing, float* lse_rows, float* loss_rows, int* correct_rows, float* out, hipStream_t st);
int ce_bwd(const float* logits, const int* targets, const float* lse_rows, int P, int V, float smoothing, const float* gscale, float* dlogits, hipStream_t st);
int ds_fwd(const float* alphas, int N, int T1, int L, float gamma, float* asum, float* part, float* out, hipStream_t st);
int ds_bwd(const float* asum, const float* gscale, int N, int T1, int L, float gamma, float* dalphas, hipStream_t st);
}  // namespace sat
