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
                         const int* lengths, int step, float* alphas, int T1, float* Z, float* XZ, int B, int R, int L, int D, int A, float* sc = nullptr, void* xzb = nullptr);
size_t decoder_infer_workspace_bytes(const sat_decoder_dims& d, int Kmax);
int decoder_infer_begin(const sat_decoder_dims& d, const sat_decoder_params& p, const float* ann, int K, int Kmax, float* h, float* c,
                        char* ws, size_t ws_bytes, hipStream_t st);
int decoder_infer_step(const sat_decoder_dims& d, const sat_decoder_params& p, const float* ann, const int* tokens, int K, int Kmax,
                       float* h, float* c, float* logits, float* alpha, const float* h_noise, char* ws, size_t ws_bytes, hipStream_t st);
size_t decoder_beam_workspace_bytes(const sat_decoder_dims& d, int K);
int decoder_beam_batched(const sat_decoder_dims& d, const sat_decoder_params& p, const float* ann, int K, int max_gen_length, const float* temps_host,
                         int n_temps, const int* special_host, int* tok_in, int* prev_row, float* alpha_hist, int* fin_count, int* fin_step, int* fin_row,
                         float* fin_score, float* fin_mean, char* ws, size_t ws_bytes, hipStream_t st, const sat_beam_sampling* sampling);
int beam_scores(const float* logits, int K, int V, float temperature, const int* masked, int n_masked, const float* parent, float* scores, hipStream_t st);
int topk(const float* x, float* work, long n, int k, float* values, int* indices, hipStream_t st);
int colsum_public(const float* x, long ld, long rows, int cols, float* out, float* scratch, hipStream_t st);
int ce_fwd(const float* logits, const int* targets, int P, int V, float smoothing, float* lse_rows, float* loss_rows, int* correct_rows, float* out, hipStream_t st);
int ce_bwd(const float* logits, const int* targets, const float* lse_rows, int P, int V, float smoothing, const float* gscale, float* dlogits, hipStream_t st);
int ds_fwd(const float* alphas, int N, int T1, int L, float gamma, float* asum, float* part, float* out, hipStream_t st);
int ds_bwd(const float* asum, const float* gscale, int N, int T1, int L, float gamma, float* dalphas, hipStream_t st);
}  // namespace sat
