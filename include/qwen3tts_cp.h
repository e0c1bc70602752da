/*
 * qwen3tts_cp.h -- C ABI of the MI355X code-predictor library.
 *
 * The reference has no C ABI for this stage: the Python server calls
 * onnxruntime (`sess.run`, dual_npu/code_predictor_server.py:77-85) sixteen
 * times per frame and the native servers wrap the same loop
 * (code_predictor_cpp/code_predictor_server.cpp:259-416; the one-call form
 * qwen3_tts::TTSTransformer::predict_codes_autoregressive(hidden, code_0, out,
 * temperature, top_k) at code_predictor_ggml/code_pred_server.cpp:214-215).
 * cp_predict is that one-call form; cp_step mirrors the ONNX decode-step I/O
 * (hidden,position,past_k/v -> hidden,k/v) for step-level parity checks.
 *
 * Semantics kept from the reference: the code_0 embedding comes from the
 * TALKER codec table (code_predictor_server.py:97-98); group g>=1 embeds the
 * previous token with CP table g-1 (:134); positions 0..15; an out-of-range
 * token embeds as zeros (code_predictor_server.cpp:374-380).  temperature <=
 * 1e-6 is greedy (the reference's max(T,1e-6) softmax collapses to argmax); above it the device
 * draws from the top_k (<= 64) logits with softmax((l-max)/T) using a counter-based generator keyed by
 * `seed` (reproducible per seed; not numpy's / mt19937's stream).
 *
 * Buffers are caller-owned host memory; calls are synchronous; one caller
 * thread per handle.  No CPU fallback.
 */
#ifndef QWEN3TTS_CP_H
#define QWEN3TTS_CP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define Q3CP_NUM_GROUPS 15

/* Load from a packed weight file (Q3TTSW1) holding cp.* and
 * talker.codec_embedding, or from a reference-layout directory (contains
 * code_predictor_weights.npz; codec_embedding.npy is looked up in
 * `embeddings_dir`, which may be NULL when `weights` is a packed file).
 * max_batch bounds n_rows of cp_predict_batch.  NULL on failure. */
void* cp_load(const char* weights, const char* embeddings_dir, int max_batch);
void cp_free(void* h);
int cp_hidden_size(void* h);

/* One frame: hidden[1024] (talker post-norm hidden) + code_0 -> out_codes[15].
 * 0 ok / <0 error. */
int cp_predict(void* h, const float* hidden, int32_t code_0, float temperature, int top_k,
               uint64_t seed, int32_t* out_codes);
/* n_rows independent frames at once: hidden[n_rows][1024], code_0[n_rows],
 * out_codes[n_rows][15]. */
int cp_predict_batch(void* h, const float* hidden, const int32_t* code_0, int n_rows,
                     float temperature, int top_k, uint64_t seed, int32_t* out_codes);

/* Step-level mirror of the ONNX graph (code_predictor_server.py:77-85) for
 * row 0: feeds one embedding at `position` (KV of positions < position must
 * have been fed in order since the last position-0 call) and returns the
 * post-final-norm hidden[1024]. */
int cp_step(void* h, const float* embed, int position, float* out_hidden);
/* lm_head of group g on the device: logits[2048] = fp16(hidden) . lm_head_g^T
 * (code_predictor_server.py:129,136; matmul_neon code_predictor_server.cpp:58-86). */
int cp_lm_head(void* h, int group, const float* hidden, float* logits_out);

#ifdef __cplusplus
}
#endif
#endif /* QWEN3TTS_CP_H */
