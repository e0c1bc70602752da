/*
 * qwen3tts_talker.h -- C ABI of the MI355X talker library (drop-in for the
 * reference's llama_wrapper.so).
 *
 * The 12 wrapper_* entry points below replace, symbol for symbol, the ones the
 * reference exports from dual_npu/llama_wrapper.c and binds through ctypes in
 * dual_npu/llama_cpp_bindings.py:41-81.  Signatures use only scalars and
 * plain pointers (the reason the reference's shim exists: llama_wrapper.c:4-5).
 *
 * Ownership: model/context objects are owned by the library and released by
 * wrapper_free_*; every buffer argument is caller-owned host memory and every
 * call is synchronous (outputs valid on return).  Threading: one caller thread
 * per context, not re-entrant (same as the reference).  Errors: return codes
 * plus a line on stderr; no exceptions cross the boundary.
 *
 * `path` is this build's own packed fp16/fp32 weight file (Q3TTSW1 container,
 * see DESIGN.md), or -- parsed natively, no Python -- the safetensors file the
 * reference's own converter writes before its GGUF step (re-keyed Qwen3 talker,
 * scripts/extract_talker_as_qwen3.py:53-71), or the HF snapshot's
 * model.safetensors / its directory; not a GGUF.  There is no CPU fallback: if no HIP device is
 * usable, wrapper_load_model returns NULL after printing the reason.
 */
#ifndef QWEN3TTS_TALKER_H
#define QWEN3TTS_TALKER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* llama_wrapper.c:25-31 -- process-wide init/teardown (HIP runtime here). */
void wrapper_backend_init(void);
void wrapper_backend_free(void);

/* llama_wrapper.c:33-37 -- load weights; n_gpu_layers is accepted and ignored
 * (everything lives on the GPU).  NULL on failure. */
void* wrapper_load_model(const char* path, int n_gpu_layers);
/* llama_wrapper.c:39-41 */
void wrapper_free_model(void* model);
/* llama_wrapper.c:43-45 -- hidden size (1024 for Qwen3-TTS-0.6B), >0. */
int wrapper_model_n_embd(const void* model);

/* llama_wrapper.c:49-69 -- allocate the KV cache for n_ctx positions.
 * n_batch bounds n_tokens of one wrapper_decode_embd call (the reference
 * passes n_batch = n_ctx); n_threads is a CPU hint and is ignored;
 * embeddings must be non-zero (the library only has an embeddings mode).
 * NULL on failure. */
void* wrapper_create_context(void* model, int n_ctx, int n_batch, int n_threads, int embeddings);
/* llama_wrapper.c:71-73 */
void wrapper_free_context(void* ctx);

/* llama_wrapper.c:77-80 -- forget all cached positions of sequence 0. */
void wrapper_kv_clear(void* ctx);

/* llama_wrapper.c:84-109 -- KV prefix cache persistence (used by
 * llamacpp_talker_server.py:226-246).  The file format is this library's own
 * (versioned header + fp16 K/V of the occupied positions); get_size returns
 * the number of bytes save_file would write for the current state. */
size_t wrapper_state_get_size(void* ctx);
int wrapper_state_save_file(void* ctx, const char* path); /* 0 / -1 */
int wrapper_state_load_file(void* ctx, const char* path); /* 0 / -1 */

/* llama_wrapper.c:125-163 -- the hot call.  Feeds n_tokens embedding rows
 * (embd[n_tokens][n_embd], row-major f32) at positions pos_start..pos_start+
 * n_tokens-1 of sequence 0, causal attention over everything cached below,
 * and writes the post-final-RMSNorm hidden state of the LAST row to
 * out_hidden[n_embd].  Returns 0 ok, -1 decode failed (bad sizes, context
 * overflow, HIP error), -2 no output produced. */
int wrapper_decode_embd(void* ctx, const float* embd, int n_tokens, int n_embd,
                        int pos_start, float* out_hidden);

/* ---- extensions (not in the reference; seq_id is hard-wired to 0 there,
 * llama_wrapper.c:139-141) ------------------------------------------------ */

/* Number of independent sequences (KV slots) a context was created with. */
int wrapper_ctx_n_slots(void* ctx);
/* Re-create semantics: like wrapper_create_context but with n_slots sequences. */
void* wrapper_create_context_slots(void* model, int n_ctx, int n_batch, int n_slots);
/* One decode step for n_rows rows at once: row r feeds embd[r] at position
 * pos[r] of sequence slot_ids[r] (all slots distinct) and gets out[r].
 * Same return codes as wrapper_decode_embd. */
int wrapper_decode_embd_batch(void* ctx, const float* embd, int n_rows, int n_embd,
                              const int32_t* slot_ids, const int32_t* pos, float* out_hidden);
/* Prefill n_tokens rows into one slot (wrapper_decode_embd with a slot). */
int wrapper_decode_embd_slot(void* ctx, int slot, const float* embd, int n_tokens, int n_embd,
                             int pos_start, float* out_hidden);
/* Codec head on the device (llamacpp_talker_server.py:165): logits[vocab] =
 * fp16(hidden) . codec_head^T.  Returns vocab size or <0. */
int wrapper_codec_head(void* ctx, const float* hidden, int n_rows, float* logits_out);

#ifdef __cplusplus
}
#endif
#endif /* QWEN3TTS_TALKER_H */
