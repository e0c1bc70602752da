/*
 * qwen3tts_text.h -- C ABI of the text front-end on the GPU.
 *
 * Stands where the reference's talker server computes, in numpy on the host, the text projection
 * `_embed_text` (gather text_embedding[ids] f32 [151936, 2048] -> fc1 + bias -> SiLU -> fc2 + bias -> [n, 1024])
 * and the dual-stream prefix `_build_prefix` (dual_npu/llamacpp_talker_server.py:115-161; tables loaded at
 * :79-93 from scripts/extract_embeddings.py:47-66's files).  Here the 1.24 GB f32 table lives in HBM as fp16
 * (0.62 GB), fc1 / fc2 are fp16 MFMA GEMMs with f32 accumulation (the talker's linear kernels; the biases are
 * the accumulators' initial values), SiLU and the prefix assembly are device kernels; the host hands over
 * token ids and receives the prefix rows it passes to wrapper_decode_embd / q3e_start.
 *
 * Numerics: the table and the two weight matrices are rounded to fp16 and the fc1 output after SiLU is a
 * fp16 GEMM input, everything else is f32: results agree with the reference's f32 arithmetic to ~1e-3 relative
 * (tests/test_gpu_text.py states the tolerance).  Prefix layout (rows): 3 role tokens (text only), tts_pad +
 * codec{nothink, think_bos, think_eos}, tts_bos + codec_pad, n x (text + codec_pad), tts_eos + codec_pad,
 * tts_pad + codec_bos = n + 9 rows.
 *
 * Caller-owned host buffers, synchronous calls, one caller thread per handle, no CPU fallback.
 */
#ifndef QWEN3TTS_TEXT_H
#define QWEN3TTS_TEXT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* weights: Q3TTSW1 container (or the HF snapshot's model.safetensors) holding text.embedding, text.fc{1,2}.{weight,bias}
 * and talker.codec_embedding; embeddings_dir: the reference's own embeddings/ directory instead (may be NULL).
 * max_tokens: longest text accepted by one call.  NULL on failure (message on stderr). */
void* tfe_load(const char* weights, const char* embeddings_dir, int max_tokens);
void tfe_free(void* h);
int tfe_hidden_size(void* h);      /* 1024 */
int tfe_text_vocab(void* h);

/* _embed_text: n token ids -> out[n][hidden] f32.  0 ok / <0 error (id out of range, n > max_tokens + 6). */
int tfe_embed_text(void* h, const int32_t* token_ids, int n, float* out);

/* _build_prefix: n text token ids (n >= 0) -> out[(n + 9)][hidden]; returns the number of rows, <0 on error.
 * `special` = the 12 ids {im_start, assistant, newline, tts_pad, tts_bos, tts_eos, codec_pad, codec_bos,
 * codec_nothink, codec_think_bos, codec_think_eos, 0} (llamacpp_talker_server.py:44-55,132); NULL = the
 * container's meta / the 0.6B defaults. */
int tfe_build_prefix(void* h, const int32_t* text_token_ids, int n, const int32_t* special, float* out);

/* The projected tts_pad row (tts_client.py:58-69 computes it once; q3e_set_pad_embed takes it). */
int tfe_tts_pad_embed(void* h, float* out /*[hidden]*/);

#ifdef __cplusplus
}
#endif
#endif /* QWEN3TTS_TEXT_H */
