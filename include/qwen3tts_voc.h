/*
 * qwen3tts_voc.h -- C ABI of the MI355X vocoder library (codec ids -> 24 kHz waveform, fp32).
 *
 * The reference has no C ABI here: dual_npu/vocoder_server.py:67-71 calls onnxruntime
 * (`sess.run(None, {'audio_codes': i64[1,64,16]})[0].flatten()`), the graph being the traced
 * Qwen3TTSTokenizerV2 decoder of scripts/export_vocoder_traced.py:38-52 (input [B,T,16] int64,
 * permuted to [B,16,T]; output wav.squeeze(1), length T * total_upsample = T * 1920).  voc_decode
 * is that call; voc_synthesize is VocoderServer.synthesize + the int16 rule
 * (vocoder_server.py:73-121,175) including its chunk-length quirk.
 *
 * The decoder's layer list is NOT in the reference: the library executes
 * the op table stored in the weight container (`voc.program`, DESIGN.md "Vocoder program"), so a
 * real checkpoint only needs converting, not a rebuild.  The table's semantics are pinned to the
 * importable implementation of the decoder family (transformers' Qwen3OmniMoeCode2Wav + Mimi's split
 * RVQ: tests/golden/make_code2wav_golden.py).  That family's transposed convs trim kernel - stride
 * samples at BOTH ends, so a decode of T frames returns fewer than T * 1920 samples (64 frames ->
 * 122 325): like the ONNX model's output tensor, voc_decode's rows are voc_chunk_samples() long, and
 * the slices the reference takes of them (`audio[:n * 1920]`, vocoder_server.py:81,98-99) follow
 * numpy's rule -- as long as what is there.
 *
 * Caller-owned host buffers, synchronous, one caller thread per handle, no CPU fallback.
 */
#ifndef QWEN3TTS_VOC_H
#define QWEN3TTS_VOC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define Q3VOC_SAMPLES_PER_TOKEN 1920 /* vocoder_server.py:30 */
#define Q3VOC_SAMPLE_RATE 24000      /* vocoder_server.py:29 */

/* weights: Q3TTSW1 container holding voc.*.  chunk_tokens: frames per decode call (the ONNX model's
 * fixed input length, 64 in the reference: vocoder_server.py:45-46); max_batch chunks per call. */
void* voc_load(const char* weights, int chunk_tokens, int max_batch);
void voc_free(void* v);
int voc_chunk_tokens(void* v);
int voc_samples_per_token(void* v);   /* decoder.total_upsample (export_vocoder_traced.py:46): the product of the strides */
int voc_chunk_samples(void* v);      /* samples one decode of chunk_tokens frames returns (<= chunk_tokens * samples_per_token) */

/* codes[B][chunk_tokens][16] int64 (ids 0..2047; out-of-range ids embed as zeros) ->
 * out[B][voc_chunk_samples()] f32 in [-1, 1] (the ONNX output tensor's shape).  0 ok / <0 error. */
int voc_decode(void* v, const int64_t* codes, int B, float* out);

/* VocoderServer.synthesize + int16 conversion for one utterance: codes[n][16] -> out samples.
 * out must hold voc_synthesize_max_samples(n) int16.  Returns 0 and *n_samples, or <0.
 * (n > chunk_tokens walks chunks with a 16-frame overlap: needs chunk_tokens > 32.)
 * A chunk of fewer than chunk_tokens frames (an utterance's tail, or a short utterance) is decoded at its own length + 1
 * pad frame, rounded up to 8, instead of the reference's zero-padded chunk_tokens: the decoder is causal but for a quarter
 * frame of look-ahead, so the samples the walk keeps are bit-identical (env Q3_VOC_FULL_CHUNKS=1: pad as the reference). */
int voc_synthesize(void* v, const int64_t* codes, int n_tokens, int16_t* out, int32_t* n_samples);
/* same, float output before the int16 rule */
int voc_synthesize_f32(void* v, const int64_t* codes, int n_tokens, float* out, int32_t* n_samples);
int voc_synthesize_max_samples(void* v, int n_tokens);

/* The same for U utterances in one call (BASELINE configs[2]: "streaming overlap-crossfade vocoder" at batch): codes =
 * the utterances' frames concatenated ([sum n_tokens][16]); the chunks of ALL utterances are decoded max_batch at a time
 * and every utterance's chunk walk -- 64-frame chunks stepping by 48, the 16-frame linear cross-fade, the appended short
 * tail chunk (vocoder_server.py:84-117) -- is assembled on the device, bit-identical to voc_synthesize[_f32] per
 * utterance.  out holds voc_synthesize_batch_max_samples() samples (out_capacity of them are the caller's);
 * utterance u's samples are out[offsets[u] .. offsets[u+1]) (offsets has U + 1 entries).  0 ok / <0 error. */
int voc_synthesize_batch(void* v, const int64_t* codes, const int32_t* n_tokens, int U, int16_t* out, int64_t out_capacity,
                         int64_t* offsets);
int voc_synthesize_batch_f32(void* v, const int64_t* codes, const int32_t* n_tokens, int U, float* out, int64_t out_capacity,
                             int64_t* offsets);
int64_t voc_synthesize_batch_max_samples(void* v, const int32_t* n_tokens, int U);
/* GPU milliseconds and decoded chunks of the last voc_synthesize_batch* call */
float voc_last_batch_ms(void* v);
int voc_last_batch_chunks(void* v);

/* Arithmetic of the convolutions.  Default (0): split precision -- every f32 operand (weights once at load,
 * activations in the producing kernel's epilogue) is carried as two fp16 terms (22 mantissa bits) and each
 * product costs three fp16 MFMAs with f32 accumulation.  1: the exact-f32 MFMA (v_mfma_f32_32x32x2_f32)
 * everywhere.  Measured on MI355X against a float64 evaluation of the same table: max error 2.2e-7 (split)
 * vs 4.0e-7 (exact f32 MFMA) vs 2.1e-7 (torch CPU f32) of full scale -- the split path is fp32-grade, and
 * 2.8x faster at 32 chunks.  Values beyond the fp16 range cannot be split: an op with such a weight stays
 * exact, and a call in which an activation leaves the range is redone on the exact path before voc_decode
 * returns.  Process-wide; env Q3_VOC_EXACT=1 selects the exact path at load. */
int voc_set_exact_fp32(int on);

/* 1 (default): on the exact-f32 path a residual unit of the 96- / 192-channel decoder blocks (Snake, dilated 7-tap
 * conv, Snake, 1x1 conv, + input) runs as ONE launch whose intermediate stays in MFMA accumulators; 0: one launch
 * per conv (the 1x1 conv then sums its channels in a different order: results differ in the last f32 bit). */
int voc_set_fused_units(int on);

/* Cap the workgroups each vocoder kernel launch occupies (0 = one per output tile, the default and the fastest for a decode that
 * has the GPU to itself; -1 = one per compute unit of the current device).  With a cap the kernels walk their tiles persistently
 * -- same tiles, same sums, same bits -- and leave registers and LDS of every compute unit to a concurrently running frame loop
 * (talker / code predictor), which is latency-bound and otherwise finds room only in the tails of the vocoder's launches.  Measured
 * on MI355X, 32 utterances, frame loop of step s + 1 beside the decode of step s: exactly one workgroup per CU is the optimum
 * (the decode alone 89 -> 125 ms, the frame step beside it 2.40 -> 2.9 ms instead of starving, the whole step 244 -> 209 ms with the
 * frame loop's waves at raised priority, which the library's kernels set themselves); 320 / 384 / 512 / 768 workgroups: 239 / 228 /
 * 234 / 245 ms.  Returns the cap in effect.  Process-wide. */
int voc_set_max_workgroups(int n);

/* GPU milliseconds of the last voc_decode (HIP events on the library's stream) and its FLOP count. */
float voc_last_decode_ms(void* v);
double voc_decode_flops(void* v, int B);

#ifdef __cplusplus
}
#endif
#endif /* QWEN3TTS_VOC_H */
