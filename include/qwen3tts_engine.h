/*
 * qwen3tts_engine.h -- C ABI of the fused on-device frame loop (batch mode).
 *
 * Not present in the reference: there the autoregressive loop is closed by the Python client
 * over two sockets per frame (dual_npu/tts_client.py:144-215): talker hidden + code_0
 * (llamacpp_talker_server.py:254-293) -> code predictor (code_predictor_server.py:94-140)
 * -> feedback embedding (tts_client.py:199-208) -> next talker step.  This library keeps that
 * whole cycle on the GPU for B independent utterances: per frame one
 * hipGraph launch; the host only reads the codec ids.  Greedy decoding (the reference's
 * --temperature 0 limit); EOS rule, EOS boost, repetition penalty and the 2048..2149 / >=2151
 * mask are those of llamacpp_talker_server.py:163-206,258.
 *
 * Caller-owned host buffers, synchronous calls, one caller thread per handle, no CPU fallback.
 */
#ifndef QWEN3TTS_ENGINE_H
#define QWEN3TTS_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Number of visible HIP devices (0 = none: every loader of this library then returns NULL; there is no
 * CPU path), and selection of the device used by handles created afterwards (one process per GPU:
 * the per-GPU launcher sets HIP_VISIBLE_DEVICES, a torchrun rank passes LOCAL_RANK). */
int q3_device_count(void);
int q3_device_compute_units(void);   /* compute units of the current device (256 on an MI355X) */
int q3_set_device(int device);

/* weights: Q3TTSW1 container with talker.* and cp.*.  max_batch utterances at once, n_ctx talker
 * positions per utterance (prefix + frames), max_frames frames kept per utterance. */
void* q3e_create(const char* weights, int max_batch, int n_ctx, int max_frames);
void q3e_free(void* e);

/* Sampling.  Defaults are greedy (temperature 0 = the reference's --temperature 0 limit).  With
 * temperature > 1e-6 the device draws from top-k / temperature (/ top-p for the talker) like
 * llamacpp_talker_server.py:191-206 and code_predictor_server.py:87-92; the generator is counter based
 * (seed, request, utterance, frame, group): the first q3e_start after this call draws from `seed` itself,
 * every later one from a stream derived from (seed, request index), so a server replays nothing across
 * requests, a run is reproducible for a seed, and nothing is bit-compatible with numpy's or mt19937's streams.
 * top_k <= 0 (or >= the vocabulary) keeps every entry, as in the reference. */
int q3e_set_sampling(void* e, float talker_temperature, int talker_top_k, float talker_top_p,
                     float cp_temperature, int cp_top_k, uint64_t seed);

/* Teacher forcing: forced[f][b][16] (f < n_frames, b < B of the batch just started; entries < 0 = free-running).
 * Every decision the device takes is still recorded in the codes array, but the ids that are FED BACK (the
 * repetition window, the code predictor's inputs, the feedback embedding) are the forced ones -- continuing a
 * given codec prompt, and what the parity tests use to grade every decision of a run against the oracle, not only
 * those before the first near-tie.  Call after q3e_start; NULL switches it off; the next q3e_start resets it. */
int q3e_set_forced_codes(void* e, const int32_t* forced, int n_frames);

/* Split every frame step into n (1..8) independent row groups that run as parallel branches of the
 * captured graph: hides per-kernel launch latency behind the other groups' work at the price of
 * streaming the weights n times.  Default 1 (env Q3_CHAINS overrides): on ROCm 7.2 the
 * per-chain graphs were measured NOT to overlap, so more chains only re-stream the weights. */
int q3e_set_chains(void* e, int n);

/* tts_pad embedding added to every feedback (tts_client.py:207-208); zeros until set. */
int q3e_set_pad_embed(void* e, const float* pad_embed /*[hidden]*/);

/* Begin a batch of B utterances.  prefix: the dual-stream prefix rows of all utterances,
 * concatenated ([sum n_rows][hidden] f32, llamacpp_talker_server.py:121-161); n_rows[b] rows
 * belong to utterance b; n_text[b] = number of text tokens (EOS heuristics, :172-181).
 * ignore_eos != 0 suppresses EOS (fixed-length benchmarking); max_frames caps every utterance
 * (the server's --max_tokens).  Runs the prefill; 0 ok / <0 error. */
int q3e_start(void* e, int B, const float* prefix, const int32_t* n_rows, const int32_t* n_text,
              int ignore_eos, int max_frames);

/* Generate up to n_frames more frames for the whole batch (returns early once every utterance
 * has finished; never steps past the max_frames given to q3e_start: 0 when none is left).  Returns the
 * number of frame steps executed, <0 on error. */
int q3e_run(void* e, int n_frames);

/* GPU time of the last q3e_run / q3e_start in milliseconds (HIP events on the engine's stream). */
float q3e_last_run_ms(void* e);
float q3e_last_prefill_ms(void* e);

/* Codes so far: out[f][b][16] for f < returned frame count (<= max_out_frames); rows of finished
 * utterances hold -1 in column 0.  n_frames_per_utt[b] = frames utterance b really emitted. */
int q3e_get_codes(void* e, int32_t* out, int max_out_frames, int32_t* n_frames_per_utt);

/* Continuous batching (no counterpart in the reference, whose servers take one request at a time: SURVEY.md 8f2).
 * q3e_get_done: done[b] = 1 once utterance b has ended (EOS, or its frame budget); frames[b] (may be NULL) = frames it
 * has emitted.  q3e_refill: put n NEW utterances into the given slots of the running batch (finished or not) without
 * touching the others: the slots' counters, token history and codes column restart, their prefixes are prefilled
 * (prefix / n_rows / n_text as for q3e_start, in the order of `slots`), and the next q3e_run continues every slot --
 * the new ones with the full max_frames budget of q3e_start.  Fetch a finished utterance's codes (q3e_get_codes) BEFORE
 * refilling its slot.  After a refill q3e_get_codes returns rows up to the longest-running slot's frame count; each
 * column b holds utterance b's frames from ITS start (row f = its f-th frame).  0 ok / <0 error. */
int q3e_get_done(void* e, int32_t* done /*[B]*/, int32_t* frames /*[B] or NULL*/);
int q3e_refill(void* e, int n, const int32_t* slots, const float* prefix, const int32_t* n_rows, const int32_t* n_text);

/* Talker hidden state of every utterance after the last executed step ([B][hidden]). */
int q3e_get_hidden(void* e, float* out);

/* Algorithmic weight bytes one frame step streams (talker stack + head + 16 CP passes + 15 heads). */
double q3e_step_weight_bytes(void* e);

#ifdef __cplusplus
}
#endif
#endif /* QWEN3TTS_ENGINE_H */
