"""numpy restatement of the reference's front-end arithmetic -- TEST INFRASTRUCTURE ONLY.

Each function cites the reference lines it restates; tests/test_golden_frontend.py pins every
one of them against vectors produced by the reference's own Python (tests/golden/make_golden.py
imports /root/reference/dual_npu/*.py with stub back-ends and stores inputs + outputs).
numpy >= 2 scalar semantics (NEP 50) are assumed, as in the container the vectors were made in.
"""
from __future__ import annotations

import numpy as np

SAMPLES_PER_TOKEN = 1920  # vocoder_server.py:30


def embed_text(ids, text_embedding, fc1_w, fc1_b, fc2_w, fc2_b):
    """llamacpp_talker_server.py:115-119: table gather -> fc1 -> SiLU -> fc2, f32."""
    e = text_embedding[np.asarray(ids)]
    h = e @ fc1_w.T + fc1_b
    h = h * (1.0 / (1.0 + np.exp(-h)))
    return (h @ fc2_w.T + fc2_b).astype(np.float32)


def build_prefix(text_ids, cfg, codec_embedding, embed_fn):
    """llamacpp_talker_server.py:121-161: dual-stream prefix, n_text + 9 rows.

    rows: 3 role (text stream only) | 3 tts_pad + codec{nothink,think_bos,think_eos} |
    tts_bos + codec_pad | text(t)+codec_pad ..., tts_eos + codec_pad | tts_pad + codec_bos."""
    special = embed_fn(np.array([cfg.tts_pad, cfg.tts_bos, cfg.tts_eos]))
    pad, bos, eos = special[0], special[1], special[2]
    role = embed_fn(np.array([cfg.im_start, cfg.assistant, cfg.newline]))
    codec_prefix = codec_embedding[[cfg.codec_nothink, cfg.codec_think_bos, cfg.codec_think_eos]]
    dual_codec = np.stack([pad] * 3) + codec_prefix
    transition = (bos + codec_embedding[cfg.codec_pad])[np.newaxis]
    text = embed_fn(np.asarray(text_ids))
    text_plus_eos = np.concatenate([text, eos[np.newaxis]], axis=0)
    dual_text = text_plus_eos + np.tile(codec_embedding[cfg.codec_pad], (len(text_ids) + 1, 1))
    final = (pad + codec_embedding[cfg.codec_bos])[np.newaxis]
    return np.concatenate([role, dual_codec, transition, dual_text, final], axis=0).astype(np.float32)


def process_talker_logits(logits, past_tokens, n_text_tokens, eos=2150, audio_vocab=2048):
    """llamacpp_talker_server.py:167-189 -- mask, adaptive EOS boost, repetition penalty, in that
    order.  Returns (processed logits f32, forced) where forced is EOS when progress > 2."""
    logits = np.array(logits, dtype=np.float32, copy=True)
    logits[audio_vocab:eos] = -1e10
    if eos + 1 < len(logits):
        logits[eos + 1:] = -1e10
    forced = None
    if past_tokens is not None and n_text_tokens > 0:
        expected_len = n_text_tokens * 3
        progress = len(past_tokens) / expected_len if expected_len > 0 else 0
        if progress > 0.8:
            boost = min((progress - 0.8) / 0.7, 1.0) * 15.0
            logits[eos] += boost
        if progress > 2.0:
            forced = eos
    if past_tokens:
        for t in set(past_tokens[-30:]):
            if 0 <= t < len(logits):
                if logits[t] > 0:
                    logits[t] /= 1.2
                else:
                    logits[t] *= 1.2
    return logits, forced


def sample_talker(logits, past_tokens, n_text_tokens, temperature=0.0, top_k=50, rng=None, eos=2150):
    """llamacpp_talker_server.py:163-206.  temperature <= 1e-6 is the deterministic limit the
    reference reaches through max(T, 1e-6): the arg-max of the processed logits."""
    l, forced = process_talker_logits(logits, past_tokens, n_text_tokens, eos)
    if forced is not None:
        return int(forced)
    if temperature <= 1e-6:
        return int(np.argmax(l))
    rng = rng or np.random
    top_indices = np.argsort(l)[-top_k:]
    scaled = l[top_indices] / max(temperature, 1e-6)
    probs = np.exp(scaled - scaled.max())
    probs /= probs.sum()
    sorted_idx = np.argsort(-probs)
    cumsum = np.cumsum(probs[sorted_idx])
    cutoff = np.searchsorted(cumsum, 0.95) + 1
    keep = sorted_idx[:cutoff]
    pf = probs[keep]
    pf /= pf.sum()
    return int(top_indices[keep[rng.choice(len(keep), p=pf)]])


def sample_cp(logits, temperature=0.0, top_k=50, rng=None):
    """code_predictor_server.py:87-92; greedy limit = argmax."""
    if temperature <= 1e-6:
        return int(np.argmax(logits))
    rng = rng or np.random
    top = np.argpartition(logits, -top_k)[-top_k:]
    tl = logits[top]
    probs = np.exp((tl - tl.max()) / max(temperature, 1e-6))
    probs /= probs.sum()
    return int(top[rng.choice(len(top), p=probs)])


def feedback_embedding(code_0, codes_1_15, codec_embedding, cp_codec_embeddings, tts_pad_embed):
    """tts_client.py:199-208: copy talker row, += CP table g row (g = 0..14 in order), += tts_pad."""
    buf = np.zeros(codec_embedding.shape[1], dtype=np.float32)
    np.copyto(buf, codec_embedding[code_0])
    for gi, tok in enumerate(codes_1_15):
        buf += cp_codec_embeddings[gi][tok]
    if tts_pad_embed is not None:
        buf += tts_pad_embed
    return buf


def voc_synthesize(codes_array, chunk_fn, max_tokens=64):
    """vocoder_server.py:73-121, bug-compatible (loop runs while chunk_start < n_tokens, so a
    redundant short tail chunk is appended un-crossfaded when 1 <= n mod 48 <= 15)."""
    n_tokens = len(codes_array)
    if n_tokens <= max_tokens:
        padded = np.zeros((1, max_tokens, 16), dtype=np.int64)
        padded[0, :n_tokens, :] = codes_array[:, :16]
        return chunk_fn(padded)[:n_tokens * SAMPLES_PER_TOKEN]
    overlap = 16
    ov = overlap * SAMPLES_PER_TOKEN
    step = max_tokens - overlap
    result = np.array([], dtype=np.float32)
    start = 0
    while start < n_tokens:
        end = min(start + max_tokens, n_tokens)
        ln = end - start
        padded = np.zeros((1, max_tokens, 16), dtype=np.int64)
        padded[0, :ln, :] = codes_array[start:end, :16]
        chunk = chunk_fn(padded)[:ln * SAMPLES_PER_TOKEN]
        if start == 0:
            result = chunk
        elif len(result) >= ov and len(chunk) >= ov:
            fade_out = np.linspace(1.0, 0.0, ov, dtype=np.float32)
            fade_in = 1.0 - fade_out
            blended = result[-ov:] * fade_out + chunk[:ov] * fade_in
            result = np.concatenate([result[:-ov], blended, chunk[ov:]])
        else:
            result = np.concatenate([result, chunk])
        start += step
    return result


def to_int16(audio):
    """vocoder_server.py:175: scale, clip, truncate toward zero."""
    return np.clip(audio * 32767, -32768, 32767).astype(np.int16)


def cp_predict_loop(step_fn, hidden_state, code_0, talker_codec_embedding, cp_embeddings, lm_heads,
                    sample_fn=None, batch_prefill=False, H=1024):
    """code_predictor_server.py:94-140: the 16-position schedule around a decode-step function.

    step_fn(hidden[1,n,H], positions list) -> hidden[1,n,H] (post final norm); it owns the KV state.
    Position 0 = talker hidden, position 1 = TALKER table row of code_0 (:97-98), position g+1 (g>=1)
    = CP table g-1 row of the previous token (:134); head g after position g+1."""
    sample_fn = sample_fn or (lambda l: int(np.argmax(l)))
    code_0_embed = talker_codec_embedding[code_0]
    h0 = hidden_state.flatten()[:H].astype(np.float32)
    h1 = code_0_embed.flatten()[:H].astype(np.float32)
    if batch_prefill:
        hidden = step_fn(np.stack([h0, h1]).reshape(1, 2, H), [0, 1])
    else:
        step_fn(h0.reshape(1, 1, H), [0])
        hidden = step_fn(h1.reshape(1, 1, H), [1])
    tokens = []
    token = sample_fn(hidden[0, -1] @ lm_heads[0].T)
    tokens.append(token)
    for step in range(1, len(lm_heads)):
        embed = cp_embeddings[step - 1][token].reshape(1, 1, H)
        hidden = step_fn(embed, [step + 1])
        token = sample_fn(hidden[0, -1] @ lm_heads[step].T)
        tokens.append(token)
    return tokens
