"""Host-side front-end arithmetic of the talker server: text projection, dual-stream prefix, sampling.

Mirrors (same argument meaning and results) Qwen3TTSTalkerServer._embed_text / _build_prefix /
_sample_token of the reference (dual_npu/llamacpp_talker_server.py:115-206).  Runs once per utterance /
once per frame on small vectors; the GEMMs behind it (talker, codec head) are on the GPU.
"""
from __future__ import annotations

import numpy as np


class TextFrontEnd:
    """text ids -> prefix embedding rows [n_text + 9, hidden] (llamacpp_talker_server.py:115-161)."""

    def __init__(self, cfg, text_embedding, fc1_w, fc1_b, fc2_w, fc2_b, codec_embedding):
        self.cfg = cfg
        self.table = text_embedding
        self.fc1_w, self.fc1_b, self.fc2_w, self.fc2_b = fc1_w, fc1_b, fc2_w, fc2_b
        self.codec = codec_embedding
        sp = self.embed_text([cfg.tts_pad, cfg.tts_bos, cfg.tts_eos])
        self.tts_pad_embed, self.tts_bos_embed, self.tts_eos_embed = sp[0], sp[1], sp[2]

    def embed_text(self, token_ids):
        rows = self.table[np.asarray(token_ids, dtype=np.int64)].astype(np.float32, copy=False)
        h = rows @ self.fc1_w.T + self.fc1_b
        h = h * (1.0 / (1.0 + np.exp(-h)))          # SiLU
        return (h @ self.fc2_w.T + self.fc2_b).astype(np.float32)

    def build_prefix(self, text_token_ids, language="russian"):
        """Text stream + codec stream, summed per position.  `language` is accepted and unused, as in
        the reference (parsed, passed, never read)."""
        c, cod = self.cfg, self.codec
        n = len(text_token_ids)
        out = np.empty((n + 9, cod.shape[1]), np.float32)
        out[0:3] = self.embed_text([c.im_start, c.assistant, c.newline])            # role rows: text only
        out[3:6] = self.tts_pad_embed + cod[[c.codec_nothink, c.codec_think_bos, c.codec_think_eos]]
        out[6] = self.tts_bos_embed + cod[c.codec_pad]
        if n:
            out[7:7 + n] = self.embed_text(text_token_ids) + cod[c.codec_pad]
        out[7 + n] = self.tts_eos_embed + cod[c.codec_pad]
        out[8 + n] = self.tts_pad_embed + cod[c.codec_bos]
        return out


class DeviceTextFrontEnd:
    """The same two calls on the GPU (include/qwen3tts_text.h): fp16 text table resident in HBM, fc1 / SiLU / fc2 and
    the prefix assembly as device kernels.  Same attributes as TextFrontEnd (tts_pad_embed ...), so the talker server
    takes either (--text_on_device)."""

    def __init__(self, cfg, weights_path=None, embeddings_dir=None, max_tokens=2048):
        from . import hiplib
        self._hl, self._lib, self.cfg = hiplib, hiplib.load(), cfg
        self.h = self._lib.tfe_load(str(weights_path).encode() if weights_path else None,
                                    str(embeddings_dir).encode() if embeddings_dir else None, int(max_tokens))
        if not self.h:
            raise RuntimeError(f"tfe_load failed: {weights_path or embeddings_dir}")
        self.hidden = self._lib.tfe_hidden_size(self.h)
        c = cfg
        self._special = np.array([c.im_start, c.assistant, c.newline, c.tts_pad, c.tts_bos, c.tts_eos, c.codec_pad, c.codec_bos,
                                  c.codec_nothink, c.codec_think_bos, c.codec_think_eos, 0], np.int32)
        sp = self.embed_text([c.tts_pad, c.tts_bos, c.tts_eos])
        self.tts_pad_embed, self.tts_bos_embed, self.tts_eos_embed = sp[0], sp[1], sp[2]

    def embed_text(self, token_ids):
        ids = np.ascontiguousarray(token_ids, dtype=np.int32).reshape(-1)
        out = np.empty((len(ids), self.hidden), np.float32)
        if self._lib.tfe_embed_text(self.h, self._hl.iptr(ids), len(ids), self._hl.fptr(out)) != 0:
            raise RuntimeError("tfe_embed_text failed")
        return out

    def build_prefix(self, text_token_ids, language="russian"):
        ids = np.ascontiguousarray(text_token_ids, dtype=np.int32).reshape(-1)
        out = np.empty((len(ids) + 9, self.hidden), np.float32)
        n = self._lib.tfe_build_prefix(self.h, self._hl.iptr(ids) if len(ids) else None, len(ids),
                                       self._hl.iptr(self._special), self._hl.fptr(out))
        if n != len(ids) + 9:
            raise RuntimeError(f"tfe_build_prefix failed: {n}")
        return out

    def destroy(self):
        if self.h:
            self._lib.tfe_free(self.h)
            self.h = None


class TalkerSampler:
    """Codec-token sampling with the reference's heuristics (llamacpp_talker_server.py:163-206): mask of
    ids 2048..2149 and >= 2151, adaptive EOS boost, repetition penalty over the set of the last 30
    tokens, top-k / temperature / top-p 0.95.  temperature <= 1e-6 = arg-max (the reference's limit)."""

    def __init__(self, eos=2150, audio_vocab=2048, temperature=0.8, top_k=50, top_p=0.95, rng=None):
        self.eos, self.audio_vocab = eos, audio_vocab
        self.temperature, self.top_k, self.top_p = temperature, top_k, top_p
        self.rng = rng if rng is not None else np.random

    def process(self, logits, past_tokens, n_text_tokens):
        l = np.array(logits, dtype=np.float32, copy=True)
        l[self.audio_vocab:self.eos] = -1e10
        l[self.eos + 1:] = -1e10
        forced = None
        if past_tokens is not None and n_text_tokens > 0:
            progress = len(past_tokens) / (n_text_tokens * 3)
            if progress > 0.8:
                l[self.eos] += min((progress - 0.8) / 0.7, 1.0) * 15.0
            if progress > 2.0:
                forced = self.eos
        if past_tokens:
            for t in set(past_tokens[-30:]):
                if 0 <= t < len(l):
                    l[t] = l[t] / 1.2 if l[t] > 0 else l[t] * 1.2
        return l, forced

    def sample(self, logits, past_tokens=None, n_text_tokens=0):
        l, forced = self.process(logits, past_tokens, n_text_tokens)
        if forced is not None:
            return int(forced)
        if self.temperature <= 1e-6:
            return int(np.argmax(l))
        top = np.argsort(l)[-self.top_k:]
        z = l[top] / max(self.temperature, 1e-6)
        p = np.exp(z - z.max())
        p /= p.sum()
        order = np.argsort(-p)
        keep = order[:np.searchsorted(np.cumsum(p[order]), self.top_p) + 1]
        pk = p[keep] / p[keep].sum()
        return int(top[keep[self.rng.choice(len(keep), p=pk)]])


def feedback_embedding(code_0, codes_1_15, codec_embedding, cp_codec_embeddings, tts_pad_embed):
    """Next talker input (tts_client.py:199-208): talker table row of code_0, plus CP table g row of
    code g+1 for g = 0..14 in that order, plus the tts_pad embedding."""
    buf = codec_embedding[code_0].astype(np.float32, copy=True)
    for g, tok in enumerate(codes_1_15):
        buf += cp_codec_embeddings[g][tok]
    if tts_pad_embed is not None:
        buf += tts_pad_embed
    return buf


def load_text_front_end(model_path=None, embeddings_dir=None, cfg=None):
    """-> (cfg, TextFrontEnd).  Tables from the reference's own `embeddings/` directory when given
    (text_embedding.npy, text_projection_linear_fc{1,2}_{weight,bias}.npy, codec_embedding.npy:
    scripts/extract_embeddings.py:47-66, loaded as dual_npu/llamacpp_talker_server.py:79-93 does), else
    from the `text.*` / `talker.codec_embedding` tensors of a Q3TTSW1 container."""
    import os

    import numpy as np

    from .weights import ModelConfig, read_pack
    if embeddings_dir:
        e = lambda n: np.load(os.path.join(embeddings_dir, n), mmap_mode="r")
        f32 = lambda n: np.asarray(e(n), dtype=np.float32)
        cfg = cfg or ModelConfig()
        table = e("text_embedding.npy")             # 1.2 GB in the real model: stays memory-mapped, rows are gathered
        return cfg, TextFrontEnd(cfg, table, f32("text_projection_linear_fc1_weight.npy"),
                                 f32("text_projection_linear_fc1_bias.npy"), f32("text_projection_linear_fc2_weight.npy"),
                                 f32("text_projection_linear_fc2_bias.npy"), f32("codec_embedding.npy"))
    meta, t = read_pack(model_path)
    cfg = cfg or ModelConfig.from_meta(meta)
    g = lambda n: np.asarray(t[n], dtype=np.float32)
    return cfg, TextFrontEnd(cfg, t["text.embedding"], g("text.fc1.weight"), g("text.fc1.bias"), g("text.fc2.weight"),
                             g("text.fc2.bias"), g("talker.codec_embedding"))
