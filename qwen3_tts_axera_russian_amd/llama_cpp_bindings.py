"""ctypes bindings of the talker library -- host-side mirror of the reference's
dual_npu/llama_cpp_bindings.py (same class, method names, argument meaning and errors), bound
to the HIP build of the wrapper_* ABI (include/qwen3tts_talker.h) instead of llama.cpp.

`model_path` is a Q3TTSW1 weight container (weights.py), not a GGUF.  If the shared library is
missing or no GPU is usable this raises RuntimeError -- there is no CPU path.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import LIB_DIR
from . import hiplib

# reference: llama_cpp_bindings.py:18-21 looks for llama_wrapper.so beside itself
_WRAPPER_PATHS = [
    os.path.join(LIB_DIR, "llama_wrapper.so"),
    os.path.join(LIB_DIR, "libqwen3tts.so"),
]


def _load_wrapper():
    for path in _WRAPPER_PATHS:
        if os.path.exists(path):
            return hiplib.load(path)
    raise RuntimeError(f"llama_wrapper.so not found in {_WRAPPER_PATHS}")


_lib = None


def _get_lib():
    global _lib
    if _lib is None:
        _lib = _load_wrapper()
    return _lib


class LlamaCppModel:
    """Talker in embedding mode: feeds codec/text embeddings, returns hidden states
    (reference: llama_cpp_bindings.py:84-179)."""

    def __init__(self, model_path, n_ctx=512, n_threads=4):
        lib = _get_lib()
        self._lib = lib
        lib.wrapper_backend_init()
        self.model = lib.wrapper_load_model(str(model_path).encode(), 0)
        if not self.model:
            raise RuntimeError(f"Failed to load model: {model_path}")
        self.n_embd = lib.wrapper_model_n_embd(self.model)
        assert self.n_embd > 0, f"n_embd={self.n_embd}"
        self.ctx = lib.wrapper_create_context(self.model, n_ctx, n_ctx, n_threads, 1)  # embeddings=1
        if not self.ctx:
            raise RuntimeError("Failed to create llama context")
        self.n_ctx = n_ctx
        self._pos = 0
        self._hidden_buf = np.zeros(self.n_embd, dtype=np.float32)
        print(f"LlamaCppModel ready: n_embd={self.n_embd}, n_ctx={n_ctx}, backend=HIP/gfx950")

    def get_hidden(self, embeddings, keep_history=0):
        """[n_tokens, n_embd] or [n_embd] float32 -> [n_embd] hidden of the last token.
        keep_history: 0 = clear KV (prefill), 1 = append (decode step)."""
        if keep_history == 0:
            self._lib.wrapper_kv_clear(self.ctx)
            self._pos = 0
        embeddings = np.ascontiguousarray(embeddings, dtype=np.float32)
        if embeddings.ndim == 1:
            embeddings = embeddings.reshape(1, -1)
        n_tokens = embeddings.shape[0]
        assert embeddings.shape[1] == self.n_embd, f"Dim mismatch: {embeddings.shape[1]} vs {self.n_embd}"
        ret = self._lib.wrapper_decode_embd(self.ctx, hiplib.fptr(embeddings), n_tokens, self.n_embd, self._pos,
                                            hiplib.fptr(self._hidden_buf))
        if ret != 0:
            raise RuntimeError(f"wrapper_decode_embd failed: {ret}")
        self._pos += n_tokens
        return self._hidden_buf.copy()

    def codec_head(self, hidden):
        """Extension: codec_head GEMV on the device -> logits[talker_vocab]."""
        hidden = np.ascontiguousarray(hidden, dtype=np.float32).reshape(-1, self.n_embd)
        out = np.empty((hidden.shape[0], 3072), np.float32)
        v = self._lib.wrapper_codec_head(self.ctx, hiplib.fptr(hidden), hidden.shape[0], hiplib.fptr(out))
        if v <= 0:
            raise RuntimeError(f"wrapper_codec_head failed: {v}")
        return out[:, :v] if v != 3072 else out

    def clear_kv(self):
        self._lib.wrapper_kv_clear(self.ctx)
        self._pos = 0

    def state_get_size(self):
        return self._lib.wrapper_state_get_size(self.ctx)

    def state_save(self, path):
        ret = self._lib.wrapper_state_save_file(self.ctx, str(path).encode())
        if ret == 0:
            print(f"  KV state saved: {path} (pos={self._pos})")
        return ret

    def state_load(self, path):
        ret = self._lib.wrapper_state_load_file(self.ctx, str(path).encode())
        if ret == 0:
            print(f"  KV state loaded: {path}")
        return ret

    @property
    def pos(self):
        return self._pos

    @pos.setter
    def pos(self, value):
        self._pos = value

    def destroy(self):
        if self.ctx:
            self._lib.wrapper_free_context(self.ctx)
            self.ctx = None
        if self.model:
            self._lib.wrapper_free_model(self.model)
            self.model = None
        self._lib.wrapper_backend_free()


class CodePredictor:
    """ctypes front of include/qwen3tts_cp.h (stands where the reference calls onnxruntime:
    code_predictor_server.py:53-62,77-85)."""

    def __init__(self, weights_path, embeddings_dir=None, max_batch=1):
        self._lib = hiplib.load()
        self.h = self._lib.cp_load(str(weights_path).encode(),
                                   str(embeddings_dir).encode() if embeddings_dir else None, max_batch)
        if not self.h:
            raise RuntimeError(f"cp_load failed: {weights_path}")
        self.hidden = self._lib.cp_hidden_size(self.h)
        self.max_batch = max_batch

    def predict(self, hidden_state, code_0, temperature=0.0, top_k=50, seed=0):
        hs = np.ascontiguousarray(hidden_state, dtype=np.float32).reshape(-1)[:self.hidden]
        out = np.zeros(15, np.int32)
        rc = self._lib.cp_predict(self.h, hiplib.fptr(hs), int(code_0), float(temperature), int(top_k), int(seed),
                                  hiplib.iptr(out))
        if rc != 0:
            raise RuntimeError(f"cp_predict failed: {rc}")
        return [int(x) for x in out]

    def predict_batch(self, hidden, code_0, temperature=0.0, top_k=50, seed=0):
        hs = np.ascontiguousarray(hidden, dtype=np.float32).reshape(-1, self.hidden)
        c0 = np.ascontiguousarray(code_0, dtype=np.int32)
        out = np.zeros((hs.shape[0], 15), np.int32)
        rc = self._lib.cp_predict_batch(self.h, hiplib.fptr(hs), hiplib.iptr(c0), hs.shape[0], float(temperature),
                                        int(top_k), int(seed), hiplib.iptr(out))
        if rc != 0:
            raise RuntimeError(f"cp_predict_batch failed: {rc}")
        return out

    def step(self, embed, position):
        e = np.ascontiguousarray(embed, dtype=np.float32).reshape(-1)[:self.hidden]
        out = np.empty(self.hidden, np.float32)
        rc = self._lib.cp_step(self.h, hiplib.fptr(e), int(position), hiplib.fptr(out))
        if rc != 0:
            raise RuntimeError(f"cp_step failed: {rc}")
        return out

    def lm_head(self, group, hidden):
        hs = np.ascontiguousarray(hidden, dtype=np.float32).reshape(-1)[:self.hidden]
        out = np.empty(2048, np.float32)
        v = self._lib.cp_lm_head(self.h, int(group), hiplib.fptr(hs), hiplib.fptr(out))
        if v <= 0:
            raise RuntimeError(f"cp_lm_head failed: {v}")
        return out[:v]

    def destroy(self):
        if self.h:
            self._lib.cp_free(self.h)
            self.h = None
