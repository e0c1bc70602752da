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

def _talker_library():
    """The wrapper_* ABI lives in lib/llama_wrapper.so -- the file name the reference's loader looks for beside
    itself (dual_npu/llama_cpp_bindings.py:18-21) -- which is a copy of lib/libqwen3tts.so."""
    for name in ("llama_wrapper.so", "libqwen3tts.so"):
        path = os.path.join(LIB_DIR, name)
        if os.path.exists(path):
            return hiplib.load(path)
    raise RuntimeError(f"no talker library (llama_wrapper.so / libqwen3tts.so) under {LIB_DIR}: "
                       "run `python -m qwen3_tts_axera_russian_amd.build`; there is no CPU path")


class LlamaCppModel:
    """Talker in embeddings mode behind the wrapper_* ABI (include/qwen3tts_talker.h).

    Public surface = the reference class of the same name (dual_npu/llama_cpp_bindings.py:84-179), because
    llamacpp_talker_server.py drives it: get_hidden(emb, keep_history), clear_kv(), state_get_size(),
    state_save(path), state_load(path), the read/write `pos` property (the server sets it after a KV-cache hit,
    llamacpp_talker_server.py:231) and destroy().  Failures raise RuntimeError as there.  The position counter is
    host bookkeeping only: the library honours whatever pos_start it is handed."""

    #            method name      C symbol                     what a non-zero return means
    _STATE_IO = {"state_save": ("wrapper_state_save_file", "saved"),
                 "state_load": ("wrapper_state_load_file", "restored")}

    def __init__(self, model_path, n_ctx=512, n_threads=4):
        self._abi = _talker_library()
        self.model = self.ctx = None
        self._abi.wrapper_backend_init()
        self.model = self._abi.wrapper_load_model(os.fsencode(str(model_path)), 0)
        if not self.model:
            raise RuntimeError(f"talker weights could not be loaded from {model_path} (see stderr of the library)")
        self.n_embd = int(self._abi.wrapper_model_n_embd(self.model))
        if self.n_embd <= 0:
            raise RuntimeError(f"library reports hidden size {self.n_embd}")
        # n_batch = n_ctx and embeddings = 1, the values the reference passes (llama_cpp_bindings.py:100-104)
        self.ctx = self._abi.wrapper_create_context(self.model, int(n_ctx), int(n_ctx), int(n_threads), 1)
        if not self.ctx:
            self.destroy()
            raise RuntimeError(f"talker context of {n_ctx} positions could not be created")
        self.n_ctx, self._pos = int(n_ctx), 0
        self._out = np.empty(self.n_embd, dtype=np.float32)
        print(f"LlamaCppModel (HIP, gfx950): hidden {self.n_embd}, context {self.n_ctx}")

    # -- decode ----------------------------------------------------------------------------------------------
    def get_hidden(self, embeddings, keep_history=0):
        """embeddings f32 [n, hidden] (or [hidden]) -> post-norm hidden [hidden] of the last row.
        keep_history 0: the cache is cleared and the rows start at position 0 (prefill); 1: they are appended."""
        rows = np.ascontiguousarray(embeddings, dtype=np.float32)
        rows = rows.reshape(1, -1) if rows.ndim == 1 else rows
        if rows.ndim != 2 or rows.shape[1] != self.n_embd:
            raise AssertionError(f"embedding rows must be [n, {self.n_embd}], got {rows.shape}")
        if not keep_history:
            self.clear_kv()
        rc = self._abi.wrapper_decode_embd(self.ctx, hiplib.fptr(rows), rows.shape[0], self.n_embd, self._pos,
                                           hiplib.fptr(self._out))
        if rc != 0:
            raise RuntimeError(f"wrapper_decode_embd({rows.shape[0]} rows at position {self._pos}) returned {rc}")
        self._pos += rows.shape[0]
        return self._out.copy()

    def codec_head(self, hidden):
        """Extension (not in the reference class): codec_head GEMV on the device -> logits [rows, talker_vocab]."""
        hidden = np.ascontiguousarray(hidden, dtype=np.float32).reshape(-1, self.n_embd)
        out = np.empty((hidden.shape[0], 3072), np.float32)
        v = self._abi.wrapper_codec_head(self.ctx, hiplib.fptr(hidden), hidden.shape[0], hiplib.fptr(out))
        if v <= 0:
            raise RuntimeError(f"wrapper_codec_head returned {v}")
        return out[:, :v] if v != 3072 else out

    def clear_kv(self):
        self._abi.wrapper_kv_clear(self.ctx)
        self._pos = 0

    # -- KV state files (the server's prefix cache, llamacpp_talker_server.py:208-246) -------------------------
    def state_get_size(self):
        return int(self._abi.wrapper_state_get_size(self.ctx))

    def _state_io(self, which, path):
        symbol, verb = self._STATE_IO[which]
        rc = getattr(self._abi, symbol)(self.ctx, os.fsencode(str(path)))
        if rc == 0:
            print(f"  talker KV state {verb}: {path}" + (f" ({self._pos} positions)" if which == "state_save" else ""))
        return rc

    def state_save(self, path):
        return self._state_io("state_save", path)

    def state_load(self, path):
        return self._state_io("state_load", path)

    pos = property(lambda self: self._pos, lambda self, value: setattr(self, "_pos", int(value)),
                   doc="next position handed to wrapper_decode_embd (caller's bookkeeping, as in the reference)")

    def destroy(self):
        abi = self._abi
        if self.ctx:
            abi.wrapper_free_context(self.ctx)
        if self.model:
            abi.wrapper_free_model(self.model)
        self.ctx = self.model = None
        abi.wrapper_backend_free()


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
