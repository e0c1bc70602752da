#!/usr/bin/env python3
"""Talker server -- MI355X mirror of the reference's dual_npu/llamacpp_talker_server.py.

Same socket protocol, CLI flags, sampling heuristics and KV-prefix-cache behaviour; the talker runs
through the HIP build of the wrapper_* ABI (llama_cpp_bindings.LlamaCppModel) instead of llama.cpp,
and the codec head is a device GEMV.  Tables (text embedding, projection MLP, codec embedding) come
from the reference's own embeddings/ directory (--embeddings) or from the Q3TTSW1 container of --model.

    python -m qwen3_tts_axera_russian_amd.llamacpp_talker_server --model qwen3tts.q3w \
        [--tokenizer /path/to/local/tokenizer_dir] --socket /tmp/qwen3_talker.sock

Protocol: see protocol.py.  Extension: the request JSON may carry "token_ids" (text already
tokenised), which makes the server independent of a tokenizer installation.
"""
from __future__ import annotations

import argparse
import hashlib
import os
import signal
import socket
import time

import numpy as np

from . import protocol as P
from .frontend import TalkerSampler, load_text_front_end
from .llama_cpp_bindings import LlamaCppModel



class Qwen3TTSTalkerServer:
    def __init__(self, model_path, embeddings_dir=None, socket_path="/tmp/qwen3_talker.sock", temperature=0.8,
                 top_k=50, max_tokens=200, n_threads=4, kv_cache_dir="/tmp", tokenizer=None, n_ctx=512,
                 install_signal_handlers=True, text_on_device=False):
        self.socket_path, self.max_tokens, self.kv_cache_dir = socket_path, max_tokens, kv_cache_dir
        print("Loading embeddings...")
        # the reference's own embeddings/ directory (llamacpp_talker_server.py:79-93) or the container's text.* tensors
        self.cfg, self.front = load_text_front_end(model_path, embeddings_dir)
        self.codec_embedding = self.front.codec
        if text_on_device:
            # the text table (fp16) and the projection MLP live on the GPU (include/qwen3tts_text.h); the host copy
            # above only supplied the configuration and the codec table the client-side code reads
            from .frontend import DeviceTextFrontEnd
            self.front = DeviceTextFrontEnd(self.cfg, None if embeddings_dir else model_path, embeddings_dir)
        self.tts_pad_embed = self.front.tts_pad_embed
        self.sampler = TalkerSampler(self.cfg.codec_eos, self.cfg.cp_vocab, temperature, top_k)
        self.tokenizer = None
        if tokenizer:
            # a LOCAL directory (vocab.json / merges.txt of the model snapshot), never a hub name
            from .tokenizer import ByteLevelBPE
            self.tokenizer = ByteLevelBPE.from_dir(tokenizer)
        print(f"Loading talker: {model_path}")
        self.llm = LlamaCppModel(model_path, n_ctx=n_ctx, n_threads=n_threads)
        self._running = True
        if install_signal_handlers:
            signal.signal(signal.SIGINT, self._signal_handler)
            signal.signal(signal.SIGTERM, self._signal_handler)

    def _signal_handler(self, signum, frame):
        print(f"\nReceived signal {signum}, shutting down...")
        self._running = False

    # same names as the reference's methods
    def _embed_text(self, token_ids):
        return self.front.embed_text(token_ids)

    def _build_prefix(self, text_token_ids, language="russian"):
        return self.front.build_prefix(text_token_ids, language)

    def _sample_token(self, hidden_state, past_tokens=None, n_text_tokens=0):
        logits = self.llm.codec_head(hidden_state)[0]       # [3072] on the device
        return self.sampler.sample(logits, past_tokens, n_text_tokens)

    def _prefix_hash(self, prefix):
        return hashlib.md5(prefix.tobytes()).hexdigest()[:16]

    def _tokenize(self, msg):
        if msg.get("token_ids") is not None:
            return [int(x) for x in msg["token_ids"]]
        if self.tokenizer is None:
            raise RuntimeError("no tokenizer configured (--tokenizer DIR) and the request has no token_ids")
        return self.tokenizer.encode(msg.get("text", ""), add_special_tokens=False)

    def _generate_streaming(self, conn, text_token_ids, language="russian"):
        prefix = self._build_prefix(text_token_ids, language)
        h = self._prefix_hash(prefix)
        kv_path = os.path.join(self.kv_cache_dir, f"qwen3_kv_{h}.bin")
        hid_path = os.path.join(self.kv_cache_dir, f"qwen3_hidden_{h}.npy")
        print(f"  Text tokens: {len(text_token_ids)}  Prefix: {prefix.shape[0]} rows, hash={h}")
        t0 = time.time()
        hidden = None
        if os.path.exists(kv_path) and os.path.exists(hid_path):
            try:
                if self.llm.state_load(kv_path) == 0:
                    hidden = np.load(hid_path)
                    self.llm.pos = prefix.shape[0]
                    print(f"  KV CACHE HIT: {time.time() - t0:.3f}s")
            except Exception as e:  # same tolerance as the reference: fall back to a prefill
                print(f"  KV cache error: {e}")
                hidden = None
        if hidden is None:
            hidden = self.llm.get_hidden(prefix, keep_history=0)
            dt = time.time() - t0
            print(f"  Prefill: {dt:.3f}s ({prefix.shape[0] / max(dt, 1e-9):.0f} tok/s)")
            try:
                self.llm.state_save(kv_path)
                np.save(hid_path, hidden)
            except Exception as e:
                print(f"  KV cache save error: {e}")
        past, n_text, out_tokens = [], len(text_token_ids), 0
        t_gen = time.time()
        for i in range(self.max_tokens):
            code_0 = self._sample_token(hidden, past_tokens=past, n_text_tokens=n_text)
            if code_0 == self.cfg.codec_eos or code_0 >= self.cfg.cp_vocab:
                print(f"  EOS at step {i} (token={code_0})")
                break
            try:
                conn.sendall(P.pack_talker_frame(code_0, hidden))
            except (BrokenPipeError, ConnectionResetError):
                print("  Client disconnected")
                return out_tokens
            out_tokens += 1
            past.append(code_0)
            fb = P.recv_exact(conn, P.HIDDEN_SIZE * 4)
            if len(fb) < P.HIDDEN_SIZE * 4:
                print("  Client closed connection")
                return out_tokens
            hidden = self.llm.get_hidden(np.frombuffer(fb, dtype=np.float32).reshape(1, P.HIDDEN_SIZE), keep_history=1)
        try:
            conn.sendall(P.pack_sentinel(P.SENTINEL_DONE))
        except (BrokenPipeError, ConnectionResetError):
            pass
        if out_tokens:
            dt = time.time() - t_gen
            print(f"  Generated {out_tokens} tokens in {dt:.2f}s ({out_tokens / max(dt, 1e-9):.1f} tok/s)")
        return out_tokens

    def serve(self):
        if os.path.exists(self.socket_path):
            os.unlink(self.socket_path)
        sock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        sock.bind(self.socket_path)
        sock.listen(1)
        sock.settimeout(1.0)
        os.chmod(self.socket_path, 0o666)
        print(f"\nQwen3-TTS Talker Server listening on {self.socket_path}")
        n = 0
        while self._running:
            try:
                conn, _ = sock.accept()
            except socket.timeout:
                continue
            n += 1
            print(f"--- Request #{n} ---")
            try:
                msg = P.read_talker_request(conn)
                if msg is None:
                    continue
                self._generate_streaming(conn, self._tokenize(msg), msg.get("language", "russian"))
            except Exception as e:
                print(f"  Error: {e}")
                try:
                    conn.sendall(P.pack_sentinel(P.SENTINEL_ERROR))
                except OSError:
                    pass
            finally:
                conn.close()
        sock.close()
        if os.path.exists(self.socket_path):
            os.unlink(self.socket_path)
        print("Server stopped.")
        self.llm.destroy()


def main():
    ap = argparse.ArgumentParser(description="Qwen3-TTS Talker Server (MI355X / HIP)")
    ap.add_argument("--model", required=True, help="Q3TTSW1 container, or the re-keyed talker model.safetensors / its directory")
    ap.add_argument("--embeddings", default=None, help="the reference's embeddings/ directory of .npy tables (default: the text.* tensors of --model)")
    ap.add_argument("--socket", default="/tmp/qwen3_talker.sock")
    ap.add_argument("--temperature", type=float, default=0.8)
    ap.add_argument("--top_k", type=int, default=50)
    ap.add_argument("--max_tokens", type=int, default=200)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--kv_cache_dir", default="/tmp")
    ap.add_argument("--tokenizer", default=None, help="LOCAL tokenizer directory")
    ap.add_argument("--n_ctx", type=int, default=512)
    ap.add_argument("--text_on_device", action="store_true",
                    help="text table (fp16) + projection MLP + prefix assembly on the GPU instead of host numpy")
    a = ap.parse_args()
    Qwen3TTSTalkerServer(a.model, a.embeddings, a.socket, a.temperature, a.top_k, a.max_tokens, a.threads,
                         a.kv_cache_dir, a.tokenizer, a.n_ctx, text_on_device=a.text_on_device).serve()


if __name__ == "__main__":
    main()
