#!/usr/bin/env python3
"""Code-predictor server -- MI355X mirror of dual_npu/code_predictor_server.py (and of its native twins
code_predictor_cpp/code_predictor_server.cpp, code_predictor_ggml/code_pred_server.cpp): same protocol
(recv 4096 + 4 bytes on a fresh connection, send 60 bytes, close), the 16-position loop runs as one
hipGraph behind cp_predict (include/qwen3tts_cp.h).

    python -m qwen3_tts_axera_russian_amd.code_predictor_server --model qwen3tts.q3w
"""
from __future__ import annotations

import argparse
import os
import signal
import socket
import time

from . import protocol as P
from .llama_cpp_bindings import CodePredictor


class CodePredictorServer:
    def __init__(self, model_dir, embeddings_dir=None, socket_path="/tmp/qwen3_cp.sock", temperature=0.1, top_k=50,
                 n_threads=1, batch_prefill=False, install_signal_handlers=True):
        self.socket_path, self.temperature, self.top_k = socket_path, temperature, top_k
        self.num_groups = P.NUM_CP_CODES
        self.cp = CodePredictor(model_dir, embeddings_dir, max_batch=1)
        self._running = True
        self._seed = 42      # the C++ reference seeds mt19937 with 42 (code_predictor_server.cpp:136)
        if install_signal_handlers:
            signal.signal(signal.SIGINT, self._signal_handler)
            signal.signal(signal.SIGTERM, self._signal_handler)
        # warm-up like the native servers (code_predictor_server.cpp:510-518): hidden = 0.1, code_0 = 100
        import numpy as np
        print("  warmup codes:", self.predict(np.full(P.HIDDEN_SIZE, 0.1, np.float32), 100))

    def _signal_handler(self, signum, frame):
        self._running = False

    def predict(self, hidden_state, code_0):
        """groups 1-15 from the talker hidden state and code_0 (code_predictor_server.py:94-140)."""
        self._seed += 1
        return self.cp.predict(hidden_state, code_0, self.temperature, self.top_k, self._seed)

    def serve(self):
        if os.path.exists(self.socket_path):
            os.unlink(self.socket_path)
        sock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        sock.bind(self.socket_path)
        sock.listen(1)
        sock.settimeout(1.0)
        os.chmod(self.socket_path, 0o666)
        print(f"\nCode Predictor Server listening on {self.socket_path}")
        while self._running:
            try:
                conn, _ = sock.accept()
            except socket.timeout:
                continue
            try:
                req = P.read_cp_request(conn)
                if req is not None:
                    t0 = time.time()
                    codes = self.predict(*req)
                    conn.sendall(P.pack_cp_reply(codes))
                    print(f"  predict: {(time.time() - t0) * 1e3:.2f}ms")
            except Exception as e:
                print(f"  CP Error: {e}")
            finally:
                conn.close()
        sock.close()
        if os.path.exists(self.socket_path):
            os.unlink(self.socket_path)
        print("Code Predictor Server stopped.")
        self.cp.destroy()


def main():
    ap = argparse.ArgumentParser(description="Qwen3-TTS Code Predictor Server (MI355X / HIP)")
    ap.add_argument("--model_dir", "--model", dest="model", required=True, help="the reference's model directory (code_predictor_weights.npz, with --embeddings_dir for codec_embedding.npy) or a Q3TTSW1 container")
    ap.add_argument("--embeddings_dir", default=None)
    ap.add_argument("--socket", default="/tmp/qwen3_cp.sock")
    ap.add_argument("--temperature", type=float, default=0.1)
    ap.add_argument("--top_k", type=int, default=50)
    ap.add_argument("--threads", type=int, default=3)
    ap.add_argument("--batch_prefill", action="store_true", help="accepted; positions 0/1 always run exactly")
    a = ap.parse_args()
    CodePredictorServer(a.model, a.embeddings_dir, a.socket, a.temperature, a.top_k, a.threads, a.batch_prefill).serve()


if __name__ == "__main__":
    main()
