#!/usr/bin/env python3
"""Vocoder server -- MI355X mirror of dual_npu/vocoder_server.py: same protocol (i32 n + i64[n*16] in,
i32 n_samples + i16 out), 64-frame chunks, overlap-16 linear crossfade with the reference's chunk
walk (including its tail-chunk quirk) and its int16 rule, all behind voc_synthesize
(include/qwen3tts_voc.h).

    python -m qwen3_tts_axera_russian_amd.vocoder_server --model qwen3tts_voc.q3w
"""
from __future__ import annotations

import argparse
import os
import signal
import socket
import time

import numpy as np

from . import hiplib
from . import protocol as P

SAMPLE_RATE = 24000
SAMPLES_PER_TOKEN = 1920


class VocoderServer:
    def __init__(self, model_path, socket_path="/tmp/qwen3_voc.sock", max_tokens=64, install_signal_handlers=True, max_batch=1):
        self.socket_path = socket_path
        self._lib = hiplib.load()
        self.h = self._lib.voc_load(str(model_path).encode(), max_tokens, max_batch)
        if not self.h:
            raise RuntimeError(f"Failed to load vocoder: {model_path}")
        self.max_tokens = self._lib.voc_chunk_tokens(self.h)
        print(f"Vocoder: HIP/gfx950 fp32, max_tokens={self.max_tokens}")
        self._running = True
        if install_signal_handlers:
            signal.signal(signal.SIGINT, self._signal_handler)
            signal.signal(signal.SIGTERM, self._signal_handler)

    def _signal_handler(self, signum, frame):
        self._running = False

    def _inference_chunk(self, padded):
        # the model's output tensor: [1, voc_chunk_samples] (<= max_tokens * SAMPLES_PER_TOKEN: the decoder family's
        # transposed convs trim, include/qwen3tts_voc.h); callers slice it numpy-style like the reference does
        out = np.empty((1, self._lib.voc_chunk_samples(self.h)), np.float32)
        c = np.ascontiguousarray(padded, np.int64)
        if self._lib.voc_decode(self.h, c.ctypes.data_as(hiplib.i64p), 1, hiplib.fptr(out)) != 0:
            raise RuntimeError("voc_decode failed")
        return out[0]

    def synthesize(self, codes_array):
        """codes [n,16] -> float32 audio (vocoder_server.py:73-121 semantics, chunk walk on the library side)."""
        c = np.ascontiguousarray(np.asarray(codes_array)[:, :16], np.int64)
        out = np.empty(self._lib.voc_synthesize_max_samples(self.h, c.shape[0]), np.float32)
        ns = np.zeros(1, np.int32)
        if self._lib.voc_synthesize_f32(self.h, c.ctypes.data_as(hiplib.i64p), c.shape[0], hiplib.fptr(out), hiplib.iptr(ns)):
            raise RuntimeError("voc_synthesize_f32 failed")
        return out[:ns[0]]

    def synthesize_int16(self, codes_array):
        c = np.ascontiguousarray(np.asarray(codes_array)[:, :16], np.int64)
        out = np.empty(self._lib.voc_synthesize_max_samples(self.h, c.shape[0]), np.int16)
        ns = np.zeros(1, np.int32)
        if self._lib.voc_synthesize(self.h, c.ctypes.data_as(hiplib.i64p), c.shape[0], out.ctypes.data_as(hiplib.i16p),
                                    hiplib.iptr(ns)):
            raise RuntimeError("voc_synthesize failed")
        return out[:ns[0]]

    def synthesize_batch(self, codes_list, int16=True):
        """U utterances in one call (include/qwen3tts_voc.h: voc_synthesize_batch): codes_list[u] is [n_u, 16]; the chunks
        of all utterances are decoded together and each utterance's overlap-crossfade walk (synthesize above, per
        utterance) is assembled on the device.  -> list of int16 (or float32) arrays."""
        n = np.array([len(c) for c in codes_list], np.int32)
        cat = np.ascontiguousarray(np.concatenate([np.asarray(c)[:, :16] for c in codes_list], axis=0), np.int64)
        cap = int(self._lib.voc_synthesize_batch_max_samples(self.h, hiplib.iptr(n), len(n)))
        out = np.empty(cap, np.int16 if int16 else np.float32)
        off = np.zeros(len(n) + 1, np.int64)
        fn = self._lib.voc_synthesize_batch if int16 else self._lib.voc_synthesize_batch_f32
        ptr = out.ctypes.data_as(hiplib.i16p) if int16 else hiplib.fptr(out)
        if fn(self.h, cat.ctypes.data_as(hiplib.i64p), hiplib.iptr(n), len(n), ptr, cap, off.ctypes.data_as(hiplib.i64p)) != 0:
            raise RuntimeError("voc_synthesize_batch failed")
        return [out[off[u]:off[u + 1]].copy() for u in range(len(n))]

    def serve(self):
        if os.path.exists(self.socket_path):
            os.unlink(self.socket_path)
        sock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        sock.bind(self.socket_path)
        sock.listen(1)
        sock.settimeout(1.0)
        os.chmod(self.socket_path, 0o666)
        print(f"\nVocoder Server listening on {self.socket_path}")
        while self._running:
            try:
                conn, _ = sock.accept()
            except socket.timeout:
                continue
            try:
                codes = P.read_voc_request(conn)
                if codes is not None:
                    t0 = time.time()
                    audio = self.synthesize_int16(codes)
                    print(f"  Vocoder: {len(codes)} tokens -> {len(audio)} samples ({time.time() - t0:.3f}s)")
                    conn.sendall(P.pack_voc_reply(audio))
            except Exception as e:
                print(f"  Vocoder Error: {e}")
            finally:
                conn.close()
        sock.close()
        if os.path.exists(self.socket_path):
            os.unlink(self.socket_path)
        self._lib.voc_free(self.h)
        print("Vocoder Server stopped.")


def main():
    ap = argparse.ArgumentParser(description="Qwen3-TTS Vocoder Server (MI355X / HIP)")
    ap.add_argument("--model", required=True, help="Q3TTSW1 container holding voc.*")
    ap.add_argument("--socket", default="/tmp/qwen3_voc.sock")
    a = ap.parse_args()
    VocoderServer(a.model, a.socket).serve()


if __name__ == "__main__":
    main()
