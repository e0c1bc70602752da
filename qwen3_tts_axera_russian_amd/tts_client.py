#!/usr/bin/env python3
"""Client -- MI355X mirror of dual_npu/tts_client.py: drives the three servers over their sockets
(text -> talker -> code predictor -> feedback -> ... -> vocoder -> WAV, RTF print), same protocol and
CLI.  `--token_ids` (extension) sends pre-tokenised text.

    python -m qwen3_tts_axera_russian_amd.tts_client "Привет, как дела?" --weights qwen3tts.q3w
"""
from __future__ import annotations

import argparse
import os
import socket
import struct
import threading
import time
import wave as wavmod

import numpy as np

from . import protocol as P
from .frontend import TextFrontEnd, feedback_embedding
from .weights import ModelConfig, read_pack

SAMPLE_RATE = 24000
SAMPLES_PER_TOKEN = 1920
VOC_CHUNK_SIZE = 64


class Qwen3TTSClient:
    def __init__(self, talker_socket="/tmp/qwen3_talker.sock", cp_socket="/tmp/qwen3_cp.sock",
                 voc_socket="/tmp/qwen3_voc.sock", weights=None, embeddings_dir=None, cp_dir=None):
        """Tables of the feedback sum: from a Q3TTSW1 container (`weights`), or, like the reference's client
        (tts_client.py:39-76), from its embeddings/ directory (.npy) and code-predictor directory (.npz)."""
        self.talker_socket, self.cp_socket, self.voc_socket = talker_socket, cp_socket, voc_socket
        self.codec_embedding = self.cp_codec_embeddings = self.tts_pad_embed = None
        if embeddings_dir:
            from .frontend import load_text_front_end
            _, fe = load_text_front_end(None, embeddings_dir)
            self.codec_embedding = np.asarray(fe.codec, dtype=np.float32)
            self.tts_pad_embed = fe.tts_pad_embed
            if cp_dir:
                w = np.load(os.path.join(cp_dir, "code_predictor_weights.npz"))
                self.cp_codec_embeddings = [w[f"codec_emb_{i}"].astype(np.float32) for i in range(P.NUM_CP_CODES)]
        elif weights:
            meta, t = read_pack(weights)
            cfg = ModelConfig.from_meta(meta)
            f32 = lambda n: np.asarray(t[n], dtype=np.float32)
            self.codec_embedding = f32("talker.codec_embedding")
            self.cp_codec_embeddings = [f32(f"cp.codec_emb.{g}") for g in range(cfg.cp_groups)]
            fe = TextFrontEnd(cfg, t["text.embedding"], f32("text.fc1.weight"), f32("text.fc1.bias"),
                              f32("text.fc2.weight"), f32("text.fc2.bias"), self.codec_embedding)
            self.tts_pad_embed = fe.tts_pad_embed

    def _vocoder_chunk(self, codes_list, chunk_idx, results):
        try:
            s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            s.connect(self.voc_socket)
            s.sendall(P.pack_voc_request(np.array(codes_list, dtype=np.int64)))
            head = P.recv_exact(s, 4)
            n = struct.unpack("<i", head)[0]
            results[chunk_idx] = np.frombuffer(P.recv_exact(s, n * 2), dtype=np.int16)
            s.close()
        except Exception as e:
            print(f"  Vocoder chunk {chunk_idx} error: {e}")
            results[chunk_idx] = np.array([], dtype=np.int16)

    def synthesize(self, text, language="russian", output="output.wav", streaming=False, token_ids=None):
        t_start = time.time()
        talker = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        talker.connect(self.talker_socket)
        talker.sendall(P.pack_talker_request(text, language, token_ids))
        all_codes, pending, threads, results = [], [], [], {}
        while True:
            head = P.recv_exact(talker, 4)
            if len(head) < 4:
                break
            code_0 = struct.unpack("<i", head)[0]
            if code_0 in (P.SENTINEL_DONE, P.SENTINEL_ERROR):
                print("  Talker done" if code_0 == P.SENTINEL_DONE else "  Talker error!")
                break
            hidden = np.frombuffer(P.recv_exact(talker, P.HIDDEN_SIZE * 4), dtype=np.float32).copy()
            cp = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)      # new connection per frame
            cp.connect(self.cp_socket)
            cp.sendall(P.pack_cp_request(hidden, code_0))
            codes_1_15 = list(struct.unpack("<15i", P.recv_exact(cp, 60)))
            cp.close()
            frame = [code_0] + codes_1_15
            all_codes.append(frame)
            pending.append(frame)
            if streaming and len(pending) >= VOC_CHUNK_SIZE:
                th = threading.Thread(target=self._vocoder_chunk, args=(pending[:], len(threads), results))
                th.start()
                threads.append(th)
                pending = []
            talker.sendall(feedback_embedding(code_0, codes_1_15, self.codec_embedding, self.cp_codec_embeddings,
                                              self.tts_pad_embed).tobytes())
        talker.close()
        n_tokens = len(all_codes)
        if n_tokens == 0:
            print("No tokens generated!")
            return None
        if pending:
            if streaming and threads:
                th = threading.Thread(target=self._vocoder_chunk, args=(pending[:], len(threads), results))
                th.start()
                threads.append(th)
            else:
                self._vocoder_chunk(pending, 0, results)
        for th in threads:
            th.join()
        chunks = [results[i] for i in range(max(len(threads), 1)) if i in results and len(results[i])]
        if not chunks:
            print("No audio generated!")
            return None
        audio = np.concatenate(chunks)
        with wavmod.open(output, "w") as wf:
            wf.setnchannels(1)
            wf.setsampwidth(2)
            wf.setframerate(SAMPLE_RATE)
            wf.writeframes(audio.tobytes())
        dur, total = len(audio) / SAMPLE_RATE, time.time() - t_start
        print(f"\nAudio: {dur:.2f}s ({n_tokens} frames), saved to {output}")
        print(f"Total: {total:.2f}s (RTF={total / dur:.3f}x)")
        return np.array(all_codes, dtype=np.int32), audio


def main():
    ap = argparse.ArgumentParser(description="Qwen3-TTS Client (MI355X servers)")
    ap.add_argument("text", nargs="?", default=None)
    ap.add_argument("--text", dest="text_flag", default=None)
    ap.add_argument("--language", default="russian")
    ap.add_argument("--output", default="output.wav")
    ap.add_argument("--talker_socket", default="/tmp/qwen3_talker.sock")
    ap.add_argument("--cp_socket", default="/tmp/qwen3_cp.sock")
    ap.add_argument("--voc_socket", default="/tmp/qwen3_voc.sock")
    ap.add_argument("--weights", default=None, help="Q3TTSW1 container (tables for the feedback embedding)")
    ap.add_argument("--embeddings_dir", default=None, help="the reference's embeddings/ directory (instead of --weights)")
    ap.add_argument("--cp_dir", default=None, help="the reference's code-predictor directory (code_predictor_weights.npz)")
    ap.add_argument("--token_ids", default=None, help="comma-separated text token ids (skips the tokenizer)")
    ap.add_argument("--streaming", action="store_true")
    a = ap.parse_args()
    text = a.text or a.text_flag or "Привет, как дела? Сегодня хорошая погода для прогулки."
    ids = [int(x) for x in a.token_ids.split(",")] if a.token_ids else None
    if not a.weights and not (a.embeddings_dir and a.cp_dir):
        ap.error("give --weights, or --embeddings_dir and --cp_dir")
    Qwen3TTSClient(a.talker_socket, a.cp_socket, a.voc_socket, a.weights, a.embeddings_dir, a.cp_dir).synthesize(
        text, a.language, a.output, streaming=a.streaming, token_ids=ids)


if __name__ == "__main__":
    main()
