#!/usr/bin/env python3
"""Batched synthesis server: B utterances per request through the on-device frame loop.

The reference closes the autoregressive loop in its client, two socket round trips per frame and one
utterance at a time (dual_npu/tts_client.py:144-215; `listen(1)`, llamacpp_talker_server.py:314).  This
server keeps that front-end (tokenizer, text projection, dual-stream prefix: llamacpp_talker_server.py:
115-161; EOS heuristics and sampling on the device) and runs the whole loop for all utterances of a request
on the GPU (include/qwen3tts_engine.h), then the vocoder's chunk walk (vocoder_server.py:73-121, :175) per
utterance.  Extension of the wire protocol, same conventions (little-endian, u32 length + UTF-8 JSON):

    request   u32 len + JSON {"texts": [...], "language": "...", "max_tokens": N}   or {"token_ids": [[...], ...]}
    reply     i32 n_utterances (or -2 on error), then per utterance:
              i32 n_frames, i32[n_frames*16] codes, i32 n_samples, i16[n_samples] PCM (24 kHz)

One process per GPU (HIP_VISIBLE_DEVICES), like the other servers.

`--pipeline`: the vocoder of request k runs on a worker thread (and replies on k's connection) while the frame loop of request
k + 1 already runs -- the reference's client does the same per 64-frame block of ONE utterance (tts_client.py:188-197).  The
vocoder then launches one persistent workgroup per compute unit (voc_set_max_workgroups(-1)), which leaves the frame loop's
workgroups room beside it (244 -> 209 ms per 32 x 64-frame step); results are bit-identical.
"""
from __future__ import annotations

import argparse
import os
import signal
import socket
import struct
import time

import numpy as np

from . import hiplib
from . import protocol as P
from .engine import FrameEngine
from .frontend import TextFrontEnd
from .weights import ModelConfig, read_pack


class BatchSynthesisServer:
    def __init__(self, model_path, vocoder_path, socket_path="/tmp/qwen3_batch.sock", max_batch=32, n_ctx=512,
                 max_tokens=200, temperature=0.0, top_k=50, cp_temperature=0.0, tokenizer=None, seed=0,
                 install_signal_handlers=True, max_request=None, pipeline=False):
        self.socket_path, self.max_batch, self.max_tokens = socket_path, max_batch, max_tokens
        # utterances one request may queue (the server is single-threaded: an unbounded request holds it indefinitely)
        self.max_request = int(max_request) if max_request else 8 * max_batch
        meta, t = read_pack(model_path)
        self.cfg = ModelConfig.from_meta(meta)
        f32 = lambda n: np.asarray(t[n], dtype=np.float32)
        self.front = TextFrontEnd(self.cfg, t["text.embedding"], f32("text.fc1.weight"), f32("text.fc1.bias"),
                                  f32("text.fc2.weight"), f32("text.fc2.bias"), f32("talker.codec_embedding"))
        self.tokenizer = None
        if tokenizer:
            from .tokenizer import ByteLevelBPE
            self.tokenizer = ByteLevelBPE.from_dir(tokenizer)
        self.eng = FrameEngine(model_path, max_batch=max_batch, n_ctx=n_ctx, max_frames=max_tokens)
        self.eng.set_pad_embed(self.front.tts_pad_embed)
        self.eng.set_sampling(temperature, top_k, 0.95, cp_temperature, top_k, seed)
        self.n_ctx = n_ctx
        self._lib = hiplib.load()
        self.voc = self._lib.voc_load(str(vocoder_path).encode(), 64, min(max_batch, 32))
        if not self.voc:
            raise RuntimeError(f"voc_load failed: {vocoder_path}")
        self.pipeline = bool(pipeline)
        self._pool = None
        if self.pipeline:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=1)       # ONE worker: the vocoder handle has one caller, replies keep their order
            self._lib.voc_set_max_workgroups(-1)
        self._running = True
        if install_signal_handlers:
            signal.signal(signal.SIGINT, self._signal_handler)
            signal.signal(signal.SIGTERM, self._signal_handler)

    def _signal_handler(self, signum, frame):
        self._running = False

    def _token_ids(self, msg):
        if msg.get("token_ids") is not None:
            return [[int(x) for x in ids] for ids in msg["token_ids"]]
        if self.tokenizer is None:
            raise RuntimeError("no tokenizer configured (--tokenizer DIR) and the request has no token_ids")
        return [self.tokenizer.encode(t, add_special_tokens=False) for t in msg.get("texts", [])]

    def generate(self, token_ids, max_tokens=None):
        """The frame loop of a request -> list of codes int32 [n_frames][16] per utterance."""
        B = len(token_ids)
        if B == 0:
            raise ValueError("a request needs at least one utterance")
        if B > self.max_request:
            raise ValueError(f"a request may carry at most {self.max_request} utterances (got {B})")
        max_tokens = min(int(max_tokens or self.max_tokens), self.max_tokens)
        prefixes = [self.front.build_prefix(ids) for ids in token_ids]
        if max(p.shape[0] for p in prefixes) + max_tokens > self.n_ctx:
            raise ValueError("prefix + max_tokens exceed n_ctx")
        n_text = [len(ids) for ids in token_ids]
        if B > self.max_batch:
            # more utterances than slots: continuous batching -- a finished utterance's slot takes the next one of the
            # request (q3e_refill), longest expected first (3 frames per text token, llamacpp_talker_server.py:174)
            order = sorted(range(B), key=lambda i: -n_text[i])
            got = self.eng.generate_queue([prefixes[i] for i in order], [n_text[i] for i in order], max_tokens)
            per_utt = [None] * B
            for k, i in enumerate(order):
                per_utt[i] = got[k]
        else:
            self.eng.start(prefixes, n_text, ignore_eos=False, max_frames=max_tokens)
            self.eng.run(max_tokens)
            codes, per = self.eng.codes()
            per_utt = [codes[:int(per[b]), b, :] for b in range(B)]
        return [np.ascontiguousarray(c, dtype=np.int32) for c in per_utt]

    def vocode(self, cs):
        """The vocoder of a request: every utterance's chunk walk in ONE batched call (voc_synthesize_batch: chunks of all
        utterances decoded together, overlap-crossfade assembled on the device; per utterance = VocoderServer.synthesize +
        int16) -> list of (codes, pcm int16)."""
        B = len(cs)
        live = [b for b in range(B) if cs[b].shape[0] > 0]
        pcm = {b: np.zeros(0, np.int16) for b in range(B)}
        if live:
            n = np.array([cs[b].shape[0] for b in live], np.int32)
            cat = np.ascontiguousarray(np.concatenate([cs[b] for b in live], axis=0), dtype=np.int64)
            cap = int(self._lib.voc_synthesize_batch_max_samples(self.voc, hiplib.iptr(n), len(n)))
            buf = np.empty(cap, np.int16)
            off = np.zeros(len(n) + 1, np.int64)
            if self._lib.voc_synthesize_batch(self.voc, cat.ctypes.data_as(hiplib.i64p), hiplib.iptr(n), len(n),
                                              buf.ctypes.data_as(hiplib.i16p), cap, off.ctypes.data_as(hiplib.i64p)) != 0:
                raise RuntimeError("voc_synthesize_batch failed")
            for k, b in enumerate(live):
                pcm[b] = buf[off[k]:off[k + 1]].copy()
        return [(cs[b], pcm[b]) for b in range(B)]

    def synthesize(self, token_ids, max_tokens=None):
        """-> list of (codes int32 [n_frames][16], pcm int16) per utterance."""
        return self.vocode(self.generate(token_ids, max_tokens))

    def _finish(self, conn, cs, t0):
        """Worker side of the pipelined mode: vocode, reply on the request's own connection, close it."""
        try:
            res = self.vocode(cs)
            conn.sendall(pack_batch_reply(res))
            print(f"  {len(res)} utterances, {sum(len(c) for c, _ in res)} frames in {time.time() - t0:.3f}s")
        except Exception as e:
            print(f"Error: {e}")
            try:
                conn.sendall(P.pack_sentinel(P.SENTINEL_ERROR))
            except OSError:
                pass
        finally:
            conn.close()

    def serve(self):
        if os.path.exists(self.socket_path):
            os.unlink(self.socket_path)
        sock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        sock.bind(self.socket_path)
        sock.listen(1)
        sock.settimeout(1.0)
        os.chmod(self.socket_path, 0o666)
        print(f"Batch synthesis server listening on {self.socket_path} ({self.max_batch} slots; longer requests run through them by continuous batching)")
        while self._running:
            try:
                conn, _ = sock.accept()
            except socket.timeout:
                continue
            except OSError:
                break
            handed_over = False
            try:
                msg = P.read_talker_request(conn)
                if msg is None:
                    continue
                t0 = time.time()
                if self._pool is not None:
                    # pipelined: this request's vocoder runs on the worker while the loop accepts and generates the next one
                    cs = self.generate(self._token_ids(msg), msg.get("max_tokens"))
                    self._pool.submit(self._finish, conn, cs, t0)
                    handed_over = True
                    continue
                res = self.synthesize(self._token_ids(msg), msg.get("max_tokens"))
                conn.sendall(pack_batch_reply(res))
                frames = sum(len(c) for c, _ in res)
                print(f"  {len(res)} utterances, {frames} frames in {time.time() - t0:.3f}s")
            except Exception as e:  # like the reference's servers: report, send the error sentinel, keep serving
                print(f"Error: {e}")
                try:
                    conn.sendall(P.pack_sentinel(P.SENTINEL_ERROR))
                except OSError:
                    pass
            finally:
                if not handed_over:
                    conn.close()
        if self._pool is not None:
            self._pool.shutdown(wait=True)       # replies in flight go out before the socket disappears
            self._pool = None
        sock.close()
        if os.path.exists(self.socket_path):
            os.unlink(self.socket_path)

    def close(self):
        self._running = False
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        if self.pipeline:
            self._lib.voc_set_max_workgroups(0)
        if self.voc:
            self._lib.voc_free(self.voc)
            self.voc = None
        self.eng.destroy()


def pack_batch_request(texts=None, token_ids=None, language="russian", max_tokens=None) -> bytes:
    import json
    msg = {"language": language}
    if token_ids is not None:
        msg["token_ids"] = [[int(t) for t in ids] for ids in token_ids]
    else:
        msg["texts"] = list(texts)
    if max_tokens:
        msg["max_tokens"] = int(max_tokens)
    raw = json.dumps(msg).encode()
    return struct.pack("<I", len(raw)) + raw


def pack_batch_reply(results) -> bytes:
    parts = [struct.pack("<i", len(results))]
    for codes, pcm in results:
        c = np.ascontiguousarray(codes, dtype="<i4").reshape(-1, 16)
        a = np.ascontiguousarray(pcm, dtype="<i2")
        parts += [struct.pack("<i", c.shape[0]), c.tobytes(), struct.pack("<i", a.shape[0]), a.tobytes()]
    return b"".join(parts)


def read_batch_reply(conn):
    """-> list of (codes [n][16] int32, pcm int16); raises on the error sentinel / a short read."""
    head = P.recv_exact(conn, 4)
    if len(head) < 4:
        raise RuntimeError("connection closed")
    (n,) = struct.unpack("<i", head)
    if n < 0:
        raise RuntimeError(f"server error ({n})")
    out = []
    for _ in range(n):
        (nf,) = struct.unpack("<i", P.recv_exact(conn, 4))
        codes = np.frombuffer(P.recv_exact(conn, nf * 16 * 4), dtype="<i4").reshape(nf, 16)
        (ns,) = struct.unpack("<i", P.recv_exact(conn, 4))
        pcm = np.frombuffer(P.recv_exact(conn, ns * 2), dtype="<i2")
        out.append((codes, pcm))
    return out


def synthesize_batch(socket_path, texts=None, token_ids=None, language="russian", max_tokens=None):
    """Client side of the batched request."""
    s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    s.connect(socket_path)
    try:
        s.sendall(pack_batch_request(texts, token_ids, language, max_tokens))
        return read_batch_reply(s)
    finally:
        s.close()


def main():
    ap = argparse.ArgumentParser(description="Qwen3-TTS batched synthesis server (MI355X)")
    ap.add_argument("--model", required=True, help="Q3TTSW1 container with talker, code predictor and text tables")
    ap.add_argument("--vocoder", required=True, help="container with the vocoder program")
    ap.add_argument("--socket", default="/tmp/qwen3_batch.sock")
    ap.add_argument("--tokenizer", default=None, help="directory with vocab.json + merges.txt")
    ap.add_argument("--max_batch", type=int, default=32)
    ap.add_argument("--n_ctx", type=int, default=512)
    ap.add_argument("--max_tokens", type=int, default=200)
    ap.add_argument("--temperature", type=float, default=0.8)
    ap.add_argument("--top_k", type=int, default=50)
    ap.add_argument("--cp_temperature", type=float, default=0.1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--pipeline", action="store_true",
                    help="vocode request k on a worker thread while request k + 1 generates (one vocoder workgroup per CU)")
    a = ap.parse_args()
    srv = BatchSynthesisServer(a.model, a.vocoder, a.socket, a.max_batch, a.n_ctx, a.max_tokens, a.temperature, a.top_k,
                               a.cp_temperature, a.tokenizer, a.seed, pipeline=a.pipeline)
    try:
        srv.serve()
    finally:
        srv.close()


if __name__ == "__main__":
    main()
