"""Wire protocol of the three Unix-socket servers (SURVEY.md "Wire protocol"): little-endian raw
structs, no framing beyond what is listed.  Kept byte-for-byte so the reference's client and servers
interoperate with these (reference: llamacpp_talker_server.py:264-297,332-352,
code_predictor_server.py:161-187, vocoder_server.py:143-178, tts_client.py:84-86,128-130,169-171,211).
"""
from __future__ import annotations

import json
import struct

import numpy as np

HIDDEN_SIZE = 1024
SENTINEL_DONE = -1
SENTINEL_ERROR = -2
NUM_CP_CODES = 15
MAX_REQUEST_BYTES = 65536
MAX_VOC_TOKENS = 10000


def recv_exact(conn, n: int) -> bytes:
    """Read exactly n bytes; returns fewer only if the peer closed."""
    buf = bytearray()
    while len(buf) < n:
        chunk = conn.recv(min(65536, n - len(buf)))
        if not chunk:
            break
        buf += chunk
    return bytes(buf)


# client -> talker: u32 len + UTF-8 JSON {"text", "language"}  (+ optional "token_ids" extension)
def pack_talker_request(text: str, language: str = "russian", token_ids=None) -> bytes:
    msg = {"text": text, "language": language}
    if token_ids is not None:
        msg["token_ids"] = [int(t) for t in token_ids]
    raw = json.dumps(msg).encode()
    return struct.pack("<I", len(raw)) + raw


def read_talker_request(conn):
    head = recv_exact(conn, 4)
    if len(head) < 4:
        return None
    (n,) = struct.unpack("<I", head)
    if n > MAX_REQUEST_BYTES:
        raise ValueError(f"request of {n} bytes exceeds {MAX_REQUEST_BYTES}")
    return json.loads(recv_exact(conn, n).decode())


# talker -> client per frame: i32 code_0 + f32[1024] hidden (4100 B); end: i32 -1 / -2
def pack_talker_frame(code_0: int, hidden: np.ndarray) -> bytes:
    return struct.pack("<i", int(code_0)) + np.ascontiguousarray(hidden, dtype="<f4").tobytes()


def pack_sentinel(v: int) -> bytes:
    return struct.pack("<i", v)


# client -> CP (new connection per frame): f32[1024] hidden + i32 code_0 (4100 B); CP -> client: i32[15]
def pack_cp_request(hidden: np.ndarray, code_0: int) -> bytes:
    return np.ascontiguousarray(hidden, dtype="<f4").tobytes() + struct.pack("<i", int(code_0))


def read_cp_request(conn):
    raw = recv_exact(conn, HIDDEN_SIZE * 4 + 4)
    if len(raw) < HIDDEN_SIZE * 4 + 4:
        return None
    return np.frombuffer(raw[:HIDDEN_SIZE * 4], dtype="<f4"), struct.unpack("<i", raw[HIDDEN_SIZE * 4:])[0]


def pack_cp_reply(codes) -> bytes:
    return np.asarray(list(codes)[:NUM_CP_CODES], dtype="<i4").tobytes()


# client -> vocoder: i32 n (1..10000) + i64[n*16]; vocoder -> client: i32 n_samples + i16[n_samples]
def pack_voc_request(codes) -> bytes:
    c = np.ascontiguousarray(codes, dtype="<i8").reshape(-1, 16)
    return struct.pack("<i", c.shape[0]) + c.tobytes()


def read_voc_request(conn):
    head = recv_exact(conn, 4)
    if len(head) < 4:
        return None
    (n,) = struct.unpack("<i", head)
    if n <= 0 or n > MAX_VOC_TOKENS:
        return None
    raw = recv_exact(conn, n * 16 * 8)
    if len(raw) < n * 16 * 8:
        return None
    return np.frombuffer(raw, dtype="<i8").reshape(n, 16)


def pack_voc_reply(audio_int16: np.ndarray) -> bytes:
    a = np.ascontiguousarray(audio_int16, dtype="<i2")
    return struct.pack("<i", a.shape[0]) + a.tobytes()
