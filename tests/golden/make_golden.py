#!/usr/bin/env python3
"""Generate tests/golden/frontend_golden.npz by RUNNING the reference's own Python front-end.

Runs only in the build container (needs /root/reference; the GPU box never sees it).  The
reference's compute back-ends (llama.cpp, onnxruntime) are absent, so each server object is
created with object.__new__ and given seeded synthetic tables and a stub back-end; what is
recorded is the reference's *front-end* behaviour:

  prefix_*    Qwen3TTSTalkerServer._embed_text/_build_prefix   (llamacpp_talker_server.py:115-161)
  sample_*    Qwen3TTSTalkerServer._sample_token               (:163-206), T=0 and seeded T>0
  cps_*       CodePredictorServer._sample                      (code_predictor_server.py:87-92)
  cploop_*    CodePredictorServer.predict call schedule/tokens (:94-140) over a stub session
  client_*    Qwen3TTSClient.synthesize over scripted sockets  (tts_client.py:110-271): the bytes it
              sends (wire protocol) and the feedback embedding it computes (:199-208)
  voc_*       VocoderServer.synthesize chunking/crossfade      (vocoder_server.py:73-121)
  vocsrv_*    VocoderServer.serve over a real AF_UNIX socket   (:123-190): wire bytes + int16 rule

Usage:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import struct
import sys
import tempfile
import threading
import time
import types

import numpy as np

REF = "/root/reference/dual_npu"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "frontend_golden.npz")

sys.path.insert(0, REF)
stub = types.ModuleType("llama_cpp_bindings")
stub.LlamaCppModel = object
sys.modules["llama_cpp_bindings"] = stub

import llamacpp_talker_server as ts  # noqa: E402
import code_predictor_server as cps  # noqa: E402
import vocoder_server as vs  # noqa: E402
import tts_client as tc  # noqa: E402

G = {}  # golden dict


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ----------------------------------------------------------------------------- talker front-end
H, TD, TV, CV = 32, 16, 151936, 3072


def talker_tables(seed):
    r = np.random.default_rng(seed)
    return dict(text_embedding=(0.5 * r.standard_normal((TV, TD))).astype(np.float32),
                codec_embedding=(0.5 * r.standard_normal((CV, H))).astype(np.float32),
                codec_head=(0.5 * r.standard_normal((CV, H))).astype(np.float32),
                proj_fc1_w=(0.3 * r.standard_normal((TD, TD))).astype(np.float32),
                proj_fc1_b=(0.1 * r.standard_normal(TD)).astype(np.float32),
                proj_fc2_w=(0.3 * r.standard_normal((H, TD))).astype(np.float32),
                proj_fc2_b=(0.1 * r.standard_normal(H)).astype(np.float32))


def make_talker(seed, temperature=0.0, top_k=50):
    srv = object.__new__(ts.Qwen3TTSTalkerServer)
    for k, v in talker_tables(seed).items():
        setattr(srv, k, v)
    srv.temperature, srv.top_k = temperature, top_k
    sp = srv._embed_text(np.array([ts.TTS_PAD_TOKEN_ID, ts.TTS_BOS_TOKEN_ID, ts.TTS_EOS_TOKEN_ID]))
    srv.tts_pad_embed, srv.tts_bos_embed, srv.tts_eos_embed = sp[0], sp[1], sp[2]
    return srv


G["talker_dims"] = np.array([H, TD, TV, CV])
G["talker_seed"] = np.array(11)
srv = make_talker(11)
id_rng = np.random.default_rng(5)
for i, n in enumerate([1, 5, 20]):
    ids = id_rng.integers(0, 151000, size=n)
    G[f"prefix_{i}_ids"] = ids.astype(np.int64)
    G[f"prefix_{i}_out"] = srv._build_prefix(list(ids))

# The same two functions at the widths the DEVICE text path is built for (text dim 2048 -> 2048 -> hidden 1024:
# include/qwen3tts_text.h runs fc1 / fc2 on the talker's MFMA GEMM kernels, whose K is 1024 / 2048 / 3072), so
# tests/test_gpu_text.py can compare tfe_build_prefix with the reference's own output directly.  A 151 936 x 2048
# f32 table is 1.2 GB; the reference only ever indexes it (`self.text_embedding[token_ids]`), so it gets an object
# whose row(id) is row id % 640 of a 640-row table -- the reference's code and constants are untouched, the
# special ids it looks up (151644, 77091, 198, 151671-3) land on rows 604, 291, 198, 631-633.
WV = 640


class ModTable:
    def __init__(self, small):
        self.small = small

    def __getitem__(self, ids):
        return self.small[np.asarray(ids) % self.small.shape[0]]


def wide_tables(seed):
    r = np.random.default_rng(seed)
    return dict(text_embedding=(0.05 * r.standard_normal((WV, 2048))).astype(np.float32),
                proj_fc1_w=(0.02 * r.standard_normal((2048, 2048))).astype(np.float32),
                proj_fc1_b=(0.02 * r.standard_normal(2048)).astype(np.float32),
                proj_fc2_w=(0.02 * r.standard_normal((1024, 2048))).astype(np.float32),
                proj_fc2_b=(0.02 * r.standard_normal(1024)).astype(np.float32),
                codec_embedding=(0.05 * r.standard_normal((CV, 1024))).astype(np.float32))


wsrv = object.__new__(ts.Qwen3TTSTalkerServer)
for k, v in wide_tables(12).items():
    setattr(wsrv, k, v)
wsrv.text_embedding = ModTable(wsrv.text_embedding)
wsp = wsrv._embed_text(np.array([ts.TTS_PAD_TOKEN_ID, ts.TTS_BOS_TOKEN_ID, ts.TTS_EOS_TOKEN_ID]))
wsrv.tts_pad_embed, wsrv.tts_bos_embed, wsrv.tts_eos_embed = wsp[0], wsp[1], wsp[2]
G["wprefix_seed"] = np.array(12)
wid_rng = np.random.default_rng(6)
for i, n in enumerate([2, 9]):
    ids = wid_rng.integers(0, 600, size=n)
    G[f"wprefix_{i}_ids"] = ids.astype(np.int64)
    out = wsrv._build_prefix(list(ids))
    assert out.shape == (n + 9, 1024) and out.dtype == np.float32
    G[f"wprefix_{i}_out"] = out

# sampling: temperature 0 (greedy limit) across the EOS-boost / force / repetition regimes
cases = []
srng = np.random.default_rng(21)
for ci in range(24):
    hidden = srng.standard_normal(H).astype(np.float32) * (1.0 + ci % 3)
    n_text = [0, 3, 5, 10][ci % 4]
    n_past = [0, 4, 7, 12, 25, 40][ci % 6]
    past = list(map(int, srng.integers(0, 2048, size=n_past)))
    if n_past >= 4:
        past[-1] = past[-3]  # duplicates inside the window
    cases.append((hidden, past, n_text))
G["sample_n"] = np.array(len(cases))
for ci, (hidden, past, n_text) in enumerate(cases):
    tok = srv._sample_token(hidden.copy(), past_tokens=list(past) if ci % 5 else (list(past) or None),
                            n_text_tokens=n_text)
    G[f"sample_{ci}_hidden"] = hidden
    G[f"sample_{ci}_past"] = np.array(past, dtype=np.int64)
    G[f"sample_{ci}_ntext"] = np.array(n_text)
    G[f"sample_{ci}_tok"] = np.array(tok)
# make the positive-logit arg-max land on a repeated token at least once
hidden = cases[0][0]
base = int(np.argmax((hidden @ srv.codec_head.T)[:2048]))
tok = srv._sample_token(hidden.copy(), past_tokens=[base, 3, base], n_text_tokens=50)
G["sample_rep_hidden"], G["sample_rep_past"], G["sample_rep_tok"] = hidden, np.array([base, 3, base]), np.array(tok)
# seeded stochastic path (T=0.8, top-k 50, top-p 0.95)
srv_t = make_talker(11, temperature=0.8)
toks = []
np.random.seed(1234)
for ci in range(8):
    hidden, past, n_text = cases[ci]
    toks.append(srv_t._sample_token(hidden.copy(), past_tokens=list(past), n_text_tokens=n_text))
G["sample_stoch_toks"] = np.array(toks)

# ----------------------------------------------------------------------------- code predictor
CPH, CPV = 1024, 2048  # predict() slices [:HIDDEN_SIZE]: keep the real width, small vocab tables


class StubSession:
    """Stands in for onnxruntime: out hidden = tanh(0.9*in + 0.05*position), KV grows by n."""

    def __init__(self):
        self.calls = []

    def run(self, _, feed):
        hid, pos = feed["hidden"], feed["position"]
        n = hid.shape[1]
        self.calls.append((tuple(hid.shape), [int(p) for p in pos], int(feed["past_k_0"].shape[2])))
        out = [np.tanh(0.9 * hid + 0.05 * pos.astype(np.float32)[None, :, None]).astype(np.float32)]
        for i in range(5):
            for nm in ("past_k", "past_v"):
                p = feed[f"{nm}_{i}"]
                out.append(np.concatenate([p, np.zeros((1, 8, n, 128), np.float32)], axis=2))
        return out


def make_cp(seed, batch_prefill, temperature=0.0):
    r = np.random.default_rng(seed)
    s = object.__new__(cps.CodePredictorServer)
    s.temperature, s.top_k, s.num_groups, s.batch_prefill = temperature, 50, 15, batch_prefill
    s.codec_embeddings = [(0.5 * r.standard_normal((CPV, CPH))).astype(np.float32) for _ in range(15)]
    s.lm_heads = [(0.5 * r.standard_normal((CPV, CPH))).astype(np.float32) for _ in range(15)]
    s.codec_embedding = (0.5 * r.standard_normal((CV, CPH))).astype(np.float32)
    s.sess = StubSession()
    s.num_layers, s.head_dim, s.num_kv_heads = 5, 128, 8
    return s


G["cp_seed"] = np.array(31)
for bp in (0, 1):
    s = make_cp(31, bool(bp))
    hid = np.random.default_rng(32).standard_normal(CPH).astype(np.float32)
    toks = s.predict(hid, 777)
    G[f"cploop_{bp}_hidden"] = hid
    G[f"cploop_{bp}_tokens"] = np.array(toks)
    G[f"cploop_{bp}_calls"] = np.array(json.dumps(s.sess.calls))
s = make_cp(31, False, temperature=0.0)
lg = np.random.default_rng(33).standard_normal(CPV).astype(np.float32)
G["cps_logits"] = lg
G["cps_greedy"] = np.array(s._sample(lg.copy()))
s.temperature = 0.1
np.random.seed(99)
G["cps_stoch"] = np.array([s._sample(lg.copy()) for _ in range(6)])

# ----------------------------------------------------------------------------- client over scripted sockets
class FakeSock:
    """Scripted peer: `script` bytes are handed out by recv(); everything sent is recorded."""
    registry = {}

    def __init__(self, *a):
        self.sent, self.rx, self.path = b"", b"", None

    def connect(self, path):
        self.path = path
        FakeSock.registry.setdefault(path, []).append(self)
        self.rx = FakeSock.script(path, len(FakeSock.registry[path]) - 1)

    def sendall(self, b):
        self.sent += bytes(b)

    def recv(self, n):
        out, self.rx = self.rx[:n], self.rx[n:]
        return out

    def close(self):
        pass


N_FRAMES = 3
crng = np.random.default_rng(41)
c_hidden = crng.standard_normal((N_FRAMES, 1024)).astype(np.float32)
c_code0 = [100, 2047, 5]
c_cp = crng.integers(0, 2048, size=(N_FRAMES, 15)).astype(np.int32)


def script(path, k):
    if path == "talker":
        b = b""
        for f in range(N_FRAMES):
            b += struct.pack("<i", c_code0[f]) + c_hidden[f].tobytes()
        return b + struct.pack("<i", -1)
    if path == "cp":
        return c_cp[k].tobytes()
    if path == "voc":
        n = N_FRAMES * 1920
        return struct.pack("<i", n) + (np.arange(n) % 1000).astype(np.int16).tobytes()
    raise KeyError(path)


FakeSock.script = staticmethod(script)
real_socket = tc.socket.socket
tc.socket.socket = FakeSock
try:
    cl = tc.Qwen3TTSClient(talker_socket="talker", cp_socket="cp", voc_socket="voc", embeddings_dir=None)
    tr = np.random.default_rng(42)
    cl.codec_embedding = (0.5 * tr.standard_normal((CV, 1024))).astype(np.float32)
    cl.cp_codec_embeddings = [(0.5 * tr.standard_normal((2048, 1024))).astype(np.float32) for _ in range(15)]
    cl.tts_pad_embed = (0.5 * tr.standard_normal(1024)).astype(np.float32)
    wav = os.path.join(tempfile.mkdtemp(), "o.wav")
    cl.synthesize("Привет", "russian", wav, streaming=False)
finally:
    tc.socket.socket = real_socket
tk = FakeSock.registry["talker"][0].sent
G["client_seed"] = np.array(42)
G["client_code0"] = np.array(c_code0)
G["client_cp"] = c_cp
G["client_hidden"] = c_hidden
hdr_len = struct.unpack("<I", tk[:4])[0]
G["client_talker_request"] = np.frombuffer(tk[:4 + hdr_len], dtype=np.uint8)
G["client_feedback"] = np.frombuffer(tk[4 + hdr_len:], dtype=np.float32).reshape(N_FRAMES, 1024)
G["client_cp_request_0"] = np.frombuffer(FakeSock.registry["cp"][0].sent, dtype=np.uint8)
G["client_voc_request"] = np.frombuffer(FakeSock.registry["voc"][0].sent, dtype=np.uint8)
import wave  # noqa: E402
with wave.open(wav) as wf:
    G["client_wav_params"] = np.array([wf.getnchannels(), wf.getsampwidth(), wf.getframerate(), wf.getnframes()])

# ----------------------------------------------------------------------------- vocoder chunking
def stub_chunk(padded):
    """Deterministic stand-in for the ONNX vocoder: depends on every code of the chunk and on the
    sample index, so chunk placement, padding and trimming all show up in the output."""
    c = padded[0].astype(np.float64)  # [64,16]
    per_tok = (c @ (np.arange(16) + 1.0)) / (2048.0 * 136.0)  # [64] in [0,1)
    t = np.arange(64 * 1920, dtype=np.float64)
    a = np.repeat(per_tok, 1920) * 0.8 + 0.1 * np.sin(t * 0.001) + 0.05 * per_tok.sum()
    return a.astype(np.float32)


vsrv = object.__new__(vs.VocoderServer)
vsrv.max_tokens, vsrv.is_onnx = 64, True
vsrv._inference_chunk = stub_chunk
vrng = np.random.default_rng(51)
ns = [1, 10, 64, 65, 80, 96, 97, 112, 150, 160, 750]
G["voc_ns"] = np.array(ns)
G["voc_seed"] = np.array(51)
for n in ns:
    codes = vrng.integers(0, 2048, size=(n, 16)).astype(np.int64)
    out = vsrv.synthesize(codes)
    assert out.dtype == np.float32
    G[f"voc_{n}_len"] = np.array(len(out))
    G[f"voc_{n}_sha"] = np.array(sha(out))
    idx = np.linspace(0, len(out) - 1, 64).astype(np.int64)
    G[f"voc_{n}_probe"] = out[idx]

# a model that returns FEWER than 64 x 1920 samples per chunk -- what the traced decoder does if its transposed convs
# trim at both ends like the importable member of its family (Qwen3OmniMoeCausalTransConvNet: 64 frames -> 122 325
# samples; tests/golden/make_code2wav_golden.py): the reference's slices `audio[:n * 1920]` then follow numpy's rule
# (vocoder_server.py:81,98-99).  Three lengths: the family's 122 325, a chunk shorter than the 16-frame overlap region
# would need for the last tokens (60 * 1920 + 7), and one shorter than the overlap itself (15 * 1920 + 5).
G["vocshort_lens"] = np.array([122325, 60 * 1920 + 7, 15 * 1920 + 5])
G["vocshort_ns"] = np.array([1, 63, 64, 65, 80, 97, 112, 150, 750])
for cs in (int(x) for x in G["vocshort_lens"]):
    vsrv._inference_chunk = lambda padded, cs=cs: stub_chunk(padded)[:cs]
    srng = np.random.default_rng(53)
    for n in (int(x) for x in G["vocshort_ns"]):
        codes = srng.integers(0, 2048, size=(n, 16)).astype(np.int64)
        out = vsrv.synthesize(codes)
        G[f"vocshort_{cs}_{n}_len"] = np.array(len(out))
        G[f"vocshort_{cs}_{n}_sha"] = np.array(sha(np.ascontiguousarray(out, dtype=np.float32)))
vsrv._inference_chunk = stub_chunk

# ----------------------------------------------------------------------------- vocoder server over a real socket
sock_path = os.path.join(tempfile.mkdtemp(), "voc.sock")
vsrv.socket_path, vsrv._running = sock_path, True
vsrv._inference_chunk = lambda padded: (stub_chunk(padded) * 8.0 - 4.5)  # exercises clipping + negative truncation
th = threading.Thread(target=vsrv.serve, daemon=True)
th.start()
for _ in range(100):
    if os.path.exists(sock_path):
        break
    time.sleep(0.05)
codes = np.random.default_rng(52).integers(0, 2048, size=(7, 16)).astype(np.int64)
cl2 = tc.Qwen3TTSClient(voc_socket=sock_path)
res = {}
cl2._vocoder_chunk(codes.tolist(), 0, res)
vsrv._running = False
th.join(timeout=3)
G["vocsrv_codes"] = codes
G["vocsrv_int16"] = res[0]
G["vocsrv_float_sha"] = np.array(sha(vsrv.synthesize(codes)))

np.savez_compressed(OUT, **G)
print(f"wrote {OUT}: {len(G)} arrays, {os.path.getsize(OUT)/1024:.0f} KiB")
