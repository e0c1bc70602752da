"""ctypes front of oracle/q3_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(as the checker / the timed CPU baseline); the product path never does.  Parity of the
transformer arithmetic is UNPINNED at the third-party boundary (see q3_oracle.c header).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_float_p = ctypes.POINTER(ctypes.c_float)
c_int_p = ctypes.POINTER(ctypes.c_int)


class _Layer(ctypes.Structure):
    _fields_ = [(n, c_float_p) for n in ("in_ln", "q", "k", "v", "o", "q_norm", "k_norm", "post_ln",
                                         "gate", "up", "down")]


class _Stack(ctypes.Structure):
    _fields_ = [("hidden", ctypes.c_int), ("head_dim", ctypes.c_int), ("n_heads", ctypes.c_int),
                ("n_kv", ctypes.c_int), ("ffn", ctypes.c_int), ("n_layers", ctypes.c_int),
                ("eps", ctypes.c_float), ("layers", ctypes.POINTER(_Layer)), ("final_norm", c_float_p),
                ("rope_cos", c_float_p), ("rope_sin", c_float_p)]


def build(force: bool = False) -> str:
    so = os.path.join(HERE, "libq3oracle.so")
    src = os.path.join(HERE, "q3_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-B", "libq3oracle.so"], stdout=subprocess.DEVNULL)
    return so


def _limit_threads():
    """The GPU box gives this job a CPU share of a much larger host: cap OpenMP so the oracle does
    not oversubscribe (it gets slower by orders of magnitude when it does)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(8, n))))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def n_threads() -> int:
    _limit_threads()
    return int(os.environ["OMP_NUM_THREADS"])


def lib():
    global _LIB
    if _LIB is None:
        _limit_threads()
        L = ctypes.CDLL(build())
        L.orc_round_f16.restype = ctypes.c_float
        L.orc_round_f16.argtypes = [ctypes.c_float]
        L.orc_round_f16_array.argtypes = [c_float_p, c_float_p, ctypes.c_long]
        L.orc_rope_tables.argtypes = [ctypes.c_double, ctypes.c_int, ctypes.c_int, c_float_p, c_float_p]
        L.orc_forward.restype = ctypes.c_int
        L.orc_forward.argtypes = [ctypes.POINTER(_Stack), c_float_p, c_float_p, ctypes.c_int, c_float_p,
                                  ctypes.c_int, ctypes.c_int, c_float_p, c_float_p]
        L.orc_head.argtypes = [c_float_p, ctypes.c_int, ctypes.c_int, c_float_p, c_float_p]
        L.orc_cp_predict.restype = ctypes.c_int
        L.orc_cp_predict.argtypes = [ctypes.POINTER(_Stack), c_float_p, ctypes.c_int, ctypes.POINTER(c_float_p),
                                     ctypes.POINTER(c_float_p), ctypes.c_int, ctypes.c_int, c_float_p,
                                     ctypes.c_int, c_int_p, c_int_p, c_float_p, c_float_p]
        L.orc_set_exact.argtypes = [ctypes.c_int]
        pp = ctypes.POINTER(c_float_p)
        L.orc_forward_batch.restype = ctypes.c_int
        L.orc_forward_batch.argtypes = [ctypes.POINTER(_Stack), pp, pp, c_int_p, c_int_p, ctypes.c_int, c_float_p, c_float_p]
        L.orc_cp_predict_batch.restype = ctypes.c_int
        L.orc_cp_predict_batch.argtypes = [ctypes.POINTER(_Stack), c_float_p, ctypes.c_int, pp, pp, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, c_float_p, c_int_p, c_int_p, c_int_p, c_float_p]
        L.orc_head_batch.argtypes = [c_float_p, ctypes.c_int, ctypes.c_int, c_float_p, ctypes.c_int, c_float_p]
        L.orc_matvec.argtypes = [c_float_p, c_float_p, c_float_p, ctypes.c_int, ctypes.c_int]
        L.orc_rmsnorm_fold.restype = ctypes.c_float
        L.orc_rmsnorm_fold.argtypes = [c_float_p, c_float_p, ctypes.c_float, ctypes.c_int, c_float_p]
        _LIB = L
    return _LIB


def set_exact(on: bool) -> None:
    """Exact mode = plain fp32 Qwen3 layer (no fp16 rounding of activations / K/V): the form that is
    compared with transformers' Qwen3Model (tests/golden/make_hf_golden.py)."""
    lib().orc_set_exact(1 if on else 0)


def fp(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_float_p)


def round_f16(x: np.ndarray) -> np.ndarray:
    """Saturating fp16 rounding, the GEMM-input rounding of the numerics contract."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    lib().orc_round_f16_array(fp(x), fp(y), x.size)
    return y


def rope_tables(theta: float, head_dim: int, max_pos: int):
    cs = np.empty((max_pos, head_dim // 2), np.float32)
    sn = np.empty_like(cs)
    lib().orc_rope_tables(float(theta), head_dim, max_pos, fp(cs), fp(sn))
    return cs, sn


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


class StackOracle:
    """One transformer stack (talker or code predictor) with its own KV cache."""

    def __init__(self, cfg, tensors: dict, prefix: str, n_layers: int, ffn: int, n_ctx: int):
        self.cfg, self.n_ctx, self.n_layers = cfg, n_ctx, n_layers
        self._keep = []
        layers = (_Layer * n_layers)()
        for i in range(n_layers):
            for fld, part in (("in_ln", "input_ln"), ("q", "q_proj"), ("k", "k_proj"), ("v", "v_proj"),
                              ("o", "o_proj"), ("q_norm", "q_norm"), ("k_norm", "k_norm"),
                              ("post_ln", "post_ln"), ("gate", "gate_proj"), ("up", "up_proj"),
                              ("down", "down_proj")):
                a = tensors[f"{prefix}.layers.{i}.{part}"]
                # projections are fp16 on the device; widen the SAME fp16 values
                a = _f32(np.asarray(a).astype(np.float16)) if a.ndim == 2 else _f32(a)
                self._keep.append(a)
                setattr(layers[i], fld, fp(a))
        self._layers = layers
        self.final_norm = _f32(tensors[f"{prefix}.norm"])
        self.cos, self.sin = rope_tables(cfg.rope_theta, cfg.head_dim, max(n_ctx, 32))
        st = _Stack()
        st.hidden, st.head_dim, st.n_heads, st.n_kv = cfg.hidden, cfg.head_dim, cfg.n_heads, cfg.n_kv_heads
        st.ffn, st.n_layers, st.eps = ffn, n_layers, cfg.rms_eps
        st.layers = ctypes.cast(layers, ctypes.POINTER(_Layer))
        st.final_norm, st.rope_cos, st.rope_sin = fp(self.final_norm), fp(self.cos), fp(self.sin)
        self.st = st
        self.clear()

    def clear(self):
        shape = (self.n_layers, self.cfg.n_kv_heads, self.n_ctx, self.cfg.head_dim)
        self.kc = np.zeros(shape, np.float32)
        self.vc = np.zeros(shape, np.float32)

    def forward(self, embd: np.ndarray, pos_start: int, all_rows: bool = False):
        embd = _f32(embd).reshape(-1, self.cfg.hidden)
        n = embd.shape[0]
        last = np.empty(self.cfg.hidden, np.float32)
        allh = np.empty((n, self.cfg.hidden), np.float32) if all_rows else None
        rc = lib().orc_forward(ctypes.byref(self.st), fp(self.kc), fp(self.vc), self.n_ctx, fp(embd), n,
                               int(pos_start), fp(last), fp(allh) if all_rows else None)
        if rc != 0:
            raise RuntimeError("orc_forward failed (context overflow?)")
        return allh if all_rows else last


def head_logits(head_f32: np.ndarray, hidden: np.ndarray) -> np.ndarray:
    V, H = head_f32.shape
    out = np.empty(V, np.float32)
    lib().orc_head(fp(head_f32), V, H, fp(_f32(hidden)), fp(out))
    return out


class TalkerOracle(StackOracle):
    def __init__(self, cfg, tensors, n_ctx=512):
        super().__init__(cfg, tensors, "talker", cfg.talker_layers, cfg.talker_ffn, n_ctx)
        self.codec_embedding = _f32(tensors["talker.codec_embedding"])
        self.codec_head = _f32(np.asarray(tensors["talker.codec_head"]).astype(np.float16))

    def logits(self, hidden):
        return head_logits(self.codec_head, hidden)


def forward_batch(stack: StackOracle, caches, pos, embd: np.ndarray) -> np.ndarray:
    """One decode step of B independent sequences through `stack`'s weights: caches = [(kc, vc, n_ctx)] per row
    (arrays shaped like StackOracle.kc), pos[b] = position appended.  Per row identical to StackOracle.forward."""
    B = len(caches)
    embd = _f32(embd).reshape(B, stack.cfg.hidden)
    kcp = (c_float_p * B)(*[fp(c[0]) for c in caches])
    vcp = (c_float_p * B)(*[fp(c[1]) for c in caches])
    nctx = np.ascontiguousarray([c[2] for c in caches], dtype=np.int32)
    posa = np.ascontiguousarray(pos, dtype=np.int32)
    out = np.empty((B, stack.cfg.hidden), np.float32)
    rc = lib().orc_forward_batch(ctypes.byref(stack.st), kcp, vcp, nctx.ctypes.data_as(c_int_p),
                                 posa.ctypes.data_as(c_int_p), B, fp(embd), fp(out))
    if rc != 0:
        raise RuntimeError("orc_forward_batch failed (context overflow?)")
    return out


def head_logits_batch(head_f32: np.ndarray, hidden: np.ndarray) -> np.ndarray:
    V, H = head_f32.shape
    hidden = _f32(hidden).reshape(-1, H)
    out = np.empty((hidden.shape[0], V), np.float32)
    lib().orc_head_batch(fp(head_f32), V, H, fp(hidden), hidden.shape[0], fp(out))
    return out


class CpOracle(StackOracle):
    def __init__(self, cfg, tensors):
        super().__init__(cfg, tensors, "cp", cfg.cp_layers, cfg.cp_ffn, cfg.cp_groups + 1)
        G = cfg.cp_groups
        self.talker_emb = _f32(tensors["talker.codec_embedding"])
        self.emb = [_f32(tensors[f"cp.codec_emb.{g}"]) for g in range(G)]
        self.heads = [_f32(np.asarray(tensors[f"cp.lm_head.{g}"]).astype(np.float16)) for g in range(G)]
        self._emb_p = (c_float_p * G)(*[fp(a) for a in self.emb])
        self._head_p = (c_float_p * G)(*[fp(a) for a in self.heads])

    def predict(self, hidden, code0: int, forced=None, want_hidden=False):
        """-> (codes[15], margins[15]) greedy; `forced` teacher-forces the fed-back tokens."""
        G, H = self.cfg.cp_groups, self.cfg.hidden
        codes = np.zeros(G, np.int32)
        margins = np.zeros(G, np.float32)
        hid = np.empty((G, H), np.float32) if want_hidden else None
        f = None
        if forced is not None:
            f = np.ascontiguousarray(forced, dtype=np.int32)
        rc = lib().orc_cp_predict(ctypes.byref(self.st), fp(self.talker_emb), self.talker_emb.shape[0],
                                  self._emb_p, self._head_p, self.cfg.cp_vocab, G, fp(_f32(hidden)), int(code0),
                                  f.ctypes.data_as(c_int_p) if f is not None else None,
                                  codes.ctypes.data_as(c_int_p), fp(margins), fp(hid) if want_hidden else None)
        if rc != 0:
            raise RuntimeError("orc_cp_predict failed")
        return (codes, margins, hid) if want_hidden else (codes, margins)

    def predict_batch(self, hidden, code0, forced=None):
        """B rows at once -> (codes[B][15], margins[B][15]); per row identical to predict()."""
        G, H = self.cfg.cp_groups, self.cfg.hidden
        hidden = _f32(hidden).reshape(-1, H)
        B = hidden.shape[0]
        c0 = np.ascontiguousarray(code0, dtype=np.int32).reshape(B)
        codes = np.zeros((B, G), np.int32)
        margins = np.zeros((B, G), np.float32)
        f = None if forced is None else np.ascontiguousarray(forced, dtype=np.int32).reshape(B, G)
        rc = lib().orc_cp_predict_batch(ctypes.byref(self.st), fp(self.talker_emb), self.talker_emb.shape[0],
                                        self._emb_p, self._head_p, self.cfg.cp_vocab, G, B, fp(hidden),
                                        c0.ctypes.data_as(c_int_p), f.ctypes.data_as(c_int_p) if f is not None else None,
                                        codes.ctypes.data_as(c_int_p), fp(margins))
        if rc != 0:
            raise RuntimeError("orc_cp_predict_batch failed")
        return codes, margins
