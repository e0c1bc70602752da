"""Q3TTSW1 weight container: writer/reader, synthetic weights, converters.

The reference ships no weights; its servers load GGUF (talker), .npy/.npz
(tables, code predictor: scripts/extract_embeddings.py:47-98,
scripts/export_code_predictor_weights.py:51-78) and ONNX (vocoder).  The HIP
libraries load ONE self-describing container instead:

    FileHeader  { char magic[8] = "Q3TTSW1\\0"; u32 version; u32 n_tensors;
                  u32 n_meta; u32 reserved; u64 data_offset; }          32 B
    Meta[n_meta]{ char key[48]; f64 value; }                           56 B each
    Tensor[n]   { char name[96]; u32 dtype; u32 ndim; u64 shape[4];
                  u64 offset; u64 nbytes; }                            152 B each
    data (each tensor 256-B aligned, little-endian, C order)

dtype: 0=f32 1=f16 2=i32 3=i64.  Names follow the HF checkpoint keys the
reference's scripts read (talker.layers.N.q_proj ...), see `talker_tensor_names`.
"""
from __future__ import annotations

import hashlib
import os
import struct
from dataclasses import dataclass, asdict, field

import numpy as np

MAGIC = b"Q3TTSW1\0"
VERSION = 1
DTYPES = {0: np.float32, 1: np.float16, 2: np.int32, 3: np.int64}
DTYPE_CODE = {np.dtype(v): k for k, v in DTYPES.items()}
_HDR = struct.Struct("<8sIIIIQ")
_META = struct.Struct("<48sd")
_TENS = struct.Struct("<96sII4QQQ")
ALIGN = 256


@dataclass
class ModelConfig:
    """Numeric model description stored in the container's meta table.

    Defaults are Qwen3-TTS-12Hz-0.6B (scripts/extract_talker_as_qwen3.py:89-110,
    dual_npu/llamacpp_talker_server.py:44-55, code_predictor_server.cpp:43-47).
    """
    hidden: int = 1024
    head_dim: int = 128
    n_heads: int = 16
    n_kv_heads: int = 8
    talker_layers: int = 28
    talker_ffn: int = 3072
    talker_vocab: int = 3072
    cp_layers: int = 5
    cp_ffn: int = 3072
    cp_vocab: int = 2048
    cp_groups: int = 15
    rms_eps: float = 1e-6
    rope_theta: float = 1e6
    text_vocab: int = 151936
    text_dim: int = 2048
    # special ids (llamacpp_talker_server.py:44-55,132)
    codec_pad: int = 2148
    codec_bos: int = 2149
    codec_eos: int = 2150
    codec_nothink: int = 2155
    codec_think_bos: int = 2156
    codec_think_eos: int = 2157
    tts_pad: int = 151671
    tts_bos: int = 151672
    tts_eos: int = 151673
    im_start: int = 151644
    assistant: int = 77091
    newline: int = 198

    def meta(self) -> dict:
        return {k: float(v) for k, v in asdict(self).items()}

    @staticmethod
    def from_meta(meta: dict) -> "ModelConfig":
        c = ModelConfig()
        for k, v in meta.items():
            if hasattr(c, k):
                cur = getattr(c, k)
                setattr(c, k, float(v) if isinstance(cur, float) else int(round(v)))
        return c


def tiny_config(talker_layers=2, cp_layers=2, text_vocab=512) -> ModelConfig:
    """Small-depth variant for tests: same widths (the kernels are shaped for
    them), fewer layers, a text table small enough to generate in a test."""
    return ModelConfig(talker_layers=talker_layers, cp_layers=cp_layers, text_vocab=text_vocab,
                       tts_pad=text_vocab - 3, tts_bos=text_vocab - 2, tts_eos=text_vocab - 1,
                       im_start=text_vocab - 4, assistant=text_vocab - 5, newline=text_vocab - 6)


# ----------------------------------------------------------------------------
# container I/O
# ----------------------------------------------------------------------------

def write_pack(path: str, meta: dict, tensors: dict) -> None:
    names = list(tensors.keys())
    n_t, n_m = len(names), len(meta)
    table_end = _HDR.size + n_m * _META.size + n_t * _TENS.size
    data_off = (table_end + ALIGN - 1) // ALIGN * ALIGN
    entries, off = [], data_off
    for n in names:
        a = tensors[n]
        if a.dtype not in DTYPE_CODE:
            raise TypeError(f"{n}: unsupported dtype {a.dtype}")
        if a.ndim > 4:
            raise ValueError(f"{n}: ndim>4")
        shape = list(a.shape) + [0] * (4 - a.ndim)
        entries.append((n, DTYPE_CODE[a.dtype], a.ndim, shape, off, a.nbytes))
        off = (off + a.nbytes + ALIGN - 1) // ALIGN * ALIGN
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(_HDR.pack(MAGIC, VERSION, n_t, n_m, 0, data_off))
        for k, v in meta.items():
            f.write(_META.pack(k.encode()[:47], float(v)))
        for n, dt, nd, shape, o, nb in entries:
            f.write(_TENS.pack(n.encode()[:95], dt, nd, *shape, o, nb))
        for (n, dt, nd, shape, o, nb) in entries:
            f.seek(o)
            np.ascontiguousarray(tensors[n]).tofile(f)
        f.truncate(off)
    os.replace(tmp, path)


def read_pack(path: str, mmap: bool = True):
    """-> (meta dict, {name: ndarray}).  Arrays are read-only memmaps."""
    with open(path, "rb") as f:
        magic, ver, n_t, n_m, _, data_off = _HDR.unpack(f.read(_HDR.size))
        if magic != MAGIC or ver != VERSION:
            raise ValueError(f"{path}: not a Q3TTSW1 v{VERSION} container")
        meta = {}
        for _ in range(n_m):
            k, v = _META.unpack(f.read(_META.size))
            meta[k.split(b"\0")[0].decode()] = v
        ents = []
        for _ in range(n_t):
            rec = _TENS.unpack(f.read(_TENS.size))
            ents.append((rec[0].split(b"\0")[0].decode(), rec[1], rec[2], rec[3:7], rec[7], rec[8]))
    tensors = {}
    for n, dt, nd, shape, off, nb in ents:
        shp = tuple(int(s) for s in shape[:nd])
        if mmap:
            tensors[n] = np.memmap(path, dtype=DTYPES[dt], mode="r", offset=off, shape=shp)
        else:
            tensors[n] = np.fromfile(path, dtype=DTYPES[dt], offset=off, count=int(np.prod(shp))).reshape(shp)
    return meta, tensors


# ----------------------------------------------------------------------------
# tensor inventory
# ----------------------------------------------------------------------------

LAYER_PARTS = ("input_ln", "q_proj", "k_proj", "v_proj", "o_proj", "q_norm", "k_norm",
               "post_ln", "gate_proj", "up_proj", "down_proj")


def layer_shapes(cfg: ModelConfig, ffn: int) -> dict:
    H, D = cfg.hidden, cfg.head_dim
    return {"input_ln": (H,), "q_proj": (cfg.n_heads * D, H), "k_proj": (cfg.n_kv_heads * D, H),
            "v_proj": (cfg.n_kv_heads * D, H), "o_proj": (H, cfg.n_heads * D), "q_norm": (D,),
            "k_norm": (D,), "post_ln": (H,), "gate_proj": (ffn, H), "up_proj": (ffn, H),
            "down_proj": (H, ffn)}


def talker_tensor_shapes(cfg: ModelConfig) -> dict:
    s = {}
    for i in range(cfg.talker_layers):
        for p, shp in layer_shapes(cfg, cfg.talker_ffn).items():
            s[f"talker.layers.{i}.{p}"] = shp
    s["talker.norm"] = (cfg.hidden,)
    s["talker.codec_embedding"] = (cfg.talker_vocab, cfg.hidden)
    s["talker.codec_head"] = (cfg.talker_vocab, cfg.hidden)
    return s


def cp_tensor_shapes(cfg: ModelConfig) -> dict:
    s = {}
    for i in range(cfg.cp_layers):
        for p, shp in layer_shapes(cfg, cfg.cp_ffn).items():
            s[f"cp.layers.{i}.{p}"] = shp
    s["cp.norm"] = (cfg.hidden,)
    for g in range(cfg.cp_groups):
        s[f"cp.codec_emb.{g}"] = (cfg.cp_vocab, cfg.hidden)
        s[f"cp.lm_head.{g}"] = (cfg.cp_vocab, cfg.hidden)
    return s


def text_tensor_shapes(cfg: ModelConfig) -> dict:
    return {"text.embedding": (cfg.text_vocab, cfg.text_dim),
            "text.fc1.weight": (cfg.text_dim, cfg.text_dim), "text.fc1.bias": (cfg.text_dim,),
            "text.fc2.weight": (cfg.hidden, cfg.text_dim), "text.fc2.bias": (cfg.hidden,)}


def _rng_for(name: str, seed: int) -> np.random.Generator:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return np.random.default_rng(int.from_bytes(h[:8], "little"))


def _is_norm(name: str) -> bool:
    return name.endswith(("input_ln", "post_ln", "q_norm", "k_norm", ".norm"))


def synth_tensor(name: str, shape, seed: int, std: float = 0.02, table_dtype=np.float32):
    """Deterministic synthetic tensor: N(0, std^2) per tensor-name
    hash; norm weights 1 + N(0, 0.1^2) so the scale path is exercised.
    Projection matrices are stored fp16 (what the device streams); tables and
    norm vectors stay f32 (they are gathered / applied in f32)."""
    rng = _rng_for(name, seed)
    if _is_norm(name):
        return (1.0 + 0.1 * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
    if name.endswith("bias"):
        return (std * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
    a = std * rng.standard_normal(shape, dtype=np.float32)
    if "codec_emb" in name or name == "text.embedding":
        return a.astype(table_dtype)
    if name.startswith("text.fc"):
        return a.astype(np.float32)
    return a.astype(np.float16)


def make_synthetic(cfg: ModelConfig, seed: int = 1234, parts=("talker", "cp", "text"),
                   text_table_dtype=np.float32) -> dict:
    shapes = {}
    if "talker" in parts:
        shapes.update(talker_tensor_shapes(cfg))
    if "cp" in parts:
        shapes.update(cp_tensor_shapes(cfg))
    if "text" in parts:
        shapes.update(text_tensor_shapes(cfg))
    return {n: synth_tensor(n, shp, seed, table_dtype=text_table_dtype if n == "text.embedding" else np.float32)
            for n, shp in shapes.items()}


def write_synthetic(path: str, cfg: ModelConfig, seed: int = 1234, parts=("talker", "cp", "text"),
                    extra: dict | None = None, text_table_dtype=np.float32) -> None:
    t = make_synthetic(cfg, seed, parts, text_table_dtype)
    if extra:
        t.update(extra)
    write_pack(path, cfg.meta(), t)


# ----------------------------------------------------------------------------
# converters from the formats the reference deploys
# ----------------------------------------------------------------------------

def from_reference_dirs(embeddings_dir: str, cp_dir: str, talker_safetensors: str | None,
                        out_path: str, cfg: ModelConfig | None = None) -> None:
    """Pack the reference's deployed files (README.md:108-122):
    embeddings/{text_embedding,codec_embedding,codec_head,text_projection_*}.npy
    (scripts/extract_embeddings.py:47-72), code_predictor_weights.npz
    (scripts/export_code_predictor_weights.py:51-78) and the re-keyed talker
    safetensors (scripts/extract_talker_as_qwen3.py:53-71) into one container."""
    cfg = cfg or ModelConfig()
    t = {}
    e = lambda n: np.load(os.path.join(embeddings_dir, n))
    t["text.embedding"] = e("text_embedding.npy").astype(np.float32)
    t["text.fc1.weight"] = e("text_projection_linear_fc1_weight.npy").astype(np.float32)
    t["text.fc1.bias"] = e("text_projection_linear_fc1_bias.npy").astype(np.float32)
    t["text.fc2.weight"] = e("text_projection_linear_fc2_weight.npy").astype(np.float32)
    t["text.fc2.bias"] = e("text_projection_linear_fc2_bias.npy").astype(np.float32)
    t["talker.codec_embedding"] = e("codec_embedding.npy").astype(np.float32)
    t["talker.codec_head"] = e("codec_head.npy").astype(np.float16)
    w = np.load(os.path.join(cp_dir, "code_predictor_weights.npz"))
    for i in range(cfg.cp_layers):
        for p in LAYER_PARTS:
            a = w[f"layer_{i}_{p}"]
            t[f"cp.layers.{i}.{p}"] = a.astype(np.float32 if a.ndim == 1 else np.float16)
    t["cp.norm"] = w["final_norm"].astype(np.float32)
    for g in range(cfg.cp_groups):
        t[f"cp.codec_emb.{g}"] = w[f"codec_emb_{g}"].astype(np.float32)
        t[f"cp.lm_head.{g}"] = w[f"lm_head_{g}"].astype(np.float16)
    if talker_safetensors:
        t.update(_talker_from_safetensors(talker_safetensors, cfg, "model.layers.", "model.norm.weight"))
    write_pack(out_path, cfg.meta(), t)


_HF_PART = {"input_ln": "input_layernorm.weight", "q_proj": "self_attn.q_proj.weight",
            "k_proj": "self_attn.k_proj.weight", "v_proj": "self_attn.v_proj.weight",
            "o_proj": "self_attn.o_proj.weight", "q_norm": "self_attn.q_norm.weight",
            "k_norm": "self_attn.k_norm.weight", "post_ln": "post_attention_layernorm.weight",
            "gate_proj": "mlp.gate_proj.weight", "up_proj": "mlp.up_proj.weight",
            "down_proj": "mlp.down_proj.weight"}


def _talker_from_safetensors(path, cfg, layer_prefix, norm_key):
    from safetensors import safe_open
    t = {}
    with safe_open(path, framework="np") as f:
        for i in range(cfg.talker_layers):
            for p, hf in _HF_PART.items():
                a = f.get_tensor(f"{layer_prefix}{i}.{hf}")
                t[f"talker.layers.{i}.{p}"] = a.astype(np.float32 if a.ndim == 1 else np.float16)
        t["talker.norm"] = f.get_tensor(norm_key).astype(np.float32)
    return t


def from_hf_checkpoint(model_dir: str, out_path: str, cfg: ModelConfig | None = None) -> None:
    """Pack straight from the HF snapshot (Qwen/Qwen3-TTS-12Hz-0.6B-Base,
    model.safetensors) using the key names the reference's scripts read
    (scripts/extract_embeddings.py:47-98, export_code_predictor_weights.py:44-74)."""
    import torch  # bf16 tensors need torch to decode
    from safetensors.torch import load_file
    cfg = cfg or ModelConfig()
    w = load_file(os.path.join(model_dir, "model.safetensors"))
    extra = [k for k in w if "small_to_mtp_projection" in k]
    if extra:
        # first op of the code predictor (scripts/export_code_predictor_onnx.py:38-41): identity, without tensors, in the
        # 0.6 B model; a checkpoint that carries it needs an op this build does not have
        raise ValueError(f"{model_dir}: {extra[0]} present: only the identity talker -> code-predictor projection "
                         "(Qwen3-TTS 0.6B) is supported; refusing to ignore it")
    f32 = lambda k: w[k].float().numpy()
    t = {}
    for i in range(cfg.talker_layers):
        for p, hf in _HF_PART.items():
            a = f32(f"talker.model.layers.{i}.{hf}")
            t[f"talker.layers.{i}.{p}"] = a.astype(np.float32 if a.ndim == 1 else np.float16)
    t["talker.norm"] = f32("talker.model.norm.weight")
    t["talker.codec_embedding"] = f32("talker.model.codec_embedding.weight")
    t["talker.codec_head"] = f32("talker.codec_head.weight").astype(np.float16)
    for i in range(cfg.cp_layers):
        for p, hf in _HF_PART.items():
            a = f32(f"talker.code_predictor.model.layers.{i}.{hf}")
            t[f"cp.layers.{i}.{p}"] = a.astype(np.float32 if a.ndim == 1 else np.float16)
    t["cp.norm"] = f32("talker.code_predictor.model.norm.weight")
    for g in range(cfg.cp_groups):
        t[f"cp.codec_emb.{g}"] = f32(f"talker.code_predictor.model.codec_embedding.{g}.weight")
        t[f"cp.lm_head.{g}"] = f32(f"talker.code_predictor.lm_head.{g}.weight").astype(np.float16)
    t["text.embedding"] = f32("talker.model.text_embedding.weight")
    for fc in ("fc1", "fc2"):
        t[f"text.{fc}.weight"] = f32(f"talker.text_projection.linear_{fc}.weight")
        t[f"text.{fc}.bias"] = f32(f"talker.text_projection.linear_{fc}.bias")
    write_pack(out_path, cfg.meta(), t)


# ----------------------------------------------------------------------------
# vocoder program (include/qwen3tts_voc.h, csrc/q3_voc.hip)
# ----------------------------------------------------------------------------
VOP_RVQ, VOP_CONV, VOP_CONVT = 1, 2, 3
VOP_DWCONV, VOP_NORM, VOP_ATTN, VOP_GLU, VOP_EMBMEAN = 4, 5, 6, 7, 8
VF_SNAKE, VF_RES_ADD, VF_RES_SAVE, VF_CLAMP, VF_GELU = 1, 2, 4, 8, 16


@dataclass
class VocConfig:
    """Shape of the codec decoder.  The reference does not contain it; the
    defaults follow the published Qwen3-TTS-Tokenizer-12Hz decoder as far as it is known here:
    16 codebooks x 2048 entries (split RVQ: 1 semantic + 15 acoustic, dim 256 -> 512), causal conv
    to the latent width, two x2 transposed-conv upsamplers, then a BigVGAN-style stack with rates
    8,5,4,3 (2*2*8*5*4*3 = 1920 samples per frame) whose residual units use dilations 1,3,9 and
    Snake activations.  `pre_transformer_layers` > 0 inserts the published model's sliding-window
    transformer (input/output projections, pre-norm layers: RMSNorm -> q/k/v -> RoPE attention -> o ->
    residual, RMSNorm -> gated MLP -> residual; layer scales fold into the o / down weights at
    conversion) after the first conv, `convnext` a ConvNeXt block (depthwise k7 -> LayerNorm -> 4x
    pointwise -> GELU -> pointwise -> residual) after each x2 upsampler.  Both are ON by default (the whole
    published decoder is what the benchmark times); their hyper-parameters (8 layers, hidden 512, 16 x 64
    heads, FFN 1024, window 72) are RECOLLECTION, not reference ("Vocoder program") -- a real
    checkpoint's values come from its own config.json / tensor shapes through convert_speech_tokenizer().
    trunk_voc_config() is the convolutional trunk alone (round 1's timed table)."""
    n_q: int = 16
    codebook_size: int = 2048
    codebook_dim: int = 256
    rvq_out: int = 512
    latent: int = 1024
    pre_kernel: int = 3
    upsample_ratios: tuple = (2, 2)
    decoder_dim: int = 1536
    rates: tuple = (8, 5, 4, 3)
    dilations: tuple = (1, 3, 9)
    kernel: int = 7
    pre_transformer_layers: int = 8
    tf_hidden: int = 512
    tf_heads: int = 16
    tf_head_dim: int = 64
    tf_ffn: int = 1024
    tf_window: int = 72
    tf_rope_theta: int = 10000
    tf_eps_e9: int = 10000        # 1e-5
    convnext: bool = True
    convnext_kernel: int = 7
    convnext_eps_e9: int = 1000   # 1e-6
    # --- what the importable implementation of this decoder family pins (transformers' Qwen3OmniMoeCode2Wav,
    #     tests/golden/make_code2wav_golden.py) ---
    # ConvTranspose1d trim.  "both" (default): kernel - stride samples are cut at BOTH ends, as
    # Qwen3OmniMoeCausalTransConvNet does (left_pad = right_pad = k - s): a k = 2s block turns L columns into
    # (L - 1) * s and output column j reads inputs j // s and j // s + 1.  "right": the strictly causal form (rounds 1-2:
    # the first L * s outputs are kept).  The k = s upsamplers are the same under both.
    convt_trim: str = "both"
    front: str = "rvq"            # "rvq": split residual VQ de-quantiser; "embed": code_embedding(codes + q * size).mean(q)
    pre_conv: bool = True         # causal conv front -> latent (the Omni class has none)
    tf_proj: bool = True          # Linear input / output projections around the pre-transformer (the Omni class has none)
    tf_attn_bias: bool = False    # q/k/v/o biases (config.attention_bias)


def trunk_voc_config() -> VocConfig:
    """The convolutional trunk alone (no pre-transformer, no ConvNeXt blocks)."""
    return VocConfig(pre_transformer_layers=0, convnext=False)


def tiny_voc_config() -> VocConfig:
    return VocConfig(codebook_dim=32, rvq_out=64, latent=64, decoder_dim=128, pre_transformer_layers=0, convnext=False)


def tiny_full_voc_config() -> VocConfig:
    """Every op kind of the program at test size: transformer (window shorter than the chunk) + ConvNeXt."""
    return VocConfig(codebook_dim=32, rvq_out=64, latent=64, decoder_dim=128, pre_transformer_layers=2, tf_hidden=64,
                     tf_heads=4, tf_head_dim=16, tf_ffn=96, tf_window=24, convnext=True)


def convt_trims(vc: VocConfig, k: int, stride: int):
    """(left, right) samples cut from a ConvTranspose1d's (L - 1) * stride + k outputs."""
    if vc.convt_trim == "both":
        return k - stride, k - stride
    if vc.convt_trim == "right":
        return 0, k - stride
    raise ValueError(f"convt_trim must be 'both' or 'right', not {vc.convt_trim!r}")


def voc_program(vc: VocConfig):
    """-> list of (op, a, b, c, d, flags) rows + per-op tensor shapes."""
    prog, shapes = [], {}

    def add(row, tens):
        i = len(prog)
        prog.append(list(row) + [0] * (8 - len(row)))
        for n, shp in tens.items():
            shapes[f"voc.op{i}.{n}"] = shp

    if vc.front == "embed":
        add([VOP_EMBMEAN, vc.n_q, vc.codebook_size, vc.rvq_out], {"embedding": (vc.n_q * vc.codebook_size, vc.rvq_out)})
    else:
        add([VOP_RVQ, vc.n_q, vc.codebook_size, vc.codebook_dim, vc.rvq_out],
            {"codebook": (vc.n_q, vc.codebook_size, vc.codebook_dim), "proj_sem": (vc.rvq_out, vc.codebook_dim),
             "proj_ac": (vc.rvq_out, vc.codebook_dim)})

    def conv(cin, cout, k, dil, flags):
        t = {"weight": (cout, cin, k), "bias": (cout,)}
        if flags & VF_SNAKE:
            t.update({"alpha": (cin,), "beta": (cin,)})
        add([VOP_CONV, cin, cout, k, dil, flags], t)

    def convt(cin, cout, k, stride, flags):
        t = {"weight": (cin, cout, k), "bias": (cout,)}
        if flags & VF_SNAKE:
            t.update({"alpha": (cin,), "beta": (cin,)})
        lt, rt = convt_trims(vc, k, stride)
        add([VOP_CONVT, cin, cout, k, stride, flags, lt, rt], t)

    def linear(cin, cout, flags, bias=True):
        t = {"weight": (cout, cin, 1)}
        if bias:
            t["bias"] = (cout,)
        add([VOP_CONV, cin, cout, 1, 1, flags], t)

    def norm(c, kind, eps_e9, flags):
        t = {"weight": (c,)}
        if kind == 1:
            t["bias"] = (c,)
        add([VOP_NORM, c, c, kind, eps_e9, flags], t)

    if vc.pre_conv:
        conv(vc.rvq_out, vc.latent, vc.pre_kernel, 1, 0)
    elif vc.rvq_out != vc.latent:
        raise ValueError(f"without a pre-conv the front's width {vc.rvq_out} must be the latent width {vc.latent}")
    if vc.pre_transformer_layers > 0:
        H, nh, hd, F = vc.tf_hidden, vc.tf_heads, vc.tf_head_dim, vc.tf_ffn
        if vc.tf_proj:
            linear(vc.latent, H, 0)                                      # input projection
        elif H != vc.latent:
            raise ValueError(f"without projections the transformer width {H} must be the latent width {vc.latent}")
        for _ in range(vc.pre_transformer_layers):
            norm(H, 0, vc.tf_eps_e9, VF_RES_SAVE)
            linear(H, 3 * nh * hd, 0, bias=vc.tf_attn_bias)              # q | k | v, head-major
            add([VOP_ATTN, 3 * nh * hd, nh * hd, nh, hd, 0, vc.tf_window, vc.tf_rope_theta], {})
            linear(nh * hd, H, VF_RES_ADD, bias=vc.tf_attn_bias)         # o projection (x layer scale)
            norm(H, 0, vc.tf_eps_e9, VF_RES_SAVE)
            linear(H, 2 * F, 0, bias=False)                              # gate | up
            add([VOP_GLU, 2 * F, F, 0], {})                              # silu(gate) * up
            linear(F, H, VF_RES_ADD, bias=False)                         # down projection (x layer scale)
        norm(H, 0, vc.tf_eps_e9, 0)
        if vc.tf_proj:
            linear(H, vc.latent, 0)                                      # output projection
    for f in vc.upsample_ratios:
        convt(vc.latent, vc.latent, f, f, 0)
        if vc.convnext:
            c = vc.latent
            add([VOP_DWCONV, c, c, vc.convnext_kernel, 1, VF_RES_SAVE], {"weight": (c, 1, vc.convnext_kernel), "bias": (c,)})
            norm(c, 1, vc.convnext_eps_e9, 0)
            linear(c, 4 * c, 0)
            linear(4 * c, c, VF_GELU | VF_RES_ADD)                       # (x layer scale)
    conv(vc.latent, vc.decoder_dim, vc.kernel, 1, 0)
    c = vc.decoder_dim
    for r in vc.rates:
        convt(c, c // 2, 2 * r, r, VF_SNAKE)
        c //= 2
        for d in vc.dilations:
            conv(c, c, vc.kernel, d, VF_SNAKE | VF_RES_SAVE)
            conv(c, c, 1, 1, VF_SNAKE | VF_RES_ADD)
    conv(c, 1, vc.kernel, 1, VF_SNAKE | VF_CLAMP)
    return prog, shapes


def make_synthetic_voc(vc: VocConfig, seed: int = 1234) -> dict:
    prog, shapes = voc_program(vc)
    t = {"voc.program": np.asarray(prog, dtype=np.int32)}
    for n, shp in shapes.items():
        rng = _rng_for(n, seed)
        row = prog[int(n.split(".")[1][2:])]
        if row[0] == VOP_NORM:
            a = (1.0 + 0.1 * rng.standard_normal(shp, dtype=np.float32)) if n.endswith("weight") else \
                0.05 * rng.standard_normal(shp, dtype=np.float32)
        elif n.endswith(("alpha", "beta")):
            a = 0.3 * rng.standard_normal(shp, dtype=np.float32)
        elif n.endswith("bias"):
            a = 0.02 * rng.standard_normal(shp, dtype=np.float32)
        elif n.endswith("codebook"):
            a = 0.25 * rng.standard_normal(shp, dtype=np.float32)
        elif n.endswith("embedding"):
            a = rng.standard_normal(shp, dtype=np.float32)
        else:
            fan_in = int(np.prod(shp[1:])) if not n.endswith(("proj_sem", "proj_ac")) else shp[1]
            if ".weight" in n and len(shp) == 3 and prog[int(n.split(".")[1][2:])][0] == VOP_CONVT:
                fan_in = shp[0] * max(1, shp[2] // prog[int(n.split(".")[1][2:])][4])
            # gains < 1 keep the random stack contractive (activations O(1), like a trained decoder);
            # the 1x1 conv that closes a residual unit is damped further
            row = prog[int(n.split(".")[1][2:])]
            gain = 0.25 if (".weight" in n and row[0] == VOP_CONV and row[3] == 1 and (row[5] & VF_RES_ADD)) else 0.7
            if row[0] == VOP_CONV and row[3] == 1 and not (row[5] & (VF_RES_ADD | VF_SNAKE)):
                gain = 1.0   # plain projections (transformer q/k/v, gate/up, ConvNeXt expansion) keep the scale
            a = (gain * rng.standard_normal(shp, dtype=np.float32) / np.sqrt(fan_in)).astype(np.float32)
        t[n] = a.astype(np.float32)
    return t


def voc_total_upsample(vc: VocConfig) -> int:
    """decoder.total_upsample (scripts/export_vocoder_traced.py:46): the product of the rates -- the nominal samples
    per frame; what a chunk really yields is voc_chunk_samples()."""
    u = 1
    for f in tuple(vc.upsample_ratios) + tuple(vc.rates):
        u *= f
    return u


def voc_chunk_samples(vc: VocConfig, n_frames: int) -> int:
    """Samples one decode of n_frames frames returns: every transposed conv maps L -> (L - 1) * s + k - trims."""
    L = n_frames
    for f in vc.upsample_ratios:
        lt, rt = convt_trims(vc, f, f)
        L = (L - 1) * f + f - lt - rt
    for r in vc.rates:
        lt, rt = convt_trims(vc, 2 * r, r)
        L = (L - 1) * r + 2 * r - lt - rt
    return L


# ----------------------------------------------------------------------------
# vocoder table from a speech_tokenizer/ directory (config.json + *.safetensors)
# ----------------------------------------------------------------------------
# The reference traces `Qwen3TTSTokenizerV2Model.from_pretrained(<speech_tokenizer dir>).decoder`
# (scripts/export_vocoder_traced.py:74-79); neither the class nor a checkpoint is in the reference.  What pins the
# table since round 3: the importable implementation of this decoder family, transformers'
# `Qwen3OmniMoeCode2Wav` (pre_transformer / upsample / decoder module tree, SnakeBeta, causal convs, the trim of the
# transposed convs, ConvNeXt blocks, layer scales) and `MimiSplitResidualVectorQuantizer` (the split RVQ front):
# tests/golden/make_code2wav_golden.py runs both on seeded weights, tests/test_code2wav_golden.py maps their
# state_dict() keys through this converter and checks oracle/voc_ref.py (CPU) and voc_decode (GPU) against their
# outputs.  The tensor NAMES live in ONE table; a value may list alternatives (first match wins): the module tree of
# the Omni / Mimi classes as transformers names it, and the Qwen3-TTS-Tokenizer spelling of the quantiser / front
# (recollection: `quantizer.rvq_first / rvq_rest`, `pre_conv`, `pre_transformer.input_proj / output_proj`).  Every SIZE
# (codebooks, widths, kernel sizes, rates, layer counts) is read from the tensors' shapes; dilations and attention
# hyper-parameters that shapes cannot show come from config.json's "decoder_config" (fallback: the defaults of
# VocConfig, flagged in the report).
VOC_NAMES = {
    # split RVQ front ([size, dim] codebooks, or their EMA pair embed_sum / cluster_usage; projections [out, dim, 1])
    "codebook_first": ("decoder.quantizer.rvq_first.vq.layers.{i}._codebook.",
                       "decoder.quantizer.semantic_residual_vector_quantizer.layers.{i}.codebook."),
    "codebook_rest": ("decoder.quantizer.rvq_rest.vq.layers.{i}._codebook.",
                      "decoder.quantizer.acoustic_residual_vector_quantizer.layers.{i}.codebook."),
    "proj_first": ("decoder.quantizer.rvq_first.output_proj.weight",
                   "decoder.quantizer.semantic_residual_vector_quantizer.output_proj.weight"),
    "proj_rest": ("decoder.quantizer.rvq_rest.output_proj.weight",
                  "decoder.quantizer.acoustic_residual_vector_quantizer.output_proj.weight"),
    # embedding-mean front of the Omni class: [n_q * size, hidden]
    "code_embedding": "decoder.code_embedding.weight",
    "pre_conv": "decoder.pre_conv.conv",                                                 # .weight [latent, out, k] .bias
    "tf_in": "decoder.pre_transformer.input_proj", "tf_out": "decoder.pre_transformer.output_proj",
    "tf_norm": "decoder.pre_transformer.norm.weight",
    "tf_layer": "decoder.pre_transformer.layers.{i}.",   # + input_layernorm.weight, self_attn.{q,k,v,o}_proj.weight,
                                                         #   self_attn_layer_scale.scale, post_attention_layernorm.weight,
                                                         #   mlp.{gate,up,down}_proj.weight, mlp_layer_scale.scale
    "up_convt": "decoder.upsample.{i}.0.conv",                                            # ConvTranspose1d [cin, cout, k]
    "up_next": "decoder.upsample.{i}.1.",                # + dwconv.conv.{weight,bias}, norm.{weight,bias}, pwconv1.*, pwconv2.*, gamma
    "dec_in": "decoder.decoder.0.conv",
    "dec_block": "decoder.decoder.{b}.block.",           # b = 1..; + 0.{alpha,beta} (Snake), 1.conv (ConvTranspose1d),
                                                         #   {2+u}.act1.{alpha,beta}, .conv1.conv, .act2.{alpha,beta}, .conv2.conv
    "dec_out_act": "decoder.decoder.{b}.", "dec_out": "decoder.decoder.{b}.conv",
}


def _load_safetensors_dir(src_dir: str) -> dict:
    from safetensors import safe_open
    out = {}
    for fn in sorted(os.listdir(src_dir)):
        if fn.endswith(".safetensors"):
            with safe_open(os.path.join(src_dir, fn), framework="np") as f:
                for k in f.keys():
                    out[k] = f.get_tensor(k)
    return out


def speech_tokenizer_to_voc(src_dir: str, names: dict | None = None, convt_trim: str | None = None):
    """-> (VocConfig, tensors dict with voc.program + voc.op*.*, report lines).  See VOC_NAMES."""
    import json
    T = _load_safetensors_dir(src_dir)
    if not T:
        raise FileNotFoundError(f"no *.safetensors under {src_dir}")
    cfgj = {}
    cj = os.path.join(src_dir, "config.json")
    if os.path.exists(cj):
        with open(cj) as f:
            j = json.load(f)
        cfgj = j.get("decoder_config", j)
    return state_to_voc(T, cfgj, names, convt_trim, where=src_dir)


def state_to_voc(T: dict, cfgj: dict | None = None, names: dict | None = None, convt_trim: str | None = None,
                 where: str = "state dict"):
    """A decoder state dict (numpy arrays under the checkpoint's names) + its config -> (VocConfig, table tensors,
    report).  `convt_trim` overrides config.json's "convt_trim" ("both", the default = the Omni class's trim of the
    transposed convs, or "right" = strictly causal; VocConfig)."""
    cfgj = dict(cfgj or {})
    N = dict(VOC_NAMES, **(names or {}))
    report = []
    f32 = lambda a: np.ascontiguousarray(np.asarray(a), dtype=np.float32)
    has = lambda k: k in T
    used = set()
    # a decoder.* module this table does not know (renamed, or a component this build has no op for) must not be
    # dropped silently
    known = {a.split(".")[1] for v in N.values() for a in (v if isinstance(v, (tuple, list)) else (v,)) if a.startswith("decoder.")}
    strange = sorted({k.split(".")[1] for k in T if k.startswith("decoder.") and k.split(".")[1] not in known})
    if strange:
        raise KeyError(f"{where}: decoder modules {strange} are not in weights.VOC_NAMES (edit that table if the checkpoint "
                       f"names things differently)")

    def pick(key, **kw):
        """the alternative of VOC_NAMES[key] that the checkpoint uses (first whose formatted prefix matches a tensor)"""
        alts = N[key] if isinstance(N[key], (tuple, list)) else (N[key],)
        for a in alts:
            pre = a.format(**kw) if kw else a.split("{")[0]
            if any(k.startswith(pre) for k in T):
                return a
        return alts[0]

    def need(k):
        if k not in T:
            raise KeyError(f"{where}: tensor {k} not found (edit weights.VOC_NAMES if the checkpoint names it differently)")
        used.add(k)
        return f32(T[k])

    def codebook(base):
        for nm in ("embed", "embedding"):
            if has(base + nm):
                return need(base + nm)
        for nm in ("embed_sum", "embedding_sum"):
            if has(base + nm) and has(base + "cluster_usage"):   # EMA form: embed = embed_sum / usage.clamp(min=1e-5)
                return need(base + nm) / np.maximum(need(base + "cluster_usage"), 1e-5)[:, None]
        raise KeyError(f"{where}: codebook {base}embed (or its embed_sum / cluster_usage pair) not found "
                       f"(edit weights.VOC_NAMES if the checkpoint names it differently)")

    def count(pattern, **kw):
        n = 0
        while any(k.startswith(pattern.format(i=n, **kw)) for k in T):
            n += 1
        return n

    vc = VocConfig()
    vc.convt_trim = convt_trim or str(cfgj.get("convt_trim", vc.convt_trim))
    # ---- front ----
    emb_key = pick("code_embedding")
    vc.front = "embed" if has(emb_key) else "rvq"
    if vc.front == "embed":
        emb = need(emb_key)
        vc.n_q = int(cfgj.get("num_quantizers", 16))
        if emb.shape[0] % vc.n_q:
            raise ValueError(f"{where}: code_embedding has {emb.shape[0]} rows, not a multiple of num_quantizers={vc.n_q}")
        vc.codebook_size, vc.rvq_out, vc.codebook_dim = emb.shape[0] // vc.n_q, emb.shape[1], emb.shape[1]
        if "num_quantizers" not in cfgj:
            report.append(f"config.json has no num_quantizers: assuming {vc.n_q}")
    else:
        cb_first, cb_rest = pick("codebook_first", i=0), pick("codebook_rest", i=0)
        n_rest = count(cb_rest)
        cb0 = codebook(cb_first.format(i=0))
        cbs = [cb0] + [codebook(cb_rest.format(i=i)) for i in range(n_rest)]
        proj_sem, proj_ac = need(pick("proj_first")), need(pick("proj_rest"))
        vc.n_q, vc.codebook_size, vc.codebook_dim = 1 + n_rest, cb0.shape[0], cb0.shape[1]
        vc.rvq_out = proj_sem.shape[0]
    vc.pre_conv = has(N["pre_conv"] + ".weight")
    if vc.pre_conv:
        pre_w = need(N["pre_conv"] + ".weight")
        vc.latent, vc.pre_kernel = pre_w.shape[0], pre_w.shape[2]
    else:
        vc.latent = vc.rvq_out
    n_tf = count(N["tf_layer"])
    n_up = count(N["up_convt"].split("{i}")[0] + "{i}.")
    dec_in_w = need(N["dec_in"] + ".weight")
    blocks = []
    b = 1
    while any(k.startswith(N["dec_block"].format(b=b)) for k in T):
        blocks.append(b)
        b += 1
    vc.pre_transformer_layers = n_tf
    kv_rep = 1
    if n_tf:
        lp = N["tf_layer"].format(i=0)
        vc.tf_proj = has(N["tf_in"] + ".weight")
        q_w = need(lp + "self_attn.q_proj.weight")
        vc.tf_hidden = q_w.shape[1]
        qd, kd = q_w.shape[0], need(lp + "self_attn.k_proj.weight").shape[0]
        if "head_dim" in cfgj or "attention_head_dim" in cfgj:
            vc.tf_head_dim = int(cfgj.get("head_dim", cfgj.get("attention_head_dim")))
        elif "num_attention_heads" in cfgj:
            vc.tf_head_dim = qd // int(cfgj["num_attention_heads"])
        else:
            report.append(f"config.json has neither head_dim nor num_attention_heads: using head_dim {vc.tf_head_dim}")
        vc.tf_heads = qd // vc.tf_head_dim
        if qd % vc.tf_head_dim or kd % vc.tf_head_dim or qd % kd:
            raise ValueError(f"{where}: q/k projections of {qd}/{kd} rows do not split into heads of {vc.tf_head_dim}")
        kv_rep = qd // kd        # grouped-query attention: every k/v head serves kv_rep query heads
        vc.tf_ffn = need(lp + "mlp.gate_proj.weight").shape[0]
        vc.tf_attn_bias = has(lp + "self_attn.q_proj.bias")
        if "sliding_window" in cfgj:
            vc.tf_window = int(cfgj["sliding_window"])
        else:
            report.append(f"config.json has no sliding_window: using the default {vc.tf_window}")
        rp = cfgj.get("rope_parameters") or {}
        if "rope_theta" in cfgj or "rope_theta" in rp:
            vc.tf_rope_theta = int(cfgj.get("rope_theta", rp.get("rope_theta")))
        else:
            report.append(f"config.json has no rope_theta: using the default {vc.tf_rope_theta}")
        if "rms_norm_eps" in cfgj:
            vc.tf_eps_e9 = int(round(float(cfgj["rms_norm_eps"]) * 1e9))
    ups = [need(N["up_convt"].format(i=i) + ".weight") for i in range(n_up)]
    vc.upsample_ratios = tuple(int(w.shape[2]) for w in ups)          # kernel = stride in these upsamplers
    vc.convnext = n_up > 0 and any(k.startswith(N["up_next"].format(i=0)) for k in T)
    if vc.convnext:
        vc.convnext_kernel = need(N["up_next"].format(i=0) + "dwconv.conv.weight").shape[2]
    vc.decoder_dim, vc.kernel = dec_in_w.shape[0], dec_in_w.shape[2]
    rates, n_units = [], 0
    for b in blocks:
        w = need(N["dec_block"].format(b=b) + "1.conv.weight")          # ConvTranspose1d [cin, cout, 2r]
        rates.append(int(w.shape[2]) // 2)
        u = 0
        while any(k.startswith(N["dec_block"].format(b=b) + f"{2 + u}.") for k in T):
            u += 1
        n_units = u
    vc.rates = tuple(rates)
    dil = cfgj.get("dilations", cfgj.get("residual_dilations"))
    if dil is None:
        dil = [3 ** i for i in range(n_units)]
        report.append(f"config.json has no dilations: assuming {dil} for the {n_units} residual units of a block")
    vc.dilations = tuple(int(d) for d in dil)
    prog, shapes = voc_program(vc)
    # ---- tensors, in program order ----
    out = {"voc.program": np.asarray(prog, dtype=np.int32)}
    flat1 = lambda a: f32(a).reshape(-1)
    lin = lambda w: f32(w).reshape(w.shape[0], -1, 1) if w.ndim == 2 else f32(w)     # Linear [out, in] -> conv [out, in, 1]
    it = iter(range(len(prog)))

    def put(i, **tens):
        for n, a in tens.items():
            want = shapes.get(f"voc.op{i}.{n}")
            if want is None or tuple(a.shape) != tuple(want):
                raise ValueError(f"voc.op{i}.{n}: checkpoint tensor has shape {tuple(a.shape)}, the table expects {want}")
            out[f"voc.op{i}.{n}"] = np.ascontiguousarray(a, dtype=np.float32)

    i = next(it)
    if vc.front == "embed":
        put(i, embedding=emb)
    else:
        put(i, codebook=np.stack(cbs), proj_sem=proj_sem.reshape(proj_sem.shape[0], -1), proj_ac=proj_ac.reshape(proj_ac.shape[0], -1))
    if vc.pre_conv:
        put(next(it), weight=pre_w, bias=need(N["pre_conv"] + ".bias"))
    if n_tf:
        if vc.tf_proj:
            put(next(it), weight=lin(need(N["tf_in"] + ".weight")), bias=need(N["tf_in"] + ".bias"))
        hd = vc.tf_head_dim

        def kv_heads(a):     # [kv_heads * hd, ...] -> [heads * hd, ...]: query head h reads k/v head h // kv_rep
            if kv_rep == 1:
                return a
            return np.repeat(a.reshape((a.shape[0] // hd, hd) + a.shape[1:]), kv_rep, axis=0).reshape((-1,) + a.shape[1:])

        for l in range(n_tf):
            lp = N["tf_layer"].format(i=l)
            sa = flat1(need(lp + "self_attn_layer_scale.scale")) if has(lp + "self_attn_layer_scale.scale") else None
            sm = flat1(need(lp + "mlp_layer_scale.scale")) if has(lp + "mlp_layer_scale.scale") else None
            put(next(it), weight=need(lp + "input_layernorm.weight"))
            qkv = dict(weight=lin(np.concatenate([need(lp + "self_attn.q_proj.weight"), kv_heads(need(lp + "self_attn.k_proj.weight")),
                                                  kv_heads(need(lp + "self_attn.v_proj.weight"))], 0)))
            if vc.tf_attn_bias:
                qkv["bias"] = np.concatenate([need(lp + "self_attn.q_proj.bias"), kv_heads(need(lp + "self_attn.k_proj.bias")),
                                              kv_heads(need(lp + "self_attn.v_proj.bias"))], 0)
            put(next(it), **qkv)
            next(it)                                                            # attention: no tensors
            o = need(lp + "self_attn.o_proj.weight")
            ot = dict(weight=lin(o * sa[:, None] if sa is not None else o))     # layer scale folds into the rows
            if vc.tf_attn_bias:
                ob = need(lp + "self_attn.o_proj.bias")
                ot["bias"] = ob * sa if sa is not None else ob
            put(next(it), **ot)
            put(next(it), weight=need(lp + "post_attention_layernorm.weight"))
            put(next(it), weight=lin(np.concatenate([need(lp + "mlp.gate_proj.weight"), need(lp + "mlp.up_proj.weight")], 0)))
            next(it)                                                            # GLU: no tensors
            d = need(lp + "mlp.down_proj.weight")
            put(next(it), weight=lin(d * sm[:, None] if sm is not None else d))
        put(next(it), weight=need(N["tf_norm"]))
        if vc.tf_proj:
            put(next(it), weight=lin(need(N["tf_out"] + ".weight")), bias=need(N["tf_out"] + ".bias"))
    for u in range(n_up):
        put(next(it), weight=ups[u], bias=need(N["up_convt"].format(i=u) + ".bias"))
        if vc.convnext:
            p = N["up_next"].format(i=u)
            g = flat1(need(p + "gamma")) if has(p + "gamma") else None
            put(next(it), weight=need(p + "dwconv.conv.weight"), bias=need(p + "dwconv.conv.bias"))
            put(next(it), weight=need(p + "norm.weight"), bias=need(p + "norm.bias"))
            put(next(it), weight=lin(need(p + "pwconv1.weight")), bias=need(p + "pwconv1.bias"))
            w2, b2 = need(p + "pwconv2.weight"), need(p + "pwconv2.bias")
            put(next(it), weight=lin(w2 * g[:, None] if g is not None else w2), bias=b2 * g if g is not None else b2)
    put(next(it), weight=dec_in_w, bias=need(N["dec_in"] + ".bias"))
    snake = lambda p: dict(alpha=flat1(need(p + "alpha")), beta=flat1(need(p + "beta")))
    for b in blocks:
        bp = N["dec_block"].format(b=b)
        put(next(it), weight=need(bp + "1.conv.weight"), bias=need(bp + "1.conv.bias"), **snake(bp + "0."))
        for u in range(n_units):
            up_ = bp + f"{2 + u}."
            put(next(it), weight=need(up_ + "conv1.conv.weight"), bias=need(up_ + "conv1.conv.bias"), **snake(up_ + "act1."))
            put(next(it), weight=need(up_ + "conv2.conv.weight"), bias=need(up_ + "conv2.conv.bias"), **snake(up_ + "act2."))
    ob = len(blocks) + 1
    put(next(it), weight=need(N["dec_out"].format(b=ob + 1) + ".weight"), bias=need(N["dec_out"].format(b=ob + 1) + ".bias"),
        **snake(N["dec_out_act"].format(b=ob)))
    missing = [n for n in shapes if n not in out]
    if missing:
        raise ValueError(f"table tensors without a source: {missing[:5]}")
    unused = sorted(k for k in T if k.startswith("decoder.") and k not in used and
                    not k.endswith(("initialized", "_initialized", "rotary_emb.inv_freq")) and
                    not (".quantizer." in k and ".input_proj." in k))      # the quantiser's encode-side projection
    if unused:
        raise KeyError(f"{where}: {len(unused)} decoder tensors have no place in the table (first: {unused[:4]}); a component "
                       f"this build has no op for, or a name weights.VOC_NAMES lacks")
    front = (f"{vc.n_q} x {vc.codebook_size} embedding rows averaged -> {vc.rvq_out}" if vc.front == "embed" else
             f"{vc.n_q} codebooks x {vc.codebook_size} x {vc.codebook_dim} -> {vc.rvq_out}")
    report.insert(0, f"{len(prog)} ops: {front}, latent {vc.latent}{'' if vc.pre_conv else ' (no pre-conv)'}, "
                     f"{n_tf} transformer layers (hidden {vc.tf_hidden}, {vc.tf_heads} x {vc.tf_head_dim}"
                     f"{f' from {vc.tf_heads // kv_rep} k/v heads' if kv_rep > 1 else ''}, ffn {vc.tf_ffn}"
                     f"{'' if vc.tf_proj or not n_tf else ', no in/out projections'}), "
                     f"upsample {vc.upsample_ratios}{' + ConvNeXt' if vc.convnext else ''}, decoder {vc.decoder_dim} rates {vc.rates} "
                     f"dilations {vc.dilations} k{vc.kernel}: x{voc_total_upsample(vc)} samples per frame nominal, "
                     f"transposed convs trimmed '{vc.convt_trim}' ({voc_chunk_samples(vc, 64)} samples per 64 frames)")
    return vc, out, report


def convert_speech_tokenizer(src_dir: str, out_path: str, chunk: int = 64, names: dict | None = None,
                             convt_trim: str | None = None):
    """speech_tokenizer/ (config.json + safetensors) -> vocoder container for voc_load()."""
    vc, t, report = speech_tokenizer_to_voc(src_dir, names, convt_trim)
    write_pack(out_path, {"voc_chunk": float(chunk)}, t)
    return vc, report


def export_speech_tokenizer_layout(tensors: dict, vc: VocConfig, dst_dir: str, layer_scales: bool = True, seed: int = 3):
    """Inverse of speech_tokenizer_to_voc for tests: write a table's tensors as a speech_tokenizer/ directory
    (VOC_NAMES layout: separate q/k/v and gate/up matrices, Linear weights 2-D, layer scales and ConvNeXt gamma as
    their own tensors -- divided out of the folded weights -- and an EMA-form first codebook)."""
    import json
    from safetensors.numpy import save_file
    N = {k: (v[0] if isinstance(v, tuple) else v) for k, v in VOC_NAMES.items()}
    prog, _ = voc_program(vc)
    rng = np.random.default_rng(seed)
    out = {}
    g = lambda i, n: np.asarray(tensors[f"voc.op{i}.{n}"], dtype=np.float32)
    it = iter(range(len(prog)))
    i = next(it)
    if vc.front == "embed":
        out[N["code_embedding"]] = g(i, "embedding")
    else:
        cb = g(i, "codebook")
        usage = (1.0 + rng.random(cb.shape[1])).astype(np.float32)
        base = N["codebook_first"].format(i=0)
        out[base + "embed_sum"] = cb[0] * usage[:, None]
        out[base + "cluster_usage"] = usage
        for q in range(1, cb.shape[0]):
            out[N["codebook_rest"].format(i=q - 1) + "embed"] = cb[q]
        out[N["proj_first"]] = g(i, "proj_sem")[:, :, None]
        out[N["proj_rest"]] = g(i, "proj_ac")[:, :, None]
    if vc.pre_conv:
        i = next(it)
        out[N["pre_conv"] + ".weight"], out[N["pre_conv"] + ".bias"] = g(i, "weight"), g(i, "bias")
    if vc.pre_transformer_layers:
        H, qd, F = vc.tf_hidden, vc.tf_heads * vc.tf_head_dim, vc.tf_ffn
        if vc.tf_proj:
            i = next(it)
            out[N["tf_in"] + ".weight"], out[N["tf_in"] + ".bias"] = g(i, "weight")[:, :, 0], g(i, "bias")
        for l in range(vc.pre_transformer_layers):
            lp = N["tf_layer"].format(i=l)
            out[lp + "input_layernorm.weight"] = g(next(it), "weight")
            qkv = g(next(it), "weight")[:, :, 0]
            for j, x in enumerate("qkv"):
                out[lp + f"self_attn.{x}_proj.weight"] = qkv[j * qd:(j + 1) * qd]
            next(it)
            o = g(next(it), "weight")[:, :, 0]
            sa = (0.5 + rng.random(H)).astype(np.float32) if layer_scales else np.ones(H, np.float32)
            out[lp + "self_attn.o_proj.weight"] = o / sa[:, None]
            out[lp + "post_attention_layernorm.weight"] = g(next(it), "weight")
            gu = g(next(it), "weight")[:, :, 0]
            out[lp + "mlp.gate_proj.weight"], out[lp + "mlp.up_proj.weight"] = gu[:F], gu[F:]
            next(it)
            d = g(next(it), "weight")[:, :, 0]
            sm = (0.5 + rng.random(H)).astype(np.float32) if layer_scales else np.ones(H, np.float32)
            out[lp + "mlp.down_proj.weight"] = d / sm[:, None]
            if layer_scales:
                out[lp + "self_attn_layer_scale.scale"], out[lp + "mlp_layer_scale.scale"] = sa, sm
        out[N["tf_norm"]] = g(next(it), "weight")
        if vc.tf_proj:
            i = next(it)
            out[N["tf_out"] + ".weight"], out[N["tf_out"] + ".bias"] = g(i, "weight")[:, :, 0], g(i, "bias")
    for u in range(len(vc.upsample_ratios)):
        i = next(it)
        out[N["up_convt"].format(i=u) + ".weight"], out[N["up_convt"].format(i=u) + ".bias"] = g(i, "weight"), g(i, "bias")
        if vc.convnext:
            p = N["up_next"].format(i=u)
            i = next(it)
            out[p + "dwconv.conv.weight"], out[p + "dwconv.conv.bias"] = g(i, "weight"), g(i, "bias")
            i = next(it)
            out[p + "norm.weight"], out[p + "norm.bias"] = g(i, "weight"), g(i, "bias")
            i = next(it)
            out[p + "pwconv1.weight"], out[p + "pwconv1.bias"] = g(i, "weight")[:, :, 0], g(i, "bias")
            i = next(it)
            gam = (0.5 + rng.random(vc.latent)).astype(np.float32)
            out[p + "pwconv2.weight"], out[p + "pwconv2.bias"] = g(i, "weight")[:, :, 0] / gam[:, None], g(i, "bias") / gam
            out[p + "gamma"] = gam
    i = next(it)
    out[N["dec_in"] + ".weight"], out[N["dec_in"] + ".bias"] = g(i, "weight"), g(i, "bias")
    for bi in range(len(vc.rates)):
        bp = N["dec_block"].format(b=bi + 1)
        i = next(it)
        out[bp + "0.alpha"], out[bp + "0.beta"] = g(i, "alpha")[None, :, None], g(i, "beta")[None, :, None]   # [1, C, 1] parameters
        out[bp + "1.conv.weight"], out[bp + "1.conv.bias"] = g(i, "weight"), g(i, "bias")
        for u in range(len(vc.dilations)):
            up_ = bp + f"{2 + u}."
            for cv, act in (("conv1", "act1"), ("conv2", "act2")):
                i = next(it)
                out[up_ + act + ".alpha"], out[up_ + act + ".beta"] = g(i, "alpha"), g(i, "beta")
                out[up_ + cv + ".conv.weight"], out[up_ + cv + ".conv.bias"] = g(i, "weight"), g(i, "bias")
    ob = len(vc.rates) + 1
    i = next(it)
    out[N["dec_out_act"].format(b=ob) + "alpha"], out[N["dec_out_act"].format(b=ob) + "beta"] = g(i, "alpha"), g(i, "beta")
    out[N["dec_out"].format(b=ob + 1) + ".weight"], out[N["dec_out"].format(b=ob + 1) + ".bias"] = g(i, "weight"), g(i, "bias")
    os.makedirs(dst_dir, exist_ok=True)
    save_file({k: np.ascontiguousarray(v) for k, v in out.items()}, os.path.join(dst_dir, "model.safetensors"))
    with open(os.path.join(dst_dir, "config.json"), "w") as f:
        json.dump({"model_type": "qwen3_tts_tokenizer_12hz", "decoder_config": {
            "head_dim": vc.tf_head_dim, "sliding_window": vc.tf_window, "rope_theta": vc.tf_rope_theta,
            "rms_norm_eps": vc.tf_eps_e9 * 1e-9, "dilations": list(vc.dilations), "num_quantizers": vc.n_q,
            "convt_trim": vc.convt_trim}}, f)
