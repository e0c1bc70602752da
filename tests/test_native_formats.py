"""Native readers of the reference's on-disk formats (csrc/q3_formats.cpp): .npy (v1/v2, f4/f8/f2), .npz
(stored as scripts/export_code_predictor_weights.py:76 writes it, deflated as scripts/extract_embeddings.py:93
does), .safetensors (F32/F16/BF16) and the reference-key -> container-name maps.  Files are written here by
numpy / safetensors themselves and read back through the C++ loader's own entry point (no GPU call)."""
import ctypes
import os

import numpy as np
import pytest

from qwen3_tts_axera_russian_amd import hiplib

PARTS = ["input_ln", "q_proj", "k_proj", "v_proj", "o_proj", "q_norm", "k_norm", "post_ln", "gate_proj",
         "up_proj", "down_proj"]


def fnv1a(b: bytes) -> int:
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def inspect(path, aux=None):
    lib = hiplib.load_test()
    buf = ctypes.create_string_buffer(1 << 20)
    n = lib.q3t_inspect_weights(str(path).encode(), str(aux).encode() if aux else None, buf, len(buf))
    if n < 0:
        return None, None
    tens, meta = {}, {}
    for line in buf.value.decode().splitlines():
        f = line.split()
        if f[0] == "meta":
            meta[f[1]] = float(f[2])
        else:
            tens[f[0]] = (int(f[1]), tuple(int(x) for x in f[3:3 + int(f[2])]), int(f[7], 16))
    assert len(tens) == n
    return tens, meta


def cp_arrays(rng, layers=2, groups=3, dtype=np.float32):
    w = {}
    for i in range(layers):
        for p in PARTS:
            shape = (8,) if p.endswith("ln") or p.endswith("norm") else (6, 8)
            w[f"layer_{i}_{p}"] = rng.standard_normal(shape).astype(dtype)
    w["final_norm"] = rng.standard_normal(8).astype(dtype)
    for g in range(groups):
        w[f"codec_emb_{g}"] = rng.standard_normal((5, 8)).astype(dtype)
        w[f"lm_head_{g}"] = rng.standard_normal((5, 8)).astype(dtype)
    return w


@pytest.mark.parametrize("compressed", [False, True])
def test_reference_cp_directory(tmp_path, compressed):
    """--model_dir/code_predictor_weights.npz + --embeddings_dir/codec_embedding.npy (code_predictor_server.py:43-51)."""
    rng = np.random.default_rng(1)
    w = cp_arrays(rng)
    w["unrelated_key"] = np.zeros(3, np.float32)
    model_dir, emb_dir = tmp_path / "cp", tmp_path / "emb"
    model_dir.mkdir()
    emb_dir.mkdir()
    (np.savez_compressed if compressed else np.savez)(model_dir / "code_predictor_weights.npz", **w)
    table = rng.standard_normal((3072, 8)).astype(np.float32)
    np.save(emb_dir / "codec_embedding.npy", table)
    tens, meta = inspect(model_dir, emb_dir)
    assert tens is not None
    assert meta["cp_layers"] == 2 and meta["cp_groups"] == 3 and meta["cp_ffn"] == 6
    want = {"talker.codec_embedding": table, "cp.norm": w["final_norm"]}
    for i in range(2):
        for p in PARTS:
            want[f"cp.layers.{i}.{p}"] = w[f"layer_{i}_{p}"]
    for g in range(3):
        want[f"cp.codec_emb.{g}"] = w[f"codec_emb_{g}"]
        want[f"cp.lm_head.{g}"] = w[f"lm_head_{g}"]
    assert set(tens) == set(want)
    for k, a in want.items():
        assert tens[k] == (0, a.shape, fnv1a(a.tobytes())), k


def test_npy_variants(tmp_path):
    rng = np.random.default_rng(2)
    d = tmp_path / "d"
    d.mkdir()
    np.savez(d / "code_predictor_weights.npz", **cp_arrays(rng, 1, 1))
    a64 = rng.standard_normal((3072, 4))                       # f8 is converted to f4, as npy_reader.h does
    np.save(d / "codec_embedding.npy", a64)
    a16 = rng.standard_normal((3072, 4)).astype(np.float16)
    with open(d / "codec_head.npy", "wb") as f:                # header version 2.0
        np.lib.format.write_array(f, a16, version=(2, 0))
    tens, _ = inspect(d)
    assert tens["talker.codec_embedding"] == (0, (3072, 4), fnv1a(a64.astype(np.float32).tobytes()))
    assert tens["talker.codec_head"] == (1, (3072, 4), fnv1a(a16.tobytes()))


@pytest.mark.parametrize("bad", ["fortran", "dtype", "truncated", "notzip"])
def test_malformed_inputs_are_refused(tmp_path, bad):
    d = tmp_path / "d"
    d.mkdir()
    rng = np.random.default_rng(3)
    if bad == "notzip":
        (d / "code_predictor_weights.npz").write_bytes(b"PK\x03\x04 definitely not a zip" * 4)
    else:
        np.savez(d / "code_predictor_weights.npz", **cp_arrays(rng, 1, 1))
        a = rng.standard_normal((3072, 4)).astype(np.float32)
        if bad == "fortran":
            np.save(d / "codec_embedding.npy", np.asfortranarray(a))
        elif bad == "dtype":
            np.save(d / "codec_embedding.npy", a.astype(np.complex64))
        else:
            np.save(d / "codec_embedding.npy", a)
            raw = (d / "codec_embedding.npy").read_bytes()
            (d / "codec_embedding.npy").write_bytes(raw[: len(raw) // 2])
    tens, _ = inspect(d)
    assert tens is None


def test_safetensors_hf_and_rekeyed_keys(tmp_path):
    """HF snapshot keys (scripts/extract_embeddings.py:47-98) and the re-keyed Qwen3 talker
    (scripts/extract_talker_as_qwen3.py:53-71, tables padded to the text vocabulary) incl. BF16."""
    torch = pytest.importorskip("torch")
    from safetensors.torch import save_file
    hf = {"input_ln": "input_layernorm.weight", "q_proj": "self_attn.q_proj.weight", "k_proj": "self_attn.k_proj.weight",
          "v_proj": "self_attn.v_proj.weight", "o_proj": "self_attn.o_proj.weight", "q_norm": "self_attn.q_norm.weight",
          "k_norm": "self_attn.k_norm.weight", "post_ln": "post_attention_layernorm.weight",
          "gate_proj": "mlp.gate_proj.weight", "up_proj": "mlp.up_proj.weight", "down_proj": "mlp.down_proj.weight"}
    g = torch.Generator().manual_seed(4)
    t, want = {}, {}

    def put(key, name, shape, dtype=torch.bfloat16):
        x = torch.randn(shape, generator=g).to(dtype)
        t[key] = x
        want[name] = x
    for i in range(2):
        for p, k in hf.items():
            shape = (8,) if p.endswith("ln") or p.endswith("norm") else (6, 8)
            put(f"talker.model.layers.{i}.{k}", f"talker.layers.{i}.{p}", shape)
    for p, k in hf.items():
        put(f"talker.code_predictor.model.layers.0.{k}", f"cp.layers.0.{p}", (8,) if "ln" in p or "norm" in p else (6, 8),
            torch.float16)
    put("talker.model.norm.weight", "talker.norm", (8,), torch.float32)
    put("talker.code_predictor.model.norm.weight", "cp.norm", (8,))
    put("talker.model.codec_embedding.weight", "talker.codec_embedding", (3072, 8))
    put("talker.codec_head.weight", "talker.codec_head", (3072, 8))
    put("talker.code_predictor.model.codec_embedding.0.weight", "cp.codec_emb.0", (5, 8))
    put("talker.code_predictor.lm_head.0.weight", "cp.lm_head.0", (5, 8))
    t["speaker_encoder.whatever"] = torch.zeros(3)
    snap = tmp_path / "snap"
    snap.mkdir()
    save_file(t, str(snap / "model.safetensors"), metadata={"format": "pt", "note": "a {nested} \"quoted\" value"})
    tens, meta = inspect(snap)
    assert tens is not None and set(tens) == set(want)
    code = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 4}
    for k, x in want.items():
        raw = x.view(torch.int16).numpy().tobytes() if x.dtype != torch.float32 else x.numpy().tobytes()
        assert tens[k] == (code[x.dtype], tuple(x.shape), fnv1a(raw)), k
    assert meta["talker_layers"] == 2 and meta["cp_layers"] == 1 and meta["talker_ffn"] == 6
    # the re-keyed talker: model.layers.*, tables padded to 151936 rows -> only the 3072 codec rows are kept
    rk = {f"model.layers.0.{k}": torch.randn((8,) if "norm" in k else (6, 8), generator=g) for k in hf.values()}
    emb = torch.randn((4000, 8), generator=g)
    rk["model.embed_tokens.weight"] = emb
    rk["model.norm.weight"] = torch.randn(8, generator=g)
    f = tmp_path / "talker_qwen3.safetensors"
    save_file(rk, str(f))
    tens, meta = inspect(f)
    assert tens["talker.codec_embedding"] == (0, (3072, 8), fnv1a(emb[:3072].numpy().tobytes()))
    assert "talker.layers.0.q_proj" in tens and "talker.norm" in tens and meta["talker_layers"] == 1


def test_checkpoint_with_a_talker_to_predictor_projection_is_refused(tmp_path):
    """The code predictor's first op, `small_to_mtp_projection` (scripts/export_code_predictor_onnx.py:38-41), is the
    identity -- and has no tensors -- in the 0.6 B model (export_code_predictor_weights.py:51-74 exports none).  A
    snapshot that carries it is a model this build has no op for: the native loader refuses it instead of dropping the
    tensor, and so does the Python converter."""
    torch = pytest.importorskip("torch")
    from safetensors.torch import save_file
    from qwen3_tts_axera_russian_amd import weights as W
    t = {"talker.model.norm.weight": torch.ones(8), "talker.codec_head.weight": torch.zeros(3072, 8),
         "talker.model.codec_embedding.weight": torch.zeros(3072, 8)}
    ok, bad = tmp_path / "ok", tmp_path / "bad"
    ok.mkdir()
    bad.mkdir()
    save_file(t, str(ok / "model.safetensors"))
    tens, _ = inspect(ok)
    assert tens is not None and "talker.norm" in tens
    t["talker.code_predictor.small_to_mtp_projection.weight"] = torch.zeros(8, 16)
    t["talker.code_predictor.small_to_mtp_projection.bias"] = torch.zeros(8)
    save_file(t, str(bad / "model.safetensors"))
    tens, _ = inspect(bad)
    assert tens is None
    with pytest.raises(ValueError, match="small_to_mtp_projection"):
        W.from_hf_checkpoint(str(bad), str(tmp_path / "out.q3w"))
