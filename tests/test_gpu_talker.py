"""Talker boundary (wrapper_* ABI through the reference-shaped bindings) against the CPU oracle
on identical fp16 weights.  Float tolerance: both sides use the same numerics contract (fp16 GEMM
inputs, f32 accumulate), so only summation order and rare fp16 rounding flips differ; observed
relative error is ~1e-4, the bound asserted here is 5e-3 of the hidden's max magnitude."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from qwen3_tts_axera_russian_amd import hiplib
from qwen3_tts_axera_russian_amd.llama_cpp_bindings import LlamaCppModel
from tests.util import rel_err, synthetic_pack

pytestmark = pytest.mark.gpu
TOL = 5e-3


@pytest.fixture(scope="module")
def tiny():
    path, cfg, tensors = synthetic_pack(2, 2)
    return path, cfg, tensors


def _prefix(rng, n, H=1024, scale=0.05):
    return (scale * rng.standard_normal((n, H))).astype(np.float32)


def test_prefill_and_decode_match_oracle(gpu_lib, tiny):
    path, cfg, tensors = tiny
    llm = LlamaCppModel(path, n_ctx=64)
    ref = orc.TalkerOracle(cfg, tensors, n_ctx=64)
    rng = np.random.default_rng(1)
    prefix = _prefix(rng, 13)
    h_gpu = llm.get_hidden(prefix, keep_history=0)
    h_ref = ref.forward(prefix, 0)
    assert h_gpu.shape == (1024,)
    errs = [rel_err(h_gpu, h_ref)]
    for step in range(6):
        fb = _prefix(rng, 1)
        h_gpu = llm.get_hidden(fb, keep_history=1)
        h_ref = ref.forward(fb, 13 + step)
        errs.append(rel_err(h_gpu, h_ref))
    print("talker rel errs:", ["%.2e" % e for e in errs])
    assert max(errs) < TOL
    assert llm.pos == 19
    # prefill again from scratch gives the same answer (keep_history=0 resets the position)
    h2 = llm.get_hidden(prefix, keep_history=0)
    assert rel_err(h2, ref.forward(prefix, 0)) < TOL
    # codec head on the device vs oracle head
    lg = llm.codec_head(h2)[0]
    lr = ref.logits(h2)
    assert rel_err(lg, lr) < 1e-3
    llm.destroy()


def test_single_token_prefill_and_edge_sizes(gpu_lib, tiny):
    path, cfg, tensors = tiny
    llm = LlamaCppModel(path, n_ctx=48)
    ref = orc.TalkerOracle(cfg, tensors, n_ctx=48)
    rng = np.random.default_rng(2)
    for n in (1, 2, 16, 17, 33, 48):  # ragged sizes around the 16/32/64-row kernel tiles, up to n_ctx
        p = _prefix(rng, n)
        ref.clear()
        assert rel_err(llm.get_hidden(p, 0), ref.forward(p, 0)) < TOL, n
    # context overflow is an error, not a crash (llama_decode failure -> -1 -> RuntimeError)
    with pytest.raises(RuntimeError):
        llm.get_hidden(_prefix(rng, 1), keep_history=1)
    with pytest.raises(AssertionError):
        llm.get_hidden(np.zeros((2, 512), np.float32), 0)
    llm.destroy()


def test_state_save_load_roundtrip(gpu_lib, tiny, tmp_path):
    path, cfg, tensors = tiny
    llm = LlamaCppModel(path, n_ctx=64)
    rng = np.random.default_rng(3)
    prefix = _prefix(rng, 11)
    fb = _prefix(rng, 1)
    llm.get_hidden(prefix, 0)
    kv = str(tmp_path / "kv.bin")
    assert llm.state_save(kv) == 0
    assert os.path.getsize(kv) == llm.state_get_size()
    h_a = llm.get_hidden(fb, 1)
    # wipe the cache with another prefix, then restore (llamacpp_talker_server.py:226-236)
    llm.get_hidden(_prefix(rng, 20), 0)
    assert llm.state_load(kv) == 0
    llm.pos = 11
    h_b = llm.get_hidden(fb, 1)
    np.testing.assert_array_equal(h_a, h_b)
    assert llm.state_load(str(tmp_path / "missing.bin")) == -1
    llm.destroy()


def test_batch_slots_match_single_sequence(gpu_lib, tiny):
    path, cfg, tensors = tiny
    lib = gpu_lib
    model = lib.wrapper_load_model(path.encode(), 0)
    assert model
    ctx = lib.wrapper_create_context_slots(model, 64, 64, 4)
    assert ctx and lib.wrapper_ctx_n_slots(ctx) == 4
    rng = np.random.default_rng(4)
    lens = [5, 9, 14]
    refs, out = [], np.empty(1024, np.float32)
    for s, n in enumerate(lens):
        p = _prefix(rng, n)
        assert lib.wrapper_decode_embd_slot(ctx, s, hiplib.fptr(p), n, 1024, 0, hiplib.fptr(out)) == 0
        r = orc.TalkerOracle(cfg, tensors, n_ctx=64)
        assert rel_err(out, r.forward(p, 0)) < TOL
        refs.append(r)
    for step in range(3):
        rows = _prefix(rng, 3)
        slots = np.array([0, 1, 2], np.int32)
        pos = np.array([n + step for n in lens], np.int32)
        outb = np.empty((3, 1024), np.float32)
        assert lib.wrapper_decode_embd_batch(ctx, hiplib.fptr(rows), 3, 1024, hiplib.iptr(slots), hiplib.iptr(pos),
                                             hiplib.fptr(outb)) == 0
        for s in range(3):
            assert rel_err(outb[s], refs[s].forward(rows[s], int(pos[s]))) < TOL
    # duplicate slots in one batch are rejected
    bad = np.array([1, 1, 2], np.int32)
    assert lib.wrapper_decode_embd_batch(ctx, hiplib.fptr(rows), 3, 1024, hiplib.iptr(bad), hiplib.iptr(pos),
                                         hiplib.fptr(outb)) == -1
    lib.wrapper_free_context(ctx)
    lib.wrapper_free_model(model)


def test_full_depth_28_layers(gpu_lib):
    """The real depth (28 layers), a short prefix and two decode steps."""
    path, cfg, tensors = synthetic_pack(28, 1, parts=("talker",))
    llm = LlamaCppModel(path, n_ctx=32)
    ref = orc.TalkerOracle(cfg, tensors, n_ctx=32)
    rng = np.random.default_rng(6)
    p = _prefix(rng, 10)
    errs = [rel_err(llm.get_hidden(p, 0), ref.forward(p, 0))]
    for s in range(2):
        fb = _prefix(rng, 1)
        errs.append(rel_err(llm.get_hidden(fb, 1), ref.forward(fb, 10 + s)))
    print("28-layer rel errs:", ["%.2e" % e for e in errs])
    assert max(errs) < 2e-2
    llm.destroy()


def test_long_context_beyond_the_reference_n_ctx(gpu_lib, tiny):
    """BASELINE config 5 (>= 60 s = 750 frames) needs positions past the reference's n_ctx = 512
    (llamacpp_talker_server.py:104): prefill 600 rows, then decode at positions 600..603 with the
    attention walking > 512 cached rows, against the oracle."""
    path, cfg, tensors = tiny
    n = 600
    llm = LlamaCppModel(path, n_ctx=640)
    ref = orc.TalkerOracle(cfg, tensors, n_ctx=640)
    rng = np.random.default_rng(11)
    prefix = _prefix(rng, n)
    errs = [rel_err(llm.get_hidden(prefix, keep_history=0), ref.forward(prefix, 0))]
    for step in range(4):
        fb = _prefix(rng, 1)
        errs.append(rel_err(llm.get_hidden(fb, keep_history=1), ref.forward(fb, n + step)))
    print("long-context rel errs:", ["%.2e" % e for e in errs])
    assert max(errs) < TOL and llm.pos == n + 4
    llm.destroy()


def test_load_talker_from_rekeyed_safetensors(gpu_lib, tiny, tmp_path):
    """wrapper_load_model on the file the reference's converter writes before its GGUF step (re-keyed
    Qwen3ForCausalLM talker, BF16, embed_tokens / lm_head padded to the text vocabulary:
    scripts/extract_talker_as_qwen3.py:53-71), parsed natively: same hidden states as the container holding
    the same values."""
    torch = pytest.importorskip("torch")
    from safetensors.torch import save_file
    path, cfg, tensors = tiny
    hf = {"input_ln": "input_layernorm.weight", "q_proj": "self_attn.q_proj.weight", "k_proj": "self_attn.k_proj.weight",
          "v_proj": "self_attn.v_proj.weight", "o_proj": "self_attn.o_proj.weight", "q_norm": "self_attn.q_norm.weight",
          "k_norm": "self_attn.k_norm.weight", "post_ln": "post_attention_layernorm.weight",
          "gate_proj": "mlp.gate_proj.weight", "up_proj": "mlp.up_proj.weight", "down_proj": "mlp.down_proj.weight"}
    t = {}
    for i in range(cfg.talker_layers):
        for p, k in hf.items():
            a = np.asarray(tensors[f"talker.layers.{i}.{p}"])
            # fp16 matrices travel as F16 (exact); norm vectors as F32
            t[f"model.layers.{i}.{k}"] = torch.from_numpy(np.array(a, dtype=a.dtype))
    t["model.norm.weight"] = torch.from_numpy(np.array(tensors["talker.norm"], dtype=np.float32))
    emb = np.zeros((4000, 1024), np.float32)
    emb[:3072] = np.asarray(tensors["talker.codec_embedding"], dtype=np.float32)
    t["model.embed_tokens.weight"] = torch.from_numpy(emb)
    head = np.zeros((4000, 1024), np.float16)
    head[:3072] = np.asarray(tensors["talker.codec_head"], dtype=np.float16)
    t["lm_head.weight"] = torch.from_numpy(head)
    d = tmp_path / "talker_as_qwen3"
    d.mkdir()
    save_file(t, str(d / "model.safetensors"), metadata={"format": "pt"})
    a = LlamaCppModel(path, n_ctx=64)
    b = LlamaCppModel(str(d), n_ctx=64)           # the directory, as the reference's scripts leave it
    rng = np.random.default_rng(3)
    prefix = _prefix(rng, 11)
    np.testing.assert_array_equal(b.get_hidden(prefix, keep_history=0), a.get_hidden(prefix, keep_history=0))
    fb = _prefix(rng, 1)
    hb, ha = b.get_hidden(fb, keep_history=1), a.get_hidden(fb, keep_history=1)
    np.testing.assert_array_equal(hb, ha)
    np.testing.assert_array_equal(b.codec_head(hb), a.codec_head(ha))
    a.destroy()
    b.destroy()
