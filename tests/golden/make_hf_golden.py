#!/usr/bin/env python3
"""Generate tests/golden/hf_qwen3_golden.npz: outputs of transformers' Qwen3Model -- the layer type the
reference names for the talker (/root/reference/scripts/extract_talker_as_qwen3.py:89-110: architectures
["Qwen3ForCausalLM"], model_type "qwen3", 16/8 heads, head_dim 128, rms_norm_eps 1e-6, rope_theta 1e6, silu,
no attention bias) and for the code predictor's core (scripts/export_code_predictor_onnx.py:30-46:
small_to_mtp_projection -> rotary_emb -> layers -> norm) -- on the seeded weights and inputs of
tests/hf_common.py, in fp32, eager attention.

What is stored (inputs are regenerated from the seeds; a digest of the weights guards the regeneration):
  talker_prefill   last_hidden_state of Qwen3Model(inputs_embeds=prefill[1,n,1024])        (post final norm)
  talker_decode    the same for 5 single-token steps continued on the KV cache (use_cache)
  cp_hidden        last_hidden_state[1:16] of the code-predictor stack over the 16-row sequence
                   [talker hidden, talker_emb[code_0], cp_emb[g][tok_g] ...] with teacher-forced tokens
                   (code_predictor_server.py:94-140 with forced samples), one causal pass = the AR loop

tests/test_oracle_vs_hf.py asserts oracle/q3_oracle.c reproduces them: in exact mode (no fp16 rounding of
activations) to fp32 round-off, in the device's rounded mode to the fp16-input tolerance of DESIGN.md 2.

Usage:  python tests/golden/make_hf_golden.py      (needs transformers + torch; runs on CPU in ~20 s)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import hf_common as C  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hf_qwen3_golden.npz")

HF_KEY = {"input_ln": "input_layernorm.weight", "q_proj": "self_attn.q_proj.weight",
          "k_proj": "self_attn.k_proj.weight", "v_proj": "self_attn.v_proj.weight",
          "o_proj": "self_attn.o_proj.weight", "q_norm": "self_attn.q_norm.weight",
          "k_norm": "self_attn.k_norm.weight", "post_ln": "post_attention_layernorm.weight",
          "gate_proj": "mlp.gate_proj.weight", "up_proj": "mlp.up_proj.weight", "down_proj": "mlp.down_proj.weight"}


def hf_stack(cfg, tensors, prefix, n_layers, ffn):
    from transformers import Qwen3Config, Qwen3Model
    hc = Qwen3Config(vocab_size=8, hidden_size=cfg.hidden, intermediate_size=ffn, num_hidden_layers=n_layers,
                     num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads, head_dim=cfg.head_dim,
                     rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta, max_position_embeddings=32768,
                     hidden_act="silu", attention_bias=False, attention_dropout=0.0, tie_word_embeddings=False,
                     use_cache=True)
    hc._attn_implementation = "eager"
    m = Qwen3Model(hc).to(torch.float32).eval()
    sd = {"embed_tokens.weight": torch.zeros(8, cfg.hidden)}
    for i in range(n_layers):
        for part, key in HF_KEY.items():
            sd[f"layers.{i}.{key}"] = torch.from_numpy(np.array(tensors[f"{prefix}.layers.{i}.{part}"]))
    sd["norm.weight"] = torch.from_numpy(np.array(tensors[f"{prefix}.norm"]))
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not [k for k in missing if "rotary" not in k] and not unexpected, (missing, unexpected)
    return m


@torch.no_grad()
def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cfg = C.hf_check_config()
    t = C.make_tensors(cfg)
    x = C.make_inputs(cfg)
    out = {"weights_sha256": np.frombuffer(C.digest(t).encode(), np.uint8)}
    # ---- talker: prefill, then decode on the cache (llama_wrapper.c:125-163 with n_tokens = n, then 1) ----
    m = hf_stack(cfg, t, "talker", cfg.talker_layers, cfg.talker_ffn)
    r = m(inputs_embeds=torch.from_numpy(x["prefill"])[None], use_cache=True)
    out["talker_prefill"] = r.last_hidden_state[0].numpy().copy()
    past, dec = r.past_key_values, []
    for i in range(x["decode"].shape[0]):
        r = m(inputs_embeds=torch.from_numpy(x["decode"][i])[None, None], past_key_values=past, use_cache=True)
        past = r.past_key_values
        dec.append(r.last_hidden_state[0, 0].numpy().copy())
    out["talker_decode"] = np.stack(dec)
    # ---- code predictor: 16-row teacher-forced sequence in one causal pass ----
    G = cfg.cp_groups
    rows = [x["cp_hidden"], t["talker.codec_embedding"][int(x["cp_code0"])]]
    for g in range(G - 1):
        rows.append(t[f"cp.codec_emb.{g}"][int(x["cp_forced"][g])])
    seq = torch.from_numpy(np.stack(rows).astype(np.float32))
    mc = hf_stack(cfg, t, "cp", cfg.cp_layers, cfg.cp_ffn)
    r = mc(inputs_embeds=seq[None], use_cache=False)
    out["cp_hidden"] = r.last_hidden_state[0, 1:].numpy().copy()      # positions 1..15 feed lm_head 0..14
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()}, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
