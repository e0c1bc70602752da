#!/usr/bin/env python3
"""Generate tests/golden/code2wav_golden.npz: outputs of the one importable implementation of the reference's
vocoder family, on seeded weights.

The reference's vocoder is `Qwen3TTSTokenizerV2Model(...).decoder(codes[B,16,T])`, traced to ONNX by
/root/reference/scripts/export_vocoder_traced.py:38-52,74-79 (wrapper: permute to [B,16,T], `wav.squeeze(1)`, lengths =
T x decoder.total_upsample) and called by dual_npu/vocoder_server.py:67-71.  `qwen_tts` is not installable here, but
transformers ships that decoder family as `Qwen3OmniMoeCode2Wav` (same forward(codes), same total_upsample, the module
tree this repo's converter expects: pre_transformer / upsample / decoder, SnakeBeta, CausalConvNet, CausalTransConvNet,
ConvNeXtBlock, LayerScale) and the split residual VQ as `MimiSplitResidualVectorQuantizer`.  This script runs them
in fp32 (eager attention) on the seeded tensors of tests/c2w_common.py and stores, per case:

    <case>.keys      JSON list of [state-dict key, shape] (the seeds regenerate the tensors; <case>.sha guards that)
    <case>.codes     int64 [1][T][16]
    <case>.wav       f32 [samples]                       the decoder's output (clamped)
    <case>.<stage>   f32 [C][kept columns]               activations after each stage (columns: c2w_common.column_subset)

tests/test_code2wav_golden.py maps the keys through weights.state_to_voc and requires oracle/voc_ref.py to reproduce
every stage and the waveform to <= 1e-5 (CPU) and voc_decode (chunk = the case's T) to <= 2e-4 (GPU, exact and split
arithmetic).

Usage (build container only; transformers + torch on CPU, ~10 s):  python tests/golden/make_code2wav_golden.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import c2w_common as C  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "code2wav_golden.npz")


def _c2w(params):
    from transformers.models.qwen3_omni_moe.configuration_qwen3_omni_moe import Qwen3OmniMoeCode2WavConfig
    from transformers.models.qwen3_omni_moe.modeling_qwen3_omni_moe import Qwen3OmniMoeCode2Wav
    cfg = Qwen3OmniMoeCode2WavConfig(**{k: (list(v) if isinstance(v, tuple) else v) for k, v in params.items()})
    cfg._attn_implementation = "eager"
    return cfg, Qwen3OmniMoeCode2Wav(cfg).to(torch.float32).eval()


class Omni:
    """Qwen3OmniMoeCode2Wav as it is."""

    def __init__(self, case):
        self.cfg, self.m = _c2w(case["c2w"])
        self.key_shapes = [("decoder." + k, tuple(v.shape)) for k, v in self.m.state_dict().items()]

    def load(self, state):
        self.m.load_state_dict({k[len("decoder."):]: torch.from_numpy(v) for k, v in state.items()}, strict=True)

    def run(self, codes):           # codes [1][T][16] -> dict of stages, wav
        m, st = self.m, {}
        c = torch.from_numpy(codes).permute(0, 2, 1)      # the export wrapper's permute (export_vocoder_traced.py:48)
        hooks = [m.pre_transformer.register_forward_hook(lambda _m, _i, o: st.__setitem__("pre_transformer", o.last_hidden_state.permute(0, 2, 1)))]
        for u, blocks in enumerate(m.upsample):
            hooks.append(blocks[-1].register_forward_hook(lambda _m, _i, o, u=u: st.__setitem__(f"upsample{u}", o)))
        hooks.append(m.decoder[0].register_forward_hook(lambda _m, _i, o: st.__setitem__("dec_in", o)))
        for b in range(len(self.cfg.upsample_rates)):
            hooks.append(m.decoder[1 + b].register_forward_hook(lambda _m, _i, o, b=b: st.__setitem__(f"block{b}", o)))
        st["front"] = m.code_embedding(c + m.code_offset).mean(1).permute(0, 2, 1)
        wav = m(c)
        for h in hooks:
            h.remove()
        return st, wav


class Tts:
    """Mimi split RVQ -> causal conv -> Linear -> Omni pre-transformer (narrower) -> Linear -> Omni upsample + decoder."""

    def __init__(self, case):
        from transformers.models.mimi.configuration_mimi import MimiConfig
        from transformers.models.mimi.modeling_mimi import MimiSplitResidualVectorQuantizer
        from transformers.models.qwen3_omni_moe.modeling_qwen3_omni_moe import (Qwen3OmniMoeCausalConvNet,
                                                                                 Qwen3OmniMoeCode2WavTransformerModel)
        r = case["rvq"]
        mc = MimiConfig(codebook_size=r["codebook_size"], codebook_dim=r["codebook_dim"],
                        vector_quantization_hidden_dimension=r["codebook_dim"], hidden_size=r["hidden"],
                        num_quantizers=r["num_quantizers"], num_semantic_quantizers=r["num_semantic"])
        self.rvq = MimiSplitResidualVectorQuantizer(mc).to(torch.float32).eval()
        self.cfg, self.m = _c2w(case["c2w"])
        latent = self.cfg.hidden_size
        self.pre_conv = Qwen3OmniMoeCausalConvNet(r["hidden"], latent, case["pre_kernel"]).eval()
        tcfg, _ = _c2w(dict(case["c2w"], hidden_size=case["tf_hidden"]))
        self.tf = Qwen3OmniMoeCode2WavTransformerModel(tcfg).to(torch.float32).eval()
        self.inp = torch.nn.Linear(latent, case["tf_hidden"]).eval()
        self.outp = torch.nn.Linear(case["tf_hidden"], latent).eval()
        self.parts = [("decoder.quantizer.", self.rvq), ("decoder.pre_conv.", self.pre_conv),
                      ("decoder.pre_transformer.input_proj.", self.inp), ("decoder.pre_transformer.", self.tf),
                      ("decoder.pre_transformer.output_proj.", self.outp)]
        self.key_shapes = [(p + k, tuple(v.shape)) for p, mod in self.parts for k, v in mod.state_dict().items()]
        self.c2w_keys = [k for k in self.m.state_dict() if k.startswith(("upsample.", "decoder."))]
        self.key_shapes += [("decoder." + k, tuple(self.m.state_dict()[k].shape)) for k in self.c2w_keys]

    def load(self, state):
        for p, mod in self.parts:
            mod.load_state_dict({k: torch.from_numpy(state[p + k]) for k in mod.state_dict()}, strict=True)
            for cb in mod.modules():            # Mimi caches embed_sum / usage on first use
                if hasattr(cb, "_embed"):
                    cb._embed = None
        sd = self.m.state_dict()
        sd.update({k: torch.from_numpy(state["decoder." + k]) for k in self.c2w_keys})
        self.m.load_state_dict(sd, strict=True)

    def run(self, codes):
        m, st = self.m, {}
        c = torch.from_numpy(codes).permute(0, 2, 1)
        h = self.rvq.decode(c)
        st["front"] = h
        h = self.pre_conv(h)
        st["pre_conv"] = h
        h = self.tf(inputs_embeds=self.inp(h.transpose(1, 2))).last_hidden_state
        h = self.outp(h).permute(0, 2, 1)
        st["pre_transformer"] = h
        for u, blocks in enumerate(m.upsample):
            for blk in blocks:
                h = blk(h)
            st[f"upsample{u}"] = h
        for i, blk in enumerate(m.decoder):
            h = blk(h)
            if i == 0:
                st["dec_in"] = h
            elif i <= len(self.cfg.upsample_rates):
                st[f"block{i - 1}"] = h
        return st, h.clamp(min=-1, max=1)


@torch.no_grad()
def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    out = {}
    for name, case in C.CASES.items():
        mod = Omni(case) if case["kind"] == "omni" else Tts(case)
        state = C.seeded_state(case["seed"], mod.key_shapes)
        mod.load(state)
        out[f"{name}.keys"] = np.frombuffer(json.dumps([[k, list(s)] for k, s in mod.key_shapes]).encode(), np.uint8)
        out[f"{name}.sha"] = np.frombuffer(C.digest(state).encode(), np.uint8)
        size = case["c2w"]["codebook_size"]
        codes = C.seeded_codes(case["seed"], case["T"], size)
        st, wav = mod.run(codes)
        out[f"{name}.codes"] = codes
        out[f"{name}.wav"] = wav[0, 0].numpy().copy()
        for k, v in st.items():
            a = v[0].numpy()
            out[f"{name}.{k}"] = a[:, C.column_subset(a.shape[1])].copy()
        print(name, "T", case["T"], "->", wav.shape[-1], "samples; total_upsample", int(mod.m.total_upsample), "; stages", {k: tuple(v.shape[1:]) for k, v in st.items()})
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
