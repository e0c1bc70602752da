"""Independent fp32 / float64 evaluation of the vocoder program (torch CPU ops) -- test infrastructure
(only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import it).
Follows the op-table semantics documented in DESIGN.md "Vocoder program".  The decoder's layer list is not in the
reference (scripts/export_vocoder_traced.py:38-52 only drives `decoder(codes[B,16,T])`); what pins
this file is the importable implementation of that decoder family: tests/test_code2wav_golden.py feeds it the
state_dict() of transformers' Qwen3OmniMoeCode2Wav (+ Mimi's split RVQ) through weights.state_to_voc and requires the
outputs stored by tests/golden/make_code2wav_golden.py to <= 1e-5.  Parity with the real Qwen3-TTS checkpoint stays
unpinned (no weights, no qwen_tts here)."""
import numpy as np
import torch
import torch.nn.functional as F

from qwen3_tts_axera_russian_amd import weights as W


def _snake(x, alpha, beta):
    a = torch.exp(alpha)[None, :, None]
    ib = 1.0 / (torch.exp(beta)[None, :, None] + 1e-9)
    return x + ib * torch.sin(a * x) ** 2


def voc_reference(tensors: dict, codes: np.ndarray, n_ops: int = -1, dtype=np.float32) -> np.ndarray:
    """codes int64 [B][T][16] -> wav f32 [B][T*upsample] (or the activation after n_ops ops)."""
    prog = np.asarray(tensors["voc.program"])
    t = lambda n: torch.from_numpy(np.array(tensors[n], dtype=dtype))
    codes_t = torch.from_numpy(np.asarray(codes, dtype=np.int64))
    x, res = None, None
    torch.set_num_threads(8)
    with torch.no_grad():
        for i, row in enumerate(prog):
            if 0 <= n_ops <= i:
                break
            op, p = int(row[0]), f"voc.op{i}."
            if op == W.VOP_RVQ:
                nq, cbs = int(row[1]), int(row[2])
                cb = t(p + "codebook")
                valid = ((codes_t >= 0) & (codes_t < cbs)).to(cb.dtype)
                idx = codes_t.clamp(0, cbs - 1)
                emb = torch.stack([cb[q][idx[..., q]] * valid[..., q:q + 1] for q in range(nq)], 0)  # [nq,B,T,dim]
                sem, ac = emb[0], emb[1:].sum(0)
                y = sem @ t(p + "proj_sem").T + ac @ t(p + "proj_ac").T
                x = y.transpose(1, 2).contiguous()
            elif op == W.VOP_EMBMEAN:   # code_embedding(codes + q * size).mean(q)   (Qwen3OmniMoeCode2Wav.forward)
                nq, cbs = int(row[1]), int(row[2])
                tab = t(p + "embedding")
                valid = ((codes_t >= 0) & (codes_t < cbs)).to(tab.dtype)
                idx = codes_t.clamp(0, cbs - 1) + torch.arange(nq)[None, None, :] * cbs
                y = (tab[idx[..., :nq]] * valid[..., :nq, None]).mean(2)            # [B,T,dim]
                x = y.transpose(1, 2).contiguous()
            elif op in (W.VOP_CONV, W.VOP_CONVT):
                k, p0, flags = int(row[3]), int(row[4]), int(row[5])
                if flags & W.VF_RES_SAVE:
                    res = x
                h = _snake(x, t(p + "alpha"), t(p + "beta")) if flags & W.VF_SNAKE else x
                if flags & W.VF_GELU:
                    h = F.gelu(h)     # exact (erf) form
                bias = t(p + "bias") if (p + "bias") in tensors else None
                if op == W.VOP_CONV:
                    y = F.conv1d(F.pad(h, ((k - 1) * p0, 0)), t(p + "weight"), bias, dilation=p0)
                else:
                    # (L - 1) * stride + k outputs, row[6] cut on the left, row[7] on the right
                    # (Qwen3OmniMoeCausalTransConvNet: both = k - stride)
                    y = F.conv_transpose1d(h, t(p + "weight"), bias, stride=p0)
                    y = y[..., int(row[6]): y.shape[-1] - int(row[7])].contiguous()
                if flags & W.VF_RES_ADD:
                    y = y + res
                if flags & W.VF_CLAMP:
                    y = y.clamp(-1.0, 1.0)
                x = y
            elif op == W.VOP_DWCONV:   # causal depthwise conv (ConvNeXt)
                k, flags = int(row[3]), int(row[5])
                if flags & W.VF_RES_SAVE:
                    res = x
                x = F.conv1d(F.pad(x, (k - 1, 0)), t(p + "weight"), t(p + "bias"), groups=x.shape[1])
            elif op == W.VOP_NORM:     # over channels, per column: 0 RMSNorm, 1 LayerNorm
                kind, eps, flags = int(row[3]), int(row[4]) * 1e-9, int(row[5])
                if flags & W.VF_RES_SAVE:
                    res = x
                w = t(p + "weight")[None, :, None]
                if kind == 0:
                    x = x * torch.rsqrt((x * x).mean(1, keepdim=True) + eps) * w
                else:
                    mu = x.mean(1, keepdim=True)
                    var = ((x - mu) ** 2).mean(1, keepdim=True)
                    x = (x - mu) * torch.rsqrt(var + eps) * w + t(p + "bias")[None, :, None]
            elif op == W.VOP_ATTN:     # [q | k | v] head-major channels -> causal sliding-window attention with RoPE
                nh, hd, window, theta = int(row[3]), int(row[4]), int(row[6]), float(row[7])
                B, _, L = x.shape
                q, k_, v = [z.reshape(B, nh, hd, L).transpose(2, 3) for z in x.split(nh * hd, dim=1)]   # [B,nh,L,hd]
                pos = torch.arange(L, dtype=x.dtype)
                inv = theta ** (-torch.arange(0, hd, 2, dtype=x.dtype) / hd)
                ang = pos[:, None] * inv[None, :]
                cos, sin = torch.cat([ang.cos(), ang.cos()], -1), torch.cat([ang.sin(), ang.sin()], -1)
                rot = lambda z: torch.cat([-z[..., hd // 2:], z[..., : hd // 2]], -1)
                q, k_ = q * cos + rot(q) * sin, k_ * cos + rot(k_) * sin
                sc = (q @ k_.transpose(2, 3)) / (hd ** 0.5)
                i, j = torch.arange(L)[:, None], torch.arange(L)[None, :]
                sc = sc.masked_fill(~((j <= i) & (j > i - window)), float("-inf"))
                o = torch.softmax(sc, -1) @ v                                   # [B,nh,L,hd]
                x = o.transpose(2, 3).reshape(B, nh * hd, L)
            elif op == W.VOP_GLU:      # act(first half) * second half; 0 SiLU, 1 GELU
                c, act = int(row[2]), int(row[3])
                g, u = x[:, :c], x[:, c:]
                x = (F.silu(g) if act == 0 else F.gelu(g)) * u
            else:
                raise ValueError(f"unknown vocoder op {op}")
    return x.numpy() if n_ops >= 0 else x[:, 0, :].numpy()
