"""Reference-shaped CPU pipeline: talker -> sample -> code predictor -> feedback, one utterance.
TEST INFRASTRUCTURE ONLY (checker for the fused engine, and the timed CPU baseline of bench.py).

The loop is the one the reference runs across its three processes: llamacpp_talker_server.py:254-293
(sample, break on EOS / >= 2048, feed the feedback back), code_predictor_server.py:94-140,
tts_client.py:199-208."""
from __future__ import annotations

import numpy as np

from . import frontend as fe
from . import oracle as orc


class CpuPipeline:
    def __init__(self, cfg, tensors, n_ctx=512):
        self.cfg = cfg
        self.talker = orc.TalkerOracle(cfg, tensors, n_ctx=n_ctx)
        self.cp = orc.CpOracle(cfg, tensors)
        self.codec_embedding = self.talker.codec_embedding
        self.cp_emb = self.cp.emb

    def generate(self, prefix, n_text, pad_embed, max_frames, ignore_eos=False, want_margins=False):
        """-> frames list of 16 ints (and, if asked, per-frame [16] top-1/top-2 logit gaps; the entry
        after the last frame holds the gap of the terminating EOS decision in column 0)."""
        cfg = self.cfg
        self.talker.clear()
        hidden = self.talker.forward(prefix, 0)
        pos = prefix.shape[0]
        past, frames, margins_all = [], [], []
        for _ in range(max_frames):
            logits = self.talker.logits(hidden)
            if ignore_eos:
                lg, _ = fe.process_talker_logits(logits, past, n_text, cfg.codec_eos)
                lg[cfg.codec_eos] = -1e10
                code0 = int(np.argmax(lg))
            else:
                lg, forced = fe.process_talker_logits(logits, past, n_text, cfg.codec_eos)
                code0 = int(forced) if forced is not None else int(np.argmax(lg))
            srt = np.sort(lg)
            m0 = float(srt[-1] - srt[-2])
            if code0 == cfg.codec_eos or code0 >= 2048:
                margins_all.append([m0] + [np.inf] * 15)
                break
            codes, margins = self.cp.predict(hidden, code0)
            margins_all.append([m0] + [float(x) for x in margins])
            frames.append([code0] + [int(c) for c in codes])
            past.append(code0)
            fb = fe.feedback_embedding(code0, codes, self.codec_embedding, self.cp_emb, pad_embed)
            hidden = self.talker.forward(fb, pos)
            pos += 1
        return (frames, margins_all) if want_margins else frames


    def generate_batch(self, prefixes, n_text, pad_embed, max_frames, ignore_eos=False):
        """The same loop for B utterances in lock step, weights read once per pass for the whole batch: per
        utterance bit-identical to generate() (rows never mix; checked in tests/test_weights_container.py).
        -> (frames[b] list of 16-int lists, margins[b] list of 16 gaps) per utterance."""
        cfg = self.cfg
        B = len(prefixes)
        caches, hid, pos = [], [], []
        for p in prefixes:
            self.talker.clear()
            hid.append(self.talker.forward(p, 0))
            caches.append((self.talker.kc, self.talker.vc, self.talker.n_ctx))
            pos.append(p.shape[0])
        hid = np.stack(hid)
        past = [[] for _ in range(B)]
        frames = [[] for _ in range(B)]
        margins = [[] for _ in range(B)]
        alive = list(range(B))
        for _ in range(max_frames):
            if not alive:
                break
            logits = orc.head_logits_batch(self.talker.codec_head, hid[alive])
            code0, keep = [], []
            for j, b in enumerate(alive):
                lg, forced = fe.process_talker_logits(logits[j], past[b], n_text[b], cfg.codec_eos)
                if ignore_eos:
                    lg[cfg.codec_eos] = -1e10
                    c0 = int(np.argmax(lg))
                else:
                    c0 = int(forced) if forced is not None else int(np.argmax(lg))
                srt = np.sort(lg)
                m0 = float(srt[-1] - srt[-2])
                if c0 == cfg.codec_eos or c0 >= 2048:
                    margins[b].append([m0] + [np.inf] * 15)
                    continue
                margins[b].append([m0])
                code0.append(c0)
                keep.append(b)
            alive = keep
            if not alive:
                break
            codes, cm = self.cp.predict_batch(hid[alive], code0)
            fb = []
            for j, b in enumerate(alive):
                margins[b][-1] += [float(x) for x in cm[j]]
                frames[b].append([code0[j]] + [int(c) for c in codes[j]])
                past[b].append(code0[j])
                fb.append(fe.feedback_embedding(code0[j], codes[j], self.codec_embedding, self.cp_emb, pad_embed))
            hid[alive] = orc.forward_batch(self.talker, [caches[b] for b in alive], [pos[b] for b in alive], np.stack(fb))
            for b in alive:
                pos[b] += 1
        self.last_hidden = hid      # talker hidden of every utterance after its last executed step
        return frames, margins
