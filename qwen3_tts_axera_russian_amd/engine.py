"""Host side of the fused on-device frame loop (include/qwen3tts_engine.h).

Stands where the reference's client closes the loop over sockets (dual_npu/tts_client.py:144-215);
the per-frame work (talker step, 15-group code predictor, feedback sum) stays on the GPU."""
from __future__ import annotations

import numpy as np

from . import hiplib


class FrameEngine:
    def __init__(self, weights_path, max_batch=32, n_ctx=512, max_frames=256):
        self._lib = hiplib.load()
        self.h = self._lib.q3e_create(str(weights_path).encode(), max_batch, n_ctx, max_frames)
        if not self.h:
            raise RuntimeError(f"q3e_create failed: {weights_path}")
        self.max_batch, self.n_ctx, self.max_frames = max_batch, n_ctx, max_frames
        self.B = 0

    def set_sampling(self, talker_temperature=0.0, talker_top_k=50, talker_top_p=0.95, cp_temperature=0.0,
                     cp_top_k=50, seed=0):
        if self._lib.q3e_set_sampling(self.h, float(talker_temperature), int(talker_top_k), float(talker_top_p),
                                      float(cp_temperature), int(cp_top_k), int(seed)) != 0:
            raise RuntimeError("q3e_set_sampling failed")

    def set_chains(self, n):
        if self._lib.q3e_set_chains(self.h, int(n)) != 0:
            raise RuntimeError("q3e_set_chains failed")

    def set_pad_embed(self, pad):
        pad = np.ascontiguousarray(pad, dtype=np.float32).reshape(-1)
        if self._lib.q3e_set_pad_embed(self.h, hiplib.fptr(pad)) != 0:
            raise RuntimeError("q3e_set_pad_embed failed")

    def start(self, prefixes, n_text, ignore_eos=False, max_frames=0):
        """prefixes: list of [n_b, hidden] f32 prefix matrices (llamacpp_talker_server.py:121-161)."""
        n_rows = np.array([p.shape[0] for p in prefixes], np.int32)
        cat = np.ascontiguousarray(np.concatenate(prefixes, axis=0), dtype=np.float32)
        nt = np.ascontiguousarray(n_text, dtype=np.int32)
        self.B = len(prefixes)
        rc = self._lib.q3e_start(self.h, self.B, hiplib.fptr(cat), hiplib.iptr(n_rows), hiplib.iptr(nt),
                                 int(bool(ignore_eos)), int(max_frames))
        if rc != 0:
            raise RuntimeError(f"q3e_start failed: {rc}")

    def set_forced_codes(self, forced):
        """Teacher forcing (after start()): forced[f][B][16] int32, entries < 0 free-running; None = off."""
        if forced is None:
            rc = self._lib.q3e_set_forced_codes(self.h, None, 0)
        else:
            f = np.ascontiguousarray(forced, dtype=np.int32)
            assert f.ndim == 3 and f.shape[1] == self.B and f.shape[2] == 16
            rc = self._lib.q3e_set_forced_codes(self.h, hiplib.iptr(f), f.shape[0])
        if rc != 0:
            raise RuntimeError("q3e_set_forced_codes failed")

    def run(self, n_frames):
        rc = self._lib.q3e_run(self.h, int(n_frames))
        if rc < 0:
            raise RuntimeError(f"q3e_run failed: {rc}")
        return rc

    @property
    def last_run_ms(self):
        return float(self._lib.q3e_last_run_ms(self.h))

    @property
    def last_prefill_ms(self):
        return float(self._lib.q3e_last_prefill_ms(self.h))

    @property
    def step_weight_bytes(self):
        return float(self._lib.q3e_step_weight_bytes(self.h))

    def codes(self):
        """-> (codes[frames][B][16] int32, frames emitted per utterance)."""
        out = np.full((self.max_frames, self.B, 16), -1, np.int32)
        per = np.zeros(self.B, np.int32)
        nf = self._lib.q3e_get_codes(self.h, hiplib.iptr(out), self.max_frames, hiplib.iptr(per))
        if nf < 0:
            raise RuntimeError("q3e_get_codes failed")
        return out[:nf], per

    def done(self):
        """-> (done[B] bool: the utterance has ended (EOS or its frame budget), frames emitted per utterance)."""
        d = np.zeros(self.B, np.int32)
        per = np.zeros(self.B, np.int32)
        if self._lib.q3e_get_done(self.h, hiplib.iptr(d), hiplib.iptr(per)) != 0:
            raise RuntimeError("q3e_get_done failed")
        return d.astype(bool), per

    def refill(self, slots, prefixes, n_text):
        """Continuous batching: put new utterances into `slots` of the running batch (q3e_refill); the other slots go on
        untouched.  Fetch the codes of a finished utterance before refilling its slot."""
        slots = np.ascontiguousarray(np.asarray(slots, np.int32))
        assert len(slots) == len(prefixes) == len(n_text) and len(slots) > 0
        cat = np.ascontiguousarray(np.concatenate([np.asarray(p, np.float32) for p in prefixes], axis=0))
        n_rows = np.array([p.shape[0] for p in prefixes], np.int32)
        nt = np.ascontiguousarray(np.asarray(n_text, np.int32))
        rc = self._lib.q3e_refill(self.h, len(slots), hiplib.iptr(slots), hiplib.fptr(cat), hiplib.iptr(n_rows), hiplib.iptr(nt))
        if rc != 0:
            raise RuntimeError(f"q3e_refill failed: {rc}")

    def generate_queue(self, prefixes, n_text, max_frames, ignore_eos=False, check_every=8, on_done=None):
        """Continuous batching over a queue of utterances: the first max_batch of them start together; every
        `check_every` frames the finished slots hand over their codes and take the next utterance of the queue
        (q3e_refill), so the frame loop never steps a batch of mostly finished rows.  -> list of int32 [frames][16] in
        queue order.  on_done(index, codes) is called as each utterance finishes (e.g. to hand it to the vocoder)."""
        n = len(prefixes)
        assert n == len(n_text) and n > 0
        B = min(self.max_batch, n)
        out = [None] * n
        owner = list(range(B))                    # queue index of the utterance in each slot
        nxt = B
        self.start(prefixes[:B], n_text[:B], ignore_eos=ignore_eos, max_frames=max_frames)
        while any(o is not None for o in owner):
            ran = self.run(check_every)
            done, per = self.done()
            # an utterance that used its whole frame budget without an EOS has ended too (q3e_get_done reports it)
            fin = [b for b in range(B) if owner[b] is not None and (done[b] or per[b] >= max_frames)]
            if not fin:
                if ran == 0:
                    raise RuntimeError("generate_queue: the engine ran no frame and no utterance finished")
                continue
            codes, _ = self.codes()
            for b in fin:
                res = np.ascontiguousarray(codes[:int(per[b]), b, :])
                out[owner[b]] = res
                if on_done is not None:
                    on_done(owner[b], res)
                owner[b] = None
            take = [b for b in fin if nxt + fin.index(b) < n]
            if take:
                idx = [nxt + i for i in range(len(take))]
                self.refill(take, [prefixes[i] for i in idx], [n_text[i] for i in idx])
                for b, i in zip(take, idx):
                    owner[b] = i
                nxt += len(take)
        return out

    def hidden(self):
        out = np.empty((self.B, 1024), np.float32)
        if self._lib.q3e_get_hidden(self.h, hiplib.fptr(out)) != 0:
            raise RuntimeError("q3e_get_hidden failed")
        return out

    def destroy(self):
        if self.h:
            self._lib.q3e_free(self.h)
            self.h = None
