"""Byte-level BPE tokenizer over a LOCAL vocab.json + merges.txt (the Qwen2 tokenizer family).

Stands where the reference loads a tokenizer by hub name with remote code
(dual_npu/llamacpp_talker_server.py:96-100) and calls `tokenizer.encode(text,
add_special_tokens=False)` (:212): the same ids from the two files of the model snapshot, with no
`transformers` import and nothing fetched.  Algorithm (the published GPT-2 / Qwen2 scheme): NFC
normalisation -> added/special tokens split out verbatim -> pre-tokenisation regex -> bytes mapped
to printable code points -> lowest-rank-first pair merges -> vocabulary lookup.
"""
from __future__ import annotations

import json
import os
import unicodedata
from functools import lru_cache

import regex as re

# Qwen2 pre-tokenisation pattern (tokenizer.json of the Qwen2/Qwen3 family)
PRETOKENIZE = (r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}| ?[^\s\p{L}\p{N}]+[\r\n]*"
               r"|\s*[\r\n]+|\s+(?!\S)|\s+")


@lru_cache()
def bytes_to_unicode() -> dict:
    """The reversible byte -> printable code point table of byte-level BPE."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, map(chr, cs)))


class ByteLevelBPE:
    def __init__(self, vocab: dict, merges: list, added_tokens: dict | None = None, normalize_nfc: bool = True):
        self.vocab = vocab
        self.ranks = {tuple(m): i for i, m in enumerate(merges)}
        self.added = dict(added_tokens or {})
        self.nfc = normalize_nfc
        self.byte_map = bytes_to_unicode()
        self.pat = re.compile(PRETOKENIZE)
        # longest-first alternation so that overlapping special tokens resolve like the HF trie
        self.added_pat = (re.compile("|".join(re.escape(t) for t in sorted(self.added, key=len, reverse=True)))
                          if self.added else None)
        self.id_to_token = {i: t for t, i in vocab.items()}
        self.id_to_token.update({i: t for t, i in self.added.items()})
        self.byte_unmap = {c: b for b, c in self.byte_map.items()}
        self._cache: dict = {}

    @classmethod
    def from_dir(cls, path: str) -> "ByteLevelBPE":
        """A model snapshot directory: vocab.json + merges.txt, added tokens from tokenizer_config.json
        (added_tokens_decoder) / added_tokens.json when present."""
        with open(os.path.join(path, "vocab.json"), encoding="utf-8") as f:
            vocab = json.load(f)
        merges = []
        with open(os.path.join(path, "merges.txt"), encoding="utf-8") as f:
            for line in f:
                line = line.rstrip("\n")
                if not line or line.startswith("#version"):
                    continue
                a, b = line.split(" ")
                merges.append((a, b))
        added = {}
        cfg = os.path.join(path, "tokenizer_config.json")
        if os.path.exists(cfg):
            with open(cfg, encoding="utf-8") as f:
                for tid, ent in (json.load(f).get("added_tokens_decoder") or {}).items():
                    added[ent["content"]] = int(tid)
        extra = os.path.join(path, "added_tokens.json")
        if os.path.exists(extra):
            with open(extra, encoding="utf-8") as f:
                added.update({k: int(v) for k, v in json.load(f).items()})
        return cls(vocab, merges, added)

    def _bpe(self, token: str) -> list:
        hit = self._cache.get(token)
        if hit is not None:
            return hit
        word = list(token)
        while len(word) > 1:
            best, best_rank = None, None
            for i in range(len(word) - 1):
                r = self.ranks.get((word[i], word[i + 1]))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = (word[i], word[i + 1]), r
            if best is None:
                break
            a, b = best
            out, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == a and word[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = out
        self._cache[token] = word
        return word

    def _encode_plain(self, text: str, ids: list) -> None:
        if self.nfc:
            text = unicodedata.normalize("NFC", text)
        for piece in self.pat.findall(text):
            mapped = "".join(self.byte_map[b] for b in piece.encode("utf-8"))
            for sub in self._bpe(mapped):
                tid = self.vocab.get(sub)
                if tid is None:
                    raise KeyError(f"token {sub!r} is not in the vocabulary (vocab.json / merges.txt mismatch)")
                ids.append(tid)

    def encode(self, text: str, add_special_tokens: bool = False) -> list:
        """Token ids of `text`.  `add_special_tokens` is accepted for call compatibility with the reference's
        `tokenizer.encode(text, add_special_tokens=False)`; this family prepends/appends nothing either way."""
        ids: list = []
        if self.added_pat is None:
            self._encode_plain(text, ids)
            return ids
        pos = 0
        for m in self.added_pat.finditer(text):
            if m.start() > pos:
                self._encode_plain(text[pos:m.start()], ids)
            ids.append(self.added[m.group(0)])
            pos = m.end()
        if pos < len(text):
            self._encode_plain(text[pos:], ids)
        return ids

    def decode(self, ids) -> str:
        out = bytearray()
        for i in ids:
            t = self.id_to_token[int(i)]
            if t in self.added:
                out += t.encode("utf-8")
            else:
                out += bytes(self.byte_unmap[c] for c in t)
        return out.decode("utf-8", errors="replace")
