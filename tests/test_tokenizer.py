"""Byte-level BPE (qwen3_tts_axera_russian_amd/tokenizer.py) against the `tokenizers` library configured as
the Qwen2 family is (NFC normaliser, the family's split regex, byte-level mapping, BPE without unknowns):
a vocabulary trained here on a small ru/en corpus, then identical ids on the benchmark prompts and on
edge cases.  The reference loads its tokenizer by hub name (llamacpp_talker_server.py:96-100), which cannot
run offline; the real vocab.json/merges.txt are not in the reference, so the real-vocabulary ids are
parity unpinned and this test pins the algorithm."""
import json

import pytest

from qwen3_tts_axera_russian_amd.tokenizer import PRETOKENIZE, ByteLevelBPE

CORPUS = [
    "Привет, как дела? Сегодня хорошая погода для прогулки.",
    "The quick brown fox jumps over the lazy dog. It's 12:30, isn't it?",
    "Синтез речи на графическом ускорителе работает быстрее реального времени.",
    "Speech synthesis on the accelerator runs 27x faster than real time!",
    "Она сказала: «Я приду в 7 часов», — и ушла…  \n\nНовая строка.\tTab.",
    "We'll, they've, I'm, you're, he'd — don't SHOUT.   Trailing spaces   ",
    "Числа 1234567890 и 3.14159; e-mail: user@example.com; URL https://example.com/a?b=c",
] * 3


@pytest.fixture(scope="module")
def trained(tmp_path_factory):
    tk = pytest.importorskip("tokenizers")
    from tokenizers import Regex, Tokenizer, decoders, models, normalizers, pre_tokenizers, trainers
    tok = Tokenizer(models.BPE())
    tok.normalizer = normalizers.NFC()
    tok.pre_tokenizer = pre_tokenizers.Sequence([
        pre_tokenizers.Split(Regex(PRETOKENIZE), behavior="isolated", invert=False),
        pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
    tok.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=600, initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), special_tokens=[])
    tok.train_from_iterator(CORPUS, trainer)
    tok.add_special_tokens(["<|im_start|>", "<|im_end|>", "<tts_pad>"])
    d = tmp_path_factory.mktemp("tok")
    tok.model.save(str(d))                      # vocab.json + merges.txt
    added = {str(tok.token_to_id(t)): {"content": t} for t in ["<|im_start|>", "<|im_end|>", "<tts_pad>"]}
    (d / "tokenizer_config.json").write_text(json.dumps({"added_tokens_decoder": added}))
    return tok, ByteLevelBPE.from_dir(str(d))


CASES = CORPUS[:7] + [
    "", " ", "  leading", "ёж и Ёлка — ИЙ й", "é vs é (NFC)", "mixed РУС/eng 42км/ч",
    "<|im_start|>assistant\nПривет<|im_end|>", "emoji 🙂 and 中文 bytes", "a\r\nb\n\n\nc   \n", "'S 'T 'Re 'VE",
]


@pytest.mark.parametrize("text", CASES)
def test_same_ids_as_tokenizers(trained, text):
    ref, mine = trained
    assert mine.encode(text) == ref.encode(text, add_special_tokens=False).ids


def test_round_trip(trained):
    _, mine = trained
    for text in CASES:
        import unicodedata
        assert mine.decode(mine.encode(text)) == unicodedata.normalize("NFC", text)
