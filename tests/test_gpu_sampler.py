"""The on-device talker sampler (greedy form of llamacpp_talker_server.py:163-206) against the
outputs of the REFERENCE's own _sample_token (tests/golden/frontend_golden.npz, made by running
the reference's Python).  Integer ids: exact."""
import numpy as np
import pytest

from qwen3_tts_axera_russian_amd import hiplib

pytestmark = pytest.mark.gpu


def _codec_head(golden):
    H, TD, TV, CV = (int(x) for x in golden["talker_dims"])
    r = np.random.default_rng(int(golden["talker_seed"]))
    r.standard_normal((TV, TD))  # text_embedding (skipped the same way the generator drew it)
    r.standard_normal((CV, H))   # codec_embedding
    return (0.5 * r.standard_normal((CV, H))).astype(np.float32)


def _device_sample(lib, logits, past, n_text):
    logits = np.ascontiguousarray(logits, np.float32)
    p = np.ascontiguousarray(past if len(past) else [0], np.int32)
    return lib.q3t_talker_sample(hiplib.fptr(logits), len(logits), hiplib.iptr(p), len(past), int(n_text), 0)


def test_device_sampler_matches_reference_outputs(test_lib, golden):
    head = _codec_head(golden)
    n = int(golden["sample_n"])
    ended = 0
    for ci in range(n):
        hidden = golden[f"sample_{ci}_hidden"]
        past = [int(x) for x in golden[f"sample_{ci}_past"]]
        n_text = int(golden[f"sample_{ci}_ntext"])
        ref_tok = int(golden[f"sample_{ci}_tok"])
        got = _device_sample(test_lib, hidden @ head.T, past, n_text)
        want = -1 if (ref_tok == 2150 or ref_tok >= 2048) else ref_tok
        assert got == want, f"case {ci}: device {got}, reference {ref_tok}"
        ended += want == -1
    assert ended >= 1
    got = _device_sample(test_lib, golden["sample_rep_hidden"] @ head.T, [int(x) for x in golden["sample_rep_past"]], 50)
    assert got == int(golden["sample_rep_tok"])
