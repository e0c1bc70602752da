"""oracle/q3_oracle.c against transformers' Qwen3Model (the layer type the reference names for the talker,
scripts/extract_talker_as_qwen3.py:89-110, and for the code predictor's core, export_code_predictor_onnx.py:30-46).

The expected values in tests/golden/hf_qwen3_golden.npz were produced by tests/golden/make_hf_golden.py running
Qwen3Model(inputs_embeds=...) in fp32 (eager attention) on the seeded weights of tests/hf_common.py, whose RMSNorm
vectors (input / post-attention / per-head q,k / final) are random so that norm placement, the q/k-norm-before-RoPE
order, the rotate-half pairing (i, i+64), the GQA head mapping (q head h -> kv head h//2) and the causal cache all
change the answer if restated wrongly.

Two tolerances, both stated here:
  * exact mode (orc_set_exact(1): no fp16 rounding of activations): fp32 round-off, rel. error <= 2e-5;
  * the device's numerics contract (GEMM inputs and K/V rounded to fp16): rel. error <= 4e-3
    (measured 1e-3), i.e. the rounding is the only difference between the contract and the HF layer.
This pins the restatement to an independent implementation; parity with llama.cpp / onnxruntime on real weights
stays unpinned at the reference boundary (no fixtures exist there)."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests import hf_common as C

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hf_qwen3_golden.npz")
EXACT_TOL, CONTRACT_TOL = 2e-5, 4e-3


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.fixture(scope="module")
def world():
    cfg = C.hf_check_config()
    t = C.make_tensors(cfg)
    g = np.load(GOLD)
    assert bytes(g["weights_sha256"]).decode() == C.digest(t), \
        "seeded weights differ from the ones the HF outputs were generated on (numpy generator changed?): " \
        "re-run tests/golden/make_hf_golden.py"
    return cfg, t, C.make_inputs(cfg), g


@pytest.mark.parametrize("exact", [True, False])
def test_talker_stack_prefill_and_decode_match_hf_qwen3(world, exact):
    cfg, t, x, g = world
    orc.set_exact(exact)
    try:
        o = orc.TalkerOracle(cfg, t, n_ctx=32)
        pre = o.forward(x["prefill"], 0, all_rows=True)
        pos = x["prefill"].shape[0]
        dec = []
        for i in range(x["decode"].shape[0]):
            dec.append(o.forward(x["decode"][i], pos + i))
    finally:
        orc.set_exact(False)
    tol = EXACT_TOL if exact else CONTRACT_TOL
    e1, e2 = rel(pre, g["talker_prefill"]), rel(np.stack(dec), g["talker_decode"])
    print(f"exact={exact}: prefill rel err {e1:.2e}, decode rel err {e2:.2e}")
    assert e1 <= tol and e2 <= tol


@pytest.mark.parametrize("exact", [True, False])
def test_code_predictor_frame_matches_hf_qwen3(world, exact):
    cfg, t, x, g = world
    orc.set_exact(exact)
    try:
        cp = orc.CpOracle(cfg, t)
        codes, margins, hid = cp.predict(x["cp_hidden"], int(x["cp_code0"]), forced=x["cp_forced"], want_hidden=True)
    finally:
        orc.set_exact(False)
    e = rel(hid, g["cp_hidden"])
    print(f"exact={exact}: code-predictor hidden (positions 1..15) rel err {e:.2e}")
    assert e <= (EXACT_TOL if exact else CONTRACT_TOL)
    # the group heads on top of the pinned hidden states: logits = lm_head_g . hidden (code_predictor_server.py:129)
    for gi in (0, 7, 14):
        want = t[f"cp.lm_head.{gi}"].astype(np.float64) @ g["cp_hidden"][gi].astype(np.float64)
        if margins[gi] > 1e-2:
            assert int(np.argmax(want)) == int(codes[gi])


def test_live_hf_model_if_importable(world):
    """Re-derive one fixture live when transformers is importable (it is in the build image): guards the
    committed file against drift of the generating script."""
    pytest.importorskip("transformers")
    torch = pytest.importorskip("torch")
    from tests.golden.make_hf_golden import hf_stack
    cfg, t, x, g = world
    with torch.no_grad():
        m = hf_stack(cfg, t, "talker", cfg.talker_layers, cfg.talker_ffn)
        r = m(inputs_embeds=torch.from_numpy(x["prefill"])[None], use_cache=False)
    assert rel(r.last_hidden_state[0].numpy(), g["talker_prefill"]) <= 1e-5
