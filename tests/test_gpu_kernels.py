"""Kernel-level parity of the HIP linear kernel (through the C ABI test hooks) against numpy on
fp16-rounded operands.  Tolerances: f32 accumulation of exact fp16 products, so only the
summation order differs: |err| <= 2e-4 * (sum |a*b|) is far looser than observed (~1e-6)."""
import numpy as np
import pytest

from qwen3_tts_axera_russian_amd import hiplib

pytestmark = pytest.mark.gpu


def _h(a):
    return np.ascontiguousarray(a.astype(np.float16))


def _u16(a):
    return a.view(np.uint16).ctypes.data_as(hiplib.u16p)


def _run_linear(lib, M, N, K, W16, gateup, pro, epi, x16=None, h=None, gamma=None, eps=1e-6, h_io=None, nt=0):
    y = np.zeros((M, N), np.float32) if epi != 1 else np.ascontiguousarray(h_io, dtype=np.float32).copy()
    ssq = np.zeros((M, N // 16), np.float32)
    act = np.zeros((M, N // 2), np.float16)
    rc = lib.q3t_linear(M, N, K, _u16(W16), gateup, pro, epi,
                        _u16(x16) if x16 is not None else None,
                        hiplib.fptr(h) if h is not None else None,
                        hiplib.fptr(gamma) if gamma is not None else None, eps,
                        hiplib.fptr(y), hiplib.fptr(ssq), _u16(act), nt)
    assert rc == 0
    return y, ssq, act


def _norm_fold(h, gamma, eps):
    """The folded RMSNorm of the numerics contract: GEMM input fp16((h*gamma)/16), GEMM results
    times 16*inv_rms(row)."""
    inv = 1.0 / np.sqrt((h.astype(np.float64) ** 2).mean(axis=1, keepdims=True) + eps)
    return ((h * gamma) * np.float32(0.0625)).astype(np.float16), (inv * 16.0).astype(np.float32)


@pytest.mark.parametrize("M", [1, 3, 16, 17, 32, 45, 64, 100, 129, 300])
@pytest.mark.parametrize("N,K", [(4096, 1024), (1024, 2048), (1024, 3072), (3072, 1024)])
def test_linear_store_f16(test_lib, M, N, K):
    rng = np.random.default_rng(M * 131 + N + K)
    W = _h(0.05 * rng.standard_normal((N, K)))
    x = _h(rng.standard_normal((M, K)))
    if K == 1024:
        y, _, _ = _run_linear(test_lib, M, N, K, W, 0, 0, 0, x16=x, nt=M % 2)
        ref = x.astype(np.float32) @ W.astype(np.float32).T
        np.testing.assert_allclose(y, ref, rtol=0, atol=2e-4 * np.abs(ref).max() + 1e-5)
    else:
        h0 = rng.standard_normal((M, N)).astype(np.float32)
        y, ssq, _ = _run_linear(test_lib, M, N, K, W, 0, 0, 1, x16=x, h_io=h0, nt=M % 2)
        ref = h0 + x.astype(np.float32) @ W.astype(np.float32).T
        np.testing.assert_allclose(y, ref, rtol=0, atol=2e-4 * np.abs(ref).max() + 1e-5)
        ref_ssq = (ref.astype(np.float64) ** 2).reshape(M, N // 16, 16).sum(-1)
        np.testing.assert_allclose(ssq, ref_ssq, rtol=1e-4)


@pytest.mark.parametrize("M", [1, 7, 16, 32, 40, 65, 128, 200, 971])
def test_linear_norm_prologue_and_swiglu(test_lib, M):
    rng = np.random.default_rng(900 + M)
    K, F = 1024, 3072
    h = (3.0 * rng.standard_normal((M, K))).astype(np.float32)
    gamma = (1.0 + 0.1 * rng.standard_normal(K)).astype(np.float32)
    xr, post = _norm_fold(h, gamma, 1e-6)
    xr = xr.astype(np.float32)
    # q/k/v-shaped store
    W = _h(0.05 * rng.standard_normal((4096, K)))
    y, _, _ = _run_linear(test_lib, M, 4096, K, W, 0, 1, 0, h=h, gamma=gamma)
    ref = (xr @ W.astype(np.float32).T) * post
    # a norm-input rounding flip (1 fp16 ulp on one element) moves an output by <= 2^-11*|x|*|w|
    np.testing.assert_allclose(y, ref, rtol=0, atol=1e-3 * np.abs(ref).max())
    # gate/up + SwiGLU; weights: rows [0,F) gate, [F,2F) up
    Wgu = _h(0.05 * rng.standard_normal((2 * F, K)))
    _, _, act = _run_linear(test_lib, M, 2 * F, K, Wgu, 1, 1, 2, h=h, gamma=gamma)
    g = (xr @ Wgu[:F].astype(np.float32).T) * post
    u = (xr @ Wgu[F:].astype(np.float32).T) * post
    ref_act = (g / (1.0 + np.exp(-g))) * u
    np.testing.assert_allclose(act.astype(np.float32), ref_act, rtol=2e-3, atol=2e-3 * np.abs(ref_act).max())


def test_linear_exact_integer_layout(test_lib):
    """Asymmetric small-integer operands: every product and sum is exact, so any fragment-layout
    mistake (row/col swap, k permutation) shows up as a hard mismatch."""
    M, N, K = 19, 1024, 2048
    rng = np.random.default_rng(5)
    W = rng.integers(-3, 4, size=(N, K)).astype(np.float16)
    x = rng.integers(-2, 3, size=(M, K)).astype(np.float16)
    h0 = np.zeros((M, N), np.float32)
    y, _, _ = _run_linear(test_lib, M, N, K, W, 0, 0, 1, x16=x, h_io=h0)
    np.testing.assert_array_equal(y, x.astype(np.float32) @ W.astype(np.float32).T)
