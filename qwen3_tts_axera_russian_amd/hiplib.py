"""Loader of the HIP shared library (the only compute path; there is no CPU fallback)."""
from __future__ import annotations

import ctypes
import os

from . import LIB_PATH, TEST_LIB_PATH

_lib = None
_test_lib = None

c_void_p, c_int, c_float, c_char_p = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_char_p
f32p = ctypes.POINTER(ctypes.c_float)
i32p = ctypes.POINTER(ctypes.c_int32)
i64p = ctypes.POINTER(ctypes.c_int64)
u16p = ctypes.POINTER(ctypes.c_uint16)
i16p = ctypes.POINTER(ctypes.c_int16)


def _sig(lib, name, restype, argtypes):
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = restype, argtypes


def load(path: str | None = None):
    """dlopen libqwen3tts.so and declare every C-ABI signature of include/*.h."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("QWEN3TTS_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise RuntimeError(f"{p} not found: build it with `python -m qwen3_tts_axera_russian_amd.build` "
                           "(the HIP library is the only compute path)")
    # One HIP hardware queue for this process (csrc/q3_common.cpp): a process-wide runtime policy, so the
    # entry point exports it -- before the first HIP call of the process; a value the user exported wins.
    if os.environ.get("Q3_KEEP_HW_QUEUES", "0") in ("", "0"):
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "1")
    lib = ctypes.CDLL(p)
    # include/qwen3tts_talker.h
    _sig(lib, "wrapper_backend_init", None, [])
    _sig(lib, "wrapper_backend_free", None, [])
    _sig(lib, "wrapper_load_model", c_void_p, [c_char_p, c_int])
    _sig(lib, "wrapper_free_model", None, [c_void_p])
    _sig(lib, "wrapper_model_n_embd", c_int, [c_void_p])
    _sig(lib, "wrapper_create_context", c_void_p, [c_void_p, c_int, c_int, c_int, c_int])
    _sig(lib, "wrapper_create_context_slots", c_void_p, [c_void_p, c_int, c_int, c_int])
    _sig(lib, "wrapper_free_context", None, [c_void_p])
    _sig(lib, "wrapper_ctx_n_slots", c_int, [c_void_p])
    _sig(lib, "wrapper_kv_clear", None, [c_void_p])
    _sig(lib, "wrapper_decode_embd", c_int, [c_void_p, f32p, c_int, c_int, c_int, f32p])
    _sig(lib, "wrapper_decode_embd_slot", c_int, [c_void_p, c_int, f32p, c_int, c_int, c_int, f32p])
    _sig(lib, "wrapper_decode_embd_batch", c_int, [c_void_p, f32p, c_int, c_int, i32p, i32p, f32p])
    _sig(lib, "wrapper_codec_head", c_int, [c_void_p, f32p, c_int, f32p])
    _sig(lib, "wrapper_state_get_size", ctypes.c_size_t, [c_void_p])
    _sig(lib, "wrapper_state_save_file", c_int, [c_void_p, c_char_p])
    _sig(lib, "wrapper_state_load_file", c_int, [c_void_p, c_char_p])
    # include/qwen3tts_cp.h
    _sig(lib, "cp_load", c_void_p, [c_char_p, c_char_p, c_int])
    _sig(lib, "cp_free", None, [c_void_p])
    _sig(lib, "cp_hidden_size", c_int, [c_void_p])
    _sig(lib, "cp_predict", c_int, [c_void_p, f32p, ctypes.c_int32, c_float, c_int, ctypes.c_uint64, i32p])
    _sig(lib, "cp_predict_batch", c_int, [c_void_p, f32p, i32p, c_int, c_float, c_int, ctypes.c_uint64, i32p])
    _sig(lib, "cp_step", c_int, [c_void_p, f32p, c_int, f32p])
    _sig(lib, "cp_lm_head", c_int, [c_void_p, c_int, f32p, f32p])
    # include/qwen3tts_engine.h
    _sig(lib, "q3e_create", c_void_p, [c_char_p, c_int, c_int, c_int])
    _sig(lib, "q3e_free", None, [c_void_p])
    _sig(lib, "q3e_set_pad_embed", c_int, [c_void_p, f32p])
    _sig(lib, "q3e_set_chains", c_int, [c_void_p, c_int])
    _sig(lib, "q3e_set_sampling", c_int, [c_void_p, c_float, c_int, c_float, c_float, c_int, ctypes.c_uint64])
    _sig(lib, "q3e_set_forced_codes", c_int, [c_void_p, i32p, c_int])
    _sig(lib, "q3e_start", c_int, [c_void_p, c_int, f32p, i32p, i32p, c_int, c_int])
    _sig(lib, "q3e_run", c_int, [c_void_p, c_int])
    _sig(lib, "q3e_last_run_ms", c_float, [c_void_p])
    _sig(lib, "q3e_last_prefill_ms", c_float, [c_void_p])
    _sig(lib, "q3e_get_codes", c_int, [c_void_p, i32p, c_int, i32p])
    _sig(lib, "q3e_get_done", c_int, [c_void_p, i32p, i32p])
    _sig(lib, "q3e_refill", c_int, [c_void_p, c_int, i32p, f32p, i32p, i32p])
    _sig(lib, "q3e_get_hidden", c_int, [c_void_p, f32p])
    _sig(lib, "q3e_step_weight_bytes", ctypes.c_double, [c_void_p])
    # include/qwen3tts_voc.h
    _sig(lib, "voc_load", c_void_p, [c_char_p, c_int, c_int])
    _sig(lib, "voc_free", None, [c_void_p])
    _sig(lib, "voc_chunk_tokens", c_int, [c_void_p])
    _sig(lib, "voc_samples_per_token", c_int, [c_void_p])
    _sig(lib, "voc_chunk_samples", c_int, [c_void_p])
    _sig(lib, "voc_decode", c_int, [c_void_p, i64p, c_int, f32p])
    _sig(lib, "voc_synthesize", c_int, [c_void_p, i64p, c_int, i16p, i32p])
    _sig(lib, "voc_synthesize_f32", c_int, [c_void_p, i64p, c_int, f32p, i32p])
    _sig(lib, "voc_synthesize_max_samples", c_int, [c_void_p, c_int])
    _sig(lib, "voc_set_max_workgroups", c_int, [c_int])
    _sig(lib, "voc_set_exact_fp32", c_int, [c_int])
    _sig(lib, "voc_set_fused_units", c_int, [c_int])
    _sig(lib, "voc_last_decode_ms", c_float, [c_void_p])
    _sig(lib, "voc_decode_flops", ctypes.c_double, [c_void_p, c_int])
    i64p_ = ctypes.POINTER(ctypes.c_int64)
    _sig(lib, "voc_synthesize_batch", c_int, [c_void_p, i64p, i32p, c_int, i16p, ctypes.c_int64, i64p_])
    _sig(lib, "voc_synthesize_batch_f32", c_int, [c_void_p, i64p, i32p, c_int, f32p, ctypes.c_int64, i64p_])
    _sig(lib, "voc_synthesize_batch_max_samples", ctypes.c_int64, [c_void_p, i32p, c_int])
    _sig(lib, "voc_last_batch_ms", c_float, [c_void_p])
    _sig(lib, "voc_last_batch_chunks", c_int, [c_void_p])
    # include/qwen3tts_text.h
    _sig(lib, "tfe_load", c_void_p, [c_char_p, c_char_p, c_int])
    _sig(lib, "tfe_free", None, [c_void_p])
    _sig(lib, "tfe_hidden_size", c_int, [c_void_p])
    _sig(lib, "tfe_text_vocab", c_int, [c_void_p])
    _sig(lib, "tfe_embed_text", c_int, [c_void_p, i32p, c_int, f32p])
    _sig(lib, "tfe_build_prefix", c_int, [c_void_p, i32p, c_int, i32p, f32p])
    _sig(lib, "tfe_tts_pad_embed", c_int, [c_void_p, f32p])
    _sig(lib, "q3_device_count", c_int, [])
    _sig(lib, "q3_device_compute_units", c_int, [])
    _sig(lib, "q3_set_device", c_int, [c_int])
    if path is None:
        _lib = lib
    return lib


def load_test():
    """dlopen libqwen3tts_test.so: kernel-level hooks (csrc/q3_test_api.hip) for tests/ and bench.py.  The
    product never calls this."""
    global _test_lib
    if _test_lib is not None:
        return _test_lib
    load()   # the product library first: the test library links against it
    if not os.path.exists(TEST_LIB_PATH):
        raise RuntimeError(f"{TEST_LIB_PATH} not found: build it with `python -m qwen3_tts_axera_russian_amd.build`")
    lib = ctypes.CDLL(TEST_LIB_PATH)
    _sig(lib, "q3t_set_linear_tuning", c_int, [c_int, c_int, c_int])
    _sig(lib, "q3t_linear", c_int, [c_int, c_int, c_int, u16p, c_int, c_int, c_int, u16p, f32p, f32p, c_float,
                                    f32p, f32p, u16p, c_int])
    _sig(lib, "q3t_talker_sample", c_int, [f32p, c_int, i32p, c_int, c_int, c_int])
    _sig(lib, "q3t_set_linear_split_rows", c_int, [c_int])
    _sig(lib, "q3t_set_linear_narrow8", c_int, [c_int])
    _sig(lib, "q3t_set_linear_wide_tiles", c_int, [c_int])
    _sig(lib, "q3t_set_attn_short", c_int, [c_int])
    _sig(lib, "q3t_set_gemm_min_rows", c_int, [c_int])
    _sig(lib, "q3t_set_gemm_glds", c_int, [c_int])
    _sig(lib, "q3t_inspect_weights", c_int, [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, c_int])
    _sig(lib, "q3t_bench_linear", c_float, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int])
    _sig(lib, "q3t_bench_chain", c_float, [c_int, c_int, c_int, c_int, c_int, c_int])
    _test_lib = lib
    return lib


def fptr(a):
    return a.ctypes.data_as(f32p)


def iptr(a):
    return a.ctypes.data_as(i32p)
