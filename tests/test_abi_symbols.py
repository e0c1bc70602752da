"""The shared library loads without a GPU and exports every symbol include/*.h declares
(no compute calls here).  CPU only."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from qwen3_tts_axera_russian_amd import build
    return ctypes.CDLL(build.build())


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b((?:wrapper|cp|voc|q3e|q3|tfe)_[a-z0-9_]+)\s*\(", src)))


@pytest.mark.parametrize("header", ["qwen3tts_talker.h", "qwen3tts_cp.h", "qwen3tts_voc.h", "qwen3tts_engine.h", "qwen3tts_text.h"])
def test_every_declared_symbol_is_exported(lib, header):
    names = _declared(header)
    assert len(names) >= 5
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_test_hooks_are_not_in_the_product_library(lib):
    """csrc/q3_test_api.hip (q3t_*) is built into lib/libqwen3tts_test.so, which links against the product
    library; neither libqwen3tts.so nor its llama_wrapper.so copy carries a test or diagnostic hook."""
    import subprocess
    from qwen3_tts_axera_russian_amd import LIB_PATH, TEST_LIB_PATH
    def exported(path):
        out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
        return {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    prod = exported(LIB_PATH)
    assert not [n for n in prod if n.startswith("q3t_")]
    assert not [n for n in prod if "handoff" in n or "bench_chain" in n]
    hooks = exported(TEST_LIB_PATH)
    for n in ("q3t_linear", "q3t_talker_sample", "q3t_bench_linear", "q3t_inspect_weights"):
        assert n in hooks
    assert {"q3_device_count", "q3_set_device"} <= prod


def test_reference_wrapper_names_present(lib):
    # the 12 entry points of dual_npu/llama_wrapper.c, bound by llama_cpp_bindings.py:41-81
    ref = ["wrapper_backend_init", "wrapper_backend_free", "wrapper_load_model", "wrapper_free_model",
           "wrapper_model_n_embd", "wrapper_create_context", "wrapper_free_context", "wrapper_kv_clear",
           "wrapper_decode_embd", "wrapper_state_get_size", "wrapper_state_save_file", "wrapper_state_load_file"]
    assert all(hasattr(lib, n) for n in ref)
    alias = os.path.join(ROOT, "qwen3_tts_axera_russian_amd", "lib", "llama_wrapper.so")
    assert os.path.exists(alias)  # the name llama_cpp_bindings.py:18-21 looks for


def test_fails_loudly_without_gpu_or_weights(lib, tmp_path):
    lib.wrapper_load_model.restype = ctypes.c_void_p
    lib.cp_load.restype = ctypes.c_void_p
    lib.voc_load.restype = ctypes.c_void_p
    lib.q3e_create.restype = ctypes.c_void_p
    bogus = str(tmp_path / "nope.q3w").encode()
    assert not lib.wrapper_load_model(bogus, 0)
    assert not lib.cp_load(bogus, None, 1)
    assert not lib.voc_load(bogus, 64, 1)
    assert not lib.q3e_create(bogus, 1, 64, 8)
    from qwen3_tts_axera_russian_amd.llama_cpp_bindings import LlamaCppModel
    with pytest.raises(RuntimeError):
        LlamaCppModel(str(tmp_path / "nope.q3w"))


def _queues_seen_by_a_fresh_process(preset, keep=False, via_hiplib=False):
    """(GPU_MAX_HW_QUEUES in the C environment of a fresh process after the library was loaded, its stderr)."""
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("GPU_MAX_HW_QUEUES", None)
    env.pop("Q3_KEEP_HW_QUEUES", None)
    if preset is not None:
        env["GPU_MAX_HW_QUEUES"] = preset
    if keep:
        env["Q3_KEEP_HW_QUEUES"] = "1"
    load = ("from qwen3_tts_axera_russian_amd import hiplib\nhiplib.load()\n" if via_hiplib else
            "from qwen3_tts_axera_russian_amd import build\nctypes.CDLL(build.build())\n")
    code = ("import ctypes\n" + load +
            "libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p\n"
            "v = libc.getenv(b'GPU_MAX_HW_QUEUES')\nprint(v.decode() if v else 'unset')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, text=True, capture_output=True, check=True)
    return r.stdout.strip().splitlines()[-1], r.stderr


def test_one_hardware_queue_is_asked_for_in_the_open():
    """GPU_MAX_HW_QUEUES=1 ('one hardware queue') is a process-wide HIP policy: the Python entry point
    exports it itself (silently: it is the host program); a host that dlopens the library directly gets it from the
    library's constructor, WITH one line on stderr; a value the user exported wins; Q3_KEEP_HW_QUEUES=1 keeps the
    runtime's default."""
    v, err = _queues_seen_by_a_fresh_process(None)
    assert v == "1" and err.count("GPU_MAX_HW_QUEUES=1 set for this process") == 1
    v, err = _queues_seen_by_a_fresh_process("4")
    assert v == "4" and "GPU_MAX_HW_QUEUES" not in err
    v, err = _queues_seen_by_a_fresh_process(None, keep=True)
    assert v == "unset" and "GPU_MAX_HW_QUEUES" not in err
    v, err = _queues_seen_by_a_fresh_process(None, via_hiplib=True)
    assert v == "1" and "GPU_MAX_HW_QUEUES" not in err
    v, err = _queues_seen_by_a_fresh_process(None, keep=True, via_hiplib=True)
    assert v == "unset"


def test_hot_kernels_do_not_spill_registers():
    """build.check_spills: the fused residual units (HBM-bound: scratch traffic is HBM traffic) and the frame loop's
    kernels keep everything in registers; read from the metadata notes of the built gfx950 code objects."""
    from qwen3_tts_axera_russian_amd import LIB_PATH, build
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("llvm-readelf not installed")
    res = build.kernel_resources(LIB_PATH)
    fused = {k: v for k, v in res.items() if "resunit_kernel" in k}
    assert len(fused) == 2, sorted(fused)
    for name, (vgprs, agprs, spilled, scratch) in fused.items():
        assert spilled == 0 and scratch == 0, (name, vgprs, spilled, scratch)
    build.check_spills(LIB_PATH)   # raises on a spill in any kernel of build.NO_SPILL
    assert any("linear_kernel" in k for k in res) and any("conv_kernel" in k for k in res)
