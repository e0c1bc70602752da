"""Build the HIP shared library in-tree for gfx950 (cross-compiles without a GPU).

    python -m qwen3_tts_axera_russian_amd.build

Outputs (git-ignored, shipped to the GPU box by gpurun):
    lib/libqwen3tts.so     every C-ABI symbol of include/*.h (the product; no test hooks inside)
    lib/libqwen3tts_test.so  kernel-level hooks for tests/ and bench.py (csrc/q3_test_api.hip), linked
                           against the product library
    lib/llama_wrapper.so   the same file under the name the reference's
                           llama_cpp_bindings.py:18-35 looks for
    lib/qwen3_cp_server    native code-predictor server over the cp_* ABI (the reference's
                           code_predictor_cpp / code_predictor_ggml binaries)
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
SOURCES = ["q3_common.cpp", "q3_formats.cpp", "q3_kernels.hip", "q3_model.hip", "q3_talker_api.hip", "q3_cp_api.hip",
           "q3_engine.hip", "q3_voc.hip", "q3_text_api.hip"]
TEST_SOURCES = ["q3_test_api.hip"]
ARCH = os.environ.get("Q3_OFFLOAD_ARCH", "gfx950")
# kernarg preload: the leading scalar kernel arguments arrive in SGPRs at wave launch (gfx940+)
EXTRA = os.environ.get("Q3_EXTRA_HIPCC_FLAGS", "-mllvm -amdgpu-kernarg-preload-count=16").split()


def _newer(dst: str, srcs) -> bool:
    if not os.path.exists(dst):
        return False
    t = os.path.getmtime(dst)
    return all(os.path.getmtime(s) <= t for s in srcs)


# kernels that must not spill registers to scratch: the fused residual units are HBM-bound stages, scratch traffic is
# HBM traffic (round 1 shipped a 96-channel conv variant with 12 spilled registers = +0.19 GB per launch)
NO_SPILL = ("resunit_kernel", "linear_kernelILi1ELi1E", "linear_kernelILi1ELi2E", "linear_kernelILi2ELi1E",
            "linear_kernelILi2ELi2E", "attn_kernel", "attn_short_kernel", "cp_argmax_kernel", "talker_sample_kernel")


def kernel_resources(lib_path: str):
    """-> {kernel symbol: (vgprs, agprs, spilled vgprs, scratch bytes)} of the gfx950 code objects inside a built
    shared library (the metadata notes of the fat binary's AMDGPU ELF images)."""
    import re
    import struct
    import tempfile
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    d = open(lib_path, "rb").read()
    res = {}
    i = d.find(b"__CLANG_OFFLOAD_BUNDLE__")
    while i >= 0:
        n = struct.unpack_from("<Q", d, i + 24)[0]
        off = i + 32
        for _ in range(n):
            o, sz, ts = struct.unpack_from("<QQQ", d, off)
            off += 24
            triple = d[off:off + ts].decode()
            off += ts
            if ARCH in triple and sz:
                with tempfile.NamedTemporaryFile(suffix=".elf") as f:
                    f.write(d[i + o:i + o + sz])
                    f.flush()
                    notes = subprocess.run([readelf, "--notes", f.name], capture_output=True, text=True).stdout
                for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
                    def g(key, blk=blk):
                        m = re.search(r"\." + key + r":\s+(\S+)", blk)
                        return m.group(1) if m else "0"
                    res[g("name")] = (int(g("vgpr_count")), int(blk.split()[0]), int(g("vgpr_spill_count")),
                                      int(g("private_segment_fixed_size")))
        i = d.find(b"__CLANG_OFFLOAD_BUNDLE__", i + 1)
    return res


def check_spills(lib_path: str, verbose: bool = False) -> None:
    """Fail the build when a kernel of NO_SPILL spills registers; list the others that do."""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        return
    bad, others = [], []
    for name, (v, a, spill, scratch) in sorted(kernel_resources(lib_path).items()):
        if spill:
            (bad if any(k in name for k in NO_SPILL) else others).append(f"{name}: {spill} spilled VGPRs ({scratch} B scratch)")
    if verbose and others:
        print("kernels with spilled registers (not on the no-spill list):\n  " + "\n  ".join(others), flush=True)
    if bad:
        raise RuntimeError("register spills in kernels that must not spill:\n  " + "\n  ".join(bad))


def build(force: bool = False, verbose: bool = False, timeline: bool = False, refresh_timeline: bool = True) -> str:
    """timeline=True builds lib/libqwen3tts_tl.so with in-kernel time stamps (diagnostics only).  refresh_timeline=False: the
    product build leaves a stale timeline library alone (a caller that builds both side by side, __graft_entry__.build)."""
    os.makedirs(LIB, exist_ok=True)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs += [os.path.join(HERE, "..", "include", f) for f in os.listdir(os.path.join(HERE, "..", "include"))]
    tag = "_tl" if timeline else ""
    out = os.path.join(LIB, f"libqwen3tts{tag}.so")
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(LIB, os.path.basename(s) + tag + ".o")
        objs.append(o)
        if not force and _newer(o, [s] + hdrs):
            continue
        jobs.append([hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", s, "-o", o,
                     "-Wno-unused-result", "-Wno-unused-value", "-Wno-pass-failed"] + (["-DQ3_TIMELINE"] if timeline else []) + EXTRA)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:   # independent translation units: a few at a time (each hipcc is one core and < 2 GiB)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as pool:
            list(pool.map(run, jobs))
    if force or not _newer(out, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out] + objs + ["-lz"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    # test hooks: a separate library over the product one (tests/ and bench.py only)
    tout = os.path.join(LIB, f"libqwen3tts_test{tag}.so")
    tsrcs = [os.path.join(CSRC, s) for s in TEST_SOURCES]
    if force or not _newer(tout, tsrcs + hdrs + [out]):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip"] + tsrcs + \
              ["-o", tout, "-Wno-unused-result", "-Wno-unused-value", "-L" + LIB, f"-lqwen3tts{tag}", "-Wl,-rpath,$ORIGIN"] + \
              (["-DQ3_TIMELINE"] if timeline else []) + EXTRA
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    if not timeline:
        # the native code-predictor server (the reference's code_predictor_cpp / code_predictor_ggml binaries)
        srv_src = os.path.join(CSRC, "cp_server_main.cpp")
        srv = os.path.join(LIB, "qwen3_cp_server")
        if force or not _newer(srv, [srv_src, out]):
            cmd = ["g++", "-O2", "-std=c++17", srv_src, "-o", srv, "-L" + LIB, "-lqwen3tts", "-Wl,-rpath,$ORIGIN"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        check_spills(out, verbose)
        # an existing diagnostic (timeline) build must not go stale: bench.py's in-graph launch durations come from it,
        # and it must export every symbol hiplib declares
        tl_out = os.path.join(LIB, "libqwen3tts_tl.so")
        if refresh_timeline and os.path.exists(tl_out) and not _newer(tl_out, srcs + hdrs):
            build(force=force, verbose=verbose, timeline=True)
        alias = os.path.join(LIB, "llama_wrapper.so")
        if not os.path.exists(alias) or os.path.getmtime(alias) < os.path.getmtime(out):
            shutil.copyfile(out, alias)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, timeline="--timeline" in sys.argv))
