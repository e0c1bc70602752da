#!/usr/bin/env python3
"""Per-GPU launcher -- stands where the reference's dual_npu/launch_qwen3_tts.sh does: starts the three
servers (talker, code predictor, vocoder) for one GPU, waits for their sockets while watching the
processes (launch_qwen3_tts.sh:85-104), runs the client or stays up as a daemon, and tears everything
down on exit (:70-83).  CPU pinning (taskset) is replaced by HIP_VISIBLE_DEVICES; sockets carry a
per-GPU suffix so 8 trios can share a node.

    python -m qwen3_tts_axera_russian_amd.launch_qwen3_tts --weights qwen3tts.q3w --vocoder voc.q3w \
        --gpu 0 "Привет, как дела?"            # one shot
    python -m qwen3_tts_axera_russian_amd.launch_qwen3_tts ... --daemon
"""
from __future__ import annotations

import argparse
import os
import signal
import subprocess
import sys
import time


def wait_for_socket(path, proc, name, timeout=600.0):
    t0 = time.time()
    while time.time() - t0 < timeout:
        if os.path.exists(path):
            return
        if proc.poll() is not None:
            raise RuntimeError(f"{name} exited with code {proc.returncode} before creating {path}")
        time.sleep(0.2)
    raise RuntimeError(f"{name}: {path} did not appear within {timeout:.0f}s")


def main():
    ap = argparse.ArgumentParser(description="Qwen3-TTS launcher (one GPU)")
    ap.add_argument("text", nargs="?", default=None)
    ap.add_argument("--weights", default=os.environ.get("Q3_WEIGHTS"), required=os.environ.get("Q3_WEIGHTS") is None)
    ap.add_argument("--vocoder", default=os.environ.get("VOCODER_MODEL"), required=os.environ.get("VOCODER_MODEL") is None)
    ap.add_argument("--tokenizer", default=os.environ.get("Q3_TOKENIZER"))
    ap.add_argument("--gpu", type=int, default=int(os.environ.get("Q3_GPU", "0")))
    ap.add_argument("--temperature", type=float, default=float(os.environ.get("TEMPERATURE", "0.8")))
    ap.add_argument("--top_k", type=int, default=int(os.environ.get("TOP_K", "50")))
    ap.add_argument("--max_tokens", type=int, default=int(os.environ.get("MAX_TOKENS", "200")))
    ap.add_argument("--language", default=os.environ.get("LANGUAGE", "russian"))
    ap.add_argument("--output", default="output.wav")
    ap.add_argument("--token_ids", default=None)
    ap.add_argument("--daemon", action="store_true")
    ap.add_argument("--cp_backend", choices=["auto", "native", "python"], default=os.environ.get("CP_BACKEND", "auto"),
                    help="code-predictor server: the native binary (lib/qwen3_cp_server) when built, else the Python one "
                         "(the reference's launcher picks GGML > C++ > Python the same way, launch_qwen3_tts.sh:109-115)")
    ap.add_argument("--cp_temperature", type=float, default=float(os.environ.get("CP_TEMPERATURE", "0.1")))
    a = ap.parse_args()
    env = dict(os.environ, HIP_VISIBLE_DEVICES=str(a.gpu), PYTHONUNBUFFERED="1")
    sfx = f"_gpu{a.gpu}"
    socks = {k: f"/tmp/qwen3_{k}{sfx}.sock" for k in ("talker", "cp", "voc")}
    for p in socks.values():
        if os.path.exists(p):
            os.unlink(p)
    mod = "qwen3_tts_axera_russian_amd."
    cmds = {
        "talker": [sys.executable, "-m", mod + "llamacpp_talker_server", "--model", a.weights, "--socket", socks["talker"],
                   "--temperature", str(a.temperature), "--top_k", str(a.top_k), "--max_tokens", str(a.max_tokens)] +
                  (["--tokenizer", a.tokenizer] if a.tokenizer else []),
        "cp": [sys.executable, "-m", mod + "code_predictor_server", "--model", a.weights, "--socket", socks["cp"]],
        "voc": [sys.executable, "-m", mod + "vocoder_server", "--model", a.vocoder, "--socket", socks["voc"]],
    }
    native_cp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "qwen3_cp_server")
    if a.cp_backend == "native" or (a.cp_backend == "auto" and os.path.exists(native_cp)):
        if not os.path.exists(native_cp):
            raise SystemExit(f"{native_cp} is not built (python -m qwen3_tts_axera_russian_amd.build)")
        cmds["cp"] = [native_cp, "--weights", a.weights, "--socket", socks["cp"], "--temperature", str(a.cp_temperature),
                      "--top_k", str(a.top_k)]
    procs = {}

    def cleanup(*_):
        for p in procs.values():
            if p.poll() is None:
                p.terminate()
        for p in procs.values():
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        for s in socks.values():
            if os.path.exists(s):
                os.unlink(s)

    signal.signal(signal.SIGTERM, lambda *x: (cleanup(), sys.exit(1)))
    rc = 0
    try:
        for name, cmd in cmds.items():
            procs[name] = subprocess.Popen(cmd, env=env)
        for name in cmds:
            wait_for_socket(socks[name], procs[name], name)
        print(f"All servers up on GPU {a.gpu}: {socks}")
        if a.daemon:
            while all(p.poll() is None for p in procs.values()):
                time.sleep(1.0)
            rc = 1
        else:
            client = [sys.executable, "-m", mod + "tts_client", "--weights", a.weights, "--talker_socket", socks["talker"],
                      "--cp_socket", socks["cp"], "--voc_socket", socks["voc"], "--language", a.language,
                      "--output", a.output] + (["--token_ids", a.token_ids] if a.token_ids else []) + \
                     ([a.text] if a.text else [])
            rc = subprocess.call(client, env=env)
    except KeyboardInterrupt:
        rc = 130
    finally:
        cleanup()
    sys.exit(rc)


if __name__ == "__main__":
    main()
