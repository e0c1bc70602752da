"""End-to-end over the reference's socket protocol: the three servers (talker, code predictor,
vocoder) in threads, the client driving them; the codec ids must equal the fused engine's (same
greedy decode, same kernels) and the WAV must equal the vocoder library's output for those ids."""
import os
import threading
import time
import wave

import numpy as np
import pytest

from qwen3_tts_axera_russian_amd import weights as W
from tests.util import CACHE

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def packs():
    os.makedirs(CACHE, exist_ok=True)
    cfg = W.tiny_config(2, 2, text_vocab=512)
    cfg.text_dim = 64
    main = os.path.join(CACHE, "srv_tiny_t2c2.q3w")
    if not os.path.exists(main):
        W.write_synthetic(main, cfg, seed=1234, parts=("talker", "cp", "text"))
    voc = os.path.join(CACHE, "srv_voc_tiny.q3w")
    if not os.path.exists(voc):
        W.write_pack(voc, {"voc_chunk": 64.0}, W.make_synthetic_voc(W.tiny_voc_config(), seed=7))
    return main, voc, cfg


def _wait(path):
    for _ in range(200):
        if os.path.exists(path):
            return
        time.sleep(0.05)
    raise RuntimeError(f"{path} did not appear")


def test_three_servers_and_client(gpu_lib, packs, tmp_path):
    from qwen3_tts_axera_russian_amd.code_predictor_server import CodePredictorServer
    from qwen3_tts_axera_russian_amd.engine import FrameEngine
    from qwen3_tts_axera_russian_amd.llamacpp_talker_server import Qwen3TTSTalkerServer
    from qwen3_tts_axera_russian_amd.tts_client import Qwen3TTSClient
    from qwen3_tts_axera_russian_amd.vocoder_server import VocoderServer
    main, voc, cfg = packs
    socks = {k: str(tmp_path / f"{k}.sock") for k in ("talker", "cp", "voc")}
    talker = Qwen3TTSTalkerServer(main, socket_path=socks["talker"], temperature=0.0, max_tokens=70,
                                  kv_cache_dir=str(tmp_path), n_ctx=128, install_signal_handlers=False)
    cp = CodePredictorServer(main, socket_path=socks["cp"], temperature=0.0, install_signal_handlers=False)
    vs = VocoderServer(voc, socks["voc"], install_signal_handlers=False)
    threads = [threading.Thread(target=s.serve, daemon=True) for s in (talker, cp, vs)]
    for t in threads:
        t.start()
    for p in socks.values():
        _wait(p)
    ids = [5, 17, 200, 33, 41, 7, 90, 120, 64, 3, 11, 250, 77, 8, 19, 300, 45, 60, 2, 150, 99, 21, 13, 55, 180]
    client = Qwen3TTSClient(socks["talker"], socks["cp"], socks["voc"], weights=main)
    wav = str(tmp_path / "out.wav")
    codes, audio = client.synthesize("ignored", "russian", wav, streaming=False, token_ids=ids)
    n = codes.shape[0]
    assert 1 <= n <= 70 and codes.shape[1] == 16 and (codes >= 0).all() and (codes < 2048).all()
    with wave.open(wav) as wf:
        assert (wf.getnchannels(), wf.getsampwidth(), wf.getframerate()) == (1, 2, 24000)
        assert wf.getnframes() == len(audio)
    # same ids from the fused on-device loop (identical kernels, greedy)
    prefix = talker._build_prefix(ids)
    eng = FrameEngine(main, max_batch=1, n_ctx=128, max_frames=80)
    eng.set_pad_embed(talker.tts_pad_embed)
    eng.start([prefix], [len(ids)], ignore_eos=False, max_frames=70)
    eng.run(70)
    ecodes, per = eng.codes()
    assert int(per[0]) == n
    np.testing.assert_array_equal(ecodes[:n, 0, :], codes)
    eng.destroy()
    # audio = the vocoder server's synthesize of those ids (multi-chunk path when n > 64)
    np.testing.assert_array_equal(audio, vs.synthesize_int16(codes.astype(np.int64)))
    # second request hits the KV prefix cache (llamacpp_talker_server.py:226-236) and reproduces the ids
    codes2, _ = client.synthesize("ignored", "russian", wav, streaming=True, token_ids=ids)
    np.testing.assert_array_equal(codes2, codes)
    for s in (talker, cp, vs):
        s._running = False
    for t in threads:
        t.join(timeout=5)


def test_batch_server_matches_single_utterance_path(gpu_lib, packs, tmp_path):
    """The batched request (one socket message, B utterances through the on-device frame loop + vocoder) returns,
    for every utterance, the codec ids of the fused engine run on the same batch and the PCM of the vocoder's chunk
    walk over those ids -- ragged lengths and an empty text included (utterances of one batch share the ragged
    prefill, whose tile shapes depend on the total row count, so ids are compared batch against the same batch); a
    request with more utterances than slots is served by continuous batching; a malformed request gets the error
    sentinel and the server keeps serving."""
    import socket
    import struct
    from qwen3_tts_axera_russian_amd import batch_server as bs
    from qwen3_tts_axera_russian_amd.engine import FrameEngine
    from qwen3_tts_axera_russian_amd.vocoder_server import VocoderServer
    main, voc, cfg = packs
    sock = str(tmp_path / "batch.sock")
    srv = bs.BatchSynthesisServer(main, voc, sock, max_batch=4, n_ctx=128, max_tokens=70, temperature=0.0,
                                  cp_temperature=0.0, install_signal_handlers=False)
    th = threading.Thread(target=srv.serve, daemon=True)
    th.start()
    _wait(sock)
    reqs = [[5, 17, 200, 33, 41, 7, 90, 120, 64, 3, 11, 250, 77, 8, 19, 300, 45, 60, 2, 150, 99, 21, 13, 55, 180],
            [9, 8, 7], [], [301, 302, 303, 304, 305, 306, 307, 308, 309, 310, 311, 312]]
    res = bs.synthesize_batch(sock, token_ids=reqs)
    assert len(res) == 4
    eng = FrameEngine(main, max_batch=4, n_ctx=128, max_frames=70)
    eng.set_pad_embed(srv.front.tts_pad_embed)
    eng.start([srv.front.build_prefix(ids) for ids in reqs], [len(ids) for ids in reqs], ignore_eos=False, max_frames=70)
    eng.run(70)
    ecodes, per = eng.codes()
    eng.destroy()
    vs = VocoderServer(voc, str(tmp_path / "unused.sock"), install_signal_handlers=False)
    for b, (codes, pcm) in enumerate(res):
        n = int(per[b])
        assert codes.shape == (n, 16) and n >= 1
        np.testing.assert_array_equal(codes, ecodes[:n, b, :])
        assert (codes >= 0).all() and (codes < 2048).all()
        np.testing.assert_array_equal(pcm, vs.synthesize_int16(codes.astype(np.int64)))
        assert len(pcm) >= min(n, 63) * 1920          # (a full 64-frame chunk is 122 325 samples: the family's trim)
    assert len({int(x) for x in per}) > 1          # the utterances really ended at different frames
    # more utterances than slots: continuous batching (q3e_refill) -- every utterance comes back, in request order,
    # with the codes it gets in a batch of its own (slots are independent; greedy decode)
    many = [reqs[0], reqs[1], reqs[3], [9, 8, 7, 6, 5], reqs[1], [301, 302, 303]]
    res6 = bs.synthesize_batch(sock, token_ids=many)
    assert len(res6) == 6
    # (a row's f32 sums depend on how many rows share its pass -- the split of K over waves follows the row count -- so
    # equality with the batch-of-4 codes holds only up to near-tie flips; the reference here is the same queue run on an
    # engine of its own: same order (longest first), same refills, bit-identical)
    eng = FrameEngine(main, max_batch=4, n_ctx=128, max_frames=70)
    eng.set_pad_embed(srv.front.tts_pad_embed)
    order = sorted(range(6), key=lambda i: -len(many[i]))
    q = eng.generate_queue([srv.front.build_prefix(many[i]) for i in order], [len(many[i]) for i in order], 70)
    eng.destroy()
    for k, i in enumerate(order):
        np.testing.assert_array_equal(res6[i][0], q[k])
        np.testing.assert_array_equal(res6[i][1], vs.synthesize_int16(q[k].astype(np.int64)))
    assert abs(len(res6[0][0]) - len(res[0][0])) <= 8 and abs(len(res6[2][0]) - len(res[3][0])) <= 8
    for codes, pcm in res6:
        assert codes.shape[0] >= 1 and (codes >= 0).all() and (codes < 2048).all() and len(pcm) >= min(codes.shape[0], 63) * 1920
    # an empty request -> error sentinel; the next request is served
    s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    s.connect(sock)
    s.sendall(bs.pack_batch_request(token_ids=[]))
    assert struct.unpack("<i", s.recv(4))[0] == -2
    s.close()
    assert len(bs.synthesize_batch(sock, token_ids=[[9, 8, 7]])) == 1
    srv._running = False
    th.join(timeout=5)
    srv.close()


def test_pipelined_batch_server_replies_are_the_unpipelined_ones(gpu_lib, packs, tmp_path):
    """--pipeline: request k is vocoded on a worker thread (one persistent vocoder workgroup per CU) while request k + 1 already
    generates.  Three clients connect at once; every one gets, on its own connection, exactly what the synchronous server
    computes for its request (ids and PCM bit for bit); a bad request in between gets the error sentinel."""
    import socket
    import struct
    from qwen3_tts_axera_russian_amd import batch_server as bs
    main, voc, cfg = packs
    reqs = [[[5, 17, 200, 33, 41, 7, 90, 120, 64, 3, 11, 250, 77, 8, 19, 300, 45, 60, 2, 150, 99, 21, 13, 55, 180], [9, 8, 7]],
            [[301, 302, 303, 304, 305, 306, 307, 308, 309, 310, 311, 312], [], [4, 4, 4, 4]],
            [[9, 8, 7, 6, 5]]]
    want = []
    ref = bs.BatchSynthesisServer(main, voc, str(tmp_path / "ref.sock"), max_batch=4, n_ctx=128, max_tokens=70, temperature=0.0,
                                  cp_temperature=0.0, install_signal_handlers=False)
    for r in reqs:
        want.append(ref.synthesize(r))
    ref.close()
    sock = str(tmp_path / "pipe.sock")
    srv = bs.BatchSynthesisServer(main, voc, sock, max_batch=4, n_ctx=128, max_tokens=70, temperature=0.0, cp_temperature=0.0,
                                  install_signal_handlers=False, pipeline=True)
    assert gpu_lib.voc_set_max_workgroups(-1) == gpu_lib.q3_device_compute_units()      # the setting the server made
    th = threading.Thread(target=srv.serve, daemon=True)
    th.start()
    _wait(sock)
    got = [None] * len(reqs)

    def client(i):
        got[i] = bs.synthesize_batch(sock, token_ids=reqs[i])
    ts = [threading.Thread(target=client, args=(i,)) for i in range(len(reqs))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    for i, r in enumerate(reqs):
        assert got[i] is not None and len(got[i]) == len(r)
        for (codes, pcm), (wc, wp) in zip(got[i], want[i]):
            np.testing.assert_array_equal(codes, wc)
            np.testing.assert_array_equal(pcm, wp)
    s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    s.connect(sock)
    s.sendall(bs.pack_batch_request(token_ids=[]))
    assert struct.unpack("<i", s.recv(4))[0] == -2
    s.close()
    again = bs.synthesize_batch(sock, token_ids=reqs[2])
    np.testing.assert_array_equal(again[0][1], want[2][0][1])
    srv._running = False
    th.join(timeout=10)
    srv.close()
    assert gpu_lib.voc_set_max_workgroups(0) == 0


def test_native_cp_server_binary(gpu_lib, packs, tmp_path):
    """The native code-predictor server (csrc/cp_server_main.cpp, the reference's code_predictor_cpp /
    code_predictor_ggml binaries): 4100 bytes in, 60 bytes out, one connection per frame; greedy codes equal
    the library's cp_predict; SIGTERM stops it and removes the socket."""
    import signal
    import socket
    import subprocess
    from qwen3_tts_axera_russian_amd import LIB_DIR, protocol as P
    from qwen3_tts_axera_russian_amd.llama_cpp_bindings import CodePredictor
    main, _, cfg = packs
    exe = os.path.join(LIB_DIR, "qwen3_cp_server")
    assert os.path.exists(exe), "build it with python -m qwen3_tts_axera_russian_amd.build"
    sock = str(tmp_path / "cp_native.sock")
    proc = subprocess.Popen([exe, "--weights", main, "--socket", sock, "--temperature", "0"],
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    try:
        _wait(sock)
        cp = CodePredictor(main, max_batch=1)
        rng = np.random.default_rng(23)
        for i in range(5):
            hidden = rng.standard_normal(1024).astype(np.float32)
            code0 = int(rng.integers(0, 2048)) if i else 100
            s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            s.connect(sock)
            s.sendall(P.pack_cp_request(hidden, code0))
            raw = P.recv_exact(s, 60)
            assert P.recv_exact(s, 1) == b""          # the server closes after the reply
            s.close()
            got = np.frombuffer(raw, dtype="<i4")
            assert got.shape == (15,)
            np.testing.assert_array_equal(got, np.asarray(cp.predict(hidden, code0)))
        cp.destroy()
        # a short request is dropped without a reply, the server keeps serving
        s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        s.connect(sock)
        s.sendall(b"\0" * 100)
        s.shutdown(socket.SHUT_WR)
        assert s.recv(4) == b""
        s.close()
    finally:
        proc.send_signal(signal.SIGTERM)
        out, _ = proc.communicate(timeout=20)
    text = out.decode(errors="replace")
    assert proc.returncode == 0, text
    assert "warmup result" in text and "Server stopped." in text
    assert not os.path.exists(sock)


def test_launcher_starts_trio_and_tears_down(gpu_lib, packs, tmp_path):
    """The per-GPU launcher (the reference's launch_qwen3_tts.sh:70-104,134-190): --daemon starts the three servers
    as processes pinned with HIP_VISIBLE_DEVICES, their per-GPU sockets appear while the launcher watches the pids,
    a request goes through, SIGTERM tears everything down and removes the sockets."""
    import signal
    import subprocess
    import sys
    main, voc, cfg = packs
    gpu = 0
    socks = [f"/tmp/qwen3_{k}_gpu{gpu}.sock" for k in ("talker", "cp", "voc")]
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    log = open(tmp_path / "launcher.log", "w")
    p = subprocess.Popen([sys.executable, "-m", "qwen3_tts_axera_russian_amd.launch_qwen3_tts", "--weights", main,
                          "--vocoder", voc, "--gpu", str(gpu), "--temperature", "0", "--cp_temperature", "0",
                          "--max_tokens", "12", "--daemon"], env=env, stdout=log, stderr=subprocess.STDOUT)
    try:
        t0 = time.time()
        while not all(os.path.exists(s) for s in socks):
            assert p.poll() is None, f"launcher exited early ({p.returncode}): {open(tmp_path / 'launcher.log').read()[-2000:]}"
            assert time.time() - t0 < 300, "sockets did not appear"
            time.sleep(0.2)
        from qwen3_tts_axera_russian_amd.tts_client import Qwen3TTSClient
        client = Qwen3TTSClient(*socks, weights=main)
        codes, audio = client.synthesize("ignored", "russian", str(tmp_path / "l.wav"), streaming=False,
                                         token_ids=[5, 17, 200, 33, 41, 7])
        assert 1 <= codes.shape[0] <= 12 and len(audio) == codes.shape[0] * 1920
    finally:
        p.send_signal(signal.SIGTERM)
        try:
            p.wait(timeout=60)
        except subprocess.TimeoutExpired:
            p.kill()
            raise
    assert p.returncode == 1                       # the launcher's SIGTERM path (cleanup, then exit 1)
    assert not any(os.path.exists(s) for s in socks)
