"""The committed measurement artefacts under profiles/ are consistent with each other and reproducible from their raw
rows (CPU only; nothing here runs a kernel)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


TAG = "r03"     # the round whose artefacts were regenerated from the final build (scripts/profile_round.sh + profile_collect.py)


def test_vocoder_pmc_table_regenerates_from_its_raw_counter_rows():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "pmc_vocoder_table.py"),
                                   os.path.join(P, f"{TAG}_pmc_vocoder_fetch_counter_collection.csv"),
                                   os.path.join(P, f"{TAG}_pmc_vocoder_write_counter_collection.csv"),
                                   os.path.join(P, f"{TAG}_pmc_vocoder_mfma_counter_collection.csv"), TAG], text=True)
    doc = open(os.path.join(P, f"{TAG}_pmc_vocoder.md")).read()
    assert doc.rstrip("\n") == out.rstrip("\n"), f"profiles/{TAG}_pmc_vocoder.md is not what scripts/pmc_vocoder_table.py prints"
    # the fused residual units write every element exactly once (WRITE_SIZE is exact): 4.00 B per element
    rows = [ln for ln in out.splitlines() if "resunit_kernel" in ln and "B/element" in ln]
    assert len(rows) == 2 and all("+ 4.00 B/element" in ln for ln in rows), rows


def test_linear_pmc_json_regenerates_from_its_raw_counter_rows(tmp_path):
    """profiles/r03_pmc_linear.json (what bench.py reports as roofline.traffic) = FETCH_SIZE x 2 + WRITE_SIZE of the raw rows."""
    import shutil
    for f in (f"{TAG}_pmc_fetch_counter_collection.csv", f"{TAG}_pmc_write_counter_collection.csv"):
        shutil.copyfile(os.path.join(P, f), tmp_path / f)
    os.makedirs(tmp_path / "profiles")
    subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "pmc_linear_table.py"), TAG,
                             str(tmp_path / f"{TAG}_pmc_fetch_counter_collection.csv"),
                             str(tmp_path / f"{TAG}_pmc_write_counter_collection.csv")], text=True, cwd=tmp_path)
    new = json.load(open(tmp_path / "profiles" / f"{TAG}_pmc_linear.json"))
    old = json.load(open(os.path.join(P, f"{TAG}_pmc_linear.json")))
    assert new == old
    gu = old["linear_kernel<2, 2, 4, 8, 1, 2, true>"]
    assert 0.95 < gu["hbm_bytes_per_launch"] / gu["algorithmic_bytes"] < 1.15      # the talker variant reads its weights once
    cp = old["linear_kernel<2, 2, 4, 8, 1, 2, false>"]
    assert cp["hbm_bytes_per_launch"] < gu["hbm_bytes_per_launch"]                 # the in-graph variant is served from the Infinity Cache
    sys.path.insert(0, ROOT)
    import bench
    t, c = bench.pmc_traffic_gateup(32)
    assert t == gu["hbm_bytes_per_launch"] and c == cp["hbm_bytes_per_launch"]


def test_profiled_kernel_names_are_kernels_of_the_built_library():
    """Stale evidence guard: every q3 kernel named in the round's PMC tables and kernel-stats CSV is a symbol of the library
    as it is built now (template arguments included) -- a table from an older build names variants that no longer exist."""
    import csv
    import re
    from qwen3_tts_axera_russian_amd import LIB_PATH, build
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        import pytest
        pytest.skip("llvm-readelf not installed")
    import re  # noqa: F401
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from kname import pretty
    syms = set(build.kernel_resources(LIB_PATH)) | set(build.kernel_resources(LIB_PATH.replace("libqwen3tts.so", "libqwen3tts_test.so")))
    built = {pretty(s_) for s_ in syms}
    named = set()
    for f in (f"{TAG}_pmc_vocoder_fetch_counter_collection.csv", f"{TAG}_pmc_vocoder_mfma_counter_collection.csv",
              f"{TAG}_pmc_fetch_counter_collection.csv"):
        for r in csv.DictReader(open(os.path.join(P, f))):
            named.add(pretty(r["Kernel_Name"]))
    for r in csv.DictReader(open(os.path.join(P, f"{TAG}_bench_kernel_stats.csv"))):
        named.add(pretty(r.get("Name") or list(r.values())[0]))
    named = {n for n in named if n and not n.startswith("__amd") and "rocclr" not in n}
    missing = sorted(n for n in named if n not in built)
    assert len(named) >= 15 and not missing, missing


def test_committed_bench_line_carries_the_contract_keys():
    d = json.loads(open(os.path.join(P, f"{TAG}_bench_line.json")).readline())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "verified", "ragged"):
        assert k in d, k
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["higher_is_better"] is True
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1.0
    assert r["measured"].startswith("in-graph")                      # not the stand-alone launch loop
    assert 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.2   # PMC traffic ~ algorithmic bytes: no re-reads
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    # value = utterances x frames x steps / wall
    assert abs(d["value"] - 32 * 64 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    assert d["longform"]["frames"] == 768 and d["longform"]["audio_s"] > 60.0 and d["longform"]["rtf"] < 0.1
    assert d["batch1"]["value"] > 0
    v = d["verified"]
    assert v["checked"] and v["ok"] and v["utterances"] == 32 and v["frames"] == 64      # the timed step's codes were checked
    assert d["ragged"]["utterances"] == 96 and d["ragged"]["value"] > 0


def test_frame_node_table_covers_the_whole_graph():
    import csv
    rows = list(csv.DictReader(open(os.path.join(P, f"{TAG}_frame_nodes_b32.csv"))))
    assert len(rows) == 553, len(rows)     # talker 28 layers + 15 code-predictor passes + heads
