"""The committed measurement artefacts under profiles/ are consistent with each other and reproducible from their raw
rows (CPU only; nothing here runs a kernel)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def test_vocoder_pmc_table_regenerates_from_its_raw_counter_rows():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "pmc_vocoder_table.py"),
                                   os.path.join(P, "r02_pmc_vocoder_fetch_counter_collection.csv"),
                                   os.path.join(P, "r02_pmc_vocoder_write_counter_collection.csv")], text=True)
    doc = open(os.path.join(P, "r02_pmc_vocoder.md")).read()
    assert doc.startswith(out.rstrip("\n")), "profiles/r02_pmc_vocoder.md is not what scripts/pmc_vocoder_table.py prints"
    # the fused residual units write every element exactly once (WRITE_SIZE is exact): 4.00 B per element
    rows = [ln for ln in out.splitlines() if "resunit_kernel" in ln]
    assert len(rows) == 2 and all("+ 4.00 B/element" in ln for ln in rows), rows


def test_committed_bench_line_carries_the_contract_keys():
    d = json.loads(open(os.path.join(P, "r02_bench_line.json")).readline())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
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


def test_frame_node_table_covers_the_whole_graph():
    import csv
    rows = list(csv.DictReader(open(os.path.join(P, "r02_frame_nodes_b32.csv"))))
    assert len(rows) == 553, len(rows)     # talker 28 layers + 15 code-predictor passes + heads
