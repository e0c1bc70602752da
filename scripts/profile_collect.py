#!/usr/bin/env python3
"""Copy the summaries scripts/profile_round.sh left under gpurun_out/<tag>/ into profiles/ (tracked) and regenerate the
tables derived from raw rows.   python scripts/profile_collect.py r03"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def first(pattern):
    g = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)     # gpurun MERGES into gpurun_out/: take the newest run
    if not g:
        raise SystemExit(f"nothing matches {pattern}")
    return g[-1]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    O, P = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
    cp = lambda src, dst: shutil.copyfile(src, os.path.join(P, dst))
    # kernel-stats summary with readable names (rocprofv3 prints kernels with plain-pointer arguments mangled, with a .kd suffix)
    import csv as _csv
    rows = list(_csv.DictReader(open(os.path.join(O, "bench_kernel_stats.csv"))))
    key = "Name" if rows and "Name" in rows[0] else list(rows[0].keys())[0]
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from kname import pretty
    for r in rows:
        r[key] = pretty(r[key])
    with open(os.path.join(P, f"{tag}_bench_kernel_stats.csv"), "w", newline="") as fh:
        w = _csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    cp(first(os.path.join(O, "pmc_lin_FETCH_SIZE", "**", "*counter_collection.csv")), f"{tag}_pmc_fetch_counter_collection.csv")
    cp(first(os.path.join(O, "pmc_lin_WRITE_SIZE", "**", "*counter_collection.csv")), f"{tag}_pmc_write_counter_collection.csv")
    cp(first(os.path.join(O, "pmc_voc_FETCH_SIZE", "**", "*counter_collection.csv")), f"{tag}_pmc_vocoder_fetch_counter_collection.csv")
    cp(first(os.path.join(O, "pmc_voc_WRITE_SIZE", "**", "*counter_collection.csv")), f"{tag}_pmc_vocoder_write_counter_collection.csv")
    cp(first(os.path.join(O, "pmc_voc_mfma", "**", "*counter_collection.csv")), f"{tag}_pmc_vocoder_mfma_counter_collection.csv")
    cp(os.path.join(O, "frame_nodes_b32.csv"), f"{tag}_frame_nodes_b32.csv")
    cp(os.path.join(O, "frame_nodes_b1.csv"), f"{tag}_frame_nodes_b1.csv")
    cp(os.path.join(O, "voc_per_op.log"), f"{tag}_vocoder_per_op.txt")
    line = [ln for ln in open(os.path.join(O, "bench_default.log")) if ln.startswith("{")][-1]
    json.loads(line)
    open(os.path.join(P, f"{tag}_bench_line.json"), "w").write(line)
    # the counter CSVs keep only the columns the tables read (rocprofv3 writes ~20; the raw rows stay reproducible)
    import csv
    for f in glob.glob(os.path.join(P, f"{tag}_pmc_*counter_collection.csv")):
        rows = list(csv.DictReader(open(f)))
        keep = [c for c in ("Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
                            "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp")
                if rows and c in rows[0]]
        with open(f, "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=keep)
            w.writeheader()
            for r in rows:
                if "q3::" in r["Kernel_Name"] or "q3" in r["Kernel_Name"]:
                    w.writerow({k: r[k] for k in keep})
    run = lambda *a: subprocess.check_output([sys.executable] + list(a), text=True, cwd=ROOT)
    run("scripts/pmc_linear_table.py", tag, f"profiles/{tag}_pmc_fetch_counter_collection.csv", f"profiles/{tag}_pmc_write_counter_collection.csv")
    md = run("scripts/pmc_vocoder_table.py", f"profiles/{tag}_pmc_vocoder_fetch_counter_collection.csv",
             f"profiles/{tag}_pmc_vocoder_write_counter_collection.csv", f"profiles/{tag}_pmc_vocoder_mfma_counter_collection.csv", tag)
    open(os.path.join(P, f"{tag}_pmc_vocoder.md"), "w").write(md)
    tl = {}
    for b in (32, 1):
        log = open(os.path.join(O, f"tl{b}.log")).read()
        tl[b] = log[log.index(f"B={b}:"):] if f"B={b}:" in log else log
    open(os.path.join(P, f"{tag}_frame_nodes.md"), "w").write(
        f"# In-graph per-node timing of one replayed frame step (scripts/frame_timeline.py, timeline build; MI355X), final {tag} build\n\n"
        "```\n" + tl[32].rstrip() + "\n" + tl[1].rstrip() + "\n```\n\n"
        "The timeline build stamps the device clock twice per workgroup, which lengthens every node by ~0.8 us (the product build's\n"
        f"frame time is `roofline_step.avg_launch_ms` of {tag}_bench_line.json).  Per-node rows: {tag}_frame_nodes_b32.csv, {tag}_frame_nodes_b1.csv.\n")
    print("collected into profiles/:", sorted(os.path.basename(f) for f in glob.glob(os.path.join(P, f"{tag}_*"))))


if __name__ == "__main__":
    main()
