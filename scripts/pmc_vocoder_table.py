#!/usr/bin/env python3
"""profiles/rNN_pmc_vocoder.md from the two counter_collection.csv files of the vocoder's PMC passes.
    python scripts/pmc_vocoder_table.py profiles/r03_pmc_vocoder_fetch_counter_collection.csv \
           profiles/r03_pmc_vocoder_write_counter_collection.csv [profiles/r03_pmc_vocoder_mfma_counter_collection.csv] r03 \
           > profiles/r03_pmc_vocoder.md"""
import collections
import csv
import sys


def load(f):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def mfma_section(path, tag):
    """SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128) per kernel: busy cycles of the 1 024 matrix pipes over the active cycles of
    the 8 XCDs x 128 pipes each."""
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(path)):
        per[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            n[r["Kernel_Name"]] += 1
    tot_act = sum(v["GRBM_GUI_ACTIVE"] for v in per.values())
    tot_busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] for v in per.values())
    out = ["\n## MFMA utilisation by counters (same workload, one more pass)\n",
           "`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -- python3 scripts/voc_pmc_cmd.py`",
           f"(raw rows: {tag}_pmc_vocoder_mfma_counter_collection.csv).  `SQ_VALU_MFMA_BUSY_CYCLES` sums the busy cycles of the 1 024 matrix",
           "pipes, `GRBM_GUI_ACTIVE` sums the active cycles of the 8 XCDs, so utilisation = MFMA_BUSY / (GUI_ACTIVE x 128).\n",
           "| kernel | launches | share of the vocoder's GPU cycles | MFMA pipes busy |", "|---|---|---|---|"]
    for k in sorted(per, key=lambda k: -per[k]["GRBM_GUI_ACTIVE"])[:14]:
        v = per[k]
        name = k.replace("void ", "").replace("q3::", "")[:58]
        out.append(f"| `{name}` | {n[k]} | {100 * v['GRBM_GUI_ACTIVE'] / tot_act:.1f} % | {100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] * 128):.1f} % |")
    out.append(f"| all kernels of a decode | | 100 % | {100 * tot_busy / (tot_act * 128):.1f} % |")
    return out


def main():
    F, W = load(sys.argv[1]), load(sys.argv[2])
    tag = sys.argv[-1] if not sys.argv[-1].endswith(".csv") else "r03"
    mfma_csv = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3].endswith(".csv") else None
    # columns of the 96- / 192-channel stages and of the output for 64 frames with the decoder family's trims
    el96, el192, elout = 96 * 122325 * 32, 192 * 40776 * 32, 122325 * 32

    def per(k, acc):
        v = acc.get(k, [])
        return (sum(v) / len(v) * 1024 if v else 0.0), len(v)

    out = [f"# {tag} PMC traffic of the vocoder's kernels, exact-fp32 path, whole decoder table, 32 chunks (rocprofv3 --pmc, separate passes, final {tag} build)\n",
           "Commands (GPU box, `cd /tmp && export TMPDIR=/tmp`): `rocprofv3 --pmc FETCH_SIZE --output-format csv -d DIR -- python3 scripts/voc_pmc_cmd.py`",
           "and the same with `--pmc WRITE_SIZE` (3 decodes of 32 chunks each); table: `scripts/pmc_vocoder_table.py`.  Counter unit KiB.",
           "WRITE_SIZE is exact for these stores.  FETCH_SIZE halves 16-byte-per-lane reads on gfx950 (MI355X_MICROARCH.md) -- the last",
           "conv (`conv_out1_kernel`: reads its 1.51 GB input exactly once, with 16-byte loads) is the calibration in this very run:",
           "counted 0.77 GB = half -- and is uncalibrated for the 4-byte-per-lane reads the MFMA kernels stage their input with, so the",
           f"read column is given as counted and doubled.  Raw rows: {tag}_pmc_vocoder_fetch_counter_collection.csv,",
           f"{tag}_pmc_vocoder_write_counter_collection.csv.\n",
           "| kernel | launches | FETCH per launch (as counted / doubled) | WRITE per launch | note |", "|---|---|---|---|---|"]
    names = sorted(set(F) | set(W), key=lambda k: -(per(k, F)[0] + per(k, W)[0]) * max(per(k, F)[1], 1))
    for k in names[:15]:
        f, n = per(k, F)
        w, _ = per(k, W)
        note = ""
        if "resunit_kernel<3>" in k:
            note = (f"fused residual unit, 96 ch x 122 325 x 32: {el96 * 4 / 1e9:.2f} GB in, same out -> {f / el96:.2f} "
                    f"(x2: {2 * f / el96:.2f}) + {w / el96:.2f} B/element")
        if "resunit_kernel<6>" in k:
            note = (f"fused residual unit, 192 ch x 40 776 x 32: {el192 * 4 / 1e9:.2f} GB in, same out -> {f / el192:.2f} "
                    f"(x2: {2 * f / el192:.2f}) + {w / el192:.2f} B/element")
        if "conv_out1" in k:
            note = f"last conv: 1.51 GB in (16-byte loads: counted at half), {elout * 4 / 1e6:.1f} MB out"
        if "voc_attn_tile" in k:
            note = "8.4 MB of output per launch (round 2's 305 MB of scratch writes are gone: accumulators in registers)"
        name = k.replace("void ", "").replace("q3::", "")[:60]
        out.append(f"| `{name}` | {n} | {f / 1e6:.1f} / {2 * f / 1e6:.1f} MB | {w / 1e6:.1f} MB | {note} |")
    out.append("\nThe fused residual units write each element once (4.00 B, exact) and read ~4-5 B per element as counted: the residual")
    out.append("re-read and the halo columns of neighbouring tiles are served from L2 / Infinity Cache.  No kernel of `build.NO_SPILL` spills")
    out.append("(the build fails otherwise).")
    if mfma_csv:
        out += mfma_section(mfma_csv, tag)
    print("\n".join(out))


if __name__ == "__main__":
    main()
