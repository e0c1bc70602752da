#!/usr/bin/env python3
"""profiles/rNN_pmc_vocoder.md from the two counter_collection.csv files of the vocoder's PMC passes.
    python scripts/pmc_vocoder_table.py profiles/r02_pmc_vocoder_fetch_counter_collection.csv \
           profiles/r02_pmc_vocoder_write_counter_collection.csv > profiles/r02_pmc_vocoder.md"""
import collections
import csv
import sys


def load(f):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    F, W = load(sys.argv[1]), load(sys.argv[2])
    el96, el192, elout = 96 * 122880 * 32, 192 * 40960 * 32, 122880 * 32

    def per(k, acc):
        v = acc.get(k, [])
        return (sum(v) / len(v) * 1024 if v else 0.0), len(v)

    out = ["# r02 PMC traffic of the vocoder's kernels, exact-fp32 path, whole decoder table, 32 chunks (rocprofv3 --pmc, separate passes)\n",
           "Commands (GPU box, `cd /tmp && export TMPDIR=/tmp`): `rocprofv3 --pmc FETCH_SIZE --output-format csv -d DIR -- python3 scripts/voc_pmc_cmd.py`",
           "and the same with `--pmc WRITE_SIZE` (3 decodes of 32 chunks each); table: `scripts/pmc_vocoder_table.py`.  Counter unit KiB.",
           "WRITE_SIZE is exact for these stores.  FETCH_SIZE halves 16-byte-per-lane reads on gfx950 (MI355X_MICROARCH.md) -- the last",
           "conv (`conv_out1_kernel`: reads its 1.51 GB input exactly once, with 16-byte loads) is the calibration in this very run:",
           "counted 0.77 GB = half -- and is uncalibrated for the 4-byte-per-lane reads the MFMA kernels stage their input with, so the",
           "read column is given as counted and doubled.  Raw rows: r02_pmc_vocoder_fetch_counter_collection.csv,",
           "r02_pmc_vocoder_write_counter_collection.csv.\n",
           "| kernel | launches | FETCH per launch (as counted / doubled) | WRITE per launch | note |", "|---|---|---|---|---|"]
    names = sorted(set(F) | set(W), key=lambda k: -(per(k, F)[0] + per(k, W)[0]) * max(per(k, F)[1], 1))
    for k in names[:15]:
        f, n = per(k, F)
        w, _ = per(k, W)
        note = ""
        if "resunit_kernel<3>" in k:
            note = (f"fused residual unit, 96 ch x 122 880 x 32: {el96 * 4 / 1e9:.2f} GB in, same out -> {f / el96:.2f} "
                    f"(x2: {2 * f / el96:.2f}) + {w / el96:.2f} B/element")
        if "resunit_kernel<6>" in k:
            note = (f"fused residual unit, 192 ch x 40 960 x 32: {el192 * 4 / 1e9:.2f} GB in, same out -> {f / el192:.2f} "
                    f"(x2: {2 * f / el192:.2f}) + {w / el192:.2f} B/element")
        if "conv_out1" in k:
            note = f"last conv: 1.51 GB in (16-byte loads: counted at half), {elout * 4 / 1e6:.1f} MB out"
        if "voc_attn_tile" in k:
            note = "8.4 MB of output; the rest is its 272 B/thread private array (scratch)"
        name = k.replace("void ", "").replace("q3::", "")[:60]
        out.append(f"| `{name}` | {n} | {f / 1e6:.1f} / {2 * f / 1e6:.1f} MB | {w / 1e6:.1f} MB | {note} |")
    out.append("\nRound 1 ran a residual unit as two launches (three with the copy kept for the residual) at 24 B/element with a 3.2-3.8x")
    out.append("re-read of the 7-tap conv's input (profiles/r01_pmc_vocoder.md).  The fused unit writes each element once (4.00 B, exact)")
    out.append("and reads 3.9-5.3 B per element as counted (7.9-10.5 if these reads are tallied at half like wide reads): 8-14.5 B/element,")
    out.append("the residual re-read and the halo columns of neighbouring tiles being served from L2 / Infinity Cache.  No kernel of")
    out.append("`build.NO_SPILL` spills (the build fails otherwise); round 1's 12 spilled registers of the 96-channel variant are gone.")
    print("\n".join(out))


if __name__ == "__main__":
    main()
