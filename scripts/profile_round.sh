#!/bin/bash
# Evidence of a round from the CURRENT build, on the GPU box:  gpurun -- 'bash scripts/profile_round.sh r03'
# Writes everything under gpurun_out/<tag>/ ; scripts/profile_collect.py (run in the build container afterwards) copies the
# summaries into profiles/ and regenerates the tables.  One rocprofv3 run per counter set (never --pmc with trace domains).
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# (a failed step must not leave the trace database behind: gpurun copies gpurun_out/ back only below 64 MiB)
fail() { echo "$1 failed"; find $O -name "*.db" -size +20M -delete; find $O -name "*_kernel_trace.csv" -size +20M -delete; exit 1; }
# 1. per-kernel time of the benchmark command, eager launches (rocprofv3's kernel trace cannot follow the graph replay);
#    the profiler run keeps the runtime's four hardware queues (it hangs with one)
GPU_MAX_HW_QUEUES=4 Q3_NO_GRAPH=1 timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --no-timeline --no-cpu --no-longform --no-ragged > $O/bench_prof.log 2> $O/bench_prof.err || { tail -3 $O/bench_prof.err; fail "kernel trace"; }
echo "kernel trace done"
# 2. PMC passes: HBM traffic of the weight-streaming kernels and of the vocoder, MFMA utilisation of the vocoder
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_lin_$c -- python3 $R/profiles/${TAG}_pmc_linear_cmd.py > $O/pmc_lin_$c.log 2>&1 || fail "pmc linear $c"
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_voc_$c -- python3 $R/scripts/voc_pmc_cmd.py > $O/pmc_voc_$c.log 2>&1 || fail "pmc vocoder $c"
done
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/pmc_voc_mfma -- python3 $R/scripts/voc_pmc_cmd.py > $O/pmc_voc_mfma.log 2>&1 || fail "pmc vocoder mfma"
echo "pmc done"
cd $R
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python scripts/rocpd_stats.py $DB > $O/bench_kernel_stats.csv
[ -z "$DB" ] && cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
find $O/prof -name "*.db" -size +20M -delete
find $O -name "*_kernel_trace.csv" -size +20M -delete
# 3. in-graph node table of the replayed frame step (timeline build), 32 rows and one
python scripts/frame_timeline.py --batch 32 --csv $O/frame_nodes_b32.csv > $O/tl32.log 2>&1 || fail "timeline 32"
python scripts/frame_timeline.py --batch 1 --csv $O/frame_nodes_b1.csv > $O/tl1.log 2>&1 || fail "timeline 1"
echo "timeline done"
python scripts/voc_profile.py --batch 32 --exact 1 > $O/voc_per_op.log 2>&1 || fail "voc per-op"
# 4. the benchmark line itself, default flags (what the driver runs)
python bench.py > $O/bench_default.log 2> $O/bench_default.err || { tail -5 $O/bench_default.err; fail "bench"; }
echo "bench done"
tail -c 600 $O/bench_default.log
