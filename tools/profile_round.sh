#!/bin/bash
# Evidence passes of a round, on the GPU box:  bash tools/profile_round.sh round3
# 1 x kernel trace + stats of the DEFAULT bench command, 4 x PMC passes of a short single-stream eager run (separate
# passes: gpurun refuses --pmc together with the other trace domains; FETCH_SIZE and WRITE_SIZE do not fit one pass).
# Summaries are written into profiles/<tag>_* by the tools/pmc_* scripts.  The program itself comes after `--`.
set -u
TAG=${1:-round3}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_prof
mkdir -p "$OUT" profiles
PMCARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-roofline --no-overlap --no-graph --no-other-configs --chains 1"
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace" -o run -- python3 bench.py --no-cpu-baseline --no-vae --no-other-configs --chains 1 --shapes > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace1s" -o run -- python3 bench.py --no-cpu-baseline --no-vae --no-roofline --no-overlap --no-graph --no-other-configs --chains 1 > "$OUT/trace1s_bench.json" 2> "$OUT/trace1s.err"
rocprofv3 --pmc FETCH_SIZE -f csv -d "$OUT/fetch" -o run -- python3 bench.py $PMCARGS > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE -f csv -d "$OUT/write" -o run -- python3 bench.py $PMCARGS > /dev/null 2> "$OUT/write.err"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -f csv -d "$OUT/sq" -o run -- python3 bench.py $PMCARGS > /dev/null 2> "$OUT/sq.err"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d "$OUT/mfma" -o run -- python3 bench.py $PMCARGS > /dev/null 2> "$OUT/mfma.err"
find "$OUT" -name "*.csv" | head -40
T=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1); [ -n "$T" ] && python3 tools/timeline_gaps.py "$T" > profiles/${TAG}_timeline_gaps.txt 2>&1
grep "^{" "$OUT/trace.err" > profiles/${TAG}_shapes.txt
S=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp "$S" profiles/${TAG}_bench_kernel_stats.csv && cp "$OUT/trace_bench.json" profiles/${TAG}_bench_kernel_stats_run.json
S=$(find "$OUT/trace1s" -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp "$S" profiles/${TAG}_bench_kernel_stats_single_stream.csv
F=$(find "$OUT/fetch" -name "*counter_collection.csv" | head -1); W=$(find "$OUT/write" -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python3 tools/pmc_traffic_json.py "$F" "$W" config2 fp16 > profiles/${TAG}_pmc_traffic.json
Q=$(find "$OUT/sq" -name "*counter_collection.csv" | head -1); [ -n "$Q" ] && python3 tools/pmc_sq_summary.py "$Q" > profiles/${TAG}_pmc_sq_summary.txt
M=$(find "$OUT/mfma" -name "*counter_collection.csv" | head -1); [ -n "$M" ] && python3 tools/pmc_mfma_summary.py "$M" > profiles/${TAG}_mfma_util.txt
for f in bench_kernel_stats.csv bench_kernel_stats_run.json bench_kernel_stats_single_stream.csv pmc_traffic.json pmc_sq_summary.txt mfma_util.txt timeline_gaps.txt shapes.txt; do cp profiles/${TAG}_$f gpurun_out/ 2>/dev/null; done  # (only what this script wrote)
ls -la profiles/${TAG}_*
# the raw traces stay on the box: gpurun merges at most 64 MiB back
rm -rf "$OUT/trace" "$OUT/trace1s" "$OUT/fetch" "$OUT/write" "$OUT/sq" "$OUT/mfma"
du -sh gpurun_out
