#!/bin/bash
# After `bash tools/profile_round.sh <tag>` on the GPU box (same call): copies the summaries that
# DESIGN.md and bench.py's roofline object cite into profiles/ under the round's tag.
set -e -o pipefail
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p profiles
cp $(find $out/kt -name "*kernel_stats.csv" | sort | tail -1) profiles/${tag}_bench_M2_engine_kernel_stats.csv
cp $(find $out/kt_x3 -name "*kernel_stats.csv" | sort | tail -1) profiles/${tag}_bench_M2_engine_bf16x3_kernel_stats.csv
tail -1 $out/line_under_rocprof.json > profiles/${tag}_bench_M2_engine_line_under_rocprof.json
tail -1 $out/line_x3_under_rocprof.json > profiles/${tag}_bench_M2_engine_bf16x3_line_under_rocprof.json
tail -1 $out/default_line.json > profiles/${tag}_bench_M2_default_line.json
python3 tools/trace_frame.py $(find $out/kt -name "*kernel_trace.csv" | sort | tail -1) > profiles/${tag}_frame_timeline_under_rocprof.txt
python3 tools/pmc_summary.py $tag
ls -la profiles | grep $tag
