#!/bin/bash
# Profiles behind DESIGN.md section 5 / bench.py's roofline object (run on the GPU box from the
# repo root: bash tools/profile_round.sh <tag>).  Writes under gpurun_out/<tag>/; the summaries
# are then copied into profiles/ by hand (tools/pmc_summary.py, tools/trace_frame.py).
set -e -o pipefail
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --steps 120 --warmup 20 --train-steps 0 --no-cpu-baseline --no-alt"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o p -- $B > $out/line_under_rocprof.json
echo "kernel trace done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_x3 -o p -- $B --conv-math bf16x3 > $out/line_x3_under_rocprof.json
echo "kernel trace (bf16x3) done"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o p -- python3 tools/prof_frame.py M2 6 > $out/pmc_fetch.log 2>&1
echo "pmc fetch done"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o p -- python3 tools/prof_frame.py M2 6 > $out/pmc_write.log 2>&1
echo "pmc write done"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq -o p -- python3 tools/prof_frame.py M2 6 > $out/pmc_sq.log 2>&1
echo "pmc sq done"
timeout -k 10 400 python3 bench.py > $out/default_line.json
echo "default bench done"
tail -c 1200 $out/default_line.json
