#!/bin/bash
# Profiles of the batch-8 training step (BASELINE.json configs[2]) behind bench.py's
# train.roofline (run on the GPU box from the repo root: bash tools/profile_train.sh <tag>).
# Kernel stats over steady-state steps + the three PMC passes of MI355X_MICROARCH.md's HBM
# recipe; summaries -> profiles/<tag>_train_* (tools/pmc_summary.py <tag> train).
set -e -o pipefail
tag=${1:-r03}
steps=${2:-12}
out=gpurun_out/${tag}_train
mkdir -p $out profiles
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o p -- python3 tools/prof_train.py $steps pmc > $out/kt.log 2>&1
echo "train kernel trace done"
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o p -- python3 tools/prof_train.py 2 pmc > $out/pmc_fetch.log 2>&1
echo "train pmc fetch done"
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o p -- python3 tools/prof_train.py 2 pmc > $out/pmc_write.log 2>&1
echo "train pmc write done"
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq -o p -- python3 tools/prof_train.py 2 pmc > $out/pmc_sq.log 2>&1
echo "train pmc sq done"
cp $(find $out/kt -name "*kernel_stats.csv" | sort | tail -1) profiles/${tag}_train_kernel_stats.csv
grep "ms/step" $out/kt.log > profiles/${tag}_train_ms_per_step_under_rocprof.txt
python3 tools/pmc_summary.py $tag train
ls -la profiles | grep ${tag}_train
