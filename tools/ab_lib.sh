#!/bin/bash
# A/B of two builds of liblcf.so in one call: bench.py's frame rate with the in-tree library and
# with an experimental one (LCF_LIB), alternating processes.  Usage: tools/ab_lib.sh <exp.so> [rounds]
set -e -o pipefail
exp=${1:-tools/ubench/liblcf_exp.so}
B="python3 bench.py --steps 400 --warmup 30 --train-steps 0 --no-cpu-baseline --no-alt"
for r in $(seq 1 ${2:-3}); do
    timeout -k 10 120 $B | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base', d['value'], d['ms_per_step'], d['roofline']['frac'])"
    LCF_LIB=$exp timeout -k 10 120 $B | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('exp ', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
