#!/bin/bash
# A/B of one environment variable's values in one call (alternating processes of bench.py).
# Usage: tools/ab_env.sh VAR v1 v2 ...
var=$1; shift
B="python3 bench.py --steps 400 --warmup 30 --train-steps 0 --no-cpu-baseline --no-alt"
for r in 1 2; do
    for v in "$@"; do
        env $var=$v timeout -k 10 120 $B | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', d['value'], d['ms_per_step'])"
    done
done
