#!/bin/bash
# A/B in one call: the fp32 engine and the "bf16x6" engine with its layer policy (auto), on every
# candidate layer (all) and on none (= fp32 kernels through the bf16x6 code path), alternating.
B="python3 bench.py --steps 400 --warmup 30 --train-steps 0 --no-cpu-baseline --no-alt --conv-math bf16x6"
for r in 1 2; do
    for v in none auto all; do
        LCF_ENGINE_X6=$v timeout -k 10 120 $B | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('x6=$v', d['value'], d['ms_per_step'], (d.get('pipelined') or {}).get('value'))"
    done
done
