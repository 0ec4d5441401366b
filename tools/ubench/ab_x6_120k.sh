B="python3 bench.py --steps 200 --warmup 20 --train-steps 0 --no-cpu-baseline --no-alt --shape synthia --points 120000 --max-voxels 160000"
for r in 1 2; do
  timeout -k 10 200 $B | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32', d['value'], d['ms_per_step'])"
  for v in auto all; do
    LCF_ENGINE_X6=$v timeout -k 10 200 $B --conv-math bf16x6 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('x6=$v', d['value'], d['ms_per_step'])"
  done
done
