#!/usr/bin/env python3
"""torch.profiler view of steady-state training steps: device time per ATen op (which torch ops
launch the elementwise / copy / reduce kernels around the liblcf kernels).
Usage: python tools/tprof_train.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import synth  # noqa: E402
from objectdetection_amd.train import TrainStep  # noqa: E402

dev = torch.device("cuda:0")
step = TrainStep(dev, "M2", conv_math="bf16")
frames = [torch.from_numpy(synth.vkitti_frame(1000 + i, 20000)).to(dev) for i in range(16)]
for i in range(4):
    step(frames[:8])
torch.cuda.synchronize()
n = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    for i in range(n):
        step(frames[8 * (i % 2):8 * (i % 2) + 8])
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45,
                                max_name_column_width=60))
