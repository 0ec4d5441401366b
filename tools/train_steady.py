#!/usr/bin/env python3
"""Per-window step time of the training step over a longer run (is the 10-step figure a burst?).
Usage: python tools/train_steady.py [windows] [steps_per_window]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import synth  # noqa: E402
from objectdetection_amd.train import TrainStep  # noqa: E402

windows = int(sys.argv[1]) if len(sys.argv) > 1 else 8
per = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
step = TrainStep(dev, "M2", seed=0)
frames = [torch.from_numpy(synth.vkitti_frame(1000 + i, 20000)).to(dev) for i in range(16)]
for i in range(6):
    step(frames[8 * (i % 2):8 * (i % 2) + 8])
torch.cuda.synchronize()
if os.environ.get("LCF_GC_FREEZE") == "1":
    import gc
    gc.collect()
    gc.freeze()
    if os.environ.get("LCF_GC_DISABLE") == "1":
        gc.disable()
out = []
for w in range(windows):
    t = time.perf_counter()
    for i in range(per):
        loss = step(frames[8 * (i % 2):8 * (i % 2) + 8])
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t) / per * 1e3)
    print(f"window {w}: {out[-1]:.2f} ms/step  loss {float(loss):.3f}  "
          f"mem {torch.cuda.memory_reserved() / 2**30:.1f} GiB", flush=True)
