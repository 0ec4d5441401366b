#!/usr/bin/env python3
"""Times the VFE's TRAINING forward + backward alone on a batch of 8 synthetic frames (the
voxel_encoder part of a training step: ~160k points in ~140k voxels), HIP events, median of reps.
Usage: python tools/bench_vfe_train.py [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import synth  # noqa: E402
from objectdetection_amd.frontend import SparseFrontEnd, randomize_  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 30
    dev = torch.device("cuda:0")
    model = randomize_(SparseFrontEnd("M2"), 0).to(dev).train()
    frames = [torch.from_numpy(synth.vkitti_frame(1000 + i, 20000)).to(dev) for i in range(8)]
    with torch.no_grad():
        voxels, num, coors = model.voxelize(frames)
    vfe = model.voxel_feature_extractor
    g = torch.randn(voxels.shape[0], 128, device=dev)
    fw, bw = [], []
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for _ in range(reps + 5):
        for p in vfe.parameters():
            p.grad = None
        e[0].record()
        out = vfe(voxels, num, coors)
        e[1].record()
        out.backward(g)
        e[2].record()
        torch.cuda.synchronize()
        fw.append(e[0].elapsed_time(e[1]))
        bw.append(e[1].elapsed_time(e[2]))
    if "--prof" in sys.argv:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            for _ in range(4):
                for p in vfe.parameters():
                    p.grad = None
                vfe(voxels, num, coors).backward(g)
            torch.cuda.synchronize()
        ev = prof.key_averages()
        tot = sum(x.self_device_time_total for x in ev) / 4
        nk = sum(x.count for x in ev if x.self_device_time_total > 0) / 4
        print(f"device time per fwd+bwd {tot / 1e3:.3f} ms in {nk:.0f} kernels")
        print(ev.table(sort_by="self_cuda_time_total", row_limit=28, max_name_column_width=55))
    print(f"VFE training path, {voxels.shape[0]} voxels, {int(num.sum())} points: forward "
          f"{np.median(fw[5:]):.3f} ms, backward {np.median(bw[5:]):.3f} ms")


if __name__ == "__main__":
    main()
