#!/usr/bin/env python3
"""Runs a few frames of the hot path eagerly (FrameEngine without graph capture) so that
rocprofv3 --pmc passes attribute counters to each kernel.  Usage: python tools/prof_frame.py [M2]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import synth  # noqa: E402
from objectdetection_amd.frontend import SparseFrontEnd, randomize_  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "M2"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda:0")
    model = randomize_(SparseFrontEnd(name).eval(), 0).to(dev)
    clouds = [torch.from_numpy(synth.vkitti_frame(i, 20000)).to(dev) for i in range(4)]
    with torch.no_grad():
        eng = model.engine(calib=clouds, use_graph=False)
        for i in range(frames):
            eng.forward(clouds[i % 4])
        torch.cuda.synchronize()
        print("counts", eng.counts())


if __name__ == "__main__":
    main()
