#!/usr/bin/env python3
"""Frames/s of engine.FrameBatch (B clouds per call) beside one FrameEngine, resident clouds.
Usage: python tools/bench_batch.py [batch] [points]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import synth  # noqa: E402
from objectdetection_amd.engine import FrameBatch, FrameEngine  # noqa: E402
from objectdetection_amd.frontend import SparseFrontEnd, randomize_  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    dev = torch.device("cuda:0")
    model = randomize_(SparseFrontEnd("M2").eval(), 0).to(dev)
    clouds = [torch.from_numpy(synth.vkitti_frame(i, n)).to(dev) for i in range(B)]
    one = FrameEngine(model, calib=clouds)
    bat = FrameBatch(model, batch=B, calib=clouds)

    def rate(fn, frames, reps=300):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return frames * reps / (time.perf_counter() - t)

    for r in range(3):
        print(f"round {r}: one engine {rate(lambda: one(clouds[0]), 1):7.0f} frames/s   "
              f"FrameBatch B={B} {rate(lambda: bat(clouds), B):7.0f} frames/s", flush=True)
    bat.check()


if __name__ == "__main__":
    main()
