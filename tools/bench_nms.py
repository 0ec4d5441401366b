#!/usr/bin/env python3
"""Predict-time rotated NMS at the shipped settings (nms_pre_max_size 1000, nms_post_max_size 100,
nms_iou_threshold 0.01: second/configs/*.yaml) on synthetic car-sized boxes: device time of
objectdetection_amd.nms.rotate_nms (top-k + pairwise matrix + sequential pass, no host copies)
and of the IoU matrix used by evaluation.  Usage: python tools/bench_nms.py"""
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import nms as N  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    for n_anchor, pre, post, thr in [(20000, 1000, 100, 0.01), (70400, 1000, 100, 0.01),
                                     (5000, 5000, 300, 0.3)]:
        c = rng.uniform([0, -40], [70.4, 40], (n_anchor, 2))
        wh = np.tile([1.6, 3.9], (n_anchor, 1)) * rng.uniform(0.8, 1.2, (n_anchor, 2))
        a = rng.uniform(-math.pi, math.pi, (n_anchor, 1))
        boxes = torch.from_numpy(np.concatenate([c, wh, a], 1).astype(np.float32)).to(dev)
        scores = torch.from_numpy(rng.random(n_anchor).astype(np.float32)).to(dev)
        for _ in range(3):
            keep = N.rotate_nms(boxes, scores, pre, post, thr)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            keep = N.rotate_nms(boxes, scores, pre, post, thr)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps * 1e3
        print(f"rotate_nms: {n_anchor} candidates, pre {pre}, post {post}, thr {thr}: "
              f"{dt:.3f} ms per call (host-synchronous: returns the kept count), kept {keep.numel()}")
    b = boxes[:2000].contiguous()
    q = boxes[2000:2500].contiguous()
    for _ in range(3):
        N.rotate_iou(b, q)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        N.rotate_iou(b, q)
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"rotate_iou 2000 x 500: {us:.1f} us  ({2000 * 500 / us:.1f} M pairs/s)")


if __name__ == "__main__":
    main()
