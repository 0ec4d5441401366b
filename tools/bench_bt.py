#!/usr/bin/env python3
"""The bf16-row training kernels alone on a batch-8 frame's real rulebooks (python tools/bench_bt.py
[level]): level 1 = 32 ch at ~348k rows, 2 = 64 ch at ~275k rows, 3 = 64 ch at ~136k rows.
LCF_BT_GROUPS / other build knobs are read by the library."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import ops, synth  # noqa: E402
from objectdetection_amd.frontend import SparseFrontEnd  # noqa: E402

level = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
front = SparseFrontEnd("M2", max_voxels=17000).to(dev)
clouds = [torch.from_numpy(synth.vkitti_frame(1000 + i, 20000)).to(dev) for i in range(8)]
_, _, coors = front.voxelize(clouds)
shape = [41, 1600, 1408]
ind = coors.int().contiguous()
specs = [(3, 2, 1), (3, 2, 1), (3, 2, [0, 1, 1])]
for lv in range(level):
    k, s, p = specs[lv]
    rb = ops.build_rulebook(ind, 8, shape, k, s, p, 1, False)
    ind, shape = rb.out_indices, rb.out_shape
c = 32 if level == 1 else 64
rb = ops.build_rulebook(ind, 8, shape, 3, 1, 1, 1, True)
k = rb.k_out
pairs = int((rb.out_nbr >= 0).sum())
print(f"level {level}: {k} rows, {pairs} pairs ({pairs / k:.1f} per row), {c} channels")
x = torch.randn((k, c), device=dev).bfloat16()
g = torch.randn((k, c), device=dev).bfloat16()
w = torch.randn((3, 3, 3, c, c), device=dev) / (27 * c) ** 0.5
wimg = ops.weight_image_bf16(w)
in_nbr = rb.in_nbr


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


t_f = timed(lambda: ops.conv_forward_b16(x, wimg, rb.out_nbr, k, 27, c, c))
t_d = timed(lambda: ops.conv_forward_b16(g, wimg, in_nbr, k, 27, c, c, out_bf16=True))
t_w = timed(lambda: ops.conv_wgrad_b16(x, g, in_nbr, 27, c, c))
xf = x.float()
wimg_old = ops.weight_image_bf16(w)
t_old = timed(lambda: ops.conv_forward_bf16(xf, wimg_old, rb.out_nbr, k, 27, c, c))
gb = pairs * c * 2 / 1e9
print(f"fwd {t_f:.1f} us  dgrad(bf16 out) {t_d:.1f} us  wgrad {t_w:.1f} us   "
      f"[tile-compacted fp32-row kernel: {t_old:.1f} us]   gather {gb / t_f * 1e6 / 1e3:.2f} TB/s "
      f"flops {2 * pairs * c * c / t_f / 1e6:.1f} TFLOP/s")
