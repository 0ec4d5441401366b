#!/usr/bin/env python3
"""Times the tile-compacted fp32 forward on the frame's real layers (HIP events, min of rounds).
LCF_LIB=<path> times an experimental build of the library (timing ablations).
Usage: python tools/bench_tc.py [points]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import _lib, ops, synth  # noqa: E402
from objectdetection_amd.frontend import SparseFrontEnd, randomize_  # noqa: E402
from objectdetection_amd.spconv import SparseConvTensor  # noqa: E402
from objectdetection_amd.spconv.modules import _bn_affine  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    if os.environ.get("LCF_LIB"):
        _lib.LIB_PATH = os.path.abspath(os.environ["LCF_LIB"])
    dev = torch.device("cuda:0")
    model = randomize_(SparseFrontEnd("M2").eval(), 0).to(dev)
    clouds = [torch.from_numpy(synth.vkitti_frame(0, n)).to(dev)]
    layers = []
    with torch.no_grad():
        voxels, num, coors = model.voxelize(clouds)
        feats = model.voxel_feature_extractor(voxels, num, coors)
        mid = model.middle_feature_extractor
        x = SparseConvTensor(feats, coors, mid.sparse_shape, 1)
        mods = list(mid.middle_conv)
        for i in range(0, len(mods), 3):
            conv, bn = mods[i], mods[i + 1]
            rb = conv.get_rulebook(x)
            scale, shift = _bn_affine(bn)
            args = (x.features, conv._weight_image(), rb.out_nbr, rb.k_out, rb.kvol,
                    conv.in_channels, conv.out_channels, scale, shift, True)
            pairs = int((rb.out_nbr >= 0).sum().item())
            if conv.in_channels >= 32 and conv.out_channels >= 32:
                layers.append((f"{'subm' if conv.subm else 'conv'} {conv.in_channels}->"
                               f"{conv.out_channels} K={rb.k_out}", args, pairs))
            x = conv(x, _scale=scale, _shift=shift, _relu=True)
    torch.cuda.synchronize()
    res = {}
    for r in range(5):
        for li, (name, args, pairs) in enumerate(layers):
            ops.conv_forward(*args)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.conv_forward(*args)
            e1.record()
            e1.synchronize()
            res.setdefault(li, []).append(e0.elapsed_time(e1) / 10 * 1e3)
    tot = 0.0
    for li, (name, args, pairs) in enumerate(layers):
        us = min(res[li])
        tot += us
        print(f"{name:30s} pairs {pairs:8d} {us:7.1f} us "
              f"{2.0 * pairs * args[5] * args[6] / us / 1e6:6.1f} TF")
    print(f"total {tot:.1f} us  ({os.environ.get('LCF_LIB', 'product build')})")


if __name__ == "__main__":
    main()
