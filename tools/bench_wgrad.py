#!/usr/bin/env python3
"""wgrad timing on the real rulebooks of one synthetic frame: fp32 MFMA vs the matrix-core forms
(bf16 operands, split hi + lo operands).  Usage: python tools/bench_wgrad.py [points] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import ops, synth  # noqa: E402
from objectdetection_amd.frontend import SparseFrontEnd, randomize_  # noqa: E402
from objectdetection_amd.spconv import SparseConvTensor  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    dev = torch.device("cuda:0")
    model = randomize_(SparseFrontEnd("M2", max_voxels=40000).eval(), 0).to(dev)
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    clouds = [torch.from_numpy(synth.vkitti_frame(i, n)).to(dev) for i in range(batch)]
    layers = []
    with torch.no_grad():
        voxels, num, coors = model.voxelize(clouds)
        feats = model.voxel_feature_extractor(voxels, num, coors)
        mid = model.middle_feature_extractor
        x = SparseConvTensor(feats, coors, mid.sparse_shape, batch)
        mods = list(mid.middle_conv)
        for i in range(0, len(mods), 3):
            conv = mods[i]
            rb = conv.get_rulebook(x)
            y = conv(x)
            g = torch.randn_like(y.features)
            pairs = int((rb.out_nbr >= 0).sum().item())
            layers.append((f"{'subm' if conv.subm else 'conv'} {conv.in_channels}->"
                           f"{conv.out_channels} K={rb.k_in}", x.features, g, rb.in_nbr, rb.kvol,
                           conv.in_channels, conv.out_channels, pairs))
            x = y
    torch.cuda.synchronize()
    print(f"{'layer':30s} {'pairs':>8s} {'fp32 us':>9s} {'bf16 us':>9s} {'x3 us':>9s}  err(bf16) err(x3)")
    tot = [0.0, 0.0, 0.0]
    for name, f, g, nbr, kvol, cin, cout, pairs in layers:
        res = []
        for planes in (0, 1, 2):
            if planes and not ops.wgrad_mc_supported(cin, cout):
                res.append(float("nan"))
                continue
            best = 1e9
            for _ in range(3):
                ops.conv_wgrad(f, g, nbr, kvol, cin, cout, planes)
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ops.conv_wgrad(f, g, nbr, kvol, cin, cout, planes)
                e1.record()
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1) * 100)
            res.append(best)
        errs = ""
        if ops.wgrad_mc_supported(cin, cout):
            d0 = ops.conv_wgrad(f, g, nbr, kvol, cin, cout, 0)
            for planes in (1, 2):
                d = ops.conv_wgrad(f, g, nbr, kvol, cin, cout, planes)
                errs += f"  {float((d - d0).abs().max() / d0.abs().max()):.1e}"
            for i in range(3):
                tot[i] += res[i]
        print(f"{name:30s} {pairs:8d} {res[0]:9.1f} {res[1]:9.1f} {res[2]:9.1f}{errs}")
    print("total over the matrix-core layers, us:", [round(t, 1) for t in tot])


if __name__ == "__main__":
    main()
