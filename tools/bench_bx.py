#!/usr/bin/env python3
"""Six-product form on three exact bf16 planes (lcf_indice_conv_fwd_x6, csrc/conv_bx.h) beside the
fp32 quad-row kernel on the real rulebooks of one synthetic frame: time per layer (interleaved
rounds, min) and both kernels' distance from a float64 reference.
Usage: python tools/bench_bx.py [points]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import ops, synth  # noqa: E402
from objectdetection_amd.frontend import SparseFrontEnd, randomize_  # noqa: E402
from objectdetection_amd.spconv import SparseConvTensor  # noqa: E402
from objectdetection_amd.spconv.modules import _bn_affine  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    dev = torch.device("cuda:0")
    model = randomize_(SparseFrontEnd("M2").eval(), 0).to(dev)
    clouds = [torch.from_numpy(synth.vkitti_frame(0, n) if n <= 40000 else
                               synth.synthia_frame(0, n)).to(dev)]
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
            if ops.x6_conv_supported(conv.in_channels, conv.out_channels, rb.kvol):
                layers.append(dict(
                    name=f"{'subm' if conv.subm else 'conv'} {conv.in_channels}->"
                         f"{conv.out_channels} K={rb.k_out} kvol={rb.kvol}",
                    x=x.features, w=conv.weight, wimg=conv._weight_image(),
                    wimg3=ops.weight_image_bf16x3(conv.weight, planes=3), nbr=rb.out_nbr,
                    k_out=rb.k_out, kvol=rb.kvol, cin=conv.in_channels, cout=conv.out_channels,
                    scale=scale, shift=shift, pairs=int((rb.out_nbr >= 0).sum().item())))
            x = conv(x, _scale=scale, _shift=shift, _relu=True)
    torch.cuda.synchronize()

    def f32(l):
        return ops.conv_forward(l["x"], l["wimg"], l["nbr"], l["k_out"], l["kvol"], l["cin"],
                                l["cout"], l["scale"], l["shift"], True)

    def x6(l, out_planes):
        return ops.conv_forward_x6(l["planes"], l["wimg3"], l["nbr"], l["k_out"], l["kvol"],
                                   l["cin"], l["cout"], l["scale"], l["shift"], True,
                                   out_planes=out_planes, out=l["out3" if out_planes else "out1"])

    for l in layers:
        l["planes"] = ops.split_planes(l["x"])
        l["out3"] = torch.zeros((3, l["k_out"], l["cout"]), dtype=torch.bfloat16, device=dev)
        l["out1"] = torch.empty((l["k_out"], l["cout"]), dtype=torch.float32, device=dev)
    variants = [("f32 q4", f32), ("x6 -> planes", lambda l: x6(l, True)),
                ("x6 -> fp32", lambda l: x6(l, False))]
    reps, rounds = 10, 5
    res = {}
    for r in range(rounds):
        for vname, fn in variants:
            for li, l in enumerate(layers):
                fn(l)
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn(l)
                e1.record()
                e1.synchronize()
                res.setdefault((li, vname), []).append(e0.elapsed_time(e1) / reps * 1e3)
    print(f"{'layer':40s} {'pairs':>8s} " + " ".join(v.rjust(14) for v, _ in variants))
    tot = {v: 0.0 for v, _ in variants}
    for li, l in enumerate(layers):
        row = f"{l['name']:40s} {l['pairs']:8d} "
        for v, _ in variants:
            us = min(res[(li, v)])
            tot[v] += us
            row += f"{us:14.1f} "
        print(row)
    print("total us:", {v: round(x, 1) for v, x in tot.items()})
    # split kernel
    l = layers[0]
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.split_planes(l["x"], out=l["planes"])
    e1.record()
    e1.synchronize()
    print(f"split_planes {tuple(l['x'].shape)}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
    # distance from float64, relative to the output's max magnitude
    for l in layers:
        wd = l["w"].detach().double().reshape(l["kvol"], l["cin"], l["cout"])
        ref = torch.zeros((l["k_out"], l["cout"]), dtype=torch.float64, device=dev)
        for o in range(l["kvol"]):
            idx = l["nbr"][:, o].long()
            v = idx >= 0
            ref[v] += l["x"][idx[v]].double() @ wd[o]
        ref = torch.relu(ref * l["scale"].double() + l["shift"].double())
        e = lambda t: float((t.double() - ref).abs().max() / ref.abs().max())
        a = f32(l)
        b = x6(l, False).clone()
        c = x6(l, True).float().sum(0)
        b2 = x6(l, False)
        print(f"err vs f64 {l['name']:40s} f32 {e(a):.2e}  x6 {e(b):.2e}  x6 planes {e(c):.2e}"
              f"{'' if torch.equal(b, b2) else '  NONDET'}")


if __name__ == "__main__":
    main()
