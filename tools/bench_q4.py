#!/usr/bin/env python3
"""Quad-row form of the indice_conv forward (LCF_OPT_CONV_IMPL = 5, csrc/conv_q4.h) against the
tile-compacted kernel on the real rulebooks of one synthetic frame: time per layer (interleaved
rounds, min) and error against a float64 reference.
Usage: python tools/bench_q4.py [points] [batch]"""
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
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dev = torch.device("cuda:0")
    model = randomize_(SparseFrontEnd("M2").eval(), 0).to(dev)
    clouds = [torch.from_numpy(synth.vkitti_frame(i, n) if n <= 40000 else
                               synth.synthia_frame(i, n)).to(dev) for i in range(batch)]
    L = _lib.lib()
    layers = []
    with torch.no_grad():
        voxels, num, coors = model.voxelize(clouds)
        feats = model.voxel_feature_extractor(voxels, num, coors)
        mid = model.middle_feature_extractor
        x = SparseConvTensor(feats, coors, mid.sparse_shape, batch)
        mods = list(mid.middle_conv)
        for i in range(0, len(mods), 3):
            conv, bn = mods[i], mods[i + 1]
            rb = conv.get_rulebook(x)
            scale, shift = _bn_affine(bn)
            args = (x.features, conv._weight_image(), rb.out_nbr, rb.k_out, rb.kvol,
                    conv.in_channels, conv.out_channels, scale, shift, True)
            pairs = int((rb.out_nbr >= 0).sum().item())
            if conv.in_channels in (32, 64) and conv.out_channels in (32, 64):
                layers.append((f"{'subm' if conv.subm else 'conv'} {conv.in_channels}->"
                               f"{conv.out_channels} K={rb.k_out} kvol={rb.kvol}", args, pairs,
                               conv.weight))
            x = conv(x, _scale=scale, _shift=shift, _relu=True)
    torch.cuda.synchronize()
    variants = [("tc", 2, 0), ("q4x1", 5, 1), ("q4x2", 5, 2), ("q4x3", 5, 3), ("q4x4", 5, 4)]
    if os.environ.get("Q4_ONLY"):  # e.g. Q4_ONLY=q4x4 (timing ablation builds: one variant)
        variants = [v for v in variants if v[0] in os.environ["Q4_ONLY"].split(",")]
    tiles = [0] + [int(t) for t in os.environ.get("Q4_TILES", "").split(",") if t]
    reps, rounds = 10, 5
    res = {}
    for r in range(rounds):
        for vname, impl, sets in variants:
            for tile in (tiles if impl == 5 else [0]):
                L.lcf_set_option(1, impl)
                L.lcf_set_option(3, sets)
                L.lcf_set_option(2, tile)
                for li, (name, args, pairs, w) in enumerate(layers):
                    ops.conv_forward(*args)
                    e0 = torch.cuda.Event(enable_timing=True)
                    e1 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(reps):
                        ops.conv_forward(*args)
                    e1.record()
                    e1.synchronize()
                    res.setdefault((li, vname, tile), []).append(e0.elapsed_time(e1) / reps * 1e3)
    keys = [(v, t) for v, impl, _ in variants for t in (tiles if impl == 5 else [0])]
    print(f"{'layer':40s} {'pairs':>8s} " + " ".join(f"{v}/t{t}".rjust(14) for v, t in keys))
    tot = {k: 0.0 for k in keys}
    for li, (name, args, pairs, w) in enumerate(layers):
        row = f"{name:40s} {pairs:8d} "
        for v, t in keys:
            us = min(res[(li, v, t)])
            tot[(v, t)] += us
            row += f"{us:7.1f}/{2.0 * pairs * args[5] * args[6] / us / 1e6:5.1f}T "
        print(row)
    print("total us:", {f"{v}/t{t}": round(x, 1) for (v, t), x in tot.items()})
    # errors against a float64 reference, relative to the output's max magnitude
    if os.environ.get("Q4_NOERR"):
        layers = []
    for li, (name, args, pairs, w) in enumerate(layers):
        x, _, nbr, k_out, kvol, cin, cout, scale, shift, relu = args
        wd = w.detach().double().reshape(kvol, cin, cout)
        ref = torch.zeros((k_out, cout), dtype=torch.float64, device=x.device)
        for o in range(kvol):
            idx = nbr[:, o].long()
            v = idx >= 0
            ref[v] += x[idx[v]].double() @ wd[o]
        ref = torch.relu(ref * scale.double() + shift.double())
        e = lambda t: float((t.double() - ref).abs().max() / ref.abs().max())
        row = f"err vs f64 {name:40s}"
        for vname, impl, sets in variants:
            L.lcf_set_option(1, impl)
            L.lcf_set_option(3, sets)
            L.lcf_set_option(2, 0)
            a = ops.conv_forward(*args)
            b = ops.conv_forward(*args)
            row += f" {vname} {e(a):.2e}{'' if torch.equal(a, b) else ' NONDET'}"
        print(row)
    L.lcf_set_option(1, 0)
    L.lcf_set_option(2, 0)
    L.lcf_set_option(3, 0)


if __name__ == "__main__":
    main()
