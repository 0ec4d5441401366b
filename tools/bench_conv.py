#!/usr/bin/env python3
"""Per-layer timing of the indice_conv forward on the real rulebooks of one synthetic frame.
Interleaved rounds in one process (MI355X guide rule 24).  Usage: python tools/bench_conv.py [points] [batch]"""
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
    if os.environ.get("LCF_LIB"):  # an experimental build of the library (timing ablations)
        _lib.LIB_PATH = os.path.abspath(os.environ["LCF_LIB"])
    dev = torch.device("cuda:0")
    model = randomize_(SparseFrontEnd("M2").eval(), 0).to(dev)
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
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
            layers.append((f"{'subm' if conv.subm else 'conv'} {conv.in_channels}->"
                           f"{conv.out_channels} K={rb.k_out}", args, pairs))
            x = conv(x, _scale=scale, _shift=shift, _relu=True)
    bf16_imgs, x3_imgs, x8_imgs = {}, {}, {}
    mods_c = [m for m in mid.middle_conv if hasattr(m, "subm")]
    for li, conv in enumerate(mods_c):
        if ops.bf16_supported(conv.in_channels, conv.out_channels):
            bf16_imgs[li] = ops.weight_image_bf16(conv.weight)
            x3_imgs[li] = ops.weight_image_bf16x3(conv.weight)
            x8_imgs[li] = ops.weight_image_bf16x3(conv.weight, planes=3)
    torch.cuda.synchronize()
    reps, rounds = 10, 5
    # slab; split-slab; tile-compacted: 1 set fixed 32-row tiles, 1 set, 2 sets (balanced tiles); bf16
    # x3_1 / x3_2: split-operand (bf16x3) form with one / two wave sets
    splits = [1, "ss", "tc1_32", "tc1", "tc2", "bf16", "x3_1", "x3_2", "x8_1", "x8_2"]
    res = {}
    for r in range(rounds):
        for sp in splits:
            L.lcf_set_option(3, 0)
            L.lcf_set_option(2, 32 if sp == "tc1_32" else 0)
            if sp == "bf16":
                L.lcf_set_option(1, 0)
            elif sp in ("x3_1", "x3_2", "x8_1", "x8_2"):
                L.lcf_set_option(3, 1 if sp.endswith("_1") else 2)
            elif sp == "ss":
                L.lcf_set_option(1, 3)
            elif sp == 1:
                L.lcf_set_option(1, 1)
            else:
                L.lcf_set_option(1, 2)
                L.lcf_set_option(3, 2 if sp == "tc2" else 1)
            for li, (name, args, pairs) in enumerate(layers):
                fwd = ops.conv_forward
                if sp == "bf16":
                    if not ops.bf16_supported(args[5], args[6]):
                        res.setdefault((li, sp), []).append(float("nan"))
                        continue
                    fwd = ops.conv_forward_bf16
                    args = (args[0], bf16_imgs[li]) + args[2:]
                if sp in ("x3_1", "x3_2", "x8_1", "x8_2"):
                    if not ops.bf16_supported(args[5], args[6]):
                        res.setdefault((li, sp), []).append(float("nan"))
                        continue
                    if sp.startswith("x8"):
                        fwd = lambda *a: ops.conv_forward_bf16x3(*a, planes=3)
                        args = (args[0], x8_imgs[li]) + args[2:]
                    else:
                        fwd = ops.conv_forward_bf16x3
                        args = (args[0], x3_imgs[li]) + args[2:]
                fwd(*args)
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fwd(*args)
                e1.record()
                e1.synchronize()
                res.setdefault((li, sp), []).append(e0.elapsed_time(e1) / reps * 1e3)
    L.lcf_set_option(1, 0)
    L.lcf_set_option(2, 0)
    L.lcf_set_option(3, 0)
    tot = {sp: 0.0 for sp in splits}
    import math
    print(f"{'layer':34s} {'pairs':>8s} " + " ".join(f"{sp}:us/TF" .rjust(16) for sp in splits))
    for li, (name, args, pairs) in enumerate(layers):
        cin, cout = args[5], args[6]
        row = f"{name:34s} {pairs:8d} "
        for sp in splits:
            us = min(res[(li, sp)])
            tot[sp] += 0.0 if math.isnan(us) else us
            row += f"{us:9.1f}/{2.0 * pairs * cin * cout / us / 1e6:5.1f} "
        print(row)
    print("total us:", {sp: round(v, 1) for sp, v in tot.items()})
    # errors against a float64 reference (per-offset gather + matmul in double), relative to the
    # output's max magnitude: fp32 MFMA kernel, bf16x3, bf16x8
    mods_w = [m.weight for m in mods_c]
    for li, (name, args, pairs) in enumerate(layers):
        if li not in x3_imgs:
            continue
        x, _, nbr, k_out, kvol, cin, cout, scale, shift, relu = args
        w = mods_w[li].detach().double().reshape(kvol, cin, cout)
        ref = torch.zeros((k_out, cout), dtype=torch.float64, device=x.device)
        for o in range(kvol):
            idx = nbr[:, o].long()
            v = idx >= 0
            ref[v] += x[idx[v]].double() @ w[o]
        ref = torch.relu(ref * scale.double() + shift.double())
        e = lambda t: float((t.double() - ref).abs().max() / ref.abs().max())
        print(f"err vs f64 {name:30s} fp32 {e(ops.conv_forward(*args)):.2e}  "
              f"x3 {e(ops.conv_forward_bf16x3(x, x3_imgs[li], *args[2:])):.2e}  "
              f"x8 {e(ops.conv_forward_bf16x3(x, x8_imgs[li], *args[2:], planes=3)):.2e}")


if __name__ == "__main__":
    main()
