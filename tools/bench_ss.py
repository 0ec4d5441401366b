#!/usr/bin/env python3
"""Runs the frame's three thin layers (128->16, 16->16, 16->32) on the split-slab kernel, 20 times
each; meant for rocprofv3 --kernel-trace (tools/sum_tc_trace.py prints cv_ss durations too).
LCF_LIB=<path> uses an experimental build; LCF_CONV_IMPL=3|4 pins the split-slab / thin-tile form."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import _lib, ops, synth  # noqa: E402
from objectdetection_amd.frontend import SparseFrontEnd, randomize_  # noqa: E402
from objectdetection_amd.spconv import SparseConvTensor  # noqa: E402
from objectdetection_amd.spconv.modules import _bn_affine  # noqa: E402

if os.environ.get("LCF_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["LCF_LIB"])
if os.environ.get("LCF_CONV_IMPL"):
    _lib.lib().lcf_set_option(_lib.OPT_CONV_IMPL, int(os.environ["LCF_CONV_IMPL"]))
dev = torch.device("cuda:0")
model = randomize_(SparseFrontEnd("M2").eval(), 0).to(dev)
with torch.no_grad():
    voxels, num, coors = model.voxelize([torch.from_numpy(synth.vkitti_frame(0, 20000)).to(dev)])
    feats = model.voxel_feature_extractor(voxels, num, coors)
    mid = model.middle_feature_extractor
    x = SparseConvTensor(feats, coors, mid.sparse_shape, 1)
    mods = list(mid.middle_conv)
    for i in range(0, 9, 3):
        conv, bn = mods[i], mods[i + 1]
        rb = conv.get_rulebook(x)
        scale, shift = _bn_affine(bn)
        args = (x.features, conv._weight_image(), rb.out_nbr, rb.k_out, rb.kvol,
                conv.in_channels, conv.out_channels, scale, shift, True)
        for _ in range(20):
            ops.conv_forward(*args)
            torch.cuda.synchronize()
        x = conv(x, _scale=scale, _shift=shift, _relu=True)
torch.cuda.synchronize()
