"""Quad-row kernel, 64 -> 64 submanifold layer of one frame: launch time against the number of
32-row tiles (the table truncated to k rows).  Shows what a second round of tiles costs
(DESIGN.md section 8, round 3).  Usage: python tools/ubench/tail_probe.py"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from objectdetection_amd import ops, synth
from objectdetection_amd.frontend import SparseFrontEnd, randomize_
from objectdetection_amd.spconv import SparseConvTensor
from objectdetection_amd.spconv.modules import _bn_affine
dev = torch.device("cuda:0")
model = randomize_(SparseFrontEnd("M2").eval(), 0).to(dev)
c = torch.from_numpy(synth.vkitti_frame(0, 20000)).to(dev)
with torch.no_grad():
    voxels, num, coors = model.voxelize([c])
    feats = model.voxel_feature_extractor(voxels, num, coors)
    mid = model.middle_feature_extractor
    x = SparseConvTensor(feats, coors, mid.sparse_shape, 1)
    mods = list(mid.middle_conv)
    for i in range(0, len(mods), 3):
        conv, bn = mods[i], mods[i + 1]
        rb = conv.get_rulebook(x)
        sc, sh = _bn_affine(bn)
        if conv.in_channels == 64 and conv.subm and rb.k_out > 30000:
            args = (x.features, conv._weight_image(), rb.out_nbr, rb.kvol, 64, 64, sc, sh, True)
            break
        x = conv(x, _scale=sc, _shift=sh, _relu=True)
f, w, nbr, kvol, ci, co, sc, sh, relu = args
def t(k):
    nb = nbr[:k].contiguous()
    for _ in range(3): ops.conv_forward(f, w, nb, k, kvol, ci, co, sc, sh, relu)
    best = 1e9
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.conv_forward(f, w, nb, k, kvol, ci, co, sc, sh, relu)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 100)
    return best
K = nbr.shape[0]
for k in (K, 36864, 34816, 32768, 30720, 28672, 24576, 16384, 8192):
    print(k, (k + 31) // 32, f"{t(k):.1f} us")
