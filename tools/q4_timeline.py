#!/usr/bin/env python3
"""TEMP: per-workgroup timestamps of cv_q4_kernel (LCF_Q4_ABLATE bit 6)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import _lib, ops, synth
from objectdetection_amd.frontend import SparseFrontEnd, randomize_
from objectdetection_amd.spconv import SparseConvTensor
from objectdetection_amd.spconv.modules import _bn_affine

dev = torch.device("cuda:0")
model = randomize_(SparseFrontEnd("M2").eval(), 0).to(dev)
clouds = [torch.from_numpy(synth.vkitti_frame(0, 20000)).to(dev)]
L = _lib.lib()
raw = ctypes.CDLL(_lib.LIB_PATH)
with torch.no_grad():
    voxels, num, coors = model.voxelize(clouds)
    feats = model.voxel_feature_extractor(voxels, num, coors)
    mid = model.middle_feature_extractor
    x = SparseConvTensor(feats, coors, mid.sparse_shape, 1)
    mods = list(mid.middle_conv)
    target = None
    for i in range(0, len(mods), 3):
        conv, bn = mods[i], mods[i + 1]
        rb = conv.get_rulebook(x)
        scale, shift = _bn_affine(bn)
        if conv.in_channels == 64 and conv.out_channels == 64 and conv.subm and target is None:
            target = (x.features, conv._weight_image(), rb.out_nbr, rb.k_out, rb.kvol, 64, 64, scale, shift, True)
        x = conv(x, _scale=scale, _shift=shift, _relu=True)
sets = int(os.environ.get("Q4_SETS", "4"))
os.environ["LCF_Q4_KS"] = os.environ.get("Q4_KS", "1")
L.lcf_set_option(1, 5); L.lcf_set_option(3, sets)
for ab in [int(a) for a in os.environ.get("Q4_ABLATES", "64").split(",")]:
    os.environ["LCF_Q4_ABLATE"] = str(ab | 64)
    for _ in range(3):
        ops.conv_forward(*target)
    torch.cuda.synchronize()
    buf = np.zeros(8 * 4096, dtype=np.uint64)
    assert raw.lcf_q4_debug_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    t = buf.reshape(4096, 8).astype(np.int64)
    t = t[t[:, 0] > 0]
    w0 = t[:, 0].min()
    start = (t[:, 0] - w0) / 100.0  # us (100 MHz)
    end = (t[:, 3] - w0) / 100.0
    setup = (t[:, 5] - t[:, 4]) / 2400.0
    loop0 = (t[:, 6] - t[:, 5]) / 2400.0
    wait = (t[:, 7] - t[:, 6]) / 2400.0
    dur = end - start
    print(f"ablate {ab}: {len(t)} workgroups; kernel span {end.max():.1f} us")
    print(f"  start: p50 {np.median(start):.1f} p90 {np.percentile(start, 90):.1f} max {start.max():.1f}")
    print(f"  duration: p10 {np.percentile(dur, 10):.1f} p50 {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} max {dur.max():.1f}")
    print(f"  setup p50 {np.median(setup):.2f}  wave-0 loop p50 {np.median(loop0):.2f} p90 {np.percentile(loop0, 90):.2f}  "
          f"wait for the other waves p50 {np.median(wait):.2f} p90 {np.percentile(wait, 90):.2f}")
    late = start > 1.0
    print(f"  workgroups starting after 1 us: {late.sum()}; their duration p50 {np.median(dur[late]) if late.any() else 0:.1f}; "
          f"first-round duration p50 {np.median(dur[~late]):.1f}")
    hist, edges = np.histogram(end, bins=10, range=(0, end.max()))
    print("  ends per decile of the span:", hist.tolist())

    b2 = np.zeros(128, dtype=np.uint64)
    assert raw.lcf_q4_debug_read2(b2.ctypes.data_as(ctypes.c_void_p)) == 0
    for name, blk in (("wg 600", b2[:64]), ("wg 1250", b2[64:])):
        st = blk.reshape(16, 4).astype(np.int64)
        st = st[st[:, 0] > 0]
        if len(st):
            t0 = st[0, 0]
            print(f"  {name}: per step (start, loads issued, end) in cycles from the first, groups: " +
                  "; ".join(f"{a - t0}/{b - t0}/{c - t0} g{g}" for a, b, c, g in st))
