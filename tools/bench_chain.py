#!/usr/bin/env python3
"""Times lcf_rulebook_chain alone on a real frame's coordinates (eager launches on one stream).
Usage: python tools/bench_chain.py [unfused] -- under rocprofv3 --kernel-trace --stats the
unfused form shows every step under its own kernel name (rbc_kernel<10 + kind>)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import _lib, ops, synth  # noqa: E402

FHD = [dict(subm=True, ksize=3), dict(subm=False, ksize=3, stride=2, padding=1),
       dict(subm=True, ksize=3), dict(subm=False, ksize=3, stride=2, padding=1),
       dict(subm=True, ksize=3), dict(subm=False, ksize=3, stride=2, padding=[0, 1, 1]),
       dict(subm=True, ksize=3), dict(subm=False, ksize=[3, 1, 1], stride=[2, 1, 1])]


def main():
    unfused = len(sys.argv) > 1 and sys.argv[1] == "unfused"
    dev = torch.device("cuda:0")
    pts = torch.from_numpy(synth.vkitti_frame(0, 20000)).to(dev)
    MV = 40000
    _, coors, _, vnum = ops.voxelize(pts, synth.VOXEL_SIZE, synth.PC_RANGE, 5, MV, batch_idx=0)
    loose = ops.RulebookChain(FHD, MV, 1, [41, 1600, 1408], device=dev)
    loose.run(coors, k_dev=vnum)
    need = [int(loose.k_out[i].item()) for i in loose.strided]
    caps = [int(1.5 * k + 2048) // 1024 * 1024 + 1024 for k in need]
    chain = ops.RulebookChain(FHD, MV, 1, [41, 1600, 1408], caps=caps, device=dev)
    ctx = _lib.Context()
    ctx.set(_lib.OPT_WS_CLEAN, 1)
    ctx.set(_lib.OPT_RB_UNFUSED, int(unfused))
    chain.reset()
    st = torch.cuda.current_stream().cuda_stream
    def frame():
        chain.run(coors, k_dev=vnum, ctx=ctx.handle)
        ctx.fill_batch_begin()
        chain.reset(stream=st, ctx=ctx.handle)
        ctx.fill_batch_flush(st)
    for _ in range(20):
        frame()
    torch.cuda.synchronize()
    n = 200
    t = time.perf_counter()
    for _ in range(n):
        frame()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t) / n * 1e6
    assert int(chain.flags.sum()) == 0
    for i in chain.strided:
        assert torch.equal(chain.out_coors[i][:need[chain.strided.index(i)]],
                           loose.out_coors[i][:need[chain.strided.index(i)]])
    print(f"voxels {int(vnum.item())} levels {need} caps {caps} "
          f"{'unfused' if unfused else 'fused'}: {us:.1f} us per chain (eager, host-paced)")


if __name__ == "__main__":
    main()
