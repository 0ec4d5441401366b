"""Times one light-curtain step's voxelisation: appending a curtain's returns to
a resident 20k-point cloud (lcf_voxelize_append) against voxelising the stacked cloud from
scratch (lcf_voxelize), device-resident inputs in both cases, and against the reference's shape of
the step (host cloud -> H2D of everything -> from scratch).  HIP events around each call, median.

    python tools/bench_append.py [curtain_points] [reps]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from objectdetection_amd import ops, synth
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    dev = torch.device("cuda:0")
    a = synth.vkitti_frame(0, 20000)
    rng = np.random.default_rng(1)
    b = a[rng.integers(0, len(a), m)] + rng.normal(0, 0.05, (m, 4)).astype(np.float32)
    b = b.astype(np.float32)
    both = np.concatenate([a, b])
    a_d, b_d, both_d = (torch.from_numpy(x).to(dev) for x in (a, b, both))
    both_pinned = torch.from_numpy(both).pin_memory()
    vs, rg = synth.VOXEL_SIZE, synth.PC_RANGE
    acc = ops.VoxelAccumulator(len(both), 4, vs, rg, 5, 40000, device=dev)
    out = ops.voxelize(both_d, vs, rg, 5, 40000)
    ws = torch.empty(ops._lib.lib().lcf_voxelize_workspace_bytes(len(both), 5), dtype=torch.uint8,
                     device=dev)
    t_app, t_full, t_ref = [], [], []
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(reps + 5):
        acc.reset()
        acc.append(a_d)
        e[0].record()
        acc.append(b_d)
        e[1].record()
        torch.cuda.synchronize()
        t_app.append(e[0].elapsed_time(e[1]) * 1e3)
        e[0].record()
        ops.voxelize(both_d, vs, rg, 5, 40000, out=out, ws=ws)
        e[1].record()
        torch.cuda.synchronize()
        t_full.append(e[0].elapsed_time(e[1]) * 1e3)
        e[0].record()
        ops.voxelize(both_pinned.to(dev, non_blocking=True), vs, rg, 5, 40000, out=out, ws=ws)
        e[1].record()
        torch.cuda.synchronize()
        t_ref.append(e[0].elapsed_time(e[1]) * 1e3)
    med = [float(np.median(t[5:])) for t in (t_app, t_full, t_ref)]
    k = int(acc.vnum.item())
    assert torch.equal(acc.coors, out[1]) and torch.equal(acc.voxels, out[0])
    print(f"cloud 20000 + {m} curtain points -> {k} voxels: append {med[0]:.1f} us, from scratch "
          f"{med[1]:.1f} us, H2D of the whole cloud + from scratch {med[2]:.1f} us")


if __name__ == "__main__":
    main()
