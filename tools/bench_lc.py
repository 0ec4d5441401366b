"""Times the light-curtain placement step on the reference-run `kitti` fixture
(K=2, 200x176 map, 512 camera rays x 80 points): entropy look-up + DP on the device (HIP events
over `reps` launches; both kernels run long enough that event timing is not host-bound for the
DP), and the numpy oracle's vectorised DP on the host beside it.

    python tools/bench_lc.py [reps]
"""
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    timing = "--timing" in sys.argv
    if timing:  # a build of csrc/lc_policy.hip with -DLCP_TIMING linked as tools/ubench/liblcf_timing.so
        sys.argv.remove("--timing")
        from objectdetection_amd import _lib
        _lib.LIB_PATH = os.path.join(ROOT, "tools", "ubench", "liblcf_timing.so")
    from objectdetection_amd import light_curtain as lc
    from oracle import lc_policy_oracle as O
    from test_lc_policy import lc_case
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    c = lc_case("kitti")
    dev = types.SimpleNamespace(CAMERA_PARAMS=c["cam"], TRANSFORMS={"cTl": c["cTl"]})
    pol = lc.DPOptimizedPolicy(dev, int(c["pts_per_cam_ray"]))
    pol.set_map_geometry(c["confidence_map"][:, :, :2])
    cls = torch.from_numpy(c["cls"]).cuda()
    mask = torch.from_numpy(c["mask"]).cuda()
    ul = pol.entropy_lookup(cls, mask)
    for _ in range(3):
        pol._solve(ul)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    for _ in range(reps):
        pol.entropy_lookup(cls, mask, out=ul)
    ev[1].record()
    for _ in range(reps):
        pol._solve(ul)
    ev[2].record()
    torch.cuda.synchronize()
    if timing:
        T, N = pol._graph["node_k"].shape
        raw = pol._graph["ws"][:T * 32].view(torch.int32).cpu().numpy().reshape(T, 8)[1:]
        st = (raw[:, 2:7].astype(np.int64) - raw[:, [0]].astype(np.int64)) & 0xFFFFFFFF
        print("cycles from step start to: point loaded %d, max known %d, candidates done %d, "
              "arg-min reduced %d, before barrier %d (medians)" % tuple(np.median(st, axis=0)))
        clk = raw[:, :2]
        dc = np.diff(clk[:, 0].astype(np.int64)) & 0xFFFFFFFF
        dw = np.diff(clk[:, 1].astype(np.int64)) & 0xFFFFFFFF
        print("shader cycles per step: median %d, mean %.0f, p90 %d, max %d; total %.1f k" %
              (np.median(dc), dc.mean(), np.percentile(dc, 90), dc.max(), dc.sum() / 1e3))
        print("wall (100 MHz) ticks per step: mean %.2f -> %.2f us; shader clock ~ %.0f MHz" %
              (dw.mean(), dw.mean() / 100, dc.sum() / (dw.sum() / 100)))
        print("steps at chunk boundaries (q %% 8 == 7): mean %.0f; others: mean %.0f" %
              (dc[6::8].mean(), np.delete(dc, np.s_[6::8]).mean()))
    t_ent = ev[0].elapsed_time(ev[1]) / reps * 1e3
    t_dp = ev[1].elapsed_time(ev[2]) / reps * 1e3
    t0 = time.perf_counter()
    for _ in range(reps):
        pol.design_points_from_scores(cls, mask, apply_sigmoid=False)
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / reps * 1e6
    nodes, th = O.build_graph(c["confidence_map"][:, :, :2], c["cam"], c["cTl"], 80)
    k = nodes[:, :, 4].astype(np.int64)
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        u = O.entropy_lookup(c["confidence_map"])
        O.dp_design_vec(u, k, th, float(c["max_dtheta"]))
    t_cpu = (time.perf_counter() - t0) / n * 1e6
    print(f"entropy look-up {t_ent:.1f} us  DP {t_dp:.1f} us  scores->design points "
          f"{t_all:.1f} us (wall, incl. launches)  numpy oracle {t_cpu:.0f} us")


if __name__ == "__main__":
    main()
