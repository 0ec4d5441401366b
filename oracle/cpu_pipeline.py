"""CPU baseline of the whole hot path -- TEST / BENCH INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference-faithful restatement of what the reference runs on a CPU (BASELINE.md "CPU-baseline
plan"): sequential dense-table voxelizer (numba points_to_voxel twin, 1 thread), the VFE of
oracle/vfe_oracle.py (numpy; pinned by reference-generated goldens -- never the product's
voxel_encoder module), dense-grid sequential get_indice_pairs, and spconv's CPU
indice_conv = per kernel offset gather -> torch.mm -> scatter-add (torch threads), then dense().
A restatement ("port"), not the original numba/spconv binaries, which are absent.
"""
import time

import numpy as np
import torch

import oracle
from oracle import vfe_oracle


def indice_conv_mm(feats, weight, pairs, pair_num, k_out):
    """spconv CPU indice_conv: offsets ascending, gather / mm / index_add."""
    w = weight.reshape(-1, weight.shape[-2], weight.shape[-1])
    out = torch.zeros((k_out, w.shape[-1]), dtype=feats.dtype)
    for o in range(w.shape[0]):
        n = int(pair_num[o])
        if n == 0:
            continue
        i = torch.from_numpy(pairs[0, o, :n].astype(np.int64))
        j = torch.from_numpy(pairs[1, o, :n].astype(np.int64))
        out.index_add_(0, j, feats.index_select(0, i) @ w[o])
    return out


def run_vfe(vfe, voxels, num):
    """The VFE stage on oracle/vfe_oracle.py.  `vfe` is only a parameter holder (state_dict keys
    are the reference's) and names the variant; eval-mode BatchNorm."""
    params = {k: v.detach().cpu().numpy() for k, v in vfe.state_dict().items()}
    kind = type(vfe).__name__
    if kind == "SimpleVoxel":
        return vfe_oracle.simple_voxel(voxels, num, voxels.shape[2])
    if kind == "SimpleVoxelRadius":
        return vfe_oracle.simple_voxel_radius(voxels, num, voxels.shape[2])
    if kind == "VoxelFeatureExtractor":
        return vfe_oracle.voxel_feature_extractor(
            voxels, num, params, with_distance=bool(getattr(vfe, "_with_distance", False)))
    raise NotImplementedError(f"cpu_pipeline: no oracle for VFE {kind}")


def run_middle(seq, feats, coors, shape, batch):
    cache = {}
    for m in seq:
        if hasattr(m, "subm") and hasattr(m, "kernel_size"):
            key = m.indice_key
            if key is not None and key in cache:
                outc, pairs, pnum, oshape = cache[key]
            elif m.subm:
                pairs, pnum = oracle.rulebook_subm(coors, batch, shape, m.kernel_size, m.dilation)
                outc, oshape = coors, shape
            else:
                outc, pairs, pnum, oshape = oracle.rulebook_conv(coors, batch, shape,
                                                                 m.kernel_size, m.stride,
                                                                 m.padding, m.dilation)
            if key is not None:
                cache[key] = (outc, pairs, pnum, oshape)
            feats = indice_conv_mm(feats, m.weight.detach(), pairs, pnum, outc.shape[0])
            if m.bias is not None:
                feats = feats + m.bias.detach()
            coors, shape = outc, oshape
        else:
            feats = m(feats)
    return feats, coors, shape


@torch.no_grad()
def frontend_forward(model_cpu, clouds):
    """model_cpu: a SparseFrontEnd on the CPU in eval mode (modules used as parameter holders
    and for the torch VFE/BN/ReLU maths); clouds: list of numpy [N,4].  -> [B,C*D,H,W]."""
    gen = model_cpu.voxel_generator
    vs, ns, cs = [], [], []
    for b, pts in enumerate(clouds):
        v, c, n = oracle.voxelize(pts, gen.voxel_size, gen.point_cloud_range,
                                  gen.max_num_points_per_voxel, model_cpu.max_voxels)
        vs.append(v)
        ns.append(n)
        cs.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
    coors = np.ascontiguousarray(np.concatenate(cs))
    feats = run_vfe(model_cpu.voxel_feature_extractor, np.concatenate(vs), np.concatenate(ns))
    feats = torch.from_numpy(np.ascontiguousarray(feats, dtype=np.float32))
    mid = model_cpu.middle_feature_extractor
    shape = [int(s) for s in mid.sparse_shape]
    feats, coors, shape = run_middle(mid.middle_conv, feats, coors, shape, len(clouds))
    d = oracle.dense(feats.numpy(), coors, len(clouds), shape)
    return d.reshape(d.shape[0], d.shape[1] * d.shape[2], d.shape[3], d.shape[4])


def time_frontend(model_cpu, clouds, budget_s=20.0, min_iters=2, max_iters=30, warmups=5):
    """-> (frames/s from the median, iterations, threads, times).  SURVEY.md section 8d's
    protocol: 5 untimed warm-ups, then up to 30 timed passes (fewer only when the budget runs
    out first); the caller reports median and p10 / p90."""
    t_w = time.perf_counter()
    for i in range(warmups):
        frontend_forward(model_cpu, clouds)
        if i >= 1 and time.perf_counter() - t_w > 0.4 * budget_s:
            break  # a slow host: keep most of the budget for timed passes
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < max_iters and (len(times) < min_iters or time.perf_counter() < t_end):
        t0 = time.perf_counter()
        frontend_forward(model_cpu, clouds)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return len(clouds) / med, len(times), torch.get_num_threads(), times
