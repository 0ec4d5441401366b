#!/usr/bin/env python3
"""bench.py -- frames/s of the SECOND sparse front end (voxelize + VFE + sparse middle forward,
batch 1) on MI355X, with the dominant kernel's roofline fraction and a same-host CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model M1|M2|M3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one frame through the whole hot path (BASELINE.json configs[1]: one ~20k-point
Virtual-KITTI-shaped frame on the 1600x1408x41 grid, fp32) with the point cloud already resident
in HBM.  N > 1 runs one replica per GPU on its own frames (scene sharding, no data-path
collective; SURVEY.md section 8e: "inference: replicas only") and reports the whole-job rate.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from objectdetection_amd import ops, synth  # noqa: E402
from objectdetection_amd.frontend import SparseFrontEnd, randomize_  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (matrix)"


BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA" (dense)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md "HBM3E peak BW" (spec)


def dominant_kernel_roofline(model, clouds, math="f32", reps=20):
    """Times every launch of the dominant kernel -- the 64->64 channel indice_conv:
    cv_q4_kernel<64,64,4>, fp32 MFMA over tile plans (fused gather-GEMM-scatter + BN + ReLU) as
    FrameEngine launches it (math "f32"; the 7 3x3x3 launches of a frame), or cv_bt_kernel<64,64,3>
    on rows stored as bf16 as the training step launches it for forward and dgrad (math "bf16";
    the 8 forward launches of a batch, the 3-offset one included) -- with HIP events on the stream
    it is launched on, after the timed region.  achieved = algorithmic flops per launch
    (2 * pairs * Cin * Cout) / average launch duration; the same launches
    priced in algorithmic bytes (K_in*Cin*e_in + K_out*Cout*4 + K_out*kvol*4 + kvol*Cin*Cout*e_w;
    e = 4 for fp32 rows / weights, 2 for bf16) against HBM."""
    from objectdetection_amd.spconv import SparseConvTensor
    from objectdetection_amd.spconv.modules import _bn_affine
    mid = model.middle_feature_extractor
    voxels, num, coors = model.voxelize(clouds)
    feats = model.voxel_feature_extractor(voxels, num, coors)
    x = SparseConvTensor(feats, coors, mid.sparse_shape, len(clouds))
    mods = list(mid.middle_conv)
    launches = []
    i = 0
    while i < len(mods):
        conv, bn = mods[i], mods[i + 1]
        rb = conv.get_rulebook(x)
        scale, shift = _bn_affine(bn)
        if conv.in_channels == 64 and conv.out_channels == 64 and (math != "f32" or
                                                                   rb.kvol == 27):
            pairs = int((rb.out_nbr >= 0).sum().item())
            nbytes = 4 * (rb.k_in * 64 + rb.k_out * 64 + rb.k_out * rb.kvol) + \
                rb.kvol * 64 * 64 * (4 if math == "f32" else 2)
            if math == "f32":
                plan, tr = ops.tile_plan(rb.out_nbr, rb.k_out, rb.kvol, shape=(64, 64, 0))
                call = (ops.conv_forward_planned,
                        (x.features, conv._weight_image(), plan, tr, rb.k_out, rb.kvol, 64, 64,
                         scale, shift, True))
            else:
                # the training step's forward / dgrad kernel: rows stored as bf16, fp32 out, no
                # epilogue (training BatchNorm needs the batch statistics first)
                nbytes = rb.k_in * 64 * 2 + rb.k_out * 64 * 4 + rb.k_out * rb.kvol * 4 + \
                    rb.kvol * 64 * 64 * 2
                call = (ops.conv_forward_b16,
                        (x.features.bfloat16(), ops.weight_image_bf16(conv.weight), rb.out_nbr,
                         rb.k_out, rb.kvol, 64, 64))
            launches.append((call, pairs, nbytes))
        x = conv(x, _scale=scale, _shift=shift, _relu=True)
        i += 3
    torch.cuda.synchronize()
    total_ms, total_flops, total_bytes, n = 0.0, 0.0, 0.0, 0
    stream = torch.cuda.current_stream()
    for (fn, args), pairs, nbytes in launches:
        fn(*args)  # warm
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn(*args)
        e1.record(stream)
        e1.synchronize()
        total_ms += e0.elapsed_time(e1) / reps
        total_flops += 2.0 * pairs * 64 * 64
        total_bytes += nbytes
        n += 1
    avg_s = total_ms / n * 1e-3
    tflops = total_flops / n / avg_s / 1e12
    gbs = total_bytes / n / avg_s / 1e9
    # HBM bytes per launch: NOT measured in this run (PMC counters need rocprofv3 passes of their
    # own) -- read from the committed summary of those passes and labelled so in the line
    traffic, traffic_source = None, None
    if math == "f32":
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_dominant_kernel.json")) as fh:
                rec = json.load(fh)
            traffic = rec["hbm_bytes_per_launch"]
            traffic_source = ("committed PMC passes (FETCH_SIZE / WRITE_SIZE, corrected as "
                              "MI355X_MICROARCH.md prescribes): " + rec.get("source", "profiles/"))
        except (OSError, KeyError, ValueError):
            pass
    common = {"kernel": "cv_q4_kernel<64,64,4>" if math == "f32" else
              "cv_bt_kernel<64,64,3> (bf16 rows, fp32 accumulate and output)",
              "launches_per_frame" if math == "f32" else "launches_timed": n,
              "avg_launch_us": round(avg_s * 1e6, 2),
              "flops_per_launch": round(total_flops / n, 0),
              "algorithmic_bytes_per_launch": round(total_bytes / n, 0),
              "intensity_flop_per_byte": round(total_flops / total_bytes, 1)}
    if math == "f32":
        # 165 flop/B against a ridge of 157.3 TF / 8 TB/s = 20 flop/B: priced against the fp32
        # MFMA peak (SURVEY.md 8d's "grade against HBM" assumed Cin, Cout <= 64 made it
        # bandwidth-bound; its own byte formula says otherwise); the HBM view is reported too
        return {"bound": "mfma", "achieved": round(tflops, 3), "peak": F32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(tflops / F32_MFMA_PEAK_TFLOPS, 4),
                "traffic": traffic, "traffic_source": traffic_source,
                "bound_rationale": "algorithmic intensity above the fp32 ridge (157.3 TFLOP/s / "
                                   "8 TB/s = 19.7 flop/B): MFMA-bound; fp32 MFMA and VALU share "
                                   "one datapath on gfx950 (DESIGN.md section 4)",
                "hbm_view": {"achieved_GBs": round(gbs, 1), "frac_of_8TBs": round(gbs / HBM_PEAK_GBS, 4)},
                **common}
    # bf16 operands: ridge 2500 TF / 8 TB/s = 312 flop/B, above the kernel's intensity: HBM-bound
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
            "bound_rationale": "algorithmic intensity below the bf16 ridge (2500 TFLOP/s / 8 TB/s "
                               "= 312 flop/B): priced against HBM",
            "mfma_view": {"achieved_TFLOPs": round(tflops, 2),
                          "frac_of_bf16_peak": round(tflops / BF16_MFMA_PEAK_TFLOPS, 4)},
            **common}


def cpu_baseline(model_name, cloud_np, budget_s, max_voxels=40000):
    from oracle import cpu_pipeline  # the CPU restatement: baseline only, never the product
    m = randomize_(SparseFrontEnd(model_name, max_voxels=max_voxels).eval(), 0)
    fps, iters, threads, times = cpu_pipeline.time_frontend(m, [cloud_np], budget_s=budget_s)
    return {"value": round(fps, 3), "unit": "frames/s", "cores": threads, "kind": "port",
            "host_cpus": os.cpu_count(),
            "sample": f"{iters} timed passes of the same N={cloud_np.shape[0]} frame (seed 0) "
                      f"through oracle/cpu_pipeline.py after up to 5 warm-ups; median "
                      f"{np.median(times) * 1e3:.0f} ms/frame, p10 {np.percentile(times, 10) * 1e3:.0f}, "
                      f"p90 {np.percentile(times, 90) * 1e3:.0f}; restatement of numba "
                      f"points_to_voxel (1 thread) + spconv CPU indice_conv (torch.mm, "
                      f"{threads} threads), not the original binaries"}


def light_curtain_measurement(dev, cloud_np, with_cpu, reps=30):
    """SURVEY.md 8f-2 / 8f-4, the two device-side pieces of the light-curtain loop around the
    detector (second/light_curtain/utils.py:88-124): placing the curtain from the class scores
    (entropy table + DPOptimizedPolicy's dynamic programme, 512 camera rays x 80 points on the
    200 x 176 map) and voxelising the cloud grown by one curtain's returns.  HIP events; the numpy
    oracle of the policy beside it on the host."""
    from objectdetection_amd import light_curtain as lc, ops
    cls_np, mask_np, anchors = synth.lc_scene(0)
    geo = lc.LCGeometry()
    pol = lc.DPOptimizedPolicy(geo, device=dev)
    K, Y, X = cls_np.shape
    xz = lc.confidence_grid(anchors, K, Y, X, geo.TRANSFORMS["wTc"])
    pol.set_map_geometry(xz)
    cls, mask = torch.from_numpy(cls_np).to(dev), torch.from_numpy(mask_np).to(dev)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    def timed(fn):
        ts = []
        for _ in range(reps + 3):
            e[0].record()
            fn()
            e[1].record()
            torch.cuda.synchronize()
            ts.append(e[0].elapsed_time(e[1]))
        return float(np.median(ts[3:]))
    t_pol = timed(lambda: pol.design_points_from_scores(cls, mask, apply_sigmoid=False))
    pts = pol.design_points_from_scores(cls, mask, apply_sigmoid=False)
    rng = np.random.default_rng(1)
    m = 3000
    new = (cloud_np[rng.integers(0, len(cloud_np), m)] +
           rng.normal(0, 0.05, (m, 4)).astype(np.float32)).astype(np.float32)
    a_d, b_d = torch.from_numpy(cloud_np).to(dev), torch.from_numpy(new).to(dev)
    acc = ops.VoxelAccumulator(len(cloud_np) + m, 4, synth.VOXEL_SIZE, synth.PC_RANGE, 5, 40000,
                               device=dev)

    def append():
        acc.reset()
        acc.append(a_d)
        e[0].record()
        acc.append(b_d)
        e[1].record()
        torch.cuda.synchronize()
        return e[0].elapsed_time(e[1])
    t_app = float(np.median([append() for _ in range(reps + 3)][3:]))
    both = torch.cat([a_d, b_d])
    t_full = timed(lambda: ops.voxelize(both, synth.VOXEL_SIZE, synth.PC_RANGE, 5, 40000))
    out = {"policy": "DPOptimizedPolicy (entropy table + DP), scores resident on the device",
           "map": [K, Y, X], "camera_rays": int(pts.shape[0]), "points_per_ray": 80,
           "policy_ms": round(t_pol, 4),
           "append_voxelize_us": round(t_app * 1e3, 1),
           "from_scratch_voxelize_us": round(t_full * 1e3, 1),
           "appended_points": m, "resident_points": int(len(cloud_np))}
    if with_cpu:
        from oracle import lc_policy_oracle as O  # the CPU restatement: baseline only
        cam = geo.CAMERA_PARAMS
        cmap = O.confidence_map(anchors, mask_np, cls_np, geo.TRANSFORMS["wTc"])
        nodes, theta = O.build_graph(cmap[:, :, :2], cam, geo.TRANSFORMS["cTl"], 80)
        k = nodes[:, :, 4].astype(np.int64)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            ul = O.entropy_lookup(O.confidence_map(anchors, mask_np, cls_np,
                                                   geo.TRANSFORMS["wTc"]))
            choice, _, _ = O.dp_design_vec(ul, k, theta, 14000 * 1.5e-5)
            ts.append(time.perf_counter() - t0)
        want = np.flip(nodes[np.arange(len(choice)), choice, :2], axis=0)
        out["policy_cpu_ms"] = round(float(np.median(ts)) * 1e3, 2)
        out["policy_cpu_kind"] = "port (numpy, 1 thread; the shape policy.py computes in)"
        out["design_points_equal_cpu"] = float((pts.cpu().numpy() == want).all(axis=1).mean())
    return out


def _murmur32(k):
    """lcf_hash32 (csrc/common.h) on an int64 tensor of 32-bit keys."""
    m = 0xFFFFFFFF
    h = k & m
    h = h ^ (h >> 16)
    h = (h * 0x85ebca6b) & m
    h = h ^ (h >> 13)
    h = (h * 0xc2b2ae35) & m
    return h ^ (h >> 16)


def _pow2_capacity(items):
    need, p = max(2 * int(items), 1024), 1
    while p < need:
        p <<= 1
    return p


def hash_table_stats(ws, capacity, wide):
    """Occupancy and probe lengths of one of liblcf's open-addressing tables as the kernels left it
    (read from the workspace, off the timed path): `ws` the uint8 workspace whose first bytes are
    the table -- 32-bit keys (voxelizer, submanifold level) or 64-bit key << 32 | stamp entries
    (strided levels) --, linear probing from lcf_hash32(key) & (capacity - 1).  A resident key's
    probe count = its displacement from the home slot + 1 = what a successful look-up of it walks
    (and at least what its insert walked)."""
    if wide:
        t = ws[:capacity * 8].view(torch.int64)
        occ = t != -1
        key = (t >> 32) & 0xFFFFFFFF
    else:
        t = ws[:capacity * 4].view(torch.int32)
        occ = t != -1
        key = t.long() & 0xFFFFFFFF
    slot = torch.nonzero(occ).flatten()
    if slot.numel() == 0:
        return {"slots": capacity, "load_factor": 0.0, "probes_mean": 0.0, "probes_max": 0}
    disp = (slot - _murmur32(key[slot])) & (capacity - 1)
    return {"slots": capacity, "load_factor": round(slot.numel() / capacity, 4),
            "probes_mean": round(float((disp + 1).float().mean()), 3),
            "probes_max": int(disp.max()) + 1}


def dense_lc_measurement(dev, model_name, steps=60):
    """BASELINE.json configs[4] at one GPU: a dense light-curtain-augmented, SYNTHIA-shaped frame
    (120,000 points, 640x480, 39.32 deg FOV, 10 curtains; second/data/synthia_dataset.py:53-55)
    through the same FrameEngine path (max_voxels 160000), with the rows of every level and --
    the configuration exists to stress them -- the occupancy and probe lengths of the coordinate
    hash tables ("report hash-table load factor and probe counts")."""
    import ctypes
    from objectdetection_amd import _lib
    from objectdetection_amd._lib import f3, f6
    n, mv = 120000, 160000
    model = randomize_(SparseFrontEnd(model_name, max_voxels=mv).eval(), 0).to(dev)
    clouds = [torch.from_numpy(synth.synthia_frame(i, n)).to(dev) for i in range(4)]
    calib = [torch.from_numpy(synth.synthia_frame(100000 + i, n)).to(dev) for i in range(2)]
    with torch.no_grad():
        eng = model.engine(calib=calib)
        for i in range(10):
            eng.forward(clouds[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            eng.forward(clouds[i % 4])
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        eng.check()
        rows = eng.counts()
        # hash tables of that last frame, rebuilt off the timed path by calls that leave them in
        # place (the engine re-arms its own behind every frame): the voxelizer's with the caller's
        # sizing (2 slots per point), the chain's with the engine's calibrated capacities
        L = _lib.lib()
        st = torch.cuda.current_stream().cuda_stream
        pts = clouds[(steps - 1) % 4]
        ws = torch.empty(L.lcf_voxelize_workspace_bytes(n, 5), dtype=torch.uint8, device=dev)
        v = torch.empty((mv, 5, 4), device=dev)
        c = torch.empty((mv, 4), dtype=torch.int32, device=dev)
        num = torch.empty(mv, dtype=torch.int32, device=dev)
        k = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(L.lcf_voxelize(pts.data_ptr(), n, None, 4, f3(synth.VOXEL_SIZE),
                                  f6(synth.PC_RANGE), 5, mv, 0, 0, 4, v.data_ptr(), c.data_ptr(),
                                  num.data_ptr(), k.data_ptr(), ws.data_ptr(), ws.numel(), st))
        tables = {"voxelizer": hash_table_stats(ws, _pow2_capacity(n), False)}
        chain = eng.chain
        chain.run(c, k_dev=k)  # no WS_CLEAN context: resets its own tables first, leaves them built
        torch.cuda.synchronize()
        tables["submanifold level 0"] = hash_table_stats(chain.ws0, _pow2_capacity(chain.k_cap),
                                                         False)
        cap = chain.k_cap
        for i in chain.strided:
            lv = chain.levels[i]
            E = 1
            for d in range(3):
                E *= (lv.ksize[d] + lv.stride[d] - 1) // lv.stride[d]
            tables[f"strided level {chain.strided.index(i) + 1}"] = hash_table_stats(
                chain.ws[i], _pow2_capacity(min(cap * E, lv.cap_out)), True)
            cap = lv.cap_out
        eng._reset_workspaces(st)
    return {"workload": "one 120000-point SYNTHIA-shaped frame per step (10 curtains), "
                        "max_voxels 160000, FrameEngine, batch 1 (BASELINE.json configs[4] at one "
                        "GPU)",
            "value": round(steps / el, 2), "unit": "frames/s",
            "ms_per_step": round(el / steps * 1e3, 4), "steps": steps,
            "rows_per_level": rows, "level_capacities": eng.level_caps,
            "hash_tables": tables,
            "note": "tables: open addressing, linear probing, capacity = pow2 >= 2 x items "
                    "(strided levels: items = the level's row capacity, 2.0 x the calibrated "
                    "count); probes = slots a look-up of a resident key walks"}


def train_measurement(args, dev, rank, world, dist, barrier):
    """BASELINE.json configs[2]/[3]: training step (VFE + sparse middle fwd/dgrad/wgrad + RPNV2 +
    loss + optimizer) on `train_batch` frames per rank, scene-sharded, ONE flat gradient
    all-reduce per step over RCCL when world > 1.  Reported beside the headline metric."""
    from objectdetection_amd.train import TrainStep
    step = TrainStep(dev, args.model, seed=0)
    b = args.train_batch
    # Two global batches of b x world scenes.  One rank: b frames of args.points points each
    # (BASELINE.json configs[2]).  Several ranks (configs[3]): scenes of MIXED sizes (0.6 .. 1.4 x
    # args.points, the same list on every rank) dealt to the ranks by cost with dist.shard_scenes
    # -- K varies per frame, so equal frame counts are not equal work;
    # a rank generates only its own shard.
    from objectdetection_amd.dist import shard_scenes
    batches, shard_info = [], []
    for bi in range(2):
        rng = np.random.default_rng(7 + bi)
        sizes = [args.points] * (b * world) if world == 1 else \
            [int(args.points * f) for f in rng.uniform(0.6, 1.4, b * world)]
        shards = shard_scenes(sizes, world)
        batches.append([torch.from_numpy(synth.vkitti_frame(1000 + bi * b * world + i,
                                                            sizes[i])).to(dev)
                        for i in shards[rank]])
        shard_info.append({"frames_per_rank": [len(sh) for sh in shards],
                           "points_per_rank": [sum(sizes[i] for i in sh) for sh in shards]})
    for i in range(6):  # MIOpen times its solvers for the RPN on the first steps
        step(batches[i % 2])
    barrier()
    t0 = time.perf_counter()
    for i in range(args.train_steps):
        loss = step(batches[i % 2])
    barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    nparams = sum(p.numel() for p in step.net.parameters())
    roof = None
    if rank == 0:  # the step's dominant kernel on one of its batches (profiles/r03*_train_*)
        with torch.no_grad():
            step.net.eval()
            roof = dominant_kernel_roofline(step.net.frontend, batches[0], math="bf16", reps=10)
            step.net.train()
    return {"roofline": roof,
            "metric": "frames/s training step (VFE + sparse middle fwd/dgrad/wgrad + RPNV2)",
            "value": round(world * b * args.train_steps / el, 2), "unit": "frames/s",
            "ms_per_step": round(el / args.train_steps * 1e3, 2), "steps": args.train_steps,
            "global_batch": world * b, "scaling": "weak", "scene_shards": shard_info,
            "dtype": "bf16 MFMA indice_conv fwd/dgrad/wgrad on bf16-stored rows at 32/64 channels "
                     "(fp32 accumulate, fp32 conv outputs and BatchNorm statistics, fp32 master "
                     "weights; the 16-channel level-0 layers fp32 MFMA), bf16 autocast RPN",
            "allreduce_bytes": 4 * nparams if world > 1 else 0,
            "loss": round(float(loss), 4)}


def launch_ranks(n, argv, script=None):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script as CHILD
    processes (one per GPU, torch.distributed.run, rendezvous on 127.0.0.1) and return their exit
    code; rank 0's JSON line reaches stdout through the inherited descriptor.  Called before this
    process has made any GPU call -- a process that has initialised the GPU must never be
    replaced or re-executed on this pool, and this one only relays.  (The reference's multi-GPU
    switch is second/pytorch/train.py:204-205, single-process DataParallel.)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={int(n)}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           script or os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL)
    return subprocess.run(cmd, env=env).returncode


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--model", default="M2", choices=["M1", "M2", "M3"])
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--frames", type=int, default=8, help="distinct synthetic frames cycled")
    ap.add_argument("--shape", default="vkitti", choices=["vkitti", "synthia"],
                    help="synthetic frame generator (synthia: BASELINE.json configs[4], use with "
                         "--points 120000 --max-voxels 160000)")
    ap.add_argument("--max-voxels", type=int, default=40000)
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-steps", type=int, default=30,
                    help="timed training steps of the second measurement (0 = skip it)")
    ap.add_argument("--train-batch", type=int, default=8, help="frames per rank per step")
    ap.add_argument("--inflight", type=int, default=1,
                    help="independent frames in flight (one FrameEngine + stream each)")
    ap.add_argument("--conv-math", default="f32", choices=["f32", "bf16x3", "bf16x8", "bf16x6"],
                    help="operand form of the 32/64/128-channel sparse convs in the timed path: "
                         "fp32 MFMA (the headline) or split bf16 operands (lcf.h: bf16x3)")
    ap.add_argument("--no-alt", action="store_true",
                    help="skip the secondary bf16x3 measurement reported beside the headline")
    ap.add_argument("--eager", action="store_true",
                    help="module-by-module path (spconv API, host syncs) instead of the "
                         "graph-captured FrameEngine")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: become the launcher.  This process stays GPU-free -- it makes no
        # HIP call at all, not even a device count (a launcher that has initialised the GPU must
        # not start other GPU programs on this pool); a rank whose LOCAL_RANK has no device
        # refuses below with exit code 2
        return launch_ranks(args.gpus, argv)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 control flow on a one-GPU box: LCF_REHEARSE=1 puts every rank on
    # cuda:0 and uses gloo (never used for reported numbers)
    rehearse = os.environ.get("LCF_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("LCF_REHEARSE_DUMP", "100")),
                                          exit=True)
    if not rehearse and local_rank >= torch.cuda.device_count():
        print(f"bench.py: rank {rank} (LOCAL_RANK {local_rank}) has no GPU: --gpus {args.gpus} but "
              f"only {torch.cuda.device_count()} GPU(s) are visible (LCF_REHEARSE=1 rehearses the "
              "control flow on one GPU over gloo)", file=sys.stderr)
        return 2
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()  # what the process group reports, not what was asked for
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); reporting "
              f"n_gpus = {world}", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    model = randomize_(SparseFrontEnd(args.model, max_voxels=args.max_voxels).eval(), 0).to(dev)
    gen = synth.vkitti_frame if args.shape == "vkitti" else synth.synthia_frame
    # each rank owns its own scenes (scene sharding): seeds rank*frames ... rank*frames+frames-1
    clouds_np = [gen(rank * args.frames + i, args.points)
                 for i in range(args.frames)]
    clouds = [torch.from_numpy(c).to(dev) for c in clouds_np]  # resident in HBM before timing
    # the engine's capacities and kernel-selection hints are calibrated on OTHER frames of the
    # same generator (seeds 100000+), never on the frames that are timed
    calib = [torch.from_numpy(gen(100000 + rank * 4 + i, args.points)).to(dev) for i in range(4)]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        if args.eager:
            step = lambda c: model([c])
        else:
            # one hipGraph replay per frame
            engine = model.engine(calib=calib, conv_math=args.conv_math)
            step = engine.forward
            if args.inflight > 1:
                # serving with several batch-1 requests in flight: one engine (own buffers, own
                # graph) per slot, each on its own stream; frames alternate over the slots
                streams = [torch.cuda.Stream(device=dev) for _ in range(args.inflight)]
                engines = [engine] + [model.engine(calib=calib, conv_math=args.conv_math)
                                      for _ in range(args.inflight - 1)]
                slot = [0]

                def step(c):
                    i = slot[0] % args.inflight
                    slot[0] += 1
                    with torch.cuda.stream(streams[i]):
                        return engines[i].forward(c)
        for i in range(args.warmup):
            out = step(clouds[i % len(clouds)])
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = step(clouds[i % len(clouds)])
        barrier()
        elapsed = time.perf_counter() - t0
        if not args.eager and args.inflight > 1:
            for e in engines:
                e.check()
            out = engines[(args.steps - 1) % args.inflight].out
        if not args.eager:
            engine.check()  # no frame may have exceeded a level capacity
            ref = model([clouds[(args.steps - 1) % len(clouds)]])
            err = ((out - ref).abs().max() / ref.abs().max()).item()
            assert err < 1e-4, f"engine output differs from the module-by-module path: {err}"
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        assert out.shape[1:] == (128, 200, 176) or args.model == "M3", out.shape
        roof = dominant_kernel_roofline(model, [clouds[0]]) if rank == 0 and args.model != "M3" \
            else None
        # secondary lines (never `value`): the same frames through engines whose wide sparse
        # convs run split-operand math on the bf16 matrix cores -- "bf16x8" (three exact planes,
        # fp32-grade) and "bf16x3" (two planes, ~2e-5) -- with their error against the fp32
        # module path.  ("bf16x6", csrc/conv_bx.h, is level with fp32 in the frame and not listed:
        # --conv-math bf16x6 times it.)
        alt = None
        if (world == 1 and not args.eager and not args.no_alt and args.conv_math == "f32"
                and args.model != "M3"):
            alt = []
            for cm in ("bf16x8", "bf16x3"):
                eng3 = model.engine(calib=calib, conv_math=cm)
                for i in range(args.warmup):
                    eng3.forward(clouds[i % len(clouds)])
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(args.steps):
                    out3 = eng3.forward(clouds[i % len(clouds)])
                torch.cuda.synchronize()
                el3 = time.perf_counter() - t1
                eng3.check()
                err3 = ((out3 - ref).abs().max() / ref.abs().max()).item()
                assert err3 < 1e-4, f"{cm} engine outside the fp32 feature tolerance: {err3}"
                alt.append({"conv_math": cm, "value": round(args.steps / el3, 2),
                            "unit": "frames/s", "ms_per_step": round(el3 / args.steps * 1e3, 4),
                            "max_err_vs_f32_rel_to_max": float(f"{err3:.3g}"),
                            "tolerance": 1e-4})
                del eng3

        # throughput with two batch-1 frames in flight (never `value`, which stays the one-frame-
        # at-a-time rate every round has reported): a second engine with its own buffers and graph
        # on a second stream, frames alternating over the two
        pipelined = None
        if (world == 1 and not args.eager and not args.no_alt and args.inflight == 1
                and args.model != "M3"):
            from objectdetection_amd.engine import FramePipeline
            pipe = FramePipeline(model, calib=calib, depth=2, conv_math=args.conv_math)
            for i in range(args.warmup):
                pipe.submit(clouds[i % len(clouds)], wait_current=False)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.steps):
                out2, _ = pipe.submit(clouds[i % len(clouds)], wait_current=False)
            torch.cuda.synchronize()
            el2 = time.perf_counter() - t1
            pipe.check()
            err2 = ((out2 - ref).abs().max() / ref.abs().max()).item()
            assert err2 < 1e-4, f"pipelined engine output differs from the module path: {err2}"
            pipelined = {"frames_in_flight": 2, "value": round(args.steps / el2, 2),
                         "unit": "frames/s", "ms_per_frame": round(el2 / args.steps * 1e3, 4),
                         "note": "objectdetection_amd.engine.FramePipeline: two FrameEngines (own "
                                 "buffers, own graph) on two streams, frames alternating; same "
                                 "kernels, same results; a frame's latency is then about twice "
                                 "ms_per_frame"}
            del pipe

    train = train_measurement(args, dev, rank, world, dist, barrier) if args.train_steps > 0 \
        and args.model != "M3" else None

    if rank == 0:
        fps = world * args.steps / elapsed
        line = {
            "metric": "frames/s voxelize+sparse backbone fwd (bs1)",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x3": "bf16x3 (f32 = hi + lo, f32 accumulate)",
                      "bf16x8": "bf16x8 (f32 = hi + mid + lo exactly, f32 accumulate)",
                      "bf16x6": "bf16x6 (f32 = hi + mid + lo exactly, six products, f32 "
                                "accumulate; fp32 MFMA on the other layers)"}[
                          args.conv_math],
            "data": "synthetic",
            "config": {"workload": f"{args.model}: one {args.points}-point "
                                   f"{'Virtual-KITTI' if args.shape == 'vkitti' else 'SYNTHIA'}-shaped "
                                   "frame per step, 1600x1408x41 grid (0.05/0.05/0.1 m, T=5), "
                                   "voxelize + VFE + SpMiddleFHD forward + dense(), batch 1 "
                                   "(BASELINE.json configs[1])",
                       "model": args.model, "points_per_frame": args.points,
                       "frame_shape": args.shape, "max_voxels": args.max_voxels,
                       "path": "eager modules" if args.eager else "FrameEngine (hipGraph replay)",
                       "frames_in_flight": 1 if args.eager else args.inflight,
                       "parallelism": f"replicas x{world} (scene-sharded, no collective)"},
        }
        if roof is not None:
            line["roofline"] = roof
        if alt is not None:
            line["alt"] = alt
        if pipelined is not None:
            line["pipelined"] = pipelined
        if train is not None:
            line["train"] = train
        if world == 1 and not args.eager and not args.no_alt:
            line["light_curtain"] = light_curtain_measurement(dev, clouds_np[0],
                                                              not args.no_cpu_baseline)
            if args.model != "M3" and args.shape == "vkitti":
                line["dense_lc"] = dense_lc_measurement(dev, args.model)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.model, clouds_np[0], args.cpu_budget,
                                                args.max_voxels)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
