"""FrameEngine: the whole hot path of one frame as ONE hipGraph replay, no host synchronisation.

The reference runs voxelisation on the host (second/data/preprocess.py:626-639), copies the
arrays to the GPU (second/pytorch/train.py:27-55) and then steps through ~40 Python modules
(VoxelNet.network_forward, second/pytorch/models/voxelnet.py:314-333); spconv synchronises once
per strided conv to size its outputs.  At batch 1 the path moves a few MB, so it is bound by
launch latency and those synchronisations, not by the kernels.

MI355X form: every row count of the path (points, voxels, active sites of each level) stays in
device memory.  liblcf's entry points take (capacity, device count) pairs (include/lcf.h), the
buffers are allocated once for the capacities -- 288 GB of HBM makes generous capacities free --
and the ~60 launches of a frame are captured into a hipGraph that is replayed per frame.
The result is bit-identical to the module-by-module path (same kernels, same order; both use the
fused VFE kernel when its shape applies).  Rulebooks depend on coordinates only, so their whole
chain runs on a side stream -- a second branch of the graph -- beside the feature path.

Every internal buffer reset is a fill KERNEL (lcf_fill32), not hipMemsetAsync: on ROCm 7.2 a
captured graph holding hipMemsetAsync nodes hung at its second replay on this path.

Capacities of the strided levels are calibrated on sample frames (x ``margin``); a frame that
exceeds them sets a device flag, reported by ``check()`` (``forward(..., sync=True)`` measures the
frame with loose capacities, re-plans from what it needs and re-runs it, once).

Thread safety: an engine owns its lcf_ctx (options, fill batch) and its buffers; two engines may
enqueue from two threads.  One engine is not meant to be driven from two threads at once.
"""
import os

import torch
from torch import nn

from . import _lib, ops
from ._lib import check, f3, f6, i3
from .spconv.conv import SparseConvolution
from .spconv.modules import _bn_affine
from .voxel_encoder import SimpleVoxel, VoxelFeatureExtractor, _affine


def _round_up(x, m):
    return (int(x) + m - 1) // m * m


class _Layer:
    pass


class FrameEngine:
    def __init__(self, model, max_points=None, calib=None, margin=2.0, use_graph=True,
                 two_streams=True, conv_math="f32", dense_out=None, dense_clear="full"):
        """model: objectdetection_amd.frontend.SparseFrontEnd in eval mode on a GPU.
        max_points: capacity of the point buffer (default: 2 x the largest calibration cloud).
        calib: list of [N,F] float32 cuda clouds used to size the strided levels.
        conv_math: "f32" (fp32 MFMA, the default) or "bf16x3" (split operands on the bf16 matrix
        cores, lcf_indice_conv_fwd_bf16x3: ~1e-5 of the features' max magnitude instead of ~1e-7,
        inside the 1e-4 feature tolerance; layers with 32/64/128 channels on both sides) or
        "bf16x8" (three exact bf16 planes, eight products: fp32-grade results, ~1e-7, at half the
        matrix-pipe cycles of the fp32 MFMA; the 64-channel layers only -- it does not pay on
        the 32-channel ones) or "bf16x6" (round 3: rows travel between the layers as three
        exact bf16 planes, six products, no compaction, no tile plans -- lcf_indice_conv_fwd_x6;
        the same distance from float64 as the fp32 kernels; used on the layers where it measured
        ahead of the fp32 quad-row kernel at the calibrated row counts, the others stay fp32).
        dense_out: a contiguous [1,C,D,H,W] float32 cuda tensor the frame's dense output is written
        to instead of a buffer of the engine's own (FrameBatch: a slice of the batch's output)."""
        self._dense_out = dense_out
        if conv_math not in ("f32", "bf16x3", "bf16x8", "bf16x6"):
            raise ValueError("conv_math must be 'f32', 'bf16x3', 'bf16x8' or 'bf16x6'")
        self.conv_math = conv_math
        p = next(model.parameters(), None)
        self.dev = p.device if p is not None else torch.device("cuda", torch.cuda.current_device())
        if self.dev.type != "cuda":
            raise RuntimeError("FrameEngine runs on the GPU only")
        if model.training:
            raise RuntimeError("FrameEngine folds BatchNorm: put the model in eval() mode")
        self.model = model
        self.L = _lib.lib()
        # this engine's own context (include/lcf.h: lcf_ctx): the options its calls run with and
        # its open fill batch live here, never in process-wide state, so engines enqueueing from
        # different threads (or beside plain ops.* callers) do not see each other
        self.ctx = _lib.Context()
        self._ws_dirty = False  # an enqueue died half-way: workspaces are not re-armed
        self.use_graph = use_graph
        self.two_streams = bool(two_streams)
        # dense() fused into the last conv's epilogue (lcf_indice_conv_fwd_dense)
        self.fuse_dense = os.environ.get("LCF_ENGINE_FUSE_DENSE", "1") != "0"
        # dense_clear="rows": the dense output is cleared by ROWS -- only the cells the previous
        # frame wrote (lcf_dense_clear_rows; SURVEY.md 8f-1: ~2 MB of stores instead of the whole
        # 18 MB buffer per frame).  Measured in round 3 (A/B on one box, 300 frames each, twice):
        # 1642 / 1641 frames/s with the whole-buffer fill, 1627 / 1626 with the row clear -- the
        # rows are ~8k x 64 isolated 4-byte stores, one cache line each, and cost the memory system
        # more than 18 MB of streaming 16-byte stores on the side stream do.  So the fill stays the
        # default and the row clear is an option (LCF_ENGINE_DENSE_CLEAR=rows|full overrides).
        self.clear_rows = os.environ.get("LCF_ENGINE_DENSE_CLEAR", dense_clear) == "rows"
        self.side = torch.cuda.Stream(device=self.dev) if self.two_streams else None
        # capacities = margin x the calibrated counts.  The hash tables of the rulebook chain are
        # sized from them, and emptier tables probe less: 2.0 is the fastest in the frame (1.1 /
        # 1.25: 0.618 ms, 1.5: 0.613, 2.0: 0.608, 2.5: 0.610, 3: 0.613, 4: 0.618 -- beyond that the
        # re-arming fills grow faster than the probes shrink; LCF_ENGINE_MARGIN overrides)
        self.margin = float(os.environ.get("LCF_ENGINE_MARGIN", margin))
        vg = model.voxel_generator
        self.vs = [float(x) for x in vg.voxel_size]
        self.rng = [float(x) for x in vg.point_cloud_range]
        self.T = int(vg.max_num_points_per_voxel)
        self.max_voxels = int(model.max_voxels)
        self.cap_mode = int(vg._cap_mode)
        self.F = 4
        vfe = model.voxel_feature_extractor
        if isinstance(vfe, VoxelFeatureExtractor) and not vfe._with_distance:
            self.vfe_kind = "vfe"
        elif type(vfe) is SimpleVoxel:
            self.vfe_kind = "mean"
        else:
            raise NotImplementedError(f"FrameEngine: unsupported VFE {type(vfe).__name__}")
        mid = model.middle_feature_extractor
        self.sparse_shape = [int(x) for x in mid.sparse_shape]
        self._parse_middle(list(mid.middle_conv))
        calib = list(calib or [])
        if max_points is None:
            max_points = 2 * max([c.shape[0] for c in calib] + [16384])
        self.max_points = int(max_points)
        self.graph = None
        self.level_caps = None
        # weight images and folded BatchNorm: once (they do not depend on the capacities)
        self._fold()
        # plan with worst-case-ish capacities, measure, then re-plan tight and capture
        self._plan(None)
        self.level_need = None  # rows per level seen in calibration (kernel-selection hint)
        if calib:
            need = None
            for c in calib:
                self._load(c)
                self._enqueue()
                ks = self.counts()
                if self.overflowed():
                    raise RuntimeError("FrameEngine: calibration frame exceeds the loose caps")
                need = ks if need is None else [max(a, b) for a, b in zip(need, ks)]
            self.level_need = list(need)
            self._plan([_round_up(self.margin * k + 2048, 1024) for k in need[1:]])
        if use_graph:
            self._capture()

    # ------------------------------------------------------------------ static description
    def _parse_middle(self, mods):
        self.layers = []
        i = 0
        shape = list(self.sparse_shape)
        while i < len(mods):
            conv = mods[i]
            if not isinstance(conv, SparseConvolution) or conv.ndim != 3:
                raise NotImplementedError("FrameEngine: middle_conv must be 3-D sparse convs")
            bn, relu = None, False
            j = i + 1
            if j < len(mods) and isinstance(mods[j], nn.BatchNorm1d):
                bn = mods[j]
                j += 1
            elif j < len(mods) and not isinstance(mods[j], (nn.ReLU, SparseConvolution)):
                j += 1  # identity placeholder (use_norm=False)
            if j < len(mods) and isinstance(mods[j], nn.ReLU):
                relu = True
                j += 1
            l = _Layer()
            l.conv, l.bn, l.relu = conv, bn, relu
            l.subm = conv.subm
            l.key = conv.indice_key
            l.cin, l.cout = conv.in_channels, conv.out_channels
            l.ksize = list(conv.kernel_size)
            l.kvol = l.ksize[0] * l.ksize[1] * l.ksize[2]
            l.stride, l.pad, l.dil = list(conv.stride), list(conv.padding), list(conv.dilation)
            l.in_shape = list(shape)
            if not l.subm:
                shape = ops.conv_out_shape(shape, l.ksize, l.stride, l.pad, l.dil)
            l.out_shape = list(shape)
            self.layers.append(l)
            i = j
        self.out_shape = shape
        self.n_strided = sum(1 for l in self.layers if not l.subm)

    def _fold(self):
        """Weight images and folded BatchNorm (eval) of every layer; VFE linear images."""
        with torch.no_grad():
            for l in self.layers:
                l.x3 = (self.conv_math == "bf16x3" and ops.bf16_supported(l.cin, l.cout)) or (
                    self.conv_math == "bf16x8" and l.cin >= 64 and
                    ops.bf16_supported(l.cin, l.cout))
                l.planes = 3 if self.conv_math == "bf16x8" else 2
                l.wimg = ops.weight_image_bf16x3(l.conv.weight, planes=l.planes) if l.x3 \
                    else ops.weight_image(l.conv.weight)
                # "bf16x6": rows as three exact bf16 planes, six products (csrc/conv_bx.h);
                # which of the candidate layers run it is decided per frame size in _plan
                l.x6_ok = (self.conv_math == "bf16x6" and l is not self.layers[-1] and
                           ops.x6_conv_supported(l.cin, l.cout, l.kvol))
                l.wimg3 = ops.weight_image_bf16x3(l.conv.weight, planes=3) if l.x6_ok else None
                scale = shift = None
                if l.bn is not None:
                    scale, shift = _bn_affine(l.bn)
                if l.conv.bias is not None:
                    b = l.conv.bias.float()
                    shift = (b * scale + shift) if scale is not None else b
                l.scale = None if scale is None else scale.contiguous()
                l.shift = None if shift is None else shift.contiguous()
            if self.vfe_kind == "vfe":
                vfe = self.model.voxel_feature_extractor
                self.lin = []
                for lin, norm in ((vfe.vfe1.linear, vfe.vfe1.norm), (vfe.vfe2.linear, vfe.vfe2.norm),
                                  (vfe.linear, vfe.norm)):
                    w = lin.weight.detach().float().t().contiguous()  # [in, out]
                    cin, cout = w.shape
                    scale, shift = _affine(norm, cout, w.device)
                    if lin.bias is not None:
                        b = lin.bias.detach().float()
                        shift = (b * scale + shift) if scale is not None else b
                    self.lin.append((ops.weight_image(w.view(1, cin, cout)), cin, cout,
                                     scale, None if shift is None else shift.contiguous()))

    # ------------------------------------------------------------------ buffers
    def _plan(self, caps):
        """Allocate every buffer of a frame.  caps: capacity of each strided level's output (rows);
        None = loose (worst-case bounded) capacities used for calibration."""
        dev, L = self.dev, self.L
        i32, f32 = torch.int32, torch.float32
        MV, T, F = self.max_voxels, self.T, self.F
        E = lambda *s, dtype=f32: torch.empty(s, dtype=dtype, device=dev)
        self.points = E(self.max_points, F)
        self.n_dev = torch.zeros(1, dtype=i32, device=dev)
        self.voxels, self.coors = E(MV, T, F), E(MV, 4, dtype=i32)
        self.num, self.vnum = E(MV, dtype=i32), torch.zeros(1, dtype=i32, device=dev)
        self.vox_ws = E(max(L.lcf_voxelize_workspace_bytes(self.max_points, T), 256),
                        dtype=torch.uint8)
        self.vfe_fused = False
        if self.vfe_kind == "vfe":
            (_, c0, u1, _, _), (_, _, u2, _, _), (_, _, u3, _, _) = self.lin
            probe = ops.vfe_fused(torch.zeros((1, T, F), device=dev),
                                  torch.ones(1, dtype=i32, device=dev), self.lin)
            self.vfe_fused = probe is not None
        if self.vfe_kind == "vfe" and self.vfe_fused:
            self.feat0 = E(MV, u3)
        elif self.vfe_kind == "vfe":
            self.kt = torch.zeros(1, dtype=i32, device=dev)
            self.aug = E(MV, T, c0)
            self.x1, self.p1 = E(MV * T, u1), E(MV, T, 2 * u1)
            self.x2, self.p2 = E(MV * T, u2), E(MV, T, 2 * u2)
            self.x3 = E(MV * T, u3)
            self.feat0 = E(MV, u3)
        else:
            self.feat0 = E(MV, self.layers[0].cin)
        # middle: all rulebooks of the stack come from ONE lcf_rulebook_chain call (one level per
        # indice_key / per strided conv); capacities: MV rows at the input level, then each
        # strided conv's output capacity
        specs, cap, keys = [], MV, {}
        for l in self.layers:
            l.k_in_cap = cap
            if l.subm:
                if l.key is not None and l.key in keys and keys[l.key].k_in_cap == cap:
                    l.rb_owner = keys[l.key]
                else:
                    l.rb_owner = l
                    l.level = len(specs)
                    specs.append(dict(subm=True, ksize=l.ksize, dilation=l.dil))
                    if l.key is not None:
                        keys[l.key] = l
                l.k_out_cap = cap
            else:
                worst = L.lcf_rulebook_conv_max_out(cap, i3(l.ksize), i3(l.stride))
                cells = l.out_shape[0] * l.out_shape[1] * l.out_shape[2]
                loose = min(worst, cells, 8 * MV)
                si = sum(1 for sp in specs if not sp["subm"])
                l.k_out_cap = loose if caps is None else min(int(caps[si]), min(worst, cells))
                l.k_out_cap = max(l.k_out_cap, 64)
                l.rb_owner = l
                l.level = len(specs)
                specs.append(dict(subm=False, ksize=l.ksize, stride=l.stride, padding=l.pad,
                                  dilation=l.dil))
                cap = l.k_out_cap
            l.out = E(l.k_out_cap, l.cout)
        # tile plans: the tile-compacted convs' per-tile set-up depends on the rulebook only, so the
        # chain builds it once per rulebook (beside the feature path) instead of every launch
        # re-deriving it; the tile height is the one the forward would pick for the rows expected
        plan_rows, lvl = {}, 0
        self.use_plans = os.environ.get("LCF_ENGINE_TILE_PLANS", "1") != "0"
        # conv_math="bf16x6": the six-product kernel on the layers where it measured ahead of the
        # fp32 quad-row kernel ALONE (tools/bench_bx.py): the 32 -> 32 layers (22 against 31 us at
        # 46k rows) and the 64 -> 64 ones while one workgroup per CU holds all rows (<= 256 tiles of
        # 64 rows: 33 against 40 us at 15.5k rows; 47 against 42 at 18.6k, 69 against 72 at 38k).
        # In the frame, beside the rulebook branch, that advantage is gone (tools/ab_x6.sh:
        # 1638 frames/s against 1641 with the fp32 kernels; on every candidate layer 1439), which
        # is why "f32" stays the default.  LCF_ENGINE_X6 = all | none overrides.  Calibration (no
        # row counts yet) runs the fp32 kernels.
        x6_mode = os.environ.get("LCF_ENGINE_X6", "auto")
        need0 = getattr(self, "level_need", None)
        for l in self.layers:
            if not l.subm:
                lvl += 1
            rows_exp = int(need0[lvl]) if need0 else 0
            l.use_x6 = bool(getattr(l, "x6_ok", False) and need0 and x6_mode != "none" and (
                x6_mode == "all" or (l.cin == 32 and l.cout == 32) or
                (l.cin == 64 and l.cout == 64 and (rows_exp + 63) // 64 <= 256)))
        for i, l in enumerate(self.layers):
            nxt = self.layers[i + 1] if i + 1 < len(self.layers) else None
            prv = self.layers[i - 1] if i > 0 else None
            l.out_planes = bool(l.use_x6 and nxt is not None and nxt.use_x6)
            l.split_in = bool(l.use_x6 and not (prv is not None and prv.use_x6))
            l.out3 = torch.zeros((3, l.k_out_cap, l.cout), dtype=torch.bfloat16, device=dev) \
                if l.out_planes else None
            l.in3 = torch.zeros((3, l.k_in_cap, l.cin), dtype=torch.bfloat16, device=dev) \
                if l.split_in else None
        lvl = 0
        for l in self.layers:
            if not l.subm:
                lvl += 1
            l.planned = (self.use_plans and l.kvol > 1 and not l.use_x6 and
                         ops.bf16_supported(l.cin, l.cout))
            if l.planned:
                need = getattr(self, "level_need", None)
                rows = int(need[lvl]) if need else l.k_out_cap
                # (the fp32 3x3x3 convs run the quad-row kernel, csrc/conv_q4.h, which cuts its own
                # tile heights; the split-operand forms keep the heights balanced over the CUs)
                tr = L.lcf_conv_tile_rows_for(min(rows, l.k_out_cap), l.kvol, l.cin, l.cout,
                                              (3 if l.planes == 3 else 2) if l.x3 else 0)
                plan_rows[l.rb_owner.level] = tr
        self.chain = ops.RulebookChain(
            specs, MV, 1, self.sparse_shape, device=dev, plan_rows=plan_rows,
            caps=[l.k_out_cap for l in self.layers if not l.subm])
        self.flags = self.chain.flags
        self.level_caps = [MV]
        for l in self.layers:
            if l.rb_owner is l:
                l.nbr = self.chain.out_nbr[l.level]
                if not l.subm:
                    l.out_coors = self.chain.out_coors[l.level]
                    l.k_out = self.chain.k_out[l.level]
                    assert l.out_coors.shape[0] == l.k_out_cap
                    self.level_caps.append(l.k_out_cap)
        # one event per rulebook: recorded by the chain as soon as that table is complete (the
        # handles must exist before the call: record once here)
        self.rb_events = []
        for i in range(self.chain.n):
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            self.chain.set_event(i, ev)
            self.rb_events.append(ev)
        C = self.layers[-1].cout
        D, H, W = self.out_shape
        if self._dense_out is None:
            self.dense = E(1, C, D, H, W)
        else:
            d = self._dense_out
            if tuple(d.shape) != (1, C, D, H, W) or d.dtype != f32 or d.device != dev \
                    or not d.is_contiguous():
                raise ValueError(f"dense_out must be a contiguous float32 [1,{C},{D},{H},{W}] "
                                 f"tensor on {dev}")
            self.dense = d
        self.out = self.dense.view(1, C * D, H, W)
        # rows-only clearing: the coordinates the rows of the dense output come from (the last
        # strided level's, or the voxels') and where the previous frame's rows went (-1 = none)
        self.final_coors, self.final_k, self.final_cap = self.coors, self.vnum, MV
        for l in self.layers:
            if not l.subm:
                self.final_coors, self.final_k, self.final_cap = l.out_coors, l.k_out, l.k_out_cap
        self.prev_coors = torch.full((self.final_cap, 4), -1, dtype=i32, device=dev)
        self.dense.zero_()
        self.graph = None
        self._reset_workspaces(torch.cuda.current_stream(dev).cuda_stream)
        self._ws_dirty = False

    def _reset_workspaces(self, st):
        """Hash workspaces of the voxelizer and of every rulebook into their reset state.  A frame
        runs with LCF_OPT_WS_CLEAN (the calls skip their own resets) and re-arms the workspaces
        at the END of its rulebook branch, where the fills overlap the convolutions instead of
        sitting in front of the voxelizer and of every rulebook."""
        L = self.L
        cx = self.ctx.handle
        check(L.lcf_voxelize_workspace_reset(self.vox_ws.data_ptr(), self.vox_ws.numel(),
                                             self.max_points, self.T, st, cx))
        self.chain.reset(stream=st, ctx=cx)

    # ------------------------------------------------------------------ one frame, enqueued
    def _enqueue(self):
        """One frame on two streams: every rulebook depends on COORDINATES only, so the whole
        chain voxel coords -> level-1 coords -> ... (hash inserts, scans, table fills: small
        latency-bound launches) runs on a side stream beside the feature path (VFE and the convs,
        which fill the chip).  Captured into a hipGraph the cross-stream events become edges of
        the graph: two branches that join at each conv."""
        self.ctx.set(_lib.OPT_WS_CLEAN, int(os.environ.get("LCF_ENGINE_WS_CLEAN", "1")))
        if self._ws_dirty:
            # the previous enqueue raised before its re-arming fills: the tables may hold a
            # half-built frame, and this frame's calls skip their own resets
            self.ctx.fill_batch_flush(torch.cuda.current_stream(self.dev).cuda_stream)
            self._reset_workspaces(torch.cuda.current_stream(self.dev).cuda_stream)
            self.dense.zero_()  # (and which of its cells hold a frame is unknown)
            self.prev_coors.fill_(-1)
        self._ws_dirty = True
        self._enqueue_clean()
        self._ws_dirty = False

    def _enqueue_clean(self):
        L = self.L
        cx = self.ctx.handle
        main = torch.cuda.current_stream(self.dev)
        side = self.side if self.two_streams else main
        st, sst = main.cuda_stream, side.cuda_stream
        P = lambda t: None if t is None else t.data_ptr()
        MV, T, F = self.max_voxels, self.T, self.F
        if self.two_streams:  # the previous frame is complete before this one's branch starts
            side.wait_stream(main)
        check(L.lcf_voxelize(P(self.points), self.max_points, P(self.n_dev), F, f3(self.vs),
                             f6(self.rng), T, MV, self.cap_mode, 0, 4, P(self.voxels),
                             P(self.coors), P(self.num), P(self.vnum), P(self.vox_ws),
                             self.vox_ws.numel(), st, cx))
        kd = self.vnum
        # ---- side stream: rulebook chain + clearing of the dense output.  The ORDER in which the
        # two branches are enqueued matters to the graph executor of ROCm 7.2 (kernel traces of
        # the replay): a node's cross-stream wait covers more of the
        # producing stream than its edge says.  With the whole chain first the VFE starts 37 us
        # after the voxelizer ends; with the VFE first it starts at once (+0.4 % frames/s).  The
        # chain can also be enqueued in pieces (lcf_rulebook_chain_parts), each in front of the
        # first conv that needs it: every join then costs ~10 us of cross-queue latency that the
        # run-ahead chain hides (-1.5 %), so that is not the default.
        ready = {}
        # LCF_ENGINE_ORDER: chain_first (whole chain, then the VFE), vfe_first (the VFE, then the
        # whole chain), split0 (VFE, part 0, level-0 convs, rest of the chain), interleave
        order = os.environ.get("LCF_ENGINE_ORDER", "vfe_first") if self.two_streams else "chain_first"
        kd0 = kd
        side_state = {"done": -1, "e_dense": None, "e_side": None}
        last_part = self.chain.num_parts - 1
        if self.two_streams:
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            for l in self.layers:
                if l.rb_owner is l:
                    ready[id(l)] = self.rb_events[l.level]

        def side_tail():
            # the dense clear and the re-arming of the hash workspaces (for the next frame) as ONE
            # fill launch at the end of the branch
            with torch.cuda.stream(side):
                batch = os.environ.get("LCF_ENGINE_FILL_BATCH", "1") != "0"
                if batch:
                    self.ctx.fill_batch_begin()
                try:
                    if self.clear_rows:
                        check(L.lcf_dense_clear_rows(
                            P(self.dense), P(self.prev_coors), P(self.final_coors),
                            self.final_cap, P(self.final_k), self.dense.shape[1], 1,
                            i3(self.out_shape), sst, cx))
                    else:
                        check(L.lcf_fill_u32(P(self.dense), 0, self.dense.numel() * 4, sst, cx))
                    if self.two_streams and not batch:
                        side_state["e_dense"] = torch.cuda.Event()
                        side_state["e_dense"].record(side)
                    self._reset_workspaces(sst)
                finally:
                    if batch:
                        self.ctx.fill_batch_flush(sst)
                if self.two_streams:
                    side_state["e_side"] = torch.cuda.Event()
                    side_state["e_side"].record(side)
                    if batch:
                        side_state["e_dense"] = side_state["e_side"]

        def side_parts(upto):
            if side_state["done"] >= upto:
                return
            with torch.cuda.stream(side):
                self.chain.run(self.coors, k_dev=kd0, stream=sst, ctx=cx,
                               parts=(side_state["done"] + 1, upto))
            side_state["done"] = upto
            if upto == last_part:
                side_tail()

        if order == "chain_first":
            side_parts(last_part)
        # ---- main stream: features
        if self.vfe_kind == "vfe" and self.vfe_fused:
            (w1, c0, u1, s1, h1), (w2, _, u2, s2, h2), (w3, _, u3, s3, h3) = self.lin
            check(L.lcf_vfe_fused(P(self.voxels), P(self.num), MV, P(kd), T, F, u1, u2, u3,
                                  P(w1), P(s1), P(h1), P(w2), P(s2), P(h2), P(w3), P(s3), P(h3),
                                  P(self.feat0), st))
        elif self.vfe_kind == "vfe":
            torch.mul(self.vnum, T, out=self.kt)
            (w1, c0, u1, s1, h1), (w2, _, u2, s2, h2), (w3, _, u3, s3, h3) = self.lin
            check(L.lcf_vfe_augment(P(self.voxels), P(self.num), MV, P(kd), T, F, P(self.aug), st))
            for x_in, cin, (w, cout, sc, sh), x_out, pool_out, mode in (
                    (self.aug, c0, (w1, u1, s1, h1), self.x1, self.p1, 0),
                    (self.p1, 2 * u1, (w2, u2, s2, h2), self.x2, self.p2, 0),
                    (self.p2, 2 * u2, (w3, u3, s3, h3), self.x3, self.feat0, 1)):
                check(L.lcf_indice_conv_fwd(P(x_in), P(w), None, MV * T, P(self.kt), 1, cin, cout,
                                            P(sc), P(sh), 1, P(x_out), st, cx))
                check(L.lcf_vfe_pool(P(x_out), P(self.num), MV, P(kd), T, cout, None, None, mode,
                                     P(pool_out), st))
        else:
            check(L.lcf_vfe_mean(P(self.voxels), P(self.num), MV, P(kd), T, F,
                                 self.layers[0].cin, P(self.feat0), st))
        if order == "vfe_first":
            side_parts(last_part)
        feat, coors = self.feat0, self.coors
        feat3 = None  # the rows as three bf16 planes, between two "bf16x6" layers
        waited = set()
        lvl = 0
        for l in self.layers:
            o = l.rb_owner
            if id(o) not in waited:
                need = self.chain.part_of_level[o.level]
                side_parts(last_part if order == "split0" and need > 0 else need)
                if self.two_streams:
                    main.wait_event(ready[id(o)])
                waited.add(id(o))
            kd_in = kd
            if l.subm:
                k_out_cap, k_out_dev = l.k_in_cap, kd
            else:
                k_out_cap, k_out_dev = l.k_out_cap, l.k_out
                lvl += 1
            # the launch is sized by the capacity; the kernel variant by the rows expected
            need = self.level_need
            self.ctx.set(_lib.OPT_CONV_ROWS_HINT, int(need[lvl]) if need else 0)
            if not l.subm:
                coors, kd = l.out_coors, l.k_out
            # dense() in the epilogue of the last conv: its rows go straight to [1,C,D,H,W]
            to_dense = l is self.layers[-1] and self.fuse_dense and l.cout in (16, 32, 64, 128)
            if to_dense and self.two_streams:
                side_parts(last_part)
                main.wait_event(side_state["e_dense"])
            if l.use_x6:
                if l.split_in:
                    check(L.lcf_split_planes(P(feat), l.k_in_cap, P(kd_in), l.cin, P(l.in3),
                                             l.k_in_cap, st))
                    feat3 = l.in3
                check(L.lcf_indice_conv_fwd_x6(
                    P(feat3), feat3.shape[1], P(l.wimg3), P(o.nbr), k_out_cap, P(k_out_dev),
                    l.kvol, l.cin, l.cout, P(l.scale), P(l.shift), int(l.relu),
                    P(l.out3 if l.out_planes else l.out), int(l.out_planes), k_out_cap, st))
                feat3 = l.out3
            elif l.planned:
                lv = o.level
                check(L.lcf_indice_conv_fwd_planned(
                    P(feat), P(l.wimg), (3 if l.planes == 3 else 2) if l.x3 else 0,
                    P(self.chain.plan[lv]), self.chain.plan_rows[lv], k_out_cap, P(k_out_dev),
                    l.kvol, l.cin, l.cout, P(l.scale), P(l.shift), int(l.relu),
                    P(self.dense if to_dense else l.out), P(coors) if to_dense else None, 1,
                    i3(self.out_shape), st, cx))
            elif l.x3:
                fwd_split = L.lcf_indice_conv_fwd_bf16x8 if l.planes == 3 \
                    else L.lcf_indice_conv_fwd_bf16x3
                check(fwd_split(
                    P(feat), P(l.wimg), P(o.nbr), k_out_cap, P(k_out_dev), l.kvol, l.cin, l.cout,
                    P(l.scale), P(l.shift), int(l.relu), P(self.dense if to_dense else l.out),
                    P(coors) if to_dense else None, 1, i3(self.out_shape), st, cx))
            elif to_dense:
                check(L.lcf_indice_conv_fwd_dense(
                    P(feat), P(l.wimg), P(o.nbr), k_out_cap, P(k_out_dev), l.kvol, l.cin, l.cout,
                    P(l.scale), P(l.shift), int(l.relu), P(coors), 1, i3(self.out_shape),
                    P(self.dense), st, cx))
            else:
                check(L.lcf_indice_conv_fwd(P(feat), P(l.wimg), P(o.nbr), k_out_cap,
                                            P(k_out_dev), l.kvol, l.cin, l.cout, P(l.scale),
                                            P(l.shift), int(l.relu), P(l.out), st, cx))
            feat = None if to_dense else l.out
        self.ctx.set(_lib.OPT_CONV_ROWS_HINT, 0)
        side_parts(last_part)
        if feat is not None:
            if self.two_streams:
                main.wait_event(side_state["e_dense"])
            check(L.lcf_dense(P(feat), P(coors), feat.shape[0], P(kd), feat.shape[1], 1,
                              i3(self.out_shape), 0, P(self.dense), st))
        if self.two_streams:
            main.wait_event(side_state["e_side"])  # joins the side stream (its last operation)

    def _capture(self):
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):  # warm-up outside capture (function attributes, lazy init)
            self._enqueue()
        torch.cuda.current_stream(self.dev).wait_stream(s)
        torch.cuda.synchronize(self.dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._enqueue()
        self.graph = g

    # ------------------------------------------------------------------ public
    def _load(self, points):
        n = int(points.shape[0])
        if n > self.max_points:
            raise RuntimeError(f"FrameEngine: {n} points exceed max_points={self.max_points}")
        if points.shape[1] != self.F or points.dtype != torch.float32 or not points.is_cuda:
            raise ValueError("points must be a [N,4] float32 cuda tensor")
        self._copy_in(points, 0, n, n)
        self.n_points = n

    def _copy_in(self, src, row0, rows, total):
        """points[row0 : row0 + rows] = src and n_dev = total, as one launch where the float4 copy
        applies (F = 4 rows are 16 bytes), else a copy and a fill."""
        if rows > 0 and src.is_contiguous() and src.data_ptr() % 16 == 0:
            rc = self.L.lcf_load_points(src.data_ptr(), rows, self.F,
                                        self.points[row0:].data_ptr(), total,
                                        self.n_dev.data_ptr(),
                                        torch.cuda.current_stream(self.dev).cuda_stream)
            if rc == 0:
                return
        if rows > 0:
            self.points[row0:row0 + rows].copy_(src, non_blocking=True)
        self.n_dev.fill_(total)

    def forward_append(self, new_points, sync=False):
        """The light-curtain loop (second/light_curtain/utils.py:121-124 stacks every curtain's
        return under the cloud and runs the detector again): the cumulative cloud stays in the
        engine's point buffer, only the NEW points [M,4] f32 cuda are copied behind it, and the
        frame is re-run.  Same result as forward(torch.cat([previous cloud, new_points]))."""
        m = int(new_points.shape[0])
        n0 = getattr(self, "n_points", 0)
        if n0 + m > self.max_points:
            raise RuntimeError(f"FrameEngine: {n0 + m} points exceed max_points={self.max_points}")
        if new_points.shape[1] != self.F or new_points.dtype != torch.float32 \
                or not new_points.is_cuda:
            raise ValueError("points must be a [M,4] float32 cuda tensor")
        self.n_points = n0 + m
        self._copy_in(new_points, n0, m, self.n_points)
        return self._run(self.points[:self.n_points], sync)

    def forward(self, points, sync=False):
        """points [N,4] f32 cuda -> [1, C*D, H, W] (the engine's static output buffer: consume or
        clone it before the next call).  Nothing here waits for the GPU unless sync=True.

        FAILURE BEHAVIOUR differs from the spconv-API path, which sizes every level from a host
        read and cannot overflow: here a frame with more active sites than a level's calibrated
        capacity (2 x the calibration frames' counts) is TRUNCATED at that level -- rows beyond
        the capacity are dropped, the returned features are those of the truncated frame -- and
        only a device flag records it.  Nothing raises unless the caller asks: ``check()`` (one
        synchronisation, e.g. once per batch of frames) raises, ``overflowed()`` tells, and
        ``forward(..., sync=True)`` checks every frame and re-plans + re-runs it by itself."""
        self._load(points)
        return self._run(points, sync)

    def _run(self, points, sync):
        if self.graph is not None:
            self.graph.replay()
        else:
            self._enqueue()
        if sync and self.overflowed():
            self._replan_for(points.clone())
        return self.out

    def _replan_for(self, points):
        """A frame exceeded a level capacity: measure it with the loose (worst-case bounded)
        capacities, re-plan tight from what it needs (x margin, merged with the calibration
        counts), re-capture, and run it again -- once; a frame that does not fit even the loose
        capacities raises."""
        self.graph = None
        self._plan(None)
        self._load(points)
        self._enqueue()
        if self.overflowed():
            raise RuntimeError("FrameEngine: the frame exceeds even the loose level capacities "
                               f"{self.level_caps} (max_voxels = {self.max_voxels})")
        ks = self.counts()
        need = ks if self.level_need is None else [max(a, b) for a, b in zip(self.level_need, ks)]
        self.level_need = list(need)
        self._plan([_round_up(self.margin * k + 2048, 1024) for k in need[1:]])
        if self.use_graph:
            self._capture()
        self._load(points)
        if self.graph is not None:
            self.graph.replay()
        else:
            self._enqueue()
        self.check()

    __call__ = forward

    def overflowed(self):
        """True if the last frame exceeded a level capacity (synchronises)."""
        return bool(self.flags.any().item())

    def check(self):
        if self.overflowed():
            raise RuntimeError(
                "FrameEngine: a strided level exceeded its capacity "
                f"{self.level_caps}; re-create the engine with a larger margin or calibrate on "
                "denser frames (or call forward(..., sync=True), which re-plans by itself)")

    def scan_fallbacks(self):
        """Blocks of the voxelizer's one-launch scan that stopped waiting for a predecessor and
        counted for themselves in the last frame (include/lcf.h: lcf_voxelize_scan_fallbacks;
        the frame is right either way, only slower).  Synchronises."""
        import ctypes
        ptr = ctypes.c_void_p()
        check(self.L.lcf_voxelize_scan_fallbacks(self.vox_ws.data_ptr(), self.vox_ws.numel(),
                                                 self.max_points, self.T, ctypes.byref(ptr)))
        off = ptr.value - self.vox_ws.data_ptr()
        return int(self.vox_ws[off:off + 4].view(torch.int32).item())

    def counts(self):
        """[voxels, active sites after each strided conv] of the last frame (synchronises)."""
        ks = [self.vnum] + [l.k_out for l in self.layers if not l.subm]
        return [int(x) for x in torch.cat(ks).tolist()]


class FramePipeline:
    """Throughput serving of independent batch-1 frames: ``depth`` FrameEngines (own buffers, own
    captured graph, own context) on ``depth`` streams, frames dealt to them in turn.  The same
    kernels and the same results as one engine; with the quad-row convs a second frame in flight
    fills what one frame leaves idle (tails of the conv launches, a wave's waits, the latency-
    bound front of a frame): 1629 -> 2035 frames/s at depth 2 (bench.py ``pipelined``; depth 3 is
    worse).  A frame's latency is then about ``depth`` x the time per frame.

        pipe = FramePipeline(front, calib=[cloud], depth=2)
        for cloud in stream_of_clouds:
            out, done = pipe.submit(cloud)   # out: that slot's static buffer, valid after `done`
            ...                              # (and until the slot is used again, depth frames on)
        pipe.check()
    """

    def __init__(self, model, calib=None, depth=2, **engine_kw):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.engines = [FrameEngine(model, calib=calib, **engine_kw) for _ in range(depth)]
        dev = self.engines[0].dev
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
        self.events = [torch.cuda.Event() for _ in range(depth)]
        self.i = 0

    def submit(self, points, wait_current=True):
        """Enqueues one frame on the next slot -> (output buffer of that slot, event recorded
        behind the frame on the slot's stream).  With ``wait_current`` the slot's stream first
        waits for the caller's current stream, so ``points`` may have been produced there; a
        caller whose clouds are already resident passes False (the cross-stream wait costs 13 %
        of the pipelined rate)."""
        k = self.i % len(self.engines)
        self.i += 1
        st = self.streams[k]
        if wait_current:
            st.wait_stream(torch.cuda.current_stream(self.engines[k].dev))
        with torch.cuda.stream(st):
            out = self.engines[k].forward(points)
            self.events[k].record(st)
        return out, self.events[k]

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def check(self):
        """Raises if any slot's last frame exceeded a level capacity (synchronises)."""
        for e in self.engines:
            e.check()


class FrameBatch:
    """A batch of B clouds per call (the reference's example batches, second/pytorch/train.py:
    example_convert_to_torch -> net(example) with batch_size > 1): in eval mode nothing couples the
    frames of a batch (BatchNorm is folded, every rulebook is per batch index), so the batch is B
    FrameEngines on B streams whose dense outputs are the B slices of ONE [B, C*D, H, W] buffer --
    row b is what SparseFrontEnd gives for cloud b in a batch.  The engines overlap as in
    FramePipeline; the caller's stream waits for all of them before forward() returns.

        eng = FrameBatch(front, batch=2, calib=[cloud])
        spatial = eng([cloud_a, cloud_b])      # [2, C*D, H, W], static buffer
    """

    def __init__(self, model, batch, calib=None, **engine_kw):
        if batch < 1:
            raise ValueError("batch must be >= 1")
        if "dense_out" in engine_kw:
            raise ValueError("FrameBatch owns the output buffer")
        p = next(model.parameters(), None)
        dev = p.device if p is not None else torch.device("cuda", torch.cuda.current_device())
        # the dense shape from the layer list alone (no buffers planned, no weights folded)
        probe = FrameEngine.__new__(FrameEngine)
        mid = model.middle_feature_extractor
        probe.sparse_shape = [int(x) for x in mid.sparse_shape]
        probe._parse_middle(list(mid.middle_conv))
        C, (D, H, W) = probe.layers[-1].cout, probe.out_shape
        self.dense = torch.empty((batch, C, D, H, W), dtype=torch.float32, device=dev)
        self.out = self.dense.view(batch, C * D, H, W)
        self.engines = [FrameEngine(model, calib=calib, dense_out=self.dense[b:b + 1], **engine_kw)
                        for b in range(batch)]
        self.dev = dev
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(batch)]

    def forward(self, clouds, sync=False):
        """clouds: B tensors [N_b,4] f32 cuda -> [B, C*D, H, W] (static buffer).  Ordered on the
        caller's current stream on both sides."""
        if len(clouds) != len(self.engines):
            raise ValueError(f"FrameBatch: {len(clouds)} clouds for batch {len(self.engines)}")
        cur = torch.cuda.current_stream(self.dev)
        for eng, st, pts in zip(self.engines, self.streams, clouds):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                eng.forward(pts, sync=sync)
        for st in self.streams:
            cur.wait_stream(st)
        return self.out

    __call__ = forward

    def check(self):
        for e in self.engines:
            e.check()
