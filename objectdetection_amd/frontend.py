"""The hot path end to end on one GPU: points -> voxelize -> VFE -> sparse middle -> dense map
(SURVEY.md call stack E; what VoxelNet.network_forward runs before the RPN,
second/pytorch/models/voxelnet.py:314-333, plus the voxelization of
second/data/preprocess.py:626-639 moved onto the device)."""
import torch
from torch import nn

from . import middle as middle_mod
from . import voxel_encoder
from .spconv.utils import VoxelGeneratorV2
from .synth import MAX_POINTS_PER_VOXEL, PC_RANGE, VOXEL_SIZE

MODELS = {
    # SURVEY.md section 8(d) model configs
    "M1": dict(vfe="SimpleVoxel", middle="SpMiddleFHD", vfe_out=4,
               voxel_size=VOXEL_SIZE, max_points=MAX_POINTS_PER_VOXEL),
    "M2": dict(vfe="VoxelFeatureExtractor", middle="SpMiddleFHD", vfe_out=128,
               voxel_size=VOXEL_SIZE, max_points=MAX_POINTS_PER_VOXEL),
    "M3": dict(vfe="VoxelFeatureExtractor", middle="SparseMiddleExtractor", vfe_out=128,
               voxel_size=(0.2, 0.2, 0.4), max_points=35),
}


class SparseFrontEnd(nn.Module):
    """voxel generator + VFE + middle extractor with the reference's shapes.

    forward(list of [N_i,4] cuda point tensors) -> [B, C*D, H, W] spatial features."""

    def __init__(self, model="M2", max_voxels=40000, pc_range=PC_RANGE, num_point_features=4):
        super().__init__()
        cfg = MODELS[model]
        self.cfg = cfg
        self.max_voxels = max_voxels
        self.voxel_generator = VoxelGeneratorV2(list(cfg["voxel_size"]), list(pc_range),
                                                cfg["max_points"], max_voxels=max_voxels)
        grid = self.voxel_generator.grid_size  # xyz
        dense_shape = [1] + grid[::-1].tolist() + [cfg["vfe_out"]]  # second_builder.py:30-31
        self.voxel_feature_extractor = voxel_encoder.get_vfe_class(cfg["vfe"])(
            num_point_features, True, [32, 128], False, voxel_size=cfg["voxel_size"],
            pc_range=pc_range)
        self.middle_feature_extractor = middle_mod.get_middle_class(cfg["middle"])(
            dense_shape, True, num_input_features=cfg["vfe_out"])

    def voxelize(self, clouds):
        """-> voxels [sumK,T,F], num_points [sumK], coors [sumK,4] (b,z,y,x)  (merge_second_batch,
        second/data/preprocess.py:21-55, on the device)."""
        # every scene is enqueued first; the voxel counts come back in ONE synchronisation
        res = [self.voxel_generator.generate_device(pts, self.max_voxels, batch_idx=b)
               for b, pts in enumerate(clouds)]
        # (with them, the number of points kept: rows past voxel_num hold num_points = 0; the
        # VFE's training path compacts the real rows without a synchronisation of its own)
        kept = [r["num_points_per_voxel"].sum(dtype=torch.int32).view(1) for r in res]
        counts = torch.cat([r["voxel_num"] for r in res] + kept).tolist()
        ks = counts[:len(res)]
        self.kept_points = int(sum(counts[len(res):]))
        vs = [r["voxels"][:k] for r, k in zip(res, ks)]
        ns = [r["num_points_per_voxel"][:k] for r, k in zip(res, ks)]
        cs = [r["coordinates"][:k] for r, k in zip(res, ks)]
        if len(clouds) == 1:
            return vs[0], ns[0], cs[0]
        return torch.cat(vs), torch.cat(ns), torch.cat(cs)

    def engine(self, calib=None, max_points=None, margin=2.0, use_graph=True, two_streams=True,
               conv_math="f32", **engine_kw):
        """Batch-1 inference engine: the whole frame as one hipGraph replay without host
        synchronisation (objectdetection_amd/engine.py).  ``calib``: sample clouds that size it."""
        from .engine import FrameEngine
        return FrameEngine(self, max_points=max_points, calib=calib, margin=margin,
                           use_graph=use_graph, two_streams=two_streams, conv_math=conv_math,
                           **engine_kw)

    def forward(self, clouds):
        voxels, num_points, coors = self.voxelize(clouds)
        self.voxel_feature_extractor.real_rows_hint = self.kept_points
        try:
            feats = self.voxel_feature_extractor(voxels, num_points, coors)
        finally:
            self.voxel_feature_extractor.real_rows_hint = None
        return self.middle_feature_extractor(feats, coors, len(clouds))


def randomize_(module, seed=0):
    """Deterministic "trained-like" parameters for tests and benchmarks (no checkpoint is
    available): He-scaled sparse-conv / linear weights and non-trivial BatchNorm statistics, so
    that activations keep O(1) magnitude through the 14 layers (spconv's default
    kaiming_uniform_ on the 5-D weight shrinks them to denormals)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in module.modules():
            if hasattr(m, "subm") and hasattr(m, "kernel_size"):
                cin = m.in_channels
                taps = 27 if m.subm else 8
                std = (2.0 / (cin * max(1.0, taps * 0.4))) ** 0.5
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * std)
                if m.bias is not None:
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            elif isinstance(m, nn.Linear):
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) *
                               (2.0 / m.in_features) ** 0.5)
                if m.bias is not None:
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            elif isinstance(m, nn.BatchNorm1d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.5 + 0.75)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) * 0.5 + 0.75)
    return module
