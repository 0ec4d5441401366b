"""Generates tests/golden/vfe_golden.npz from the REAL reference file
/root/reference/second/pytorch/models/voxel_encoder.py (run in the build container only; the
reference never travels to the GPU box).  The fixture holds data only: inputs, explicit
parameters / BatchNorm statistics, and the reference's outputs in eval and train mode.

    python tests/golden/make_vfe_golden.py

Shims: torchplus/tools.py imports numba (unused there) and two torchplus files do
`from collections import Iterable`; the module is loaded by file path so that
second/pytorch/models/__init__.py (which drags in spconv) is not executed.
"""
import collections
import collections.abc
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vfe_golden.npz")


def load_reference_module():
    collections.Iterable = collections.abc.Iterable
    sys.modules.setdefault("numba", types.ModuleType("numba"))
    sys.path.insert(0, REF)
    spec = importlib.util.spec_from_file_location(
        "ref_voxel_encoder", os.path.join(REF, "second/pytorch/models/voxel_encoder.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def fill(module, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g) * (2.0 / p.shape[1]) ** 0.5)
            elif name.endswith("weight"):
                p.copy_(torch.rand(p.shape, generator=g) * 0.5 + 0.75)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        for name, b in module.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.1)
            elif name.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=g) * 0.5 + 0.75)


def main():
    ref = load_reference_module()
    g = torch.Generator().manual_seed(1234)
    K, T = 257, 5
    num = torch.randint(1, T + 1, (K,), generator=g, dtype=torch.int32)
    vox = torch.randn((K, T, 4), generator=g)
    vox[:, :, 0] = vox[:, :, 0] * 20 + 30
    vox[:, :, 1] = vox[:, :, 1] * 15
    vox = vox * (torch.arange(T)[None, :, None] < num[:, None, None])  # zero padding rows
    coors = torch.zeros((K, 4), dtype=torch.int32)
    out = {"voxels": vox.numpy(), "num_points": num.numpy()}
    mask = ref.get_paddings_indicator(num, T, axis=0)
    out["paddings_indicator"] = mask.numpy()
    cases = {
        "SimpleVoxel": dict(),
        "SimpleVoxelRadius": dict(),
        "VoxelFeatureExtractor": dict(num_filters=[32, 128]),
        "VoxelFeatureExtractorV2": dict(num_filters=[16, 32, 64]),
        "VoxelFeatureExtractor_nonorm": dict(num_filters=[32, 128], use_norm=False),
        "VoxelFeatureExtractor_dist": dict(num_filters=[32, 128], with_distance=True),
    }
    for tag, kw in cases.items():
        cls = getattr(ref, tag.split("_")[0])
        m = cls(num_input_features=4, **kw)
        fill(m, 7)
        for k, v in m.state_dict().items():
            out[f"{tag}/param/{k}"] = v.numpy().copy()
        m.eval()
        with torch.no_grad():
            out[f"{tag}/eval"] = m(vox.clone(), num, coors).numpy()
        m.train()
        with torch.no_grad():
            out[f"{tag}/train"] = m(vox.clone(), num, coors).numpy()
        for k, v in m.state_dict().items():  # BatchNorm statistics after one training pass
            if "running" in k:
                out[f"{tag}/after/{k}"] = v.numpy().copy()
    # VFELayer alone
    layer = ref.VFELayer(7, 32)
    fill(layer, 3)
    x = torch.randn((K, T, 7), generator=g)
    for k, v in layer.state_dict().items():
        out[f"VFELayer/param/{k}"] = v.numpy().copy()
    layer.eval()
    with torch.no_grad():
        out["VFELayer/input"] = x.numpy()
        out["VFELayer/eval"] = layer(x).numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
