"""CPU: the VFE oracle and the product's VFE modules (torch path) against golden vectors captured
from the real reference file (tests/golden/make_vfe_golden.py).  Tolerance 2e-5 relative: fp32
summation order differs between numpy / torch builds."""
import os

import numpy as np
import pytest
import torch

from oracle import vfe_oracle

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "vfe_golden.npz"))
TOL = 2e-5


def rel_err(a, b):
    return np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30)


def params(tag):
    pre = tag + "/param/"
    return {k[len(pre):]: G[k] for k in G.files if k.startswith(pre)}


def test_paddings_indicator():
    got = vfe_oracle.paddings_indicator(G["num_points"], 5)
    assert np.array_equal(got, G["paddings_indicator"])
    from objectdetection_amd.voxel_encoder import get_paddings_indicator
    got = get_paddings_indicator(torch.from_numpy(G["num_points"]), 5).numpy()
    assert np.array_equal(got, G["paddings_indicator"])


def test_simple_voxels_oracle():
    v, n = G["voxels"], G["num_points"]
    assert rel_err(vfe_oracle.simple_voxel(v, n), G["SimpleVoxel/eval"]) < TOL
    assert rel_err(vfe_oracle.simple_voxel_radius(v, n), G["SimpleVoxelRadius/eval"]) < TOL


@pytest.mark.parametrize("tag,prefixes,dist", [
    ("VoxelFeatureExtractor", ("vfe1.", "vfe2."), False),
    ("VoxelFeatureExtractorV2", ("vfe_layers.0.", "vfe_layers.1.", "vfe_layers.2."), False),
    ("VoxelFeatureExtractor_nonorm", ("vfe1.", "vfe2."), False),
    ("VoxelFeatureExtractor_dist", ("vfe1.", "vfe2."), True)])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_vfe_oracle_matches_reference(tag, prefixes, dist, mode):
    got = vfe_oracle.voxel_feature_extractor(G["voxels"], G["num_points"], params(tag), prefixes,
                                             dist, train=(mode == "train"))
    assert rel_err(got, G[f"{tag}/{mode}"]) < TOL


def test_vfe_layer_oracle():
    got = vfe_oracle.vfe_layer(G["VFELayer/input"], params("VFELayer"), "")
    assert rel_err(got, G["VFELayer/eval"]) < TOL


@pytest.mark.parametrize("tag,kw", [
    ("SimpleVoxel", {}), ("SimpleVoxelRadius", {}),
    ("VoxelFeatureExtractor", dict(num_filters=[32, 128])),
    ("VoxelFeatureExtractorV2", dict(num_filters=[16, 32, 64])),
    ("VoxelFeatureExtractor_nonorm", dict(num_filters=[32, 128], use_norm=False)),
    ("VoxelFeatureExtractor_dist", dict(num_filters=[32, 128], with_distance=True))])
def test_product_modules_load_reference_state_and_match(tag, kw):
    """Same class names, same state_dict keys (strict load), same outputs in eval and train mode,
    same BatchNorm running statistics after one training pass."""
    from objectdetection_amd import voxel_encoder as ve
    m = ve.get_vfe_class(tag.split("_")[0])(num_input_features=4, **kw)
    sd = {k: torch.from_numpy(v.copy()) for k, v in params(tag).items()}
    m.load_state_dict(sd, strict=True)
    vox, num = torch.from_numpy(G["voxels"]), torch.from_numpy(G["num_points"])
    m.eval()
    with torch.no_grad():
        assert rel_err(m(vox, num, None).numpy(), G[f"{tag}/eval"]) < TOL
    m.train()
    with torch.no_grad():
        assert rel_err(m(vox, num, None).numpy(), G[f"{tag}/train"]) < TOL
    for k, v in m.state_dict().items():
        if "running" in k:
            assert rel_err(v.numpy(), G[f"{tag}/after/{k}"]) < TOL, k


@pytest.mark.parametrize("cls,kw", [("VoxelFeatureExtractor", dict(num_filters=[32, 128])),
                                    ("VoxelFeatureExtractorV2", dict(num_filters=[16, 32, 64])),
                                    ("VoxelFeatureExtractor", dict(num_filters=[32, 128],
                                                                   with_distance=True))])
def test_compact_training_forward_equals_dense_form(cls, kw):
    """The training forward over the real points only (_forward_compact: padding rows enter the
    batch statistics and the per-voxel max in closed form) against the reference's dense
    [K, T, C] form: outputs, parameter gradients and running statistics."""
    import copy
    from objectdetection_amd import voxel_encoder as ve
    torch.manual_seed(3)
    k, t = 700, 5
    num = torch.randint(1, t + 1, (k,), dtype=torch.int32)
    num[torch.rand(k) < 0.7] = 1                      # mostly one point per voxel, some full
    vox = torch.randn(k, t, 4) * (torch.arange(t)[None, :, None] < num[:, None, None])
    a = ve.get_vfe_class(cls)(num_input_features=4, **kw).train()
    for p in a.parameters():
        torch.nn.init.normal_(p, std=0.3)
    for m in a.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            torch.nn.init.uniform_(m.weight, 0.5, 1.5)
            torch.nn.init.normal_(m.bias, std=0.3)
    b = copy.deepcopy(a)
    b.compact_training = False
    g = torch.randn(k, a.linear.out_features)
    ya, yb = a(vox, num, None), b(vox, num, None)
    assert rel_err(ya.detach().numpy(), yb.detach().numpy()) < TOL
    (ya * g).sum().backward()
    (yb * g).sum().backward()
    for (na, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert rel_err(pa.grad.numpy(), pb.grad.numpy()) < 1e-4, na
    for (na, ba), (_, bb) in zip(a.named_buffers(), b.named_buffers()):
        assert rel_err(ba.numpy(), bb.numpy()) < TOL, na
