"""GPU parity of the whole hot path and of the drop-in module path vs the CPU oracle."""
import numpy as np
import pytest
import torch

from objectdetection_amd import synth
from objectdetection_amd.frontend import SparseFrontEnd, randomize_

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def rel_err(a, b):
    return np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("name", ["M1", "M2"])
def test_frontend_matches_cpu_pipeline(cuda, name):
    from oracle import cpu_pipeline
    pts = [synth.vkitti_frame(0, 20000)]
    ref = cpu_pipeline.frontend_forward(randomize_(SparseFrontEnd(name).eval(), 0), pts)
    model = randomize_(SparseFrontEnd(name).eval(), 0).to(cuda)
    with torch.no_grad():
        out = model([torch.from_numpy(p).to(cuda) for p in pts])
        assert rel_err(out.cpu().numpy(), ref) < RTOL
        # the unfused module-by-module path gives the same answer
        model.middle_feature_extractor.middle_conv.fuse_inference = False
        out2 = model([torch.from_numpy(p).to(cuda) for p in pts])
        assert rel_err(out2.cpu().numpy(), ref) < RTOL
    assert (out != 0).any()


@pytest.mark.parametrize("name", ["M1", "M2", "M3"])
def test_frontend_bf16x3_matches_cpu_pipeline_at_fp32_tolerance(cuda, name):
    """The whole front end with its wide sparse convs on split bf16 operands
    (ops.conv_math("bf16x3")) against the CPU pipeline: inside the north star's 1e-4, and its
    distance from the fp32 path reported by the assertion message."""
    from oracle import cpu_pipeline
    from objectdetection_amd import ops
    pts = [synth.vkitti_frame(5, 20000)]
    ref = cpu_pipeline.frontend_forward(randomize_(SparseFrontEnd(name).eval(), 4), pts)
    model = randomize_(SparseFrontEnd(name).eval(), 4).to(cuda)
    with torch.no_grad():
        c = [torch.from_numpy(p).to(cuda) for p in pts]
        out32 = model(c)
        with ops.conv_math("bf16x3"):
            out3 = model(c)
    e32, e3 = rel_err(out32.cpu().numpy(), ref), rel_err(out3.cpu().numpy(), ref)
    assert e32 < RTOL and e3 < RTOL, (e32, e3)
    assert e3 < 5e-5, (e32, e3)
    assert not torch.equal(out3, out32)  # the mode is really in effect


def test_frontend_batch2_and_m3(cuda):
    from oracle import cpu_pipeline
    pts = [synth.vkitti_frame(1, 20000), synth.vkitti_frame(2, 12000)]
    ref = cpu_pipeline.frontend_forward(randomize_(SparseFrontEnd("M1").eval(), 1), pts)
    model = randomize_(SparseFrontEnd("M1").eval(), 1).to(cuda)
    with torch.no_grad():
        out = model([torch.from_numpy(p).to(cuda) for p in pts])
    assert out.shape == (2, 128, 200, 176)
    assert rel_err(out.cpu().numpy(), ref) < RTOL
    # classic grid, T=35, VoxelFeatureExtractor -> SparseMiddleExtractor (north-star names)
    pts = [synth.vkitti_frame(3, 20000)]
    ref = cpu_pipeline.frontend_forward(randomize_(SparseFrontEnd("M3").eval(), 2), pts)
    model = randomize_(SparseFrontEnd("M3").eval(), 2).to(cuda)
    with torch.no_grad():
        out = model([torch.from_numpy(p).to(cuda) for p in pts])
    assert out.shape == ref.shape == (1, 64 * 2, 400, 352)
    assert rel_err(out.cpu().numpy(), ref) < RTOL


def test_training_step_gradients_match_oracle(cuda, oracle):
    """fwd + bwd through SubM -> BN -> ReLU -> strided conv -> dense in train mode: gradients of
    the sparse conv weights and of the input match a CPU torch replay over the oracle rulebooks."""
    from oracle import cpu_pipeline
    from objectdetection_amd import spconv
    torch.manual_seed(0)
    pts = synth.vkitti_frame(4, 8000)
    _, c, _ = oracle.voxelize(pts, synth.VOXEL_SIZE, synth.PC_RANGE, 5, 40000)
    coors = np.ascontiguousarray(np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], 1))
    k = coors.shape[0]
    seq = spconv.SparseSequential(
        spconv.SubMConv3d(16, 32, 3, bias=False, indice_key="a"), torch.nn.BatchNorm1d(32),
        torch.nn.ReLU(), spconv.SparseConv3d(32, 32, 3, 2, padding=1, bias=True),
        spconv.SubMConv3d(32, 16, 3, bias=False, indice_key="b"))
    randomize_(seq, 5)
    x_np = np.random.default_rng(0).standard_normal((k, 16)).astype(np.float32)
    shape = [41, 1600, 1408]
    # CPU replay with autograd over the oracle's rulebooks
    import copy
    cpu_seq = copy.deepcopy(seq).train()
    xc = torch.from_numpy(x_np).requires_grad_(True)
    feats, cc, sh = xc, coors, shape
    cache = {}
    for m in cpu_seq:
        if hasattr(m, "subm"):
            if m.subm:
                pairs, pn = oracle.rulebook_subm(cc, 1, sh, m.kernel_size)
                oc, osh = cc, sh
            else:
                oc, pairs, pn, osh = oracle.rulebook_conv(cc, 1, sh, m.kernel_size, m.stride,
                                                          m.padding)
            w = m.weight.reshape(-1, m.in_channels, m.out_channels)
            out = torch.zeros((oc.shape[0], m.out_channels))
            for o in range(w.shape[0]):
                n = int(pn[o])
                if n:
                    i = torch.from_numpy(pairs[0, o, :n].astype(np.int64))
                    j = torch.from_numpy(pairs[1, o, :n].astype(np.int64))
                    out = out.index_add(0, j, feats[i] @ w[o])
            if m.bias is not None:
                out = out + m.bias
            feats, cc, sh = out, oc, osh
        else:
            feats = m(feats)
    loss_c = (feats ** 2).sum()
    loss_c.backward()
    # GPU
    seq = seq.to(cuda).train()
    xg = torch.from_numpy(x_np).to(cuda).requires_grad_(True)
    t = spconv.SparseConvTensor(xg, torch.from_numpy(coors).to(cuda), shape, 1)
    y = seq(t)
    assert np.array_equal(y.indices.cpu().numpy(), cc)
    d = y.dense()
    loss_g = (d ** 2).sum()
    loss_g.backward()
    assert abs(loss_g.item() - loss_c.item()) / abs(loss_c.item()) < 1e-4
    assert rel_err(xg.grad.cpu().numpy(), xc.grad.numpy()) < 1e-3
    for mg, mc in zip(seq, cpu_seq):
        for pg, pc in zip(mg.parameters(), mc.parameters()):
            assert rel_err(pg.grad.cpu().numpy(), pc.grad.numpy()) < 1e-3, type(mg)


def test_vfe_kernels_match_torch(cuda):
    from objectdetection_amd import voxel_encoder as ve
    torch.manual_seed(1)
    k, t = 5000, 5
    num = torch.randint(1, t + 1, (k,), dtype=torch.int32)
    vox = torch.randn(k, t, 4)
    vox = vox * (torch.arange(t)[None, :, None] < num[:, None, None])
    for cls in (ve.VoxelFeatureExtractor, ve.VoxelFeatureExtractorV2, ve.SimpleVoxel,
                ve.SimpleVoxelRadius):
        m = randomize_(cls(4, True, [32, 128]).eval(), 3)
        with torch.no_grad():
            ref = m(vox, num, None).numpy()           # CPU: reference torch maths
            got = m.to(cuda)(vox.to(cuda), num.to(cuda), None).cpu().numpy()  # HIP kernels
        assert got.shape == ref.shape
        assert rel_err(got, ref) < 1e-5, cls.__name__


def test_vfe_hip_path_matches_reference_golden(cuda):
    """HIP streaming VFE path vs vectors captured from the real reference file."""
    import os
    from objectdetection_amd import voxel_encoder as ve
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "vfe_golden.npz"))
    vox = torch.from_numpy(G["voxels"]).to(cuda)
    num = torch.from_numpy(G["num_points"]).to(cuda)
    for tag, kw in [("SimpleVoxel", {}), ("SimpleVoxelRadius", {}),
                    ("VoxelFeatureExtractor", dict(num_filters=[32, 128])),
                    ("VoxelFeatureExtractorV2", dict(num_filters=[16, 32, 64])),
                    ("VoxelFeatureExtractor_nonorm", dict(num_filters=[32, 128], use_norm=False))]:
        m = ve.get_vfe_class(tag.split("_")[0])(num_input_features=4, **kw)
        pre = tag + "/param/"
        m.load_state_dict({k[len(pre):]: torch.from_numpy(G[k].copy()) for k in G.files
                           if k.startswith(pre)}, strict=True)
        m = m.to(cuda).eval()
        with torch.no_grad():
            got = m(vox, num, None).cpu().numpy()
        assert rel_err(got, G[f"{tag}/eval"]) < 2e-5, tag


def test_vfe_fused_kernel_matches_layer_kernels(cuda):
    """lcf_vfe_fused (one persistent kernel) vs the layer-by-layer streaming kernels + torch
    Linear, ragged sizes, device-side voxel count; T too large for LDS falls back."""
    from objectdetection_amd import ops, voxel_encoder as ve
    torch.manual_seed(5)
    m = randomize_(ve.VoxelFeatureExtractor(4, True, [32, 128]).eval(), 7).to(cuda)
    for k, t, sparse in [(1, 5, False), (17, 5, False), (5003, 5, False), (333, 8, False),
                         (40, 1, False), (6001, 5, True), (2000, 8, True)]:
        num = torch.randint(1, t + 1, (k,), dtype=torch.int32, device=cuda)
        if sparse:  # mostly one point per voxel: the kernel skips the all-padding row tiles
            few = torch.rand(k, device=cuda) < 0.9
            num = torch.where(few, torch.ones_like(num), num)
        vox = torch.randn(k, t, 4, device=cuda)
        vox = vox * (torch.arange(t, device=cuda)[None, :, None] < num[:, None, None])
        with torch.no_grad():
            lin = m._fused_lin([m.vfe1, m.vfe2])
            fused = ops.vfe_fused(vox, num, lin)
            assert fused is not None
            x = ops.vfe_augment(vox, num)
            for layer in (m.vfe1, m.vfe2):
                x = layer.forward_masked(x, num)
            x = m.linear(x)
            sc, sh = ve._affine(m.norm, 128, x.device)
            layered = ops.vfe_pool(x, num, sc, sh, 1)
        assert rel_err(fused.cpu().numpy(), layered.cpu().numpy()) < 1e-5, (k, t, sparse)
        # device-side count: only the first kd voxels are written
        kd = max(1, k // 2)
        out = torch.full((k, 128), -1.0, device=cuda)
        ops.vfe_fused(vox, num, lin, k_dev=torch.tensor([kd], dtype=torch.int32, device=cuda),
                      out=out)
        assert torch.equal(out[:kd], fused[:kd]) and bool((out[kd:] == -1).all())
    big_t = torch.zeros(4, 35, 4, device=cuda)
    assert ops.vfe_fused(big_t, torch.ones(4, dtype=torch.int32, device=cuda), lin) is None
