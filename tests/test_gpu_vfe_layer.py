"""GPU: the VFE's training path with each layer's tail as one liblcf call (lcf_vfe_layer_fwd / _bwd:
batch statistics with the padding rows in closed form + BatchNorm affine + ReLU + per-voxel max +
concatenation, and their backward through the statistics) against (a) the same compact path in
torch ops and (b) the reference's dense [K, T, C] formulation (voxel_encoder.py:113-140, which
tests/test_vfe_golden.py pins to the reference's own train-mode vectors).
The layer call itself is checked against a float64 closed form: outputs and every gradient within
2e-6 of the max magnitude.  Through the whole three-layer VFE, outputs and running statistics agree
to 2e-5; gradients are compared in L2 (2e-2): fp32 sums in another order move layer-1 outputs by an
ulp, which re-routes the gradient at near-ties of the per-voxel max of the later layers (the torch
compact form and the dense form differ from each other by the same amount)."""
import copy

import pytest
import torch
import torch.nn.functional as F

from objectdetection_amd import synth
from objectdetection_amd.frontend import SparseFrontEnd, randomize_

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def _l2(a, b):
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


def _closed_form(x, xp, wp, seg, gamma, beta, n, k, eps, mode):
    """voxel_encoder.py:65-80 / 126-139 over the real rows with the padding rows in closed form
    (objectdetection_amd/voxel_encoder.py::_forward_compact), torch ops, any dtype."""
    if mode == 0:
        w = wp.unsqueeze(1)
        mean = (x.sum(0) + (w * xp).sum(0)) / n
        var = (((x - mean) ** 2).sum(0) + (w * (xp - mean) ** 2).sum(0)) / n
    else:
        mean = x.sum(0) / n
        var = (((x - mean) ** 2).sum(0) + (n - x.shape[0]) * mean ** 2) / n
    scale = gamma * torch.rsqrt(var + eps)
    shift = beta - mean * scale
    y = F.relu(x * scale + shift)
    neg = torch.finfo(x.dtype).min
    has_pad = (wp > 0).unsqueeze(1)
    if mode == 0:
        pad = F.relu(xp * scale + shift)
        init = torch.where(has_pad, pad, torch.full_like(pad, neg))
    elif mode == 1:
        pad = F.relu(shift).expand(k, -1)
        init = torch.where(has_pad, pad, torch.full_like(pad, neg))
    else:
        init = torch.zeros(k, x.shape[1], dtype=x.dtype, device=x.device)
    idx = seg.long().unsqueeze(1).expand(-1, x.shape[1])
    agg = init.scatter_reduce(0, idx, y, "amax", include_self=True)
    if mode == 2:
        return agg, mean, var
    return torch.cat([y, agg[seg.long()]], 1), mean, var


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("c", [16, 64, 128])
def test_layer_call_matches_float64_closed_form(cuda, mode, c):
    from objectdetection_amd import ops
    torch.manual_seed(10 * mode + c)
    k, T = 3000, 5
    num = torch.randint(1, T + 1, (k,), device=cuda)
    num[::7] = T  # voxels without padding rows
    seg = torch.repeat_interleave(torch.arange(k, device=cuda), num).int()
    n = k * T
    wp = (T - num).float()
    x = torch.randn(seg.numel(), c, device=cuda, dtype=torch.float64)
    xp = torch.randn(k, c, device=cuda, dtype=torch.float64)
    gamma = torch.rand(c, device=cuda, dtype=torch.float64) + 0.5
    beta = torch.randn(c, device=cuda, dtype=torch.float64) * 0.3
    l64 = [t.clone().requires_grad_(True) for t in (x, xp, gamma, beta)]
    want, mean, var = _closed_form(l64[0], l64[1], wp.double(), seg, l64[2], l64[3], n, k, 1e-3,
                                   mode)
    g = torch.randn_like(want)
    want.backward(g)
    l32 = [t.detach().float().requires_grad_(True) for t in (x, xp, gamma, beta)]
    got, stats = ops.VFELayerFunction.apply(l32[0], l32[1] if mode == 0 else None, wp, seg, l32[2],
                                            l32[3], n, k, 1e-3, mode)
    got.backward(g.float())
    assert _rel(got.detach().double(), want.detach()) < 2e-6
    assert _rel(stats[2 * c:3 * c].double(), mean.detach()) < 2e-6
    assert _rel(stats[3 * c:4 * c].double(), var.detach()) < 2e-6
    for i, name in enumerate(("dx", "dxp", "dgamma", "dbeta")):
        if name == "dxp" and mode != 0:
            assert l32[1].grad is None
            continue
        assert _rel(l32[i].grad.double(), l64[i].grad) < 2e-6, name


def _run(vfe, voxels, num, coors, g):
    for p in vfe.parameters():
        p.grad = None
    v = voxels.clone().requires_grad_(True)
    out = vfe(v, num, coors)
    out.backward(g)
    grads = {n: p.grad.clone() for n, p in vfe.named_parameters()}
    stats = {n: b.clone() for n, b in vfe.named_buffers()}
    return out.detach(), v.grad, grads, stats


@pytest.mark.parametrize("frames,points", [(2, 20000), (1, 3000)])
def test_fused_layer_training_matches_torch_and_dense_forms(cuda, frames, points):
    model = randomize_(SparseFrontEnd("M2"), 3).to(cuda).train()
    clouds = [torch.from_numpy(synth.vkitti_frame(50 + i, points)).to(cuda) for i in range(frames)]
    with torch.no_grad():
        voxels, num, coors = model.voxelize(clouds)
    k = voxels.shape[0]
    assert int(num.max()) > 1 and int((num < voxels.shape[1]).sum()) > 0
    g = torch.randn(k, 128, device=cuda)
    real = (torch.arange(voxels.shape[1], device=cuda)[None, :] < num[:, None]).unsqueeze(-1)
    base = model.voxel_feature_extractor
    variants = {}
    for name, (compact, fused) in {"fused": (True, True), "torch": (True, False),
                                   "dense": (False, False)}.items():
        vfe = copy.deepcopy(base)
        vfe.compact_training, vfe.fused_layer_training = compact, fused
        variants[name] = _run(vfe, voxels, num, coors, g)
    out, gin, grads, stats = variants["fused"]
    for other in ("torch", "dense"):
        o2, gin2, grads2, stats2 = variants[other]
        assert _rel(out, o2) < 2e-5, other
        assert _l2(gin * real, gin2 * real) < 2e-2, other  # the dense form also feeds padding slots
        for n in grads:
            assert _l2(grads[n], grads2[n]) < 2e-2, (other, n)
        for n in stats:
            if stats[n].dtype.is_floating_point:
                assert _rel(stats[n], stats2[n]) < 2e-5, (other, n)
            else:
                assert torch.equal(stats[n], stats2[n]), (other, n)


def test_fused_layer_is_deterministic_and_rejects_bad_shapes(cuda):
    from objectdetection_amd import _lib, ops
    model = randomize_(SparseFrontEnd("M2"), 4).to(cuda).train()
    clouds = [torch.from_numpy(synth.vkitti_frame(60, 20000)).to(cuda)]
    with torch.no_grad():
        voxels, num, coors = model.voxelize(clouds)
    g = torch.randn(voxels.shape[0], 128, device=cuda)
    a = _run(copy.deepcopy(model.voxel_feature_extractor), voxels, num, coors, g)
    b = _run(copy.deepcopy(model.voxel_feature_extractor), voxels, num, coors, g)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for n in a[2]:
        if "linear" not in n:  # the Linear weight gradients use fp32 atomics (lcf_linear_wgrad)
            assert torch.equal(a[2][n], b[2][n]), n
    L = _lib.lib()
    buf = torch.zeros(1 << 20, dtype=torch.uint8, device=cuda)
    p = buf.data_ptr()
    assert L.lcf_vfe_layer_fwd(p, 100, None, p, 50, 18, 500, p, p, p, 1e-3, 1, p, p, p, p,
                               buf.numel(), 0) == -3   # LCF_ERR_UNSUPPORTED: c % 4
    assert L.lcf_vfe_layer_fwd(p, 100, None, p, 50, 16, 500, p, p, p, 1e-3, 0, p, p, p, p,
                               buf.numel(), 0) == -1   # LCF_ERR_ARG: mode 0 without xp
    assert L.lcf_vfe_layer_fwd(p, 100, None, p, 50, 16, 500, p, p, p, 1e-3, 1, p, p, p, p, 16,
                               0) == -2                # LCF_ERR_WORKSPACE
    assert not ops.vfe_layer_supported(torch.zeros(4, 18, device=cuda), torch.nn.BatchNorm1d(18))
