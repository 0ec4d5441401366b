"""GPU: the training step of bench.py's `train` leg (objectdetection_amd/train.py): the fused
paths it takes by default (compact VFE forward, fused BatchNorm1d + ReLU kernels, flat-bucket
clipping) give the same loss and gradients as the plain torch forms of the same modules."""
import copy

import pytest
import torch

from objectdetection_amd import ops, synth
from objectdetection_amd.train import Detector, TrainStep, synthetic_loss
from objectdetection_amd.frontend import randomize_

pytestmark = pytest.mark.gpu


def _grads(net, clouds, seed):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    for p in net.parameters():
        p.grad = None
    with ops.conv_math("f32"):
        loss = synthetic_loss(net(clouds), gen)
        loss.backward()
    return loss.detach(), {n: p.grad.detach().clone() for n, p in net.named_parameters()
                           if p.grad is not None}


def test_fused_training_paths_match_plain_torch_modules(cuda):
    clouds = [torch.from_numpy(synth.vkitti_frame(40 + i, 7000)).to(cuda) for i in range(2)]
    a = randomize_(Detector("M2", rpn_bf16=False), 3).to(cuda).train()
    b = copy.deepcopy(a)
    b.frontend.voxel_feature_extractor.compact_training = False
    b.frontend.middle_feature_extractor.middle_conv.fuse_training = False
    la, ga = _grads(a, clouds, 0)
    lb, gb = _grads(b, clouds, 0)
    assert torch.isfinite(la) and abs(la.item() - lb.item()) / abs(lb.item()) < 1e-4
    assert ga.keys() == gb.keys() and len(ga) > 60
    # fp32 reassociation (different BatchNorm / pooling kernels) is amplified through 14 sparse
    # layers + the RPN and can re-route a gradient at a near-tie of a max / ReLU: compare in L2
    for n in ga:
        num = (ga[n] - gb[n]).float().norm().item()
        den = max(gb[n].float().norm().item(), 1e-12)
        assert num / den < 2e-2, (n, num / den)
    # running statistics moved identically
    for (n, x), (_, y) in zip(a.named_buffers(), b.named_buffers()):
        assert torch.allclose(x.float(), y.float(), rtol=1e-4, atol=1e-6), n


def test_train_step_runs_and_learns(cuda):
    step = TrainStep(cuda, "M2", lr=1e-3, seed=1)
    clouds = [torch.from_numpy(synth.vkitti_frame(60 + i, 6000)).to(cuda) for i in range(2)]
    losses = [step(clouds).item() for _ in range(6)]
    assert all(torch.isfinite(torch.tensor(losses)))
    assert all(torch.isfinite(p.grad).all() for p in step.net.parameters() if p.grad is not None)
    # same scenes, fresh random targets each step: the regression part still has to go down
    assert min(losses[3:]) < losses[0]


class _ReplayConv(torch.autograd.Function):
    """CPU indice_conv over the oracle's pair lists.  bf16 = True rounds the operands exactly where
    the bf16 kernels do (include/lcf.h: features / dout when they enter LDS, the weight image);
    accumulation fp32, weight gradient from unrounded fp32 operands (wgrad stays fp32-grade)."""

    @staticmethod
    def forward(ctx, feats, w, pairs, pn, k_out, bf16):
        rnd = (lambda t: t.bfloat16().float()) if bf16 else (lambda t: t)
        ctx.save_for_backward(feats, w)
        ctx.meta = (pairs, pn, bf16)
        xr, wr = rnd(feats), rnd(w)
        out = torch.zeros((k_out, w.shape[-1]))
        for o in range(w.shape[0]):
            n = int(pn[o])
            if n:
                out.index_add_(0, pairs[1][o][:n], xr[pairs[0][o][:n]] @ wr[o])
        return out

    @staticmethod
    def backward(ctx, g):
        feats, w = ctx.saved_tensors
        pairs, pn, bf16 = ctx.meta
        rnd = (lambda t: t.bfloat16().float()) if bf16 else (lambda t: t)
        gr, wr = rnd(g), rnd(w)
        dx, dw = torch.zeros_like(feats), torch.zeros_like(w)
        for o in range(w.shape[0]):
            n = int(pn[o])
            if n:
                i, j = pairs[0][o][:n], pairs[1][o][:n]
                dx.index_add_(0, i, gr[j] @ wr[o].t())
                dw[o] = feats[i].t() @ g[j]
        return dx, dw, None, None, None, None


def test_batch8_bf16_middle_gradients_match_oracle_replay(cuda, oracle):
    """BASELINE.json configs[2]: batch 8, train mode, through the whole SpMiddleFHD stack (14
    sparse convs + BatchNorm1d + ReLU): the sparse-conv weight gradients of the GPU step -- fused
    training BatchNorm kernels, fp32 and then bf16 MFMA operands in forward + dgrad (what TrainStep
    runs) -- against a CPU fp32 autograd replay over the ORACLE's rulebooks.

    Tolerances, per gradient tensor in L2.  fp32: 3e-3 (measured 5e-5 at the layer next to the
    loss growing to 1.1e-3 at the first: this stack -- random weights, train-mode BatchNorm -- about
    doubles a relative perturbation per layer, 1e-7 of reassociation becomes 1e-3).  bf16: the
    stated 2e-2 at the two layers next to the loss (measured 2.6e-3 and 1e-2); below them the
    same amplification applies to the 2^-9 operand rounding of every layer above (measured 0.18 at
    the first layer; 0.09 against a replay that rounds the same operands, i.e. half of it is
    which way near-ties round), so the whole-stack bound is 0.3: a property of bf16 operands on
    this stack, not of the kernels (per-layer: tests/test_gpu_conv.py
    ::test_bf16_forward_and_dgrad_match_oracle, 2e-2; rounding is RNE, bit-exact vs torch)."""
    import copy
    import numpy as np
    from objectdetection_amd import spconv
    from objectdetection_amd.frontend import SparseFrontEnd
    torch.manual_seed(0)
    front = randomize_(SparseFrontEnd("M2", max_voxels=17000), 11).to(cuda).train()
    clouds = [torch.from_numpy(synth.vkitti_frame(70 + i, 5000)).to(cuda) for i in range(8)]
    with torch.no_grad():
        voxels, num, coors = front.voxelize(clouds)
        front.eval()
        feats0 = front.voxel_feature_extractor(voxels, num, coors).contiguous()
        front.train()
    mid = front.middle_feature_extractor
    seq = mid.middle_conv
    shape = [int(s) for s in mid.sparse_shape]
    assert coors.shape[0] > 30000
    tgt = None

    def replay(bf16):
        nonlocal tgt
        cpu_seq = copy.deepcopy(seq).cpu().float().train()
        feats = feats0.cpu().clone()
        cc, sh, cache = coors.cpu().numpy(), shape, {}
        for m in cpu_seq:
            if hasattr(m, "subm") and hasattr(m, "kernel_size"):
                key = m.indice_key
                if key is not None and key in cache:
                    oc, pairs, pn, osh = cache[key]
                elif m.subm:
                    pairs, pn = oracle.rulebook_subm(cc, 8, sh, m.kernel_size, m.dilation)
                    oc, osh = cc, sh
                else:
                    oc, pairs, pn, osh = oracle.rulebook_conv(cc, 8, sh, m.kernel_size, m.stride,
                                                              m.padding, m.dilation)
                if key is not None:
                    cache[key] = (oc, pairs, pn, osh)
                w = m.weight.reshape(-1, m.in_channels, m.out_channels)
                pt = torch.from_numpy(pairs.astype(np.int64))
                narrow = bf16 and ops.bf16_supported(m.in_channels, m.out_channels)
                feats = _ReplayConv.apply(feats, w, pt, pn, oc.shape[0], narrow)
                cc, sh = oc, osh
            else:
                feats = m(feats)
        if tgt is None:
            tgt = torch.from_numpy(np.random.default_rng(0).standard_normal(
                feats.shape).astype(np.float32))
        ((feats - tgt) ** 2).mean().backward()
        return [m.weight.grad.clone() for m in cpu_seq if hasattr(m, "subm")], cc

    ref32, cc = replay(False)

    def gpu(math):
        for p_ in seq.parameters():
            p_.grad = None
        with ops.conv_math(math):
            y = seq(spconv.SparseConvTensor(feats0.clone(), coors, shape, 8))
            assert np.array_equal(y.indices.cpu().numpy(), cc)
            ((y.features - tgt.to(cuda)) ** 2).mean().backward()
        return [m.weight.grad.float().cpu() for m in seq if hasattr(m, "subm")]

    def errs(got, ref):
        return [((g - r).norm() / r.norm()).item() for g, r in zip(got, ref)]

    g32, g16 = gpu("f32"), gpu("bf16")
    assert len(g32) == 14
    e = errs(g32, ref32)
    assert max(e) < 3e-3, ("fp32 vs fp32 replay", e)
    e = errs(g16, ref32)
    assert e[-1] < 2e-2 and e[-2] < 2e-2 and max(e) < 0.3, ("bf16 vs fp32 replay", e)


def test_rpn_fused_batchnorm_relu_matches_the_stock_modules(cuda):
    """ops.bn_relu_train_nhwc (BatchNorm2d + ReLU of the RPN on channels_last bf16 activations as
    liblcf's row kernels) against nn.BatchNorm2d + nn.ReLU on the same bf16 input under autocast:
    output (one bf16 ulp), input / weight / bias gradients, running statistics; and RPNV2 as a
    whole with and without the fusion."""
    import copy
    torch.manual_seed(0)
    bn = torch.nn.BatchNorm2d(128, eps=1e-3, momentum=0.01).to(cuda).train()
    torch.nn.init.uniform_(bn.weight, 0.5, 1.5)
    torch.nn.init.normal_(bn.bias, std=0.3)
    ref = copy.deepcopy(bn)
    x = (torch.randn(2, 128, 50, 44, device=cuda) * 2 + 0.5).bfloat16().contiguous(
        memory_format=torch.channels_last)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    assert ops.bn_relu_nhwc_supported(bn, xa)
    ya = ops.bn_relu_train_nhwc(bn, xa)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        yb = torch.relu(ref(xb))
    assert ya.dtype == torch.bfloat16 and ya.shape == yb.shape
    assert ya.is_contiguous(memory_format=torch.channels_last)
    scale = float(yb.detach().float().abs().max())
    assert float((ya.detach().float() - yb.detach().float()).abs().max()) / scale < 1e-2  # a bf16 ulp
    g = torch.randn_like(ya)
    ya.backward(g)
    yb.backward(g)
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
    assert rel(xa.grad, xb.grad) < 1e-2
    assert rel(bn.weight.grad, ref.weight.grad) < 5e-3 and rel(bn.bias.grad, ref.bias.grad) < 5e-3
    assert torch.allclose(bn.running_mean, ref.running_mean, rtol=1e-3, atol=1e-5)
    assert torch.allclose(bn.running_var, ref.running_var, rtol=1e-3, atol=1e-5)
    assert int(bn.num_batches_tracked) == 1
    # the whole head: bf16 activations through seven conv + BatchNorm layers amplify any
    # rounding difference, so the yardstick is the fp32 head -- the fused path must be as close
    # to it as the stock bf16 path is -- and the layer next to the loss, where nothing has been
    # amplified yet
    from objectdetection_amd.rpn import RPNV2
    a = RPNV2().to(cuda).train().to(memory_format=torch.channels_last)
    b, c = copy.deepcopy(a), copy.deepcopy(a)
    b.fuse_bn_relu = False
    xin = torch.randn(2, 128, 40, 48, device=cuda).bfloat16().contiguous(
        memory_format=torch.channels_last)
    grads = []
    for net, amp in ((a, True), (b, True), (c, False)):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            p = net(xin if amp else xin.float())
        loss = sum(v.float().pow(2).mean() for v in p.values())
        loss.backward()
        grads.append((loss.item(), net.blocks[0][1].weight.grad.float().clone(),
                      net.deblocks[0][0].weight.grad.float().clone()))
    (la, ga0, ga1), (lb, gb0, gb1), (lc, gc0, gc1) = grads
    assert abs(la - lc) / abs(lc) < 2e-2 and abs(lb - lc) / abs(lc) < 2e-2
    assert rel(ga1, gc1) < max(1.5 * rel(gb1, gc1), 2e-2)   # last conv in front of the heads
    assert rel(ga0, gc0) < max(1.5 * rel(gb0, gc0), 5e-2)   # first conv: fully amplified
