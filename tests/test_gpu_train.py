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
