"""One training step of the detector's GPU part with the reference's shapes: voxelize -> VFE ->
sparse middle (fwd, dgrad, wgrad on liblcf) -> RPNV2 -> loss -> backward -> one flat gradient
all-reduce -> optimizer (second/pytorch/train.py:316-343 minus the data pipeline; BASELINE.json
configs[2] and [3]).  Targets are synthetic (no dataset here): the step exercises the same
kernels and the same gradient volume as the real loss (sigmoid focal classification
second/pytorch/core/losses.py + smooth-L1 box regression)."""
import torch
from torch import nn
from torch.nn import functional as F

from .dist import FlatGradAllReduce
from .frontend import SparseFrontEnd, randomize_
from .rpn import RPNV2


class Detector(nn.Module):
    def __init__(self, model="M2", max_voxels=17000, rpn_bf16=True):
        # max_voxels 17000: training cap (base.yaml:118).  rpn_bf16: the dense 2-D RPN (stock
        # torch / MIOpen, outside the hot-path scope) runs under bf16 autocast in channels_last,
        # the MI355X analogue of the reference's apex AMP switch (train.py:217-225); the sparse
        # front end keeps fp32 features and fp32 master weights.
        super().__init__()
        self.frontend = SparseFrontEnd(model, max_voxels=max_voxels)
        self.rpn = RPNV2(num_input_features=128)
        self.rpn_bf16 = rpn_bf16
        if rpn_bf16:  # dense() lands in the RPN's input layout and dtype in one pass
            self.frontend.middle_feature_extractor.dense_to = "rpn_bf16"

    # called (no arguments) when backward has produced the gradient of the front end's output,
    # i.e. when every RPN parameter's gradient is complete (TrainStep: async all-reduce of them)
    on_rpn_grads_ready = None

    def forward(self, clouds):
        x = self.frontend(clouds)
        if self.on_rpn_grads_ready is not None and x.requires_grad:
            def _hook(g, cb=self.on_rpn_grads_ready):
                cb()
                return g
            x.register_hook(_hook)
        if not (self.rpn_bf16 and x.is_cuda):
            return self.rpn(x)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            preds = self.rpn(x.contiguous(memory_format=torch.channels_last))
        return {k: v.float() for k, v in preds.items()}


def synthetic_loss(preds, gen):
    """Focal-style classification + smooth-L1 box + direction losses against random targets of
    the reference's shapes (cls [B,A,H,W,1], box [B,A,H,W,7], dir [B,A,H,W,2])."""
    cls, box = preds["cls_preds"], preds["box_preds"]
    pos = (torch.rand(cls.shape, device=cls.device, generator=gen) < 0.01).float()
    p = torch.sigmoid(cls)
    ce = F.binary_cross_entropy_with_logits(cls, pos, reduction="none")
    mod = (pos * (1 - p) + (1 - pos) * p) ** 2.0
    alpha = pos * 0.25 + (1 - pos) * 0.75
    n_pos = pos.sum().clamp(min=1.0)
    loss = (ce * mod * alpha).sum() / n_pos
    tgt = torch.randn(box.shape, device=box.device, generator=gen)
    loss = loss + 2.0 * (F.smooth_l1_loss(box, tgt, reduction="none") *
                         pos.expand_as(box[..., :1])).sum() / n_pos
    if "dir_cls_preds" in preds:
        d = preds["dir_cls_preds"]
        lab = (torch.rand(d.shape[:-1], device=d.device, generator=gen) < 0.5).long()
        dl = F.cross_entropy(d.reshape(-1, d.shape[-1]), lab.reshape(-1), reduction="none")
        loss = loss + 0.2 * (dl * pos.reshape(-1)).sum() / n_pos
    return loss


class TrainStep:
    """Owns the model, the flat gradient bucket and the optimizer of ONE rank."""

    def __init__(self, device, model="M2", lr=3e-4, seed=0, process_group=None,
                 conv_math="bf16", freeze_gc_after=5, overlap_allreduce=True):
        # conv_math "bf16": indice_conv forward + dgrad with bf16 MFMA operands and fp32
        # accumulation (BASELINE.json configs[2]); fp32 master weights, fp32 wgrad.
        self.device = device
        self.conv_math = conv_math
        # let MIOpen time its solvers for the RPN's dense convs once per shape: its untuned
        # heuristic picks a 13 ms weight-gradient kernel for one of them on gfx950
        torch.backends.cudnn.benchmark = True
        self.net = randomize_(Detector(model), seed).to(device).train()
        self.net.rpn.to(memory_format=torch.channels_last)
        # gradients: produced by autograd as fresh tensors and gathered into ONE flat buffer with a
        # multi-tensor copy per segment (dist.py).  Two segments -- front end | RPN: the RPN's
        # gradients are complete first (backward runs the RPN first), so with more than one rank
        # their all-reduce starts from a hook at the front end's output and overlaps the front
        # end's backward; the front end's follows behind backward
        self.bucket = FlatGradAllReduce(
            None, process_group, attach=False,
            segments=[list(self.net.frontend.parameters()), list(self.net.rpn.parameters())])
        self.overlap_allreduce = overlap_allreduce
        if overlap_allreduce and self._world() > 1:
            self.net.on_rpn_grads_ready = self._rpn_grads_ready
        # fused: one multi-tensor kernel per step instead of ~8 launches per parameter (78
        # parameters: 0.8 ms of a 19 ms step); gradients share their parameter's layout (dist.py)
        self.opt = torch.optim.AdamW(self.net.parameters(), lr=lr, weight_decay=0.01,
                                     fused=torch.device(device).type == "cuda")
        self.gen = torch.Generator(device=device).manual_seed(seed)
        # Python's cyclic collector: a full (generation-2) pass over the model's, optimizer's and
        # autograd's long-lived objects stalls the host for ~80 ms every 30-100 steps (a 17.6 ms
        # step averages 20 ms over 30 steps).  Once the step has reached its steady state
        # everything alive is moved to the permanent generation (gc.freeze(); the collector stays
        # enabled): 160 steps measured without one pause.  None disables.
        self.freeze_gc_after = freeze_gc_after
        self.calls = 0

    def _world(self):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        return dist.get_world_size(self.bucket.group)

    def _rpn_grads_ready(self):
        self.bucket.collect(1)
        self.bucket.reduce(1, async_op=True)

    def __call__(self, clouds):
        """clouds: this rank's scenes (list of [N,4] cuda tensors) -> loss (device scalar)."""
        from . import ops
        self.calls += 1
        if self.freeze_gc_after is not None and self.calls == self.freeze_gc_after + 1:
            import gc
            gc.collect()
            gc.freeze()
        self.bucket.zero()
        with ops.conv_math(self.conv_math):
            loss = synthetic_loss(self.net(clouds), self.gen)
            loss.backward()
        if self.net.on_rpn_grads_ready is not None:
            self.bucket.collect(0)
            self.bucket.reduce(0)
            self.bucket.wait()
        else:
            self.bucket.collect()
            self.bucket.reduce()
        self.bucket.clip_norm_(10.0)  # clip_grad_norm_(parameters, 10.0), train.py:342
        self.opt.step()
        return loss.detach()
