"""RPNV2 head of the shipped configs (second/pytorch/models/rpn.py:334-420,467-497;
second/configs/vkitti/second/base.yaml:25-35: layer_nums [5], strides [1], 128 filters,
upsample [1] -> 128) as plain torch modules.

OUT OF the hot-path scope (dense 2-D convs, stock torch / MIOpen).  It is
here only because BASELINE.json configs[2] times a training step "VFE + sparse middle
fwd/dgrad/wgrad + RPN": the benchmark needs a consumer with the reference's shapes downstream
of the sparse front end.  State-dict keys follow the reference (blocks.N.M, deblocks.N.M,
conv_cls / conv_box / conv_dir_cls)."""
import numpy as np
import torch
from torch import nn


class RPNV2(nn.Module):
    def __init__(self, use_norm=True, num_class=1, layer_nums=(5,), layer_strides=(1,),
                 num_filters=(128,), upsample_strides=(1,), num_upsample_filters=(128,),
                 num_input_features=128, num_anchor_per_loc=2, encode_background_as_zeros=True,
                 use_direction_classifier=True, box_code_size=7, num_direction_bins=2,
                 name="rpn"):
        super().__init__()
        assert len(layer_strides) == len(layer_nums) == len(num_filters)
        assert len(num_upsample_filters) == len(upsample_strides)
        self.name = name
        self._num_anchor_per_loc = num_anchor_per_loc
        self._num_class = num_class
        self._box_code_size = box_code_size
        self._num_direction_bins = num_direction_bins
        self._use_direction_classifier = use_direction_classifier
        self._upsample_start_idx = len(layer_nums) - len(upsample_strides)
        bias = not use_norm
        norm = (lambda c: nn.BatchNorm2d(c, eps=1e-3, momentum=0.01)) if use_norm else \
            (lambda c: nn.Identity())
        in_filters = [num_input_features, *num_filters[:-1]]
        blocks, deblocks = [], []
        for i, n in enumerate(layer_nums):
            mods = [nn.ZeroPad2d(1),
                    nn.Conv2d(in_filters[i], num_filters[i], 3, stride=layer_strides[i], bias=bias),
                    norm(num_filters[i]), nn.ReLU()]
            for _ in range(n):
                mods += [nn.Conv2d(num_filters[i], num_filters[i], 3, padding=1, bias=bias),
                         norm(num_filters[i]), nn.ReLU()]
            blocks.append(nn.Sequential(*mods))
            j = i - self._upsample_start_idx
            if j >= 0:
                s = upsample_strides[j]
                if s >= 1:
                    s = int(np.round(s))
                    up = nn.ConvTranspose2d(num_filters[i], num_upsample_filters[j], s, stride=s,
                                            bias=bias)
                else:
                    s = int(np.round(1 / s))
                    up = nn.Conv2d(num_filters[i], num_upsample_filters[j], s, stride=s, bias=bias)
                deblocks.append(nn.Sequential(up, norm(num_upsample_filters[j]), nn.ReLU()))
        self.blocks = nn.ModuleList(blocks)
        self.deblocks = nn.ModuleList(deblocks)
        final = sum(num_upsample_filters) if len(num_upsample_filters) else num_filters[-1]
        num_cls = num_anchor_per_loc * (num_class if encode_background_as_zeros else num_class + 1)
        self.conv_cls = nn.Conv2d(final, num_cls, 1)
        self.conv_box = nn.Conv2d(final, num_anchor_per_loc * box_code_size, 1)
        if use_direction_classifier:
            self.conv_dir_cls = nn.Conv2d(final, num_anchor_per_loc * num_direction_bins, 1)

    # training under bf16 autocast on channels_last activations: BatchNorm2d + ReLU pairs run as
    # liblcf's two-pass row kernels (ops.bn_relu_train_nhwc) instead of MIOpen's BatchNorm plus a
    # separate ReLU; False keeps the stock modules
    fuse_bn_relu = True

    def _run(self, seq, x):
        """nn.Sequential ``seq`` on x, with (BatchNorm2d, ReLU) pairs fused where that applies."""
        from . import ops
        mods = list(seq)
        i = 0
        while i < len(mods):
            m = mods[i]
            if (self.fuse_bn_relu and torch.is_grad_enabled() and i + 1 < len(mods)
                    and isinstance(mods[i + 1], nn.ReLU) and ops.bn_relu_nhwc_supported(m, x)):
                x = ops.bn_relu_train_nhwc(m, x)
                i += 2
                continue
            x = m(x)
            i += 1
        return x

    def forward(self, x):
        ups = []
        for i, block in enumerate(self.blocks):
            x = self._run(block, x)
            if i - self._upsample_start_idx >= 0:
                ups.append(self._run(self.deblocks[i - self._upsample_start_idx], x))
        if ups:
            x = torch.cat(ups, dim=1)
        box = self.conv_box(x)
        cls = self.conv_cls(x)
        H, W = box.shape[2:]
        A = self._num_anchor_per_loc
        ret = {
            "box_preds": box.view(-1, A, self._box_code_size, H, W).permute(0, 1, 3, 4, 2)
            .contiguous(),
            "cls_preds": cls.view(-1, A, cls.shape[1] // A, H, W).permute(0, 1, 3, 4, 2)
            .contiguous(),
        }
        if self._use_direction_classifier:
            d = self.conv_dir_cls(x)
            ret["dir_cls_preds"] = d.view(-1, A, self._num_direction_bins, H, W).permute(
                0, 1, 3, 4, 2).contiguous()
        return ret
