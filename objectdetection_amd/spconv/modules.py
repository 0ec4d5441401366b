"""spconv.SparseModule / SparseSequential / ToDense (second/pytorch/models/middle.py:93,145;
second/pytorch/models/resnet.py:32,69)."""
from collections import OrderedDict

import torch
from torch import nn

from .. import ops
from .tensor import SparseConvTensor


class SparseModule(nn.Module):
    """Marker base class: SparseSequential hands the whole SparseConvTensor to subclasses."""
    pass


def is_spconv_module(module):
    return isinstance(module, SparseModule)


def is_sparse_conv(module):
    from .conv import SparseConvolution
    return isinstance(module, SparseConvolution)


def _bn_affine(bn):
    """Eval-mode BatchNorm1d as y = x*scale + shift (fp32)."""
    inv = torch.rsqrt(bn.running_var.float() + bn.eps)
    w = bn.weight.float() if bn.weight is not None else torch.ones_like(inv)
    b = bn.bias.float() if bn.bias is not None else torch.zeros_like(inv)
    scale = (w * inv).contiguous()
    shift = (b - bn.running_mean.float() * scale).contiguous()
    return scale, shift


class SparseSequential(SparseModule):
    """Sequential container that runs sparse modules on the SparseConvTensor and ordinary
    nn.Modules (BatchNorm1d, ReLU, ...) on its ``.features``.

    MI355X addition: when gradients are off and the modules are in eval
    mode, a ``sparse conv -> BatchNorm1d -> ReLU`` run is executed as ONE kernel launch with the
    normalisation folded into the conv epilogue (exact in inference: BN is affine there).
    Set ``fuse_inference = False`` to run module by module.  In training a
    ``BatchNorm1d (training) -> ReLU`` pair on the features runs as the fused streaming kernels of
    ``ops.bn_relu_train`` (same values and running statistics as the two torch modules;
    ``fuse_training = False`` to turn it off).
    """
    fuse_inference = True
    fuse_training = True
    # under ops.conv_math("bf16") in training: sparse conv -> BatchNorm1d -> ReLU triples whose
    # channels the bf16-row kernels support run as ONE autograd node on bf16 rows
    # (ops.ConvBNReLUTrainFunction); False keeps fp32 rows and the conversion at every gather
    bf16_rows = True
    # the consumer of the sequence's output takes bf16 rows (the middle extractor's
    # dense_nhwc_bf16 hand-over to an RPN under autocast)
    tail_takes_bf16 = False

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError(f"index {idx} is out of range")
        if idx < 0:
            idx += len(self)
        it = iter(self._modules.values())
        for _ in range(idx):
            next(it)
        return next(it)

    def __len__(self):
        return len(self._modules)

    def __iter__(self):
        return iter(self._modules.values())

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def _fusable(self, mods, i):
        """(n_consumed, bn, relu) for a conv at mods[i] followed by eval BN and/or ReLU."""
        bn, relu = None, False
        j = i + 1
        if j < len(mods) and isinstance(mods[j], nn.BatchNorm1d) and not mods[j].training \
                and mods[j].track_running_stats and mods[j].running_mean is not None:
            bn = mods[j]
            j += 1
        if j < len(mods) and isinstance(mods[j], nn.ReLU):
            relu = True
            j += 1
        return j - i, bn, relu

    def _next_takes_bf16(self, mods, j):
        """True if the rows produced in front of mods[j] should be stored as bf16: the next module
        is a sparse conv the bf16-row kernels support and followed by a training BatchNorm1d (it
        will run as the fused node), or the sequence ends and its consumer takes bf16 rows."""
        if j >= len(mods):
            return self.tail_takes_bf16
        m = mods[j]
        return (is_sparse_conv(m) and m.ndim == 3 and m.bias is None and not m.conv1x1
                and ops.b16_conv_supported(m.in_channels, m.out_channels)
                and j + 1 < len(mods) and isinstance(mods[j + 1], nn.BatchNorm1d)
                and mods[j + 1].training and mods[j + 1].affine)

    def forward(self, input):
        mods = list(self._modules.values())
        fuse = (self.fuse_inference and not torch.is_grad_enabled()
                and isinstance(input, SparseConvTensor) and input.features.is_cuda)
        b16 = (self.bf16_rows and self.fuse_training and torch.is_grad_enabled()
               and ops.get_conv_math() == "bf16" and isinstance(input, SparseConvTensor)
               and input.features.is_cuda)
        i = 0
        while i < len(mods):
            module = mods[i]
            if (b16 and isinstance(input, SparseConvTensor)
                    and input.features.dtype == torch.bfloat16 and input.indices.shape[0] != 0
                    and self._next_takes_bf16(mods, i)):
                # fused bf16-row node: conv + BatchNorm1d (+ ReLU)
                relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
                nxt = i + (3 if relu else 2)
                y, rb = ops.conv_bn_relu_train(module, mods[i + 1], input, relu,
                                               self._next_takes_bf16(mods, nxt))
                out = SparseConvTensor(y, rb.out_indices, rb.out_shape, input.batch_size)
                out.indice_dict = input.indice_dict
                out.grid = input.grid
                input = out
                i = nxt
                continue
            if fuse and is_sparse_conv(module) and isinstance(input, SparseConvTensor):
                n, bn, relu = self._fusable(mods, i)
                if n > 1:
                    scale = shift = None
                    if bn is not None:
                        scale, shift = _bn_affine(bn)
                    input = module(input, _scale=scale, _shift=shift, _relu=relu)
                    i += n
                    continue
            if is_spconv_module(module):
                input = module(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:
                    if (self.fuse_training and isinstance(module, nn.BatchNorm1d)
                            and module.training and torch.is_grad_enabled()
                            and ops.bn_relu_supported(input.features)):
                        relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                        nxt = i + (2 if relu else 1)
                        input.features = ops.bn_relu_train(
                            module, input.features, relu,
                            y_bf16=b16 and self._next_takes_bf16(mods, nxt)
                            and input.features.shape[1] % 8 == 0)
                        i = nxt
                        continue
                    input.features = module(input.features)
            else:
                input = module(input)
            i += 1
        return input


class ToDense(SparseModule):
    def forward(self, x: SparseConvTensor):
        return x.dense()
