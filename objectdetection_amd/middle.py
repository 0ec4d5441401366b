"""Sparse middle extractors with the reference's class names, constructor signatures, registry
and state_dict keys (second/pytorch/models/middle.py), built on objectdetection_amd.spconv.

The reference file itself imports unmodified once ``dropin.install()`` has run; this module
exists so that the extractors are available where the reference tree is not (the GPU box), and
so that ``SparseMiddleExtractor`` -- which cannot be constructed in the reference because of an
undefined name (middle.py:53) -- is usable.

Every network is ``middle_conv = SparseSequential(conv, BatchNorm1d, ReLU, ...)`` followed by
``dense()`` and a view to [N, C*D, H, W] (middle.py:196-210); layer tables below list
(kind, out_channels, ...) in the reference's order, so parameter names line up
(``middle_conv.0.weight``, ``middle_conv.1.weight``, ...).
"""
import numpy as np
import torch
from torch import nn

from . import ops, spconv

REGISTERED_MIDDLE_CLASSES = {}


def register_middle(cls, name=None):
    global REGISTERED_MIDDLE_CLASSES
    if name is None:
        name = cls.__name__
    assert name not in REGISTERED_MIDDLE_CLASSES, f"exist class: {REGISTERED_MIDDLE_CLASSES}"
    REGISTERED_MIDDLE_CLASSES[name] = cls
    return cls


def get_middle_class(name):
    global REGISTERED_MIDDLE_CLASSES
    assert name in REGISTERED_MIDDLE_CLASSES, f"available class: {REGISTERED_MIDDLE_CLASSES}"
    return REGISTERED_MIDDLE_CLASSES[name]


class _Identity(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, x):
        return x


def S(cout, key):          # SubMConv3d(cin, cout, 3, indice_key=key)
    return ("subm", cout, key)


def C(cout, k=3, s=2, p=0):  # SparseConv3d(cin, cout, k, s, padding=p)
    return ("conv", cout, k, s, p)


_DOWN_Z = C(64, (3, 1, 1), (2, 1, 1))

LAYERS = {
    # middle.py:145-192
    "SpMiddleFHD": [S(16, "subm0"), S(16, "subm0"), C(32, p=1), S(32, "subm1"), S(32, "subm1"),
                    C(64, p=1), S(64, "subm2"), S(64, "subm2"), S(64, "subm2"),
                    C(64, p=[0, 1, 1]), S(64, "subm3"), S(64, "subm3"), S(64, "subm3"), _DOWN_Z],
    # middle.py:247-281
    "SpMiddleFHDPeople": [S(16, "subm0"), S(16, "subm0"), C(32, p=1), S(32, "subm1"),
                          S(32, "subm1"), C(64, p=[0, 1, 1]), S(64, "subm2"), S(64, "subm2"),
                          S(64, "subm2"), _DOWN_Z],
    # middle.py:336-395
    "SpMiddle2K": [S(8, "subm0"), S(8, "subm0"), C(16, p=1), S(16, "subm1"), S(16, "subm1"),
                   C(32, p=1), S(32, "subm2"), S(32, "subm2"), C(64, p=1), S(64, "subm3"),
                   S(64, "subm3"), S(64, "subm3"), C(64, p=[0, 1, 1]), S(64, "subm4"),
                   S(64, "subm4"), S(64, "subm4"), _DOWN_Z],
    # middle.py:452-469
    "SpMiddleFHDLite": [C(16, p=1), C(32, p=1), C(64, p=[0, 1, 1]), _DOWN_Z],
    # middle.py:520-541
    "SpMiddleFHDLiteHRZ": [C(32, p=1), C(64, p=1), C(64, p=1), _DOWN_Z, _DOWN_Z],
    # middle.py:580-631
    "SpMiddleFHDHRZ": [S(16, "subm0"), S(16, "subm0"), C(32, p=1), S(32, "subm1"),
                       S(32, "subm1"), C(64, p=1), S(64, "subm2"), S(64, "subm2"), C(64, p=1),
                       S(64, "subm3"), S(64, "subm3"), _DOWN_Z, S(64, "subm4"), S(64, "subm4"),
                       _DOWN_Z],
}


def build_middle_conv(layers, num_input_features, use_norm=True):
    mods = []
    cin = num_input_features
    for spec in layers:
        if spec[0] == "subm":
            mods.append(spconv.SubMConv3d(cin, spec[1], 3, bias=not use_norm,
                                          indice_key=spec[2]))
        else:
            mods.append(spconv.SparseConv3d(cin, spec[1], spec[2], spec[3], padding=spec[4],
                                            bias=not use_norm))
        cin = spec[1]
        mods.append(nn.BatchNorm1d(cin, eps=1e-3, momentum=0.01) if use_norm else _Identity())
        mods.append(nn.ReLU())
    return spconv.SparseSequential(*mods)


class _MiddleBase(nn.Module):
    layers_name = None
    # None: dense() + view as in the reference (fp32 [N, C*D, H, W]).  "rpn_bf16": the same tensor
    # as channels_last bf16, written in one pass (ops.dense_nhwc_bf16) for an RPN under autocast.
    dense_to = None

    def __init__(self, output_shape, use_norm=True, num_input_features=128,
                 num_filters_down1=[64], num_filters_down2=[64, 64], name=None):
        super().__init__()
        self.name = name or type(self).__name__
        sparse_shape = np.array(output_shape[1:4]) + [1, 0, 0]
        self.sparse_shape = sparse_shape
        self.voxel_output_shape = output_shape
        self.middle_conv = self._make(num_input_features, use_norm, num_filters_down1,
                                      num_filters_down2)

    def _make(self, num_input_features, use_norm, d1, d2):
        return build_middle_conv(LAYERS[self.layers_name], num_input_features, use_norm)

    def forward(self, voxel_features, coors, batch_size):
        coors = coors.int()
        ret = spconv.SparseConvTensor(voxel_features, coors, self.sparse_shape, batch_size)
        # the bf16 hand-over below takes the last layer's rows as bf16 (no fp32 copy of them)
        self.middle_conv.tail_takes_bf16 = self.dense_to == "rpn_bf16" and voxel_features.is_cuda
        ret = self.middle_conv(ret)
        if self.dense_to == "rpn_bf16" and ret.features.is_cuda:
            # straight into the RPN's channels_last bf16 input (no fp32 NCDHW tensor in between)
            return ops.dense_nhwc_bf16(ret.features, ret.indices, batch_size, ret.spatial_shape)
        ret = ret.dense()
        N, C_, D, H, W = ret.shape
        return ret.view(N, C_ * D, H, W)


@register_middle
class SparseMiddleExtractor(_MiddleBase):
    """middle.py:31-108 (classic 0.2/0.2/0.4 m grid): SubM(->d1) on "subm0", z-downsample,
    SubM(->d2...) on "subm1", z-downsample."""

    def _make(self, num_input_features, use_norm, d1, d2):
        layers = [S(o, "subm0") for o in d1]
        last = d1[-1] if len(d1) else num_input_features
        layers.append(C(last, (3, 1, 1), (2, 1, 1)))
        layers += [S(o, "subm1") for o in d2]
        last = d2[-1] if len(d2) else last
        layers.append(C(last, (3, 1, 1), (2, 1, 1)))
        return build_middle_conv(layers, num_input_features, use_norm)


def _variant(name):
    cls = type(name, (_MiddleBase,), {"layers_name": name, "__doc__":
                                      f"second/pytorch/models/middle.py {name}"})
    return register_middle(cls)


SpMiddleFHD = _variant("SpMiddleFHD")
SpMiddleFHDPeople = _variant("SpMiddleFHDPeople")
SpMiddle2K = _variant("SpMiddle2K")
SpMiddleFHDLite = _variant("SpMiddleFHDLite")
SpMiddleFHDLiteHRZ = _variant("SpMiddleFHDLiteHRZ")
SpMiddleFHDHRZ = _variant("SpMiddleFHDHRZ")
