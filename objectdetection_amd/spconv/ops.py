"""spconv.ops: the functional layer under the modules (get_indice_pairs, indice_conv) plus the
post-processing names that must exist at import time (second/pytorch/core/box_torch_ops.py:488
calls spconv.ops.nms -- out of the front-end scope)."""
import torch

from .. import ops as _ops


def get_conv_output_size(input_size, kernel_size, stride, padding, dilation):
    ndim = len(input_size)
    out = []
    for i in range(ndim):
        size = (input_size[i] + 2 * padding[i] - dilation[i] * (kernel_size[i] - 1) - 1) \
            // stride[i] + 1
        out.append(1 if kernel_size[i] == -1 else size)
    return out


def get_indice_pairs(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0,
                     dilation=1, out_padding=0, subm=False, transpose=False, grid=None,
                     use_hash=True):
    """-> (out_indices, indice_pairs [2,kvol,K_in] int32, indice_pair_num [kvol] int32), in
    spconv's canonical CPU order (bit-identical to the oracle's restatement)."""
    if transpose:
        raise NotImplementedError("transposed sparse convolution is not on the front-end path")
    rb = _ops.build_rulebook(indices.int(), batch_size, list(spatial_shape), ksize, stride,
                             padding, dilation, subm)
    return rb.out_indices, rb.indice_pairs, rb.indice_pair_num


def get_rulebook(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1,
                 subm=False):
    """Native form: the Rulebook object (neighbour tables + lazy canonical pairs)."""
    return _ops.build_rulebook(indices.int(), batch_size, list(spatial_shape), ksize, stride,
                               padding, dilation, subm)


def indice_conv(features, filters, rulebook):
    """features [K_in,Cin], filters [kD,kH,kW,Cin,Cout], rulebook from get_rulebook."""
    return _ops.indice_conv(features, filters, rulebook)


def nms(boxes, scores, pre_max_size, post_max_size, thresh, eps):
    raise NotImplementedError(
        "spconv.ops.nms is post-processing, outside the sparse front end this package provides; "
        "use second.pytorch.core.box_torch_ops.nms")
