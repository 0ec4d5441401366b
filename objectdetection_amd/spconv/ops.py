"""spconv.ops: the functional layer under the modules (get_indice_pairs, indice_conv) plus the
axis-aligned NMS that second/pytorch/core/box_torch_ops.py:488 calls."""
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
    """Axis-aligned NMS as called by box_torch_ops.nms (box_torch_ops.py:488): boxes [N,4]
    (x1,y1,x2,y2), pre / post sizes <= 0 meaning "all"; the reference passes CPU tensors, the work
    runs on the GPU and the kept indices come back on the input's device."""
    from .. import nms as _nms
    dev = boxes.device
    if not boxes.is_cuda:
        boxes, scores = boxes.to(_nms._dev()), scores.to(_nms._dev())
    keep = _nms.nms(boxes, scores, pre_max_size if pre_max_size > 0 else None,
                    post_max_size if post_max_size > 0 else None, thresh, eps)
    return keep.to(dev)
