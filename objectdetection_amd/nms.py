"""Rotated / axis-aligned IoU and non-maximum suppression on the device (include/lcf.h:
lcf_rotate_iou, lcf_nms).

Reference names, same argument meaning:
  * ``nms`` / ``rotate_nms``          second/pytorch/core/box_torch_ops.py:473-515 (torch tensors;
    the reference copies boxes and scores to the host and runs spconv's CPU routines, here
    top-k, the pairwise matrix and the sequential pass all stay on the GPU),
  * ``nms_gpu`` / ``rotate_nms_gpu`` / ``rotate_iou_gpu`` / ``rotate_iou_gpu_eval``
    second/core/non_max_suppression/nms_gpu.py:129-163, 440-475, 511-546, 605-640 (numpy in /
    out, numba-CUDA in the reference),
  * ``nms_cc`` / ``rotate_nms_cc``    second/core/non_max_suppression/nms_cpu.py:11-32.
Boxes are (cx, cy, w, h, angle) for the rotated forms and (x1, y1, x2, y2) for the axis-aligned
ones.  Ties in the scores are broken by position (earlier box first); numpy's
``argsort()[::-1]`` of the reference leaves that order unspecified."""
import numpy as np
import torch

from . import _lib
from ._lib import check


def _dev(device_id=0):
    if not torch.cuda.is_available():
        raise RuntimeError("objectdetection_amd.nms needs a GPU: there is no CPU path")
    return torch.device("cuda", int(device_id))


def _stream():
    return torch.cuda.current_stream().cuda_stream


def rotate_iou(boxes, query_boxes, criterion=-1, corners=False):
    """boxes [N,5] / query_boxes [K,5] cuda float32 (or [N,4,2] / [K,4,2] corners) -> [N,K]."""
    w = 8 if corners else 5
    b = boxes.reshape(-1, w).float().contiguous()
    q = query_boxes.reshape(-1, w).float().contiguous()
    if not (b.is_cuda and q.is_cuda):
        raise RuntimeError("rotate_iou: cuda tensors expected")
    n, k = b.shape[0], q.shape[0]
    out = torch.zeros((n, k), dtype=torch.float32, device=b.device)
    check(_lib.lib().lcf_rotate_iou(b.data_ptr(), n, q.data_ptr(), k, int(corners),
                                    int(criterion), out.data_ptr(), _stream()))
    return out


def nms_sorted(dets, mode, thresh, eps=1.0, inclusive=True, max_keep=None):
    """dets [N,>=4|5|8] cuda float32 sorted by descending score -> kept positions (cuda int64).
    mode 0: axis-aligned (x1,y1,x2,y2), 1: rotated (cx,cy,w,h,angle), 2: rotated, 8 corners."""
    d = dets.float().contiguous()
    n = d.shape[0]
    if n == 0:
        return torch.zeros((0,), dtype=torch.int64, device=d.device)
    L = _lib.lib()
    keep = torch.empty((n,), dtype=torch.int32, device=d.device)
    num = torch.zeros((1,), dtype=torch.int32, device=d.device)
    ws = torch.empty((L.lcf_nms_workspace_bytes(n),), dtype=torch.uint8, device=d.device)
    mk = -1 if max_keep is None or max_keep < 0 else int(max_keep)
    check(L.lcf_nms(d.data_ptr(), n, d.shape[1], int(mode), float(thresh), float(eps),
                    int(bool(inclusive)), mk, keep.data_ptr(), num.data_ptr(), ws.data_ptr(),
                    ws.numel(), _stream()))
    return keep[:int(num.item())].long()


def _order(scores):
    """Descending score, earlier position first among equal scores."""
    return torch.sort(scores, descending=True, stable=True).indices


def _nms_torch(boxes, scores, pre_max_size, post_max_size, iou_threshold, mode, eps):
    if pre_max_size is not None and pre_max_size >= 0:
        k = min(scores.shape[0], int(pre_max_size))
        order = _order(scores)[:k]
    else:
        order = _order(scores)
    if order.numel() == 0:
        return torch.zeros([0], dtype=torch.int64, device=boxes.device)
    post = None if post_max_size is None or post_max_size < 0 else int(post_max_size)
    keep = nms_sorted(boxes[order], mode, iou_threshold, eps=eps, inclusive=True, max_keep=post)
    return order[keep]


def nms(bboxes, scores, pre_max_size=None, post_max_size=None, iou_threshold=0.5, eps=1.0):
    """box_torch_ops.nms (box_torch_ops.py:473-489): axis-aligned boxes [N,4] (x1,y1,x2,y2),
    spconv.ops.nms semantics (overlap >= threshold suppresses, +eps on widths and heights)."""
    return _nms_torch(bboxes, scores, pre_max_size, post_max_size, iou_threshold, 0, eps)


def rotate_nms(rbboxes, scores, pre_max_size=None, post_max_size=None, iou_threshold=0.5):
    """box_torch_ops.rotate_nms (box_torch_ops.py:492-515): rbboxes [N,5] (cx,cy,w,h,angle)."""
    return _nms_torch(rbboxes, scores, pre_max_size, post_max_size, iou_threshold, 1, 0.0)


# ------------------------------------------------------------------ numpy API (nms_gpu.py names)

def _np_nms(dets, thresh, mode, score_col, inclusive, device_id):
    dets = np.asarray(dets, dtype=np.float32)
    if dets.shape[0] == 0:
        return []
    d = torch.from_numpy(dets).to(_dev(device_id))
    order = _order(d[:, score_col])
    keep = nms_sorted(d[order], mode, thresh, eps=1.0, inclusive=inclusive)
    return order[keep].cpu().numpy().astype(np.int32).tolist()


def nms_gpu(dets, nms_overlap_thresh, device_id=0):
    """dets [N,5] (x1,y1,x2,y2,score) -> kept indices, highest score first (nms_gpu.py:129-163:
    overlap > threshold suppresses, +1 pixel convention)."""
    return _np_nms(dets, nms_overlap_thresh, 0, 4, False, device_id)


def rotate_nms_gpu(dets, nms_overlap_thresh, device_id=0):
    """dets [N,6] (cx,cy,w,h,angle,score) -> kept indices (nms_gpu.py:440-475)."""
    return _np_nms(dets, nms_overlap_thresh, 1, 5, False, device_id)


def nms_gpu_cc(dets, nms_overlap_thresh, device_id=0):
    """nms_gpu.py:10-19 (spconv.utils.non_max_suppression on sorted boxes)."""
    return nms_gpu(dets, nms_overlap_thresh, device_id)


def nms_jit(dets, thresh, eps=0.0):
    """nms_cpu.py:34-67: axis-aligned, `eps` added to widths and heights, overlap >= thresh."""
    dets = np.asarray(dets, dtype=np.float32)
    if dets.shape[0] == 0:
        return []
    d = torch.from_numpy(dets).to(_dev())
    order = _order(d[:, 4])
    keep = nms_sorted(d[order], 0, thresh, eps=eps, inclusive=True)
    return order[keep].cpu().numpy().astype(np.int32).tolist()


def nms_cc(dets, thresh):
    """nms_cpu.py:11-14 (spconv non_max_suppression_cpu, eps = 1, >= threshold)."""
    return _np_nms(dets, thresh, 0, 4, True, 0)


def rotate_nms_cc(dets, thresh):
    """nms_cpu.py:17-32 (spconv rotate_non_max_suppression_cpu, >= threshold)."""
    return _np_nms(dets, thresh, 1, 5, True, 0)


def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):
    """boxes [N,5], query_boxes [K,5] numpy -> [N,K] (nms_gpu.py:605-640)."""
    boxes = np.asarray(boxes)
    query_boxes = np.asarray(query_boxes)
    n, k = boxes.shape[0], query_boxes.shape[0]
    if n == 0 or k == 0:
        return np.zeros((n, k), dtype=boxes.dtype)
    dev = _dev(device_id)
    out = rotate_iou(torch.from_numpy(boxes.astype(np.float32)).to(dev),
                     torch.from_numpy(query_boxes.astype(np.float32)).to(dev), criterion)
    return out.cpu().numpy().astype(boxes.dtype)


def rotate_iou_gpu(boxes, query_boxes, device_id=0):
    """nms_gpu.py:511-546."""
    return rotate_iou_gpu_eval(boxes, query_boxes, -1, device_id)


def soft_nms_jit(boxes, sigma=0.5, Nt=0.3, threshold=0.001, method=0):
    """second/core/non_max_suppression/nms_cpu.py:63-153 (Soft-NMS, Bodla et al.): ``boxes``
    [N,5] float32 (x1, y1, x2, y2, score), re-ordered and re-scored IN PLACE as the reference does;
    returns the list of kept row indices ``[0 .. N')``.  method 0: hard NMS at ``Nt``, 1: linear
    decay ``1 - iou``, 2: gaussian ``exp(-iou^2 / sigma)``; a box whose score falls below
    ``threshold`` is swapped with the last live box and dropped.  Inherently sequential (every
    selection re-scores the boxes the next selection reads) and exported but never called by the
    reference (``non_max_suppression/__init__.py:1``): host code over a numpy array, one pass per
    selected box, the pairwise part vectorised where the reference's order allows it (no box is
    dropped -- methods 1 and 2 with threshold 0 -- or falls back to the box-by-box walk)."""
    b = np.asarray(boxes)
    if b.ndim != 2 or b.shape[1] < 5 or b.dtype != np.float32:
        raise ValueError("boxes must be a [N,5] float32 array")
    n = b.shape[0]
    i = 0
    while i < n:
        m = i + int(np.argmax(b[i:n, 4]))  # first maximum, as the reference's strict '<' scan
        if m != i:
            b[[i, m], :5] = b[[m, i], :5]
        tx1, ty1, tx2, ty2 = (np.float64(v) for v in b[i, :4])
        pos = i + 1
        while pos < n:
            x1, y1, x2, y2 = (np.float64(v) for v in b[pos, :4])
            iw = min(tx2, x2) - max(tx1, x1) + 1
            if iw > 0:
                ih = min(ty2, y2) - max(ty1, y1) + 1
                if ih > 0:
                    ua = (tx2 - tx1 + 1) * (ty2 - ty1 + 1) + (x2 - x1 + 1) * (y2 - y1 + 1) - iw * ih
                    ov = iw * ih / ua
                    if method == 1:
                        w = 1 - ov if ov > Nt else 1.0
                    elif method == 2:
                        w = np.exp(-(ov * ov) / sigma)
                    else:
                        w = 0.0 if ov > Nt else 1.0
                    b[pos, 4] = np.float32(w * np.float64(b[pos, 4]))
                    if b[pos, 4] < threshold:  # drop: the last live box takes this row
                        b[pos, :5] = b[n - 1, :5]
                        n -= 1
                        pos -= 1
            pos += 1
        i += 1
    return list(range(n))
