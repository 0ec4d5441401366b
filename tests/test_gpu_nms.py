"""GPU: rotated / axis-aligned IoU and NMS (objectdetection_amd/nms.py -> lcf_rotate_iou, lcf_nms)
against the CPU oracle (oracle/nms_oracle.py).  IoU values: 2e-5 absolute (fp32 geometry on the
device, float64 in the oracle).  Kept indices: exact, on inputs whose pairwise overlaps all stay
1e-3 away from the threshold (a decision at the threshold itself is a rounding question)."""
import math

import numpy as np
import pytest
import torch

from objectdetection_amd import nms as N

pytestmark = pytest.mark.gpu
IOU_ATOL = 2e-5


def random_rboxes(rng, n, spread):
    c = rng.uniform(-spread, spread, (n, 2))
    wh = rng.uniform(0.8, 4.5, (n, 2))
    a = rng.uniform(-math.pi, math.pi, (n, 1))
    return np.concatenate([c, wh, a], 1).astype(np.float32)


@pytest.mark.parametrize("criterion", [-1, 0, 1, 2])
def test_rotate_iou_matches_oracle(cuda, criterion):
    from oracle import nms_oracle as O
    rng = np.random.default_rng(1 + criterion)
    boxes, q = random_rboxes(rng, 70, 4.0), random_rboxes(rng, 45, 4.0)
    ref = O.rotate_iou_matrix(boxes, q, criterion)
    got = N.rotate_iou_gpu_eval(boxes, q, criterion)
    assert got.shape == (70, 45) and got.dtype == np.float32
    assert (ref > 0).sum() > 300
    assert np.abs(got - ref).max() < IOU_ATOL * max(1.0, ref.max())
    if criterion == -1:
        assert np.array_equal(N.rotate_iou_gpu(boxes, q), got)
        # corner form (spconv.utils.rbbox_iou)
        from objectdetection_amd.spconv import utils as U
        bc = np.array([O.rbbox_to_corners(b) for b in boxes], np.float32)
        qc = np.array([O.rbbox_to_corners(b) for b in q], np.float32)
        assert np.abs(U.rbbox_iou(bc, qc) - ref).max() < 5e-5
        assert np.abs(U.rbbox_intersection(bc, qc) - O.rotate_iou_matrix(boxes, q, 2)).max() < 2e-4
    assert N.rotate_iou_gpu_eval(boxes[:0], q).shape == (0, 45)


def _margin_ok(seen, thresh):
    return seen.size == 0 or np.abs(seen - thresh).min() > 1e-3


def _safe_thresh(dets, thresh, rotated):
    """The requested threshold, nudged up in steps of 2e-3 until no evaluated overlap is within
    1e-3 of it (random inputs at threshold 0.01 do produce such pairs)."""
    from oracle import nms_oracle as O
    for i in range(50):
        t = thresh + 2e-3 * i
        _, seen = O.nms(dets, t, rotated=rotated, inclusive=False)
        if _margin_ok(seen, t):
            return t
    raise AssertionError("no threshold with a safe margin")


@pytest.mark.parametrize("n,thresh", [(1, 0.3), (63, 0.1), (64, 0.01), (65, 0.5), (300, 0.01),
                                      (700, 0.3)])
def test_rotate_nms_matches_oracle(cuda, n, thresh):
    from oracle import nms_oracle as O
    rng = np.random.default_rng(n)
    boxes = random_rboxes(rng, n, 2.0 + 0.08 * n)
    scores = rng.permutation(n).astype(np.float32) / n
    dets = np.concatenate([boxes, scores[:, None]], 1)
    thresh = _safe_thresh(dets, thresh, True)
    ref_gt, seen = O.nms(dets, thresh, rotated=True, inclusive=False)
    ref_ge, _ = O.nms(dets, thresh, rotated=True, inclusive=True)
    assert _margin_ok(seen, thresh)
    assert N.rotate_nms_gpu(dets, thresh) == ref_gt          # nms_gpu.py: overlap > thresh
    assert N.rotate_nms_cc(dets, thresh) == ref_ge           # spconv: overlap >= thresh
    # torch API of box_torch_ops.rotate_nms: top-k, post_max, indices into the input
    b, s = torch.from_numpy(boxes).to(cuda), torch.from_numpy(scores).to(cuda)
    keep = N.rotate_nms(b, s, pre_max_size=None, post_max_size=None, iou_threshold=thresh)
    assert keep.dtype == torch.int64 and keep.is_cuda and keep.tolist() == ref_ge
    pre = max(1, n // 2)
    top = np.argsort(-scores, kind="stable")[:pre]
    ref_pre, _ = O.nms(dets[top], thresh, rotated=True, inclusive=True, max_keep=5)
    keep = N.rotate_nms(b, s, pre_max_size=pre, post_max_size=5, iou_threshold=thresh)
    assert keep.tolist() == [int(top[i]) for i in ref_pre]
    # corner form through the spconv.utils name
    from objectdetection_amd.spconv import utils as U
    corners = np.array([O.rbbox_to_corners(x) for x in boxes], np.float32)
    order = np.argsort(-scores, kind="stable").astype(np.int32)
    assert U.rotate_non_max_suppression_cpu(corners, order, None, thresh) == ref_ge


@pytest.mark.parametrize("n,thresh", [(2, 0.5), (129, 0.3), (500, 0.7)])
def test_axis_aligned_nms_matches_oracle(cuda, n, thresh):
    from oracle import nms_oracle as O
    from objectdetection_amd.spconv import ops as sops, utils as U
    rng = np.random.default_rng(7 * n)
    xy = rng.uniform(0, 20 + 0.6 * n, (n, 2))
    wh = rng.uniform(4, 40, (n, 2))
    scores = rng.permutation(n).astype(np.float32) / n
    dets = np.concatenate([xy, xy + wh, scores[:, None]], 1).astype(np.float32)
    thresh = _safe_thresh(dets, thresh, False)
    ref_gt, seen = O.nms(dets, thresh, rotated=False, inclusive=False, eps=1.0)
    ref_ge, _ = O.nms(dets, thresh, rotated=False, inclusive=True, eps=1.0)
    assert _margin_ok(seen, thresh)
    assert N.nms_gpu(dets, thresh) == ref_gt
    assert N.nms_cc(dets, thresh) == ref_ge
    order = np.argsort(-scores, kind="stable").astype(np.int32)
    assert U.non_max_suppression_cpu(dets, order, thresh, 1.0) == ref_ge
    keep_out = np.zeros(n, np.int32)
    num = U.non_max_suppression(dets[order], keep_out, thresh)
    assert [int(order[i]) for i in keep_out[:num]] == ref_gt
    # spconv.ops.nms as box_torch_ops.nms calls it: CPU tensors in, indices on the same device
    b, s = torch.from_numpy(dets[:, :4]), torch.from_numpy(scores)
    res = sops.nms(b, s, -1, -1, thresh, 1.0)
    assert not res.is_cuda and res.tolist() == ref_ge
    res = sops.nms(b.to(cuda), s.to(cuda), 50, 7, thresh, 1.0)
    top = np.argsort(-scores, kind="stable")[:50]
    ref_pre, _ = O.nms(dets[top], thresh, rotated=False, inclusive=True, eps=1.0, max_keep=7)
    assert res.is_cuda and res.tolist() == [int(top[i]) for i in ref_pre]


def test_nms_empty_and_ties(cuda):
    assert N.rotate_nms_gpu(np.zeros((0, 6), np.float32), 0.1) == []
    e = N.rotate_nms(torch.zeros((0, 5), device=cuda), torch.zeros((0,), device=cuda), 10, 5, 0.1)
    assert e.shape == (0,) and e.dtype == torch.int64
    # identical boxes, identical scores: the first one survives
    d = np.tile(np.array([[1.0, 2.0, 3.0, 1.5, 0.4, 0.9]], np.float32), (5, 1))
    assert N.rotate_nms_gpu(d, 0.5) == [0]
