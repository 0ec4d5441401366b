"""CPU restatement of the reference's rotated IoU / NMS (second/core/non_max_suppression/
nms_gpu.py) -- TEST INFRASTRUCTURE ONLY.

The reference kernels are numba-CUDA and numba is not installed here, and spconv's C++ NMS is not
part of the reference tree, so nothing of the original can be executed: PARITY UNPINNED against the
reference's binaries.  What pins this file instead: (1) it follows the reference's algorithm line
by line (citations below), in float64 on float32 inputs; (2) tests/test_nms_oracle.py checks it
against an independent algorithm (Sutherland-Hodgman polygon clipping) and against hand-computed
known answers.  Pure-Python loops: use on a few thousand pairs at most.
"""
import math

import numpy as np


def rbbox_to_corners(rbbox):
    """nms_gpu.py:353-376: clockwise corners of (cx, cy, w, h, angle)."""
    cx, cy, xd, yd, angle = [float(v) for v in rbbox[:5]]
    a_cos, a_sin = math.cos(angle), math.sin(angle)
    xs = [-xd / 2, -xd / 2, xd / 2, xd / 2]
    ys = [-yd / 2, yd / 2, yd / 2, -yd / 2]
    out = []
    for i in range(4):
        out.append((a_cos * xs[i] + a_sin * ys[i] + cx, -a_sin * xs[i] + a_cos * ys[i] + cy))
    return out


def _point_in_quadrilateral(px, py, c):
    """nms_gpu.py:308-325."""
    ab0, ab1 = c[1][0] - c[0][0], c[1][1] - c[0][1]
    ad0, ad1 = c[3][0] - c[0][0], c[3][1] - c[0][1]
    ap0, ap1 = px - c[0][0], py - c[0][1]
    abab = ab0 * ab0 + ab1 * ab1
    abap = ab0 * ap0 + ab1 * ap1
    adad = ad0 * ad0 + ad1 * ad1
    adap = ad0 * ap0 + ad1 * ap1
    eps = -1e-6
    return abab - abap >= eps and abap >= eps and adad - adap >= eps and adap >= eps


def _segment_intersection(p, q, i, j):
    """nms_gpu.py:225-264."""
    A, B = p[i], p[(i + 1) % 4]
    C, D = q[j], q[(j + 1) % 4]
    BA0, BA1 = B[0] - A[0], B[1] - A[1]
    DA0, CA0 = D[0] - A[0], C[0] - A[0]
    DA1, CA1 = D[1] - A[1], C[1] - A[1]
    acd = DA1 * CA0 > CA1 * DA0
    bcd = (D[1] - B[1]) * (C[0] - B[0]) > (C[1] - B[1]) * (D[0] - B[0])
    if acd != bcd:
        abc = CA1 * BA0 > BA1 * CA0
        abd = DA1 * BA0 > BA1 * DA0
        if abc != abd:
            DC0, DC1 = D[0] - C[0], D[1] - C[1]
            ABBA = A[0] * B[1] - B[0] * A[1]
            CDDC = C[0] * D[1] - D[0] * C[1]
            DH = BA1 * DC0 - BA0 * DC1
            return ((ABBA * DC0 - BA0 * CDDC) / DH, (ABBA * DC1 - BA1 * CDDC) / DH)
    return None


def inter_corners(p, q):
    """Intersection area of two rectangles given as corner lists (nms_gpu.py:329-393)."""
    pts = []
    for i in range(4):
        if _point_in_quadrilateral(p[i][0], p[i][1], q):
            pts.append(p[i])
        if _point_in_quadrilateral(q[i][0], q[i][1], p):
            pts.append(q[i])
    for i in range(4):
        for j in range(4):
            t = _segment_intersection(p, q, i, j)
            if t is not None:
                pts.append(t)
    n = len(pts)
    if n < 3:
        return 0.0
    # nms_gpu.py:183-219: order by the unit vector from the centroid
    mx = sum(t[0] for t in pts) / n
    my = sum(t[1] for t in pts) / n
    keyed = []
    for t in pts:
        vx, vy = t[0] - mx, t[1] - my
        d = math.sqrt(vx * vx + vy * vy)
        vx, vy = vx / d, vy / d
        keyed.append((-2 - vx if vy < 0 else vx, t))
    keyed.sort(key=lambda kv: kv[0])  # stable, like the reference's insertion sort
    pts = [kv[1] for kv in keyed]
    area = 0.0  # nms_gpu.py:166-179
    for i in range(n - 2):
        a, b, c = pts[0], pts[i + 1], pts[i + 2]
        area += abs(((a[0] - c[0]) * (b[1] - c[1]) - (a[1] - c[1]) * (b[0] - c[0])) / 2.0)
    return area


def inter(rbox1, rbox2):
    return inter_corners(rbbox_to_corners(rbox1), rbbox_to_corners(rbox2))


def rotate_iou_eval(rbox1, rbox2, criterion=-1):
    """nms_gpu.py:549-561 (devRotateIoUEval)."""
    a1 = float(rbox1[2]) * float(rbox1[3])
    a2 = float(rbox2[2]) * float(rbox2[3])
    ai = inter(rbox1, rbox2)
    if criterion == -1:
        return ai / (a1 + a2 - ai)
    if criterion == 0:
        return ai / a1
    if criterion == 1:
        return ai / a2
    return ai


def rotate_iou_matrix(boxes, query_boxes, criterion=-1):
    """nms_gpu.py:564-640: iou[n, k] = devRotateIoUEval(query_boxes[k], boxes[n])."""
    out = np.zeros((len(boxes), len(query_boxes)), dtype=np.float64)
    for n in range(len(boxes)):
        for k in range(len(query_boxes)):
            out[n, k] = rotate_iou_eval(query_boxes[k], boxes[n], criterion)
    return out


def iou_axis_aligned(a, b, eps=1.0):
    """nms_gpu.py:21-32 (eps = 1) / nms_cpu.py:34-67 nms_jit (eps parameter)."""
    w = max(min(a[2], b[2]) - max(a[0], b[0]) + eps, 0.0)
    h = max(min(a[3], b[3]) - max(a[1], b[1]) + eps, 0.0)
    s = w * h
    sa = (a[2] - a[0] + eps) * (a[3] - a[1] + eps)
    sb = (b[2] - b[0] + eps) * (b[3] - b[1] + eps)
    return s / (sa + sb - s)


def nms(dets, thresh, rotated, inclusive=False, eps=1.0, max_keep=None):
    """Sort by descending score (earlier box first among ties), then the greedy pass of
    nms_postprocess (nms_gpu.py:109-126) over `overlap > thresh` (inclusive: >=).
    dets [N,5] (x1,y1,x2,y2,score) or, rotated, [N,6] (cx,cy,w,h,angle,score).
    Returns (kept indices, pairwise overlaps that were evaluated: for margin checks)."""
    dets = np.asarray(dets, dtype=np.float32).astype(np.float64)
    sc = dets[:, 5 if rotated else 4]
    order = np.argsort(-sc, kind="stable")
    removed = np.zeros(len(dets), dtype=bool)
    keep, seen = [], []
    for a in range(len(order)):
        if removed[a]:
            continue
        keep.append(int(order[a]))
        if max_keep is not None and len(keep) >= max_keep:
            break
        for b in range(a + 1, len(order)):
            if removed[b]:
                continue
            if rotated:
                ov = rotate_iou_eval(dets[order[a]], dets[order[b]])
            else:
                ov = iou_axis_aligned(dets[order[a]], dets[order[b]], eps)
            seen.append(ov)
            if (ov >= thresh) if inclusive else (ov > thresh):
                removed[b] = True
    return keep, np.asarray(seen)


# ---------------------------------------------------------------- independent cross-check

def _clip(subject, a, b):
    """Sutherland-Hodgman: keep the part of polygon `subject` on the inner side of edge a->b of
    a clockwise clip polygon."""
    def side(p):
        return (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
    out = []
    for i in range(len(subject)):
        p, q = subject[i], subject[(i + 1) % len(subject)]
        sp, sq = side(p), side(q)
        if sp <= 0:
            out.append(p)
        if (sp < 0 < sq) or (sq < 0 < sp):
            t = sp / (sp - sq)
            out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    return out


def inter_sutherland_hodgman(rbox1, rbox2):
    """Intersection area by polygon clipping + the shoelace formula (not the reference's
    algorithm: used to pin the restatement above)."""
    poly = rbbox_to_corners(rbox1)
    clip = rbbox_to_corners(rbox2)
    # orientation of the clip polygon decides which side is inside
    area2 = sum(clip[i][0] * clip[(i + 1) % 4][1] - clip[(i + 1) % 4][0] * clip[i][1]
                for i in range(4))
    if area2 > 0:
        clip = clip[::-1]
    for i in range(4):
        if not poly:
            return 0.0
        poly = _clip(poly, clip[i], clip[(i + 1) % 4])
    if len(poly) < 3:
        return 0.0
    s = 0.0
    for i in range(len(poly)):
        s += poly[i][0] * poly[(i + 1) % len(poly)][1] - poly[(i + 1) % len(poly)][0] * poly[i][1]
    return abs(s) / 2.0
