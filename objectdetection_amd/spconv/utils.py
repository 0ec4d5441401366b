"""spconv.utils: VoxelGeneratorV2 (second/builder/voxel_builder.py:23-33; used at
second/data/preprocess.py:621-639, second/pytorch/train.py:60,
second/pytorch/builder/second_builder.py:30-31, second/pytorch/models/voxelnet.py:147-148) and
the IoU / NMS helpers (rbbox_iou, non_max_suppression*, on the GPU through objectdetection_amd.nms)."""
import numpy as np
import torch

from .. import ops


# ---- IoU / NMS helpers of spconv.utils, on the GPU (objectdetection_amd/nms.py; SURVEY.md
# section 8f-3).  spconv's C++ source is not part of the reference tree: argument meaning follows
# the call sites (second/core/box_np_ops.py:10-34, second/core/non_max_suppression/nms_cpu.py:11-32,
# nms_gpu.py:10-19); exact tie / threshold behaviour of the originals is unpinned.

def _corners(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1, 8))


def _pair_matrix(box_corners, qbox_corners, standup_iou, standup_thresh, criterion):
    from .. import nms as _nms
    b, q = _corners(box_corners), _corners(qbox_corners)
    if b.shape[0] == 0 or q.shape[0] == 0:
        return np.zeros((b.shape[0], q.shape[0]), dtype=np.float32)
    dev = _nms._dev()
    out = _nms.rotate_iou(torch.from_numpy(b).to(dev), torch.from_numpy(q).to(dev), criterion,
                          corners=True).cpu().numpy()
    if standup_iou is not None:  # the originals skip pairs whose stand-up boxes do not overlap
        out[np.asarray(standup_iou) <= standup_thresh] = 0.0
    return out


def rbbox_iou(box_corners, qbox_corners, standup_iou=None, standup_thresh=0.0):
    """box_corners [N,4,2], qbox_corners [K,4,2] -> rotated IoU [N,K] (box_np_ops.py:10-21)."""
    return _pair_matrix(box_corners, qbox_corners, standup_iou, standup_thresh, -1)


def rbbox_intersection(box_corners, qbox_corners, standup_iou=None, standup_thresh=0.0):
    """-> intersection areas [N,K] (box_np_ops.py:23-34)."""
    return _pair_matrix(box_corners, qbox_corners, standup_iou, standup_thresh, 2)


def _nms_np(rows, order, thresh, eps, mode):
    from .. import nms as _nms
    rows = np.asarray(rows, dtype=np.float32)
    order = np.asarray(order).astype(np.int64)
    if order.shape[0] == 0:
        return []
    d = torch.from_numpy(np.ascontiguousarray(rows[order].reshape(order.shape[0], -1)))
    keep = _nms.nms_sorted(d.to(_nms._dev()), mode, thresh, eps=eps, inclusive=True)
    return order[keep.cpu().numpy()].astype(np.int32).tolist()


def non_max_suppression_cpu(boxes, order, thresh, eps=0.0):
    """boxes [N,>=4] (x1,y1,x2,y2,...), order: indices by descending score -> kept indices."""
    return _nms_np(boxes, order, thresh, eps, 0)


def rotate_non_max_suppression_cpu(box_corners, order, standup_iou, thresh):
    """box_corners [N,4,2] -> kept indices (nms_cpu.py:17-32).  ``standup_iou`` (a pre-filter in
    the original) is not needed: every pair is evaluated."""
    return _nms_np(np.asarray(box_corners).reshape(-1, 8), order, thresh, 0.0, 2)


def non_max_suppression(boxes, keep_out, nms_overlap_thresh, device_id=0):
    """boxes [N,5] sorted by descending score; fills keep_out, returns the number kept
    (nms_gpu.py:10-19 nms_gpu_cc)."""
    from .. import nms as _nms
    boxes = np.asarray(boxes, dtype=np.float32)
    if boxes.shape[0] == 0:
        return 0
    keep = _nms.nms_sorted(torch.from_numpy(boxes).to(_nms._dev(device_id)), 0,
                           nms_overlap_thresh, eps=1.0, inclusive=False).cpu().numpy()
    keep_out[:keep.shape[0]] = keep
    return int(keep.shape[0])


class VoxelGeneratorV2:
    """GPU voxel generator with the reference's constructor and result contract.

    ``generate(points, max_voxels)`` takes a host ``np.ndarray [N,F] float32`` like the
    reference's data pipeline does and returns host arrays sliced to ``voxel_num``;
    ``generate_multi_gpu`` returns arrays padded to ``max_voxels``; ``generate_device`` takes and
    returns GPU tensors (no host round trip) for callers that keep the cloud on the device.

    ``cap_mode``: 0 = stop at the first point that would open voxel number ``max_voxels`` (the
    numba ``points_to_voxel`` twin's ``break``, second/utils/simplevis.py:47-48; default),
    1 = spconv's C++ behaviour (skip that point, later points still join existing voxels).
    """

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000,
                 full_mean=False, block_filtering=False, block_factor=8, block_size=3,
                 height_threshold=0.1, height_high_threshold=2.0, cap_mode=0, device=None):
        if block_filtering:
            raise NotImplementedError(
                "block_filtering is not used by any shipped config (voxel_generator.proto:10 "
                "defaults to false) and is not implemented")
        point_cloud_range = np.array(point_cloud_range, dtype=np.float32)
        voxel_size = np.array(voxel_size, dtype=np.float32)
        grid_size = (point_cloud_range[3:] - point_cloud_range[:3]) / voxel_size
        grid_size = np.round(grid_size).astype(np.int64)
        self._voxel_size = voxel_size
        self._point_cloud_range = point_cloud_range
        self._max_num_points = int(max_num_points)
        self._max_voxels = int(max_voxels)
        self._grid_size = grid_size
        self._full_mean = bool(full_mean)
        self._cap_mode = int(cap_mode)
        self._device = device

    # ---- device path
    def generate_device(self, points, max_voxels=None, batch_idx=None):
        """points: [N,F] float32 cuda tensor -> dict of cuda tensors padded to max_voxels, with
        ``voxel_num`` a [1] int32 cuda tensor (no synchronisation).  ``batch_idx`` not None:
        coordinates come out as [max_voxels,4] (batch_idx, z, y, x), merge_second_batch's form."""
        mv = int(max_voxels or self._max_voxels)
        voxels, coors, num, vnum = ops.voxelize(points, self._voxel_size, self._point_cloud_range,
                                                self._max_num_points, mv, self._cap_mode,
                                                batch_idx=batch_idx)
        if self._full_mean:
            # fill the empty slots of a voxel with the mean of its points
            n = num.clamp(min=1).view(-1, 1, 1).float()
            mean = voxels.sum(dim=1, keepdim=True) / n
            t = torch.arange(self._max_num_points, device=voxels.device).view(1, -1, 1)
            empty = (t >= num.view(-1, 1, 1)) & (num.view(-1, 1, 1) > 0)
            voxels = torch.where(empty, mean.expand_as(voxels), voxels)
        return {"voxels": voxels, "coordinates": coors, "num_points_per_voxel": num,
                "voxel_num": vnum}

    def accumulator(self, capacity, num_features=4, max_voxels=None, batch_idx=None):
        """A device-resident cumulative cloud voxelised incrementally with this generator's
        geometry (ops.VoxelAccumulator): for the light-curtain loop, which grows
        the cloud by one curtain's returns per step (second/light_curtain/utils.py:121-124)."""
        if self._full_mean:
            raise NotImplementedError("full_mean rewrites the padding of every voxel; use "
                                      "generate_device on the accumulator's cloud instead")
        return ops.VoxelAccumulator(capacity, num_features, self._voxel_size,
                                    self._point_cloud_range, self._max_num_points,
                                    int(max_voxels or self._max_voxels), self._cap_mode,
                                    batch_idx, self._dev())

    def _dev(self):
        if self._device is not None:
            return torch.device(self._device)
        if not torch.cuda.is_available():
            raise RuntimeError("VoxelGeneratorV2 (objectdetection_amd) needs a GPU: no CPU path")
        return torch.device("cuda", torch.cuda.current_device())

    def _run(self, points, max_voxels):
        pts = np.ascontiguousarray(points, dtype=np.float32)
        if self._device is not None and torch.device(self._device).type == "cpu":
            # asked for by name (device="cpu"): the GPU-free twin, for forked DataLoader workers
            # (objectdetection_amd/cpu.py; same results as the HIP voxelizer, bit for bit)
            if self._full_mean:
                raise NotImplementedError("full_mean is not part of the CPU twin")
            from .. import cpu
            v, c, n, k = cpu.voxelize(pts, self._voxel_size, self._point_cloud_range,
                                      self._max_num_points, int(max_voxels or self._max_voxels),
                                      self._cap_mode)
            return {"voxels": torch.from_numpy(v), "coordinates": torch.from_numpy(c),
                    "num_points_per_voxel": torch.from_numpy(n)}, k
        res = self.generate_device(torch.from_numpy(pts).to(self._dev()), max_voxels)
        k = int(res["voxel_num"].item())
        return res, k

    # ---- reference API
    def generate(self, points, max_voxels=None):
        res, k = self._run(points, max_voxels)
        voxels = res["voxels"][:k].cpu().numpy()
        coors = res["coordinates"][:k].cpu().numpy()
        num = res["num_points_per_voxel"][:k].cpu().numpy()
        mask = (np.arange(self._max_num_points)[None, :] < num[:, None])
        return {"voxels": voxels, "coordinates": coors, "num_points_per_voxel": num,
                "voxel_point_mask": mask[..., None].astype(voxels.dtype), "voxel_num": k}

    def generate_multi_gpu(self, points, max_voxels=None):
        res, k = self._run(points, max_voxels)
        voxels = res["voxels"].cpu().numpy()
        coors = res["coordinates"].cpu().numpy()
        num = res["num_points_per_voxel"].cpu().numpy()
        mask = (np.arange(self._max_num_points)[None, :] < num[:, None])
        return {"voxels": voxels, "coordinates": coors, "num_points_per_voxel": num,
                "voxel_point_mask": mask[..., None].astype(voxels.dtype), "voxel_num": k}

    @property
    def voxel_size(self):
        return self._voxel_size

    @property
    def max_num_points_per_voxel(self):
        return self._max_num_points

    @property
    def point_cloud_range(self):
        return self._point_cloud_range

    @property
    def grid_size(self):
        return self._grid_size


class VoxelGenerator(VoxelGeneratorV2):
    """spconv.utils.VoxelGenerator (v1 signature); returns the tuple form."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000):
        super().__init__(voxel_size, point_cloud_range, max_num_points, max_voxels)

    def generate(self, points, max_voxels=None):
        res = super().generate(points, max_voxels)
        return res["voxels"], res["coordinates"], res["num_points_per_voxel"]
