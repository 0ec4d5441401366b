"""second.core.point_cloud.point_cloud_ops.points_to_voxel -- the module the reference's viewers
import (second/kittiviewer/viewer.py:37,389-396) but does not ship.  Same
signature and return contract as second.pytorch's numba implementation; runs the HIP voxelizer.
"""
import numpy as np
import torch

from . import ops


def points_to_voxel(points, voxel_size, coors_range, max_points=35, reverse_index=True,
                    max_voxels=20000):
    """convert kitti points(N, >=3) to voxels.

    Args:
        points: [N, ndim] float array. points[:, :3] contain xyz.
        voxel_size: [3] list/tuple or array, float. xyz, indicate voxel size
        coors_range: [6] list/tuple or array, float. xyzxyz, minmax
        max_points: int. maximum number of points kept per voxel.
        reverse_index: bool. True returns zyx coordinates (what every caller passes);
            False returns xyz.
        max_voxels: int. the loop stops at the first point that would create one more.

    Returns:
        voxels: [M, max_points, ndim] float array (zero padded).
        coordinates: [M, 3] int32 array.
        num_points_per_voxel: [M] int32 array.
    """
    if not torch.cuda.is_available():
        raise RuntimeError("points_to_voxel (objectdetection_amd) needs a GPU: no CPU path")
    pts = np.ascontiguousarray(points, dtype=np.float32)
    vs = np.asarray(voxel_size, dtype=np.float32)
    rg = np.asarray(coors_range, dtype=np.float32)
    voxels, coors, num, vnum = ops.voxelize(torch.from_numpy(pts).cuda(), vs, rg, int(max_points),
                                            int(max_voxels), 0)
    k = int(vnum.item())
    coors = coors[:k].cpu().numpy()
    if not reverse_index:
        coors = np.ascontiguousarray(coors[:, ::-1])
    return voxels[:k].cpu().numpy().astype(points.dtype, copy=False), coors, num[:k].cpu().numpy()
