"""The GPU-free twin of the voxelizer (include/lcf_cpu.h, csrc/cpu/lcf_cpu.c -> liblcf_cpu.so) for
processes that must not touch the GPU: the reference voxelizes in forked DataLoader workers
(second/configs/vkitti/second/base.yaml:120 `num_workers: 3`, second/data/preprocess.py:626-639).
numpy + ctypes only -- no torch, no HIP; nothing here is a fallback: the GPU path never calls it,
a caller asks for it by name (``VoxelGeneratorV2(..., device="cpu")`` or the functions below).
Same results as the HIP voxelizer, bit for bit (tests/test_cpu_twin.py)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblcf_cpu.so")
_lib = None

_ERR = {-1: "bad argument", -2: "out of host memory for the cell table",
        -4: "grid too large for 32-bit cell keys"}


def lib():
    """The loaded liblcf_cpu.so; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with "
                               "`make -C objectdetection_amd/csrc`")
        L = ctypes.CDLL(LIB_PATH)
        fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
        L.lcf_cpu_voxelize.restype = ctypes.c_int
        L.lcf_cpu_voxelize.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp, fp, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       fp, ip, ip, ip]
        _lib = L
    return _lib


def voxelize(points, voxel_size, coors_range, max_points, max_voxels, cap_mode=0, batch_idx=None):
    """points [N,F] -> (voxels [max_voxels,max_points,F] f32, coors [max_voxels,3|4] i32 (z,y,x) or
    (batch_idx,z,y,x), num_points [max_voxels] i32, voxel_num int): lcf_cpu_voxelize, the contract
    of ops.voxelize on host arrays."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    if pts.ndim != 2 or pts.shape[1] < 3:
        raise ValueError("points must be [N, >=3]")
    n, f = pts.shape
    vs = np.ascontiguousarray(voxel_size, dtype=np.float32)
    rg = np.ascontiguousarray(coors_range, dtype=np.float32)
    if vs.shape != (3,) or rg.shape != (6,):
        raise ValueError("voxel_size must have 3 and coors_range 6 entries")
    stride = 3 if batch_idx is None else 4
    voxels = np.empty((max_voxels, max_points, f), dtype=np.float32)
    coors = np.empty((max_voxels, stride), dtype=np.int32)
    num = np.empty((max_voxels,), dtype=np.int32)
    vnum = np.zeros((1,), dtype=np.int32)
    fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    rc = lib().lcf_cpu_voxelize(
        pts.ctypes.data_as(fp), n, f, vs.ctypes.data_as(fp), rg.ctypes.data_as(fp),
        int(max_points), int(max_voxels), int(cap_mode), int(batch_idx or 0), stride,
        voxels.ctypes.data_as(fp), coors.ctypes.data_as(ip), num.ctypes.data_as(ip),
        vnum.ctypes.data_as(ip))
    if rc != 0:
        raise (ValueError if rc == -1 else RuntimeError)(
            f"lcf_cpu_voxelize: {_ERR.get(rc, rc)}")
    return voxels, coors, num, int(vnum[0])


def points_to_voxel(points, voxel_size, coors_range, max_points=35, reverse_index=True,
                    max_voxels=20000):
    """second.core.point_cloud.point_cloud_ops.points_to_voxel on the host (signature and returns
    of objectdetection_amd.point_cloud_ops.points_to_voxel, which runs on the GPU)."""
    voxels, coors, num, k = voxelize(points, voxel_size, coors_range, max_points, max_voxels, 0)
    coors = coors[:k]
    if not reverse_index:
        coors = np.ascontiguousarray(coors[:, ::-1])
    return voxels[:k].astype(np.asarray(points).dtype, copy=False), coors, num[:k]
