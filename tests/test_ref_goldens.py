"""CPU: the oracle against fixtures produced by RUNNING the reference's own files in the build
container (tests/golden/make_ref_goldens.py): second/utils/simplevis.py:8-59 (voxelizer loop),
torchplus/ops/array_ops.py:7-30 (scatter_nd / gather_nd = SparseConvTensor.dense()), and
second/core/non_max_suppression/nms_gpu.py (rotated IoU, rotated and axis-aligned NMS kernels with
their host drivers).  tests/test_gpu_ref_goldens.py checks the HIP path against the same files."""
import os

import numpy as np
import pytest

import oracle
from oracle import nms_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VOXEL_CASES = ["fhd_uncapped", "fhd_break700", "fhd_edges", "toy_dups", "toy_break"]
IOU_ATOL = 2e-5  # fp32 geometry in the reference's kernels, float64 in the oracle


def voxel_case(tag):
    z = np.load(os.path.join(G, "voxel_golden.npz"))
    return {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(tag + "/")}


def check_voxels_against_reference(c, coors, num_points):
    """coors [K,3] zyx / num_points [K] from a voxelizer run with max_points >= any cell's
    population, against what the reference's loop left in coor_to_voxelidx and bev_map[-1]."""
    ref = c["coors_by_id"]
    assert coors.shape[0] == ref.shape[0]            # voxel_num, incl. the max_voxels break
    assert np.array_equal(coors, ref)                # zyx coordinates in first-appearance order
    # points consumed per (y, x) column until the loop ended
    H, W = int(c["grid_zyx"][1]), int(c["grid_zyx"][2])
    got = np.zeros(H * W, np.int64)
    np.add.at(got, coors[:, 1].astype(np.int64) * W + coors[:, 2], num_points)
    want = np.zeros(H * W, np.int64)
    want[c["bev_yx"][:, 0].astype(np.int64) * W + c["bev_yx"][:, 1]] = c["bev_count"]
    assert np.array_equal(got, want)


@pytest.mark.parametrize("tag", VOXEL_CASES)
def test_oracle_voxelizer_matches_reference_loop(tag):
    c = voxel_case(tag)
    v, coors, num = oracle.voxelize(c["points"], c["voxel_size"], c["range"], 64,
                                    int(c["max_voxels"]), cap_mode=0)
    assert num.max() < 64  # no cell was truncated, so num_points is the consumed count
    check_voxels_against_reference(c, coors, num)
    # the kept points of a voxel are the first ones in cloud order
    first = v[:, 0, :]
    for k in (0, len(first) // 2, len(first) - 1):
        cell = np.floor((first[k, :3] - c["range"][:3]) / c["voxel_size"]).astype(np.int32)
        assert np.array_equal(cell[::-1], coors[k])


def dense_case(tag):
    z = np.load(os.path.join(G, "dense_golden.npz"))
    return {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(tag + "/")}


@pytest.mark.parametrize("tag", ["small", "tail"])
def test_oracle_dense_matches_scatter_nd(tag):
    c = dense_case(tag)
    B, D, H, W, C = [int(x) for x in c["shape"]]
    got = oracle.dense(c["features"], c["indices"], B, [D, H, W])
    # spconv's dense(): scatter_nd(...) is [B,D,H,W,C]; channels_first permutes to [B,C,D,H,W]
    want = np.ascontiguousarray(c["scatter_nd"].transpose(0, 4, 1, 2, 3))
    assert got.shape == want.shape and np.array_equal(got, want)
    assert np.array_equal(c["gather_nd"], c["features"])


def nms_golden():
    return np.load(os.path.join(G, "nms_golden.npz"))


@pytest.mark.parametrize("criterion", [-1, 0, 1, 2])
def test_oracle_rotate_iou_matches_reference_kernel(criterion):
    z = nms_golden()
    ref = z[f"iou/criterion{criterion}"]
    got = O.rotate_iou_matrix(z["iou/boxes"], z["iou/qboxes"], criterion)
    assert (ref > 0).sum() > 1000
    assert np.abs(got - ref).max() < IOU_ATOL * max(1.0, float(ref.max()))
    if criterion == -1:
        assert np.array_equal(z["iou/plain"], ref)


@pytest.mark.parametrize("tag", ["rnms_a", "rnms_b", "rnms_c"])
def test_oracle_rotate_nms_matches_reference(tag):
    z = nms_golden()
    keep, _ = O.nms(z[f"{tag}/dets"], float(z[f"{tag}/thresh"]), rotated=True, inclusive=False)
    assert keep == z[f"{tag}/keep"].tolist()


@pytest.mark.parametrize("tag", ["nms_a", "nms_b"])
def test_oracle_nms_matches_reference(tag):
    z = nms_golden()
    keep, _ = O.nms(z[f"{tag}/dets"], float(z[f"{tag}/thresh"]), rotated=False, inclusive=False)
    assert keep == z[f"{tag}/keep"].tolist()
