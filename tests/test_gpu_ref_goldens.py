"""GPU: the HIP path (through the C ABI) against fixtures produced by RUNNING the reference's own
files (tests/golden/make_ref_goldens.py; see tests/test_ref_goldens.py for what each holds):
lcf_voxelize vs second/utils/simplevis.py:8-59, lcf_dense / lcf_dense_bwd vs
torchplus/ops/array_ops.py:7-30, lcf_rotate_iou / lcf_nms vs
second/core/non_max_suppression/nms_gpu.py."""
import numpy as np
import pytest
import torch

from objectdetection_amd import nms as N
from objectdetection_amd import ops
from test_ref_goldens import (IOU_ATOL, VOXEL_CASES, check_voxels_against_reference, dense_case,
                              nms_golden, voxel_case)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", VOXEL_CASES)
def test_hip_voxelizer_matches_reference_loop(cuda, tag):
    c = voxel_case(tag)
    mv = int(c["max_voxels"])
    voxels, coors, num, vnum = ops.voxelize(torch.from_numpy(c["points"]).to(cuda),
                                            c["voxel_size"].tolist(), c["range"].tolist(), 64,
                                            mv, 0)
    k = int(vnum.item())
    coors, num = coors.cpu().numpy()[:k], num.cpu().numpy()[:k]
    assert num.max() < 64
    check_voxels_against_reference(c, coors, num)


@pytest.mark.parametrize("tag", ["small", "tail"])
def test_hip_dense_matches_scatter_nd(cuda, tag):
    c = dense_case(tag)
    B, D, H, W, C = [int(x) for x in c["shape"]]
    feat = torch.from_numpy(c["features"]).to(cuda).requires_grad_(True)
    idx = torch.from_numpy(c["indices"]).to(cuda)
    got = ops.dense(feat, idx, B, [D, H, W])
    want = np.ascontiguousarray(c["scatter_nd"].transpose(0, 4, 1, 2, 3))
    assert np.array_equal(got.detach().cpu().numpy(), want)
    # backward = gather_nd of the upstream gradient at the same indices
    g = torch.randn(got.shape, device=cuda, generator=torch.Generator(cuda).manual_seed(0))
    got.backward(g)
    gi = idx.long()
    want_g = g.permute(0, 2, 3, 4, 1)[gi[:, 0], gi[:, 1], gi[:, 2], gi[:, 3]]
    assert torch.equal(feat.grad, want_g)


@pytest.mark.parametrize("criterion", [-1, 0, 1, 2])
def test_hip_rotate_iou_matches_reference_kernel(cuda, criterion):
    z = nms_golden()
    ref = z[f"iou/criterion{criterion}"]
    got = N.rotate_iou_gpu_eval(z["iou/boxes"], z["iou/qboxes"], criterion)
    assert np.abs(got - ref).max() < IOU_ATOL * max(1.0, float(ref.max()))


@pytest.mark.parametrize("tag", ["rnms_a", "rnms_b", "rnms_c"])
def test_hip_rotate_nms_matches_reference(cuda, tag):
    z = nms_golden()
    assert N.rotate_nms_gpu(z[f"{tag}/dets"], float(z[f"{tag}/thresh"])) == \
        z[f"{tag}/keep"].tolist()


@pytest.mark.parametrize("tag", ["nms_a", "nms_b"])
def test_hip_nms_matches_reference(cuda, tag):
    z = nms_golden()
    assert N.nms_gpu(z[f"{tag}/dets"], float(z[f"{tag}/thresh"])) == z[f"{tag}/keep"].tolist()
