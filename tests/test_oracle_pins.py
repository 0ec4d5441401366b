"""CPU: pins for the parts of the oracle the reference cannot pin (no tests, no fixtures, spconv /
numba absent -- SURVEY.md F6, section 8c): hand-derived known answers, an independent pure-Python
restatement of the voxelizer loop, the reference's documented shape chain, and equivalence with
dense torch.nn.functional.conv3d."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from objectdetection_amd import synth


def py_points_to_voxel(points, voxel_size, coors_range, max_points, max_voxels):
    """Pure-Python loop with numpy float32 scalars, shaped like second/utils/simplevis.py:23-50."""
    vs = np.asarray(voxel_size, np.float32)
    rg = np.asarray(coors_range, np.float32)
    grid = np.round((rg[3:] - rg[:3]) / vs).astype(np.int32)
    lut = {}
    voxels, coors, nums = [], [], []
    for i in range(points.shape[0]):
        c = [0, 0, 0]
        failed = False
        for j in range(3):
            v = np.floor((points[i, j] - rg[j]) / vs[j])
            if v < 0 or v >= grid[j]:
                failed = True
                break
            c[2 - j] = int(v)
        if failed:
            continue
        key = tuple(c)
        if key not in lut:
            if len(voxels) >= max_voxels:
                break
            lut[key] = len(voxels)
            voxels.append(np.zeros((max_points, points.shape[1]), np.float32))
            coors.append(c)
            nums.append(0)
        vid = lut[key]
        if nums[vid] < max_points:
            voxels[vid][nums[vid]] = points[i]
            nums[vid] += 1
    return (np.stack(voxels) if voxels else np.zeros((0, max_points, points.shape[1]), np.float32),
            np.array(coors, np.int32).reshape(-1, 3), np.array(nums, np.int32))


@pytest.mark.parametrize("max_voxels", [40000, 700])
def test_voxelizer_matches_python_loop(max_voxels):
    pts = synth.vkitti_frame(0, 3000)
    ev, ec, en = py_points_to_voxel(pts, synth.VOXEL_SIZE, synth.PC_RANGE, 5, max_voxels)
    gv, gc, gn = oracle.voxelize(pts, synth.VOXEL_SIZE, synth.PC_RANGE, 5, max_voxels)
    assert np.array_equal(gv, ev) and np.array_equal(gc, ec) and np.array_equal(gn, en)


def test_voxelizer_known_answer():
    vs, rg = (1.0, 1.0, 1.0), (0, 0, 0, 4, 4, 2)
    pts = np.array([[0.5, 0.5, 0.5, 1], [3.5, 0.5, 1.5, 2], [0.6, 0.4, 0.1, 3], [4.0, 1, 1, 4],
                    [0.1, 0.9, 0.9, 5], [-0.1, 1, 1, 6], [3.2, 0.2, 1.9, 7]], np.float32)
    v, c, n = oracle.voxelize(pts, vs, rg, 2, 10)
    assert c.tolist() == [[0, 0, 0], [1, 0, 3]]          # zyx, first-appearance order
    assert n.tolist() == [2, 2]                            # third point of voxel 0 is dropped
    assert v[0, :, 3].tolist() == [1, 3] and v[1, :, 3].tolist() == [2, 7]
    # max_voxels: break (mode 0) stops at the point that would open voxel #1
    v, c, n = oracle.voxelize(pts, vs, rg, 2, 1, cap_mode=0)
    assert n.tolist() == [1]
    v, c, n = oracle.voxelize(pts, vs, rg, 2, 1, cap_mode=1)  # continue: later points still join
    assert n.tolist() == [2] and v[0, :, 3].tolist() == [1, 3]


def test_shape_chain_of_spmiddlefhd():
    """middle.py:153-189 comments: 41->21->11->5->2, 1600->800->400->200, 1408->704->352->176."""
    s = [41, 1600, 1408]
    s = oracle.conv_out_shape(s, 3, 2, 1, 1)
    assert s == [21, 800, 704]
    s = oracle.conv_out_shape(s, 3, 2, 1, 1)
    assert s == [11, 400, 352]
    s = oracle.conv_out_shape(s, 3, 2, [0, 1, 1], 1)
    assert s == [5, 200, 176]
    s = oracle.conv_out_shape(s, [3, 1, 1], [2, 1, 1], 0, 1)
    assert s == [2, 200, 176]


def test_subm_known_answers():
    pairs, pn = oracle.rulebook_subm(np.array([[0, 5, 5, 5]], np.int32), 1, [11, 11, 11], 3)
    assert pn.sum() == 1 and pn[13] == 1
    two = np.array([[0, 5, 5, 5], [0, 5, 5, 6]], np.int32)
    pairs, pn = oracle.rulebook_subm(two, 1, [11, 11, 11], 3)
    assert pn.sum() == 4 and pn[12] == pn[14] == 1 and pn[13] == 2
    # offset 14 = (kz,ky,kx) = (1,1,2): input x = output x + 1
    assert (pairs[0, 14, 0], pairs[1, 14, 0]) == (1, 0)
    assert (pairs[0, 12, 0], pairs[1, 12, 0]) == (0, 1)
    # different batches never pair
    two[1, 0] = 1
    pairs, pn = oracle.rulebook_subm(two, 2, [11, 11, 11], 3)
    assert pn.sum() == 2


def _random_sparse(rng, batch, shape, n, c):
    cells = np.unique(rng.integers(0, batch * shape[0] * shape[1] * shape[2], n))
    rng.shuffle(cells)
    x = cells % shape[2]
    y = (cells // shape[2]) % shape[1]
    z = (cells // (shape[2] * shape[1])) % shape[0]
    b = cells // (shape[2] * shape[1] * shape[0])
    coors = np.ascontiguousarray(np.stack([b, z, y, x], 1).astype(np.int32))
    feats = rng.standard_normal((coors.shape[0], c)).astype(np.float32)
    dense = np.zeros((batch, c, *shape), np.float32)
    dense[coors[:, 0], :, coors[:, 1], coors[:, 2], coors[:, 3]] = feats
    return coors, feats, dense


CASES = [(3, 1, 1, 1), (3, 2, 1, 1), (3, 2, 0, 1), ([3, 1, 1], [2, 1, 1], 0, 1), (2, 2, 0, 1),
         ([1, 3, 3], 1, [0, 1, 1], 1), (3, 1, 2, 2), (1, 1, 0, 1), ([3, 3, 2], [2, 1, 2], [1, 0, 1], 1)]


@pytest.mark.parametrize("ks,st,pd,dl", CASES)
def test_sparse_conv_equals_dense_conv3d(ks, st, pd, dl):
    """SparseConv3d == dense conv3d evaluated at the active outputs (every output with >= 1 active
    input); inactive outputs of the dense conv are exactly zero when there is no bias."""
    rng = np.random.default_rng(5)
    batch, shape, cin, cout = 2, [6, 9, 10], 5, 7
    coors, feats, dense_in = _random_sparse(rng, batch, shape, 260, cin)
    k3 = [ks] * 3 if isinstance(ks, int) else ks
    w = rng.standard_normal((*k3, cin, cout)).astype(np.float32)
    oc, pairs, pn, oshape = oracle.rulebook_conv(coors, batch, shape, ks, st, pd, dl)
    out = oracle.indice_conv_fwd(feats, w, pairs, pn, oc.shape[0])
    ref = F.conv3d(torch.from_numpy(dense_in), torch.from_numpy(w).permute(4, 3, 0, 1, 2),
                   stride=st, padding=pd, dilation=dl).numpy()
    assert list(ref.shape[2:]) == oshape
    got = oracle.dense(out, oc, batch, oshape)
    assert np.abs(got - ref).max() < 1e-4
    # the active set is exactly the support of the dense conv of the occupancy mask
    occ = (np.abs(dense_in).sum(1, keepdims=True) > 0).astype(np.float32)
    sup = F.conv3d(torch.from_numpy(occ), torch.ones((1, 1, *k3)), stride=st, padding=pd,
                   dilation=dl).numpy()[:, 0] > 0
    act = np.zeros_like(sup)
    act[oc[:, 0], oc[:, 1], oc[:, 2], oc[:, 3]] = True
    assert np.array_equal(act, sup)
    assert len({tuple(r) for r in oc.tolist()}) == oc.shape[0]  # no duplicate outputs


@pytest.mark.parametrize("ks,dl", [(3, 1), ([1, 3, 3], 1), (3, 2), (1, 1)])
def test_subm_conv_equals_masked_dense_conv3d(ks, dl):
    rng = np.random.default_rng(9)
    batch, shape, cin, cout = 2, [6, 9, 10], 4, 6
    coors, feats, dense_in = _random_sparse(rng, batch, shape, 300, cin)
    k3 = [ks] * 3 if isinstance(ks, int) else ks
    w = rng.standard_normal((*k3, cin, cout)).astype(np.float32)
    pairs, pn = oracle.rulebook_subm(coors, batch, shape, ks, dl)
    out = oracle.indice_conv_fwd(feats, w, pairs, pn, coors.shape[0])
    pad = [(k // 2) * dl for k in k3]
    ref = F.conv3d(torch.from_numpy(dense_in), torch.from_numpy(w).permute(4, 3, 0, 1, 2),
                   padding=pad, dilation=dl).numpy()
    ref_rows = ref[coors[:, 0], :, coors[:, 1], coors[:, 2], coors[:, 3]]
    assert np.abs(out - ref_rows).max() < 1e-4


def test_backward_matches_autograd_of_dense_conv():
    rng = np.random.default_rng(11)
    batch, shape, cin, cout = 1, [5, 8, 8], 3, 4
    coors, feats, dense_in = _random_sparse(rng, batch, shape, 120, cin)
    w = rng.standard_normal((3, 3, 3, cin, cout)).astype(np.float32)
    oc, pairs, pn, oshape = oracle.rulebook_conv(coors, batch, shape, 3, 2, 1)
    g = rng.standard_normal((oc.shape[0], cout)).astype(np.float32)
    din, dw = oracle.indice_conv_bwd(feats, w, g, pairs, pn)
    x = torch.from_numpy(dense_in).double().requires_grad_(True)
    wt = torch.from_numpy(w).double().requires_grad_(True)
    y = F.conv3d(x, wt.permute(4, 3, 0, 1, 2), stride=2, padding=1)
    gd = torch.zeros_like(y)
    gd[oc[:, 0], :, oc[:, 1], oc[:, 2], oc[:, 3]] = torch.from_numpy(g).double()
    y.backward(gd)
    ref_din = x.grad[coors[:, 0], :, coors[:, 1], coors[:, 2], coors[:, 3]].numpy()
    assert np.abs(din - ref_din).max() < 1e-4
    assert np.abs(dw - wt.grad.numpy()).max() < 1e-4


def test_strided_discovery_order_known_answer():
    two = np.array([[0, 5, 5, 5], [0, 5, 5, 6]], np.int32)
    oc, pairs, pn, oshape = oracle.rulebook_conv(two, 1, [11, 11, 11], 3, 2, 1)
    assert oshape == [6, 6, 6] and oc.shape[0] == 8 and pn.sum() == 12
    assert oc.tolist()[:4] == [[0, 3, 3, 3], [0, 3, 3, 2], [0, 3, 2, 3], [0, 3, 2, 2]]


def test_run_sequential_matches_cpu_pipeline():
    from objectdetection_amd.frontend import SparseFrontEnd, randomize_
    from oracle import cpu_pipeline
    m = randomize_(SparseFrontEnd("M1").eval(), 0)
    pts = synth.vkitti_frame(0, 4000)
    v, c, n = oracle.voxelize(pts, synth.VOXEL_SIZE, synth.PC_RANGE, 5, 40000)
    coors = np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], 1)
    feats = v.sum(1) / n[:, None].astype(np.float32)
    f1, c1, s1 = oracle.run_sequential(m.middle_feature_extractor.middle_conv, feats, coors,
                                       [41, 1600, 1408], 1)
    d1 = oracle.dense(f1, c1, 1, s1).reshape(1, 128, 200, 176)
    d2 = cpu_pipeline.frontend_forward(m, [pts])
    assert np.abs(d1 - d2).max() / np.abs(d2).max() < 1e-5
