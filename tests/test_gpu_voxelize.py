"""GPU parity: HIP voxelizer vs the CPU oracle (bit-exact), through the C ABI."""
import numpy as np
import pytest
import torch

from objectdetection_amd import ops, synth

pytestmark = pytest.mark.gpu


def _run(pts, vs, rg, T, mv, cap_mode, dev):
    voxels, coors, num, vnum = ops.voxelize(torch.from_numpy(pts).to(dev), vs, rg, T, mv, cap_mode)
    torch.cuda.synchronize()
    return voxels.cpu().numpy(), coors.cpu().numpy(), num.cpu().numpy(), int(vnum.item())


def _check(oracle, pts, vs, rg, T, mv, cap_mode, dev):
    ev, ec, en, ek = oracle.voxelize(pts, vs, rg, T, mv, cap_mode, pad=True)
    gv, gc, gn, gk = _run(pts, vs, rg, T, mv, cap_mode, dev)
    assert gk == ek
    # bit-exact, including the zero padding beyond voxel_num
    assert np.array_equal(gc, ec)
    assert np.array_equal(gn, en)
    assert np.array_equal(gv.view(np.uint32), ev.view(np.uint32))
    return ek


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_vkitti_frame_bit_exact(oracle, cuda, seed):
    pts = synth.vkitti_frame(seed, 20000)
    k = _check(oracle, pts, synth.VOXEL_SIZE, synth.PC_RANGE, 5, 40000, 0, cuda)
    assert 10000 < k <= 20000


def test_synthia_frame_bit_exact(oracle, cuda):
    pts = synth.synthia_frame(0, 120000)
    _check(oracle, pts, synth.VOXEL_SIZE, synth.PC_RANGE, 5, 200000, 0, cuda)


@pytest.mark.parametrize("cap_mode", [0, 1])
@pytest.mark.parametrize("max_voxels", [1, 100, 17000])
def test_max_voxels_cap(oracle, cuda, cap_mode, max_voxels):
    pts = synth.vkitti_frame(3, 20000)
    k = _check(oracle, pts, synth.VOXEL_SIZE, synth.PC_RANGE, 5, max_voxels, cap_mode, cuda)
    assert k == max_voxels


@pytest.mark.parametrize("cap_mode", [0, 1])
def test_classic_grid_dense_voxels(oracle, cuda, cap_mode):
    # 0.2/0.2/0.4 m voxels, T=35: many points per voxel, full voxels, duplicates
    pts = synth.synthia_frame(1, 120000)
    pts[1000:3000] = pts[0:2000]  # exact duplicates
    _check(oracle, pts, (0.2, 0.2, 0.4), synth.PC_RANGE, 35, 6000, cap_mode, cuda)
    _check(oracle, pts, (0.2, 0.2, 0.4), synth.PC_RANGE, 35, 40000, cap_mode, cuda)


def test_edges_empty_ragged(oracle, cuda):
    vs, rg = synth.VOXEL_SIZE, synth.PC_RANGE
    # empty cloud
    _check(oracle, np.zeros((0, 4), np.float32), vs, rg, 5, 100, 0, cuda)
    # every point out of range
    far = np.full((257, 4), 1e6, np.float32)
    assert _check(oracle, far, vs, rg, 5, 100, 0, cuda) == 0
    # points exactly on cell and range edges, negative denormal, NaN, F = 3 and F = 7
    rng = np.random.default_rng(7)
    base = rng.integers(0, 1408, 5000).astype(np.float32) * np.float32(0.05)
    for f in (3, 4, 7):
        p = np.zeros((5000, f), np.float32)
        p[:, 0] = base
        p[:, 1] = rng.integers(-800, 801, 5000).astype(np.float32) * np.float32(0.05)
        p[:, 2] = rng.integers(-30, 11, 5000).astype(np.float32) * np.float32(0.1)
        p[0, :3] = (70.4, 40.0, 1.0)          # on the upper bound: outside
        p[1, :3] = (0.0, -40.0, -3.0)         # on the lower bound: inside
        p[2, :3] = (-1e-40, 0.0, 0.0)         # negative denormal: floor -> -1 -> outside
        p[3, :3] = (np.nan, 0.0, 0.0)
        p[4, :3] = (np.nextafter(np.float32(70.4), np.float32(0)), 0.0, 0.0)
        _check(oracle, p, vs, rg, 5, 8000, 0, cuda)
    # single point, tiny N, one voxel holding everything
    _check(oracle, np.array([[1.0, 1.0, 0.0, 0.5]], np.float32), vs, rg, 5, 10, 0, cuda)
    same = np.tile(np.array([[10.01, 2.01, -1.01, 1.0]], np.float32), (3000, 1))
    same[:, 3] = np.arange(3000)
    _check(oracle, same, vs, rg, 5, 10, 0, cuda)
    _check(oracle, same, vs, rg, 35, 10, 1, cuda)


def test_order_sensitivity(oracle, cuda):
    """Voxel ids follow first appearance: the same cloud in another order gives another
    (but oracle-identical) numbering, and idempotent results on a re-run."""
    pts = synth.vkitti_frame(5, 20000)
    a = _run(pts, synth.VOXEL_SIZE, synth.PC_RANGE, 5, 40000, 0, cuda)
    b = _run(pts, synth.VOXEL_SIZE, synth.PC_RANGE, 5, 40000, 0, cuda)
    for x, y in zip(a[:3], b[:3]):
        assert np.array_equal(x, y)
    rev = np.ascontiguousarray(pts[::-1])
    _check(oracle, rev, synth.VOXEL_SIZE, synth.PC_RANGE, 5, 40000, 0, cuda)


def test_generator_api(oracle, cuda):
    from objectdetection_amd.spconv.utils import VoxelGeneratorV2
    from objectdetection_amd.point_cloud_ops import points_to_voxel
    gen = VoxelGeneratorV2(list(synth.VOXEL_SIZE), list(synth.PC_RANGE), 5, max_voxels=20000)
    assert gen.grid_size.tolist() == [1408, 1600, 40]
    assert gen.grid_size[::-1].tolist() == [40, 1600, 1408]
    assert gen.point_cloud_range[[0, 1, 3, 4]].tolist() == pytest.approx([0, -40, 70.4, 40])
    pts = synth.vkitti_frame(0, 20000)
    ev, ec, en = oracle.voxelize(pts, synth.VOXEL_SIZE, synth.PC_RANGE, 5, 17000)
    res = gen.generate(pts, 17000)
    assert res["voxel_num"] == ev.shape[0]
    assert np.array_equal(res["voxels"], ev) and np.array_equal(res["coordinates"], ec)
    assert np.array_equal(res["num_points_per_voxel"], en)
    assert res["coordinates"].dtype == np.int32 and res["voxels"].dtype == np.float32
    res2 = gen.generate_multi_gpu(pts, 17000)
    assert res2["voxels"].shape == (17000, 5, 4) and res2["voxel_num"] == ev.shape[0]
    v, c, n = points_to_voxel(pts, np.array(synth.VOXEL_SIZE, np.float32),
                              np.array(synth.PC_RANGE, np.float32), 5, reverse_index=True,
                              max_voxels=17000)
    assert np.array_equal(v, ev) and np.array_equal(c, ec) and np.array_equal(n, en)


@pytest.mark.gpu
@pytest.mark.parametrize("cap_mode,max_vox", [(0, 40000), (0, 2500), (1, 2500)])
def test_cpu_twin_equals_the_hip_voxelizer(cuda, cap_mode, max_vox):
    """include/lcf_cpu.h: the GPU-free twin for forked DataLoader workers gives what the HIP
    voxelizer gives, bit for bit, padding rows included."""
    from objectdetection_amd import cpu, ops, synth
    pts = synth.vkitti_frame(21, 20000)
    v, c, n, k = cpu.voxelize(pts, synth.VOXEL_SIZE, synth.PC_RANGE, 5, max_vox, cap_mode,
                              batch_idx=1)
    gv, gc, gn, gk = ops.voxelize(torch.from_numpy(pts).to(cuda), synth.VOXEL_SIZE, synth.PC_RANGE,
                                  5, max_vox, cap_mode, batch_idx=1)
    assert int(gk.item()) == k
    assert np.array_equal(gv.cpu().numpy(), v) and np.array_equal(gn.cpu().numpy(), n)
    assert np.array_equal(gc.cpu().numpy()[:k], c[:k])
