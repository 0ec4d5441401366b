"""GPU: incremental voxelisation of a growing cloud (lcf_voxelize_append) against
the CPU oracle run FROM SCRATCH on the concatenated cloud -- what the reference's light-curtain
loop computes (second/light_curtain/utils.py:121-124 stacks the curtain returns under the cloud,
then preprocess.py:626-639 voxelises all of it).  Bit-exact after every append: voxel order,
slots, counts, padding, and the max_voxels cut-off wherever it falls."""
import numpy as np
import pytest
import torch

from objectdetection_amd import ops, synth
from objectdetection_amd.spconv.utils import VoxelGeneratorV2

pytestmark = pytest.mark.gpu


def _curtain(seed, n):
    """Curtain-like returns: points on a vertical ruled surface inside the range, many of them in
    cells the lidar-like cloud already occupies near the ground."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(5, 60, n).astype(np.float32)
    y = (0.3 * x - 10 + rng.normal(0, 0.02, n)).astype(np.float32)
    z = rng.uniform(-1.75, 1.0, n).astype(np.float32)
    i = rng.uniform(50 / 255, 1, n).astype(np.float32)
    return np.stack([x, y, z, i], axis=1)


def _session(oracle, cuda, chunks, vs, rg, T, mv, cap_mode, batch_idx=None):
    cap = sum(len(c) for c in chunks) + 17
    acc = ops.VoxelAccumulator(cap, 4, vs, rg, T, mv, cap_mode, batch_idx, cuda)
    acc.reset()
    seen = np.zeros((0, 4), np.float32)
    ks = []
    for c in chunks:
        seen = np.concatenate([seen, c])
        gv, gc, gn, gk = acc.append(torch.from_numpy(c).to(cuda))
        torch.cuda.synchronize()
        ev, ec, en, ek = oracle.voxelize(seen, vs, rg, T, mv, cap_mode, pad=True)
        assert int(gk.item()) == ek
        gc = gc.cpu().numpy()
        if batch_idx is not None:
            assert np.all(gc[:ek, 0] == batch_idx)
            gc = gc[:, 1:]
        assert np.array_equal(gc, ec)
        assert np.array_equal(gn.cpu().numpy(), en)
        assert np.array_equal(gv.cpu().numpy().view(np.uint32), ev.view(np.uint32))
        ks.append(ek)
    assert acc.n == len(seen) and np.array_equal(acc.cloud.cpu().numpy(), seen)
    return ks


@pytest.mark.parametrize("cap_mode", [0, 1])
def test_lidar_then_curtains(oracle, cuda, cap_mode):
    chunks = [synth.vkitti_frame(0, 20000)] + [_curtain(s, 3000 + 500 * s) for s in range(4)]
    ks = _session(oracle, cuda, chunks, synth.VOXEL_SIZE, synth.PC_RANGE, 5, 40000, cap_mode)
    assert all(b > a for a, b in zip(ks, ks[1:]))  # every curtain opens new voxels


@pytest.mark.parametrize("cap_mode", [0, 1])
@pytest.mark.parametrize("max_voxels", [9000, 15500, 17500])
def test_cut_off_falls_in_any_chunk(oracle, cuda, cap_mode, max_voxels):
    """max_voxels reached inside the first cloud, inside a later append, or just before the end."""
    chunks = [synth.vkitti_frame(1, 12000), _curtain(7, 4000), _curtain(8, 4000)]
    ks = _session(oracle, cuda, chunks, synth.VOXEL_SIZE, synth.PC_RANGE, 5, max_voxels, cap_mode)
    assert ks[-1] == max_voxels


def test_full_voxels_duplicates_and_empty_appends(oracle, cuda):
    """Classic 0.2/0.2/0.4 grid with T = 8: appended points find full voxels (dropped), half-full
    ones (later slots), and exact duplicates of resident points; empty and 1-point appends."""
    base = synth.synthia_frame(2, 30000)
    dup = base[100:2100].copy()
    chunks = [base[:10000], np.zeros((0, 4), np.float32), base[10000:10001], base[10001:], dup,
              _curtain(3, 5000), np.array([[1e9, 0, 0, 1]], np.float32)]
    _session(oracle, cuda, chunks, (0.2, 0.2, 0.4), synth.PC_RANGE, 8, 40000, 0)


def test_session_from_empty_and_batch_column(oracle, cuda):
    chunks = [np.zeros((0, 4), np.float32), _curtain(1, 700), _curtain(2, 5)]
    _session(oracle, cuda, chunks, synth.VOXEL_SIZE, synth.PC_RANGE, 5, 1000, 0, batch_idx=3)


def test_reset_starts_over_and_generator_front_door(oracle, cuda):
    gen = VoxelGeneratorV2(synth.VOXEL_SIZE, synth.PC_RANGE, 5, 40000)
    acc = gen.accumulator(30000)
    a, b = synth.vkitti_frame(4, 20000), _curtain(5, 6000)
    acc.append(a)
    acc.append(b)  # host arrays are accepted: only the new rows cross PCIe
    first = [t.clone() for t in acc.outputs]
    acc.reset()
    acc.append(b)
    acc.append(a)  # another order: another first-appearance numbering
    assert not torch.equal(first[1], acc.outputs[1])
    acc.reset()
    acc.append(a)
    acc.append(b)
    for x, y in zip(first, acc.outputs):
        assert torch.equal(x, y)
    want = gen.generate_device(torch.from_numpy(np.concatenate([a, b])).to(cuda))
    assert torch.equal(want["voxels"], acc.voxels) and torch.equal(want["coordinates"], acc.coors)
    assert torch.equal(want["num_points_per_voxel"], acc.num)
    with pytest.raises(ValueError):
        acc.append(np.zeros((30000, 4), np.float32))
