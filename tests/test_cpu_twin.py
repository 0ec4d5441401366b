"""CPU: the GPU-free twin of the voxelizer (include/lcf_cpu.h, objectdetection_amd/cpu.py; SURVEY.md
section 8b) against the fixtures made by RUNNING the reference's own loop
(second/utils/simplevis.py:8-59, tests/golden/make_ref_goldens.py), against the oracle on the edge
cases the domain has, and from a forked worker process, which is where the reference voxelizes
(base.yaml:120 num_workers: 3)."""
import ctypes
import multiprocessing as mp
import os
import re

import numpy as np
import pytest

import oracle
from objectdetection_amd import cpu, synth
from test_ref_goldens import VOXEL_CASES, check_voxels_against_reference, voxel_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_what_the_header_declares_and_links_no_gpu_runtime():
    hdr = open(os.path.join(ROOT, "include", "lcf_cpu.h")).read()
    names = re.findall(r"^int (lcf_cpu_\w+)\(", hdr, flags=re.M)
    assert names == ["lcf_cpu_voxelize"]
    L = ctypes.CDLL(cpu.LIB_PATH)
    for n in names:
        assert getattr(L, n) is not None
    needed = os.popen(f"readelf -d {cpu.LIB_PATH}").read()
    assert "amdhip" not in needed and "hsa" not in needed


@pytest.mark.parametrize("tag", VOXEL_CASES)
def test_cpu_twin_matches_reference_loop(tag):
    c = voxel_case(tag)
    v, coors, num, k = cpu.voxelize(c["points"], c["voxel_size"], c["range"], 64,
                                    int(c["max_voxels"]), cap_mode=0)
    assert num.max() < 64
    check_voxels_against_reference(c, coors[:k], num[:k])
    assert not v[k:].any() and not coors[k:].any() and not num[k:].any()


@pytest.mark.parametrize("cap_mode", [0, 1])
@pytest.mark.parametrize("n,max_pts,max_vox", [(20000, 5, 40000), (20000, 5, 3000), (6000, 1, 700),
                                               (3000, 35, 20000), (1, 5, 10), (0, 5, 10)])
def test_cpu_twin_equals_oracle(n, max_pts, max_vox, cap_mode):
    pts = synth.vkitti_frame(11, n) if n else np.zeros((0, 4), np.float32)
    if n > 100:  # a few points outside the range and on its upper faces
        pts[::97, 0] = 70.4
        pts[5::101, 2] = -3.0001
    v, c, num, k = cpu.voxelize(pts, synth.VOXEL_SIZE, synth.PC_RANGE, max_pts, max_vox, cap_mode)
    ev, ec, en = oracle.voxelize(pts, synth.VOXEL_SIZE, synth.PC_RANGE, max_pts, max_vox,
                                 cap_mode=cap_mode)
    assert k == ev.shape[0]
    assert np.array_equal(v[:k], ev) and np.array_equal(c[:k], ec) and np.array_equal(num[:k], en)


def test_batch_column_and_points_to_voxel_signature():
    pts = synth.vkitti_frame(3, 4000)
    v, c4, num, k = cpu.voxelize(pts, synth.VOXEL_SIZE, synth.PC_RANGE, 5, 40000, batch_idx=2)
    _, c3, _, _ = cpu.voxelize(pts, synth.VOXEL_SIZE, synth.PC_RANGE, 5, 40000)
    assert c4.shape[1] == 4 and np.all(c4[:k, 0] == 2) and np.array_equal(c4[:k, 1:], c3[:k])
    vox, coors, n = cpu.points_to_voxel(pts.astype(np.float64), synth.VOXEL_SIZE, synth.PC_RANGE,
                                        max_points=5, reverse_index=False, max_voxels=40000)
    assert vox.dtype == np.float64 and np.array_equal(coors, c3[:k, ::-1]) and len(n) == k
    with pytest.raises(ValueError):
        cpu.voxelize(pts[:, :2], synth.VOXEL_SIZE, synth.PC_RANGE, 5, 100)
    with pytest.raises(ValueError):
        cpu.voxelize(pts, synth.VOXEL_SIZE, synth.PC_RANGE, 0, 100)


def _worker(seed, q):
    # a forked child: no GPU call may happen here, and none does
    from objectdetection_amd.spconv.utils import VoxelGeneratorV2
    gen = VoxelGeneratorV2(synth.VOXEL_SIZE, synth.PC_RANGE, 5, max_voxels=40000, device="cpu")
    res = gen.generate(synth.vkitti_frame(seed, 5000))
    q.put((seed, res["voxel_num"], res["coordinates"].sum(), res["voxels"].sum()))


def test_voxel_generator_on_the_cpu_twin_in_forked_workers():
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(s, q)) for s in (1, 2, 3)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=60) for _ in ps)
    for p in ps:
        p.join(30)
        assert p.exitcode == 0
    for seed, k, csum, vsum in got:
        ev, ec, en = oracle.voxelize(synth.vkitti_frame(seed, 5000), synth.VOXEL_SIZE,
                                     synth.PC_RANGE, 5, 40000)
        assert k == ev.shape[0] and csum == ec.sum() and vsum == ev.sum()
