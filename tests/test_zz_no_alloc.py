"""GPU: lcf_voxelize_static -- the one entry point that allocates nothing (its hash workspace is
a file-scope device array, include/lcf.h) -- against lcf_voxelize, through the C ABI, bit for
bit, on two calls in a row, the second with a smaller input (the array keeps the first call's
contents and must be re-armed by the call itself).  Collected last (test_zz_*): the round's end
runs `pytest -x`."""
import numpy as np
import pytest
import torch

from objectdetection_amd import _lib, synth
from objectdetection_amd._lib import f3, f6

pytestmark = pytest.mark.gpu


def _outputs(dev, mv, T, F, cs):
    return (torch.full((mv, T, F), 7.0, dtype=torch.float32, device=dev),
            torch.full((mv, cs), 7, dtype=torch.int32, device=dev),
            torch.full((mv,), 7, dtype=torch.int32, device=dev),
            torch.full((1,), 7, dtype=torch.int32, device=dev))


def _call(L, static, pts, vs, rg, T, mv, cap_mode, cs, dev):
    n, F = pts.shape
    v, c, num, vn = _outputs(dev, mv, T, F, cs)
    st = torch.cuda.current_stream().cuda_stream
    head = (pts.data_ptr(), n, None, F, f3(vs), f6(rg), T, mv, cap_mode, 3, cs, v.data_ptr(),
            c.data_ptr(), num.data_ptr(), vn.data_ptr())
    if static:
        rc = L.lcf_voxelize_static(*head, st)
    else:
        ws = torch.empty(max(L.lcf_voxelize_workspace_bytes(n, T), 256), dtype=torch.uint8,
                         device=dev)
        rc = L.lcf_voxelize(*head, ws.data_ptr(), ws.numel(), st)
    assert rc == 0, L.lcf_strerror(rc)
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in (v, c, num, vn)]


def _same(a, b):
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x,
                              y.view(np.uint32) if y.dtype == np.float32 else y)


def test_static_entry_matches_lcf_voxelize_twice_second_call_smaller(cuda):
    L = _lib.lib()
    vs, rg = synth.VOXEL_SIZE, synth.PC_RANGE
    big = torch.from_numpy(synth.synthia_frame(0, 120000)).to(cuda)
    small = torch.from_numpy(synth.vkitti_frame(1, 6000)).to(cuda)
    for pts, mv, cap_mode, cs in ((big, 160000, 0, 4), (small, 40000, 0, 3),
                                  (small[:37], 10, 1, 4), (big[:50000], 30000, 0, 3)):
        got = _call(L, True, pts, vs, rg, 5, mv, cap_mode, cs, cuda)
        want = _call(L, False, pts, vs, rg, 5, mv, cap_mode, cs, cuda)
        assert int(want[3][0]) > 0
        _same(got, want)
    # classic grid, T = 35, then an empty cloud
    got = _call(L, True, big[:30000], (0.2, 0.2, 0.4), rg, 35, 20000, 0, 3, cuda)
    _same(got, _call(L, False, big[:30000], (0.2, 0.2, 0.4), rg, 35, 20000, 0, 3, cuda))
    got = _call(L, True, big[:0], vs, rg, 5, 64, 0, 3, cuda)
    _same(got, _call(L, False, big[:0], vs, rg, 5, 64, 0, 3, cuda))


def test_static_entry_rejects_inputs_beyond_its_array(cuda):
    L = _lib.lib()
    cap = L.lcf_voxelize_static_max_bytes()
    assert cap == 16 << 20
    n = 1 << 20
    assert L.lcf_voxelize_workspace_bytes(n, 5) > cap
    pts = torch.zeros((n, 4), dtype=torch.float32, device=cuda)
    v, c, num, vn = _outputs(cuda, 16, 5, 4, 3)
    rc = L.lcf_voxelize_static(pts.data_ptr(), n, None, 4, f3(synth.VOXEL_SIZE),
                               f6(synth.PC_RANGE), 5, 16, 0, 0, 3, v.data_ptr(), c.data_ptr(),
                               num.data_ptr(), vn.data_ptr(),
                               torch.cuda.current_stream().cuda_stream)
    assert rc == -6 and b"static scratch" in L.lcf_strerror(rc)
    torch.cuda.synchronize()
    assert int(vn.item()) == 7 and int(num[0].item()) == 7  # nothing ran
