"""CPU oracle for the SECOND sparse front end -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product (``objectdetection_amd``) never does.

ctypes wrapper over ``oracle/liblcf_oracle.so`` (plain C, see lcf_oracle.c for the reference
file:line each routine follows) plus numpy/torch restatements of the VFE
(``oracle/vfe_oracle.py``, pinned by tests/golden/vfe_*.npz generated from the real
/root/reference/second/pytorch/models/voxel_encoder.py).

Parity pin status (fixtures under tests/golden/, each made by running the reference's own files
in the build container; generators committed beside them):
  voxelizer      PINNED  voxel_golden.npz   <- second/utils/simplevis.py:8-59 (the in-tree twin of
                         the missing points_to_voxel loop), run as plain Python
  dense / gather PINNED  dense_golden.npz   <- torchplus/ops/array_ops.py:7-30
  VFE            PINNED  vfe_golden.npz     <- second/pytorch/models/voxel_encoder.py
  IoU / NMS      PINNED  nms_golden.npz     <- second/core/non_max_suppression/nms_gpu.py, its
                         numba-CUDA kernels under a SIMT emulator (oracle/nms_oracle.py)
  LC policy      PINNED  lc_policy_golden.npz <- second/light_curtain/policy.py, utils.py
                         (oracle/lc_policy_oracle.py; the pylc device geometry alone is unpinned)
  rulebook / indice_conv  "parity unpinned" against spconv's binaries (absent, un-vendored,
                         version not pinned: SURVEY.md F2); pinned by known-answer tests and by
                         equivalence with dense torch conv3d (tests/test_oracle_pins.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile liblcf_oracle.so with gcc (no-op if up to date)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liblcf_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def _i3(v):
    if isinstance(v, (int, np.integer)):
        v = [int(v)] * 3
    a = np.ascontiguousarray(np.asarray(v, dtype=np.int32).reshape(3))
    return a


_f = ctypes.c_float
_i = ctypes.c_int


def voxelize(points, voxel_size, pc_range, max_points, max_voxels, cap_mode=0, pad=False):
    """-> voxels [K,T,F] f32, coors [K,3] i32 (z,y,x), num_points [K] i32 (padded if pad)."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    n, f = pts.shape
    vs = np.ascontiguousarray(voxel_size, dtype=np.float32)
    rg = np.ascontiguousarray(pc_range, dtype=np.float32)
    voxels = np.empty((max_voxels, max_points, f), np.float32)
    coors = np.empty((max_voxels, 3), np.int32)
    num = np.empty((max_voxels,), np.int32)
    nv = ctypes.c_int(0)
    rc = lib().orc_voxelize(_p(pts, _f), n, f, _p(vs, _f), _p(rg, _f), int(max_points),
                            int(max_voxels), int(cap_mode), _p(voxels, _f), _p(coors, _i),
                            _p(num, _i), ctypes.byref(nv))
    if rc != 0:
        raise RuntimeError(f"orc_voxelize rc={rc}")
    k = nv.value
    if pad:
        return voxels, coors, num, k
    return voxels[:k].copy(), coors[:k].copy(), num[:k].copy()


def conv_out_shape(shape, ksize, stride, pad, dil):
    out = np.zeros(3, np.int32)
    lib().orc_conv_out_shape(_p(_i3(shape), _i), _p(_i3(ksize), _i), _p(_i3(stride), _i),
                             _p(_i3(pad), _i), _p(_i3(dil), _i), _p(out, _i))
    return [int(x) for x in out]


def rulebook_subm(coors, batch, shape, ksize, dil=1):
    """coors [K,4] (b,z,y,x) -> pairs i32 [2,kvol,K] (-1 filled), pair_num i32 [kvol]."""
    c = np.ascontiguousarray(coors, dtype=np.int32)
    k = c.shape[0]
    ks = _i3(ksize)
    kvol = int(np.prod(ks))
    pairs = np.empty((2, kvol, k), np.int32)
    pnum = np.zeros(kvol, np.int32)
    if k == 0:
        return pairs, pnum
    rc = lib().orc_rulebook_subm(_p(c, _i), k, int(batch), _p(_i3(shape), _i), _p(ks, _i),
                                 _p(_i3(dil), _i), _p(pairs, _i), _p(pnum, _i))
    if rc != 0:
        raise RuntimeError(f"orc_rulebook_subm rc={rc}")
    return pairs, pnum


def rulebook_conv(coors, batch, shape, ksize, stride, pad, dil=1):
    """-> out_coors [K_out,4], pairs [2,kvol,K_in], pair_num [kvol], out_shape."""
    c = np.ascontiguousarray(coors, dtype=np.int32)
    k = c.shape[0]
    ks = _i3(ksize)
    kvol = int(np.prod(ks))
    oshape = conv_out_shape(shape, ksize, stride, pad, dil)
    pairs = np.empty((2, kvol, k), np.int32)
    pnum = np.zeros(kvol, np.int32)
    if k == 0:
        return np.zeros((0, 4), np.int32), pairs, pnum, oshape
    outc = np.empty((k * kvol, 4), np.int32)
    ko = ctypes.c_int(0)
    rc = lib().orc_rulebook_conv(_p(c, _i), k, int(batch), _p(_i3(shape), _i), _p(ks, _i),
                                 _p(_i3(stride), _i), _p(_i3(pad), _i), _p(_i3(dil), _i),
                                 _p(outc, _i), ctypes.byref(ko), _p(pairs, _i), _p(pnum, _i))
    if rc != 0:
        raise RuntimeError(f"orc_rulebook_conv rc={rc}")
    return outc[:ko.value].copy(), pairs, pnum, oshape


def indice_conv_fwd(feat, w, pairs, pair_num, k_out):
    """feat [K_in,Cin] f32, w [kD,kH,kW,Cin,Cout] (or [kvol,Cin,Cout]) -> out [K_out,Cout]."""
    x = np.ascontiguousarray(feat, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    cin, cout = w.shape[-2], w.shape[-1]
    kvol = w.size // (cin * cout)
    k_in = x.shape[0]
    out = np.zeros((k_out, cout), np.float32)
    pr = np.ascontiguousarray(pairs, dtype=np.int32)
    pn = np.ascontiguousarray(pair_num, dtype=np.int32)
    assert pr.shape == (2, kvol, k_in), (pr.shape, kvol, k_in)
    if k_in and k_out:
        lib().orc_indice_conv_fwd(_p(x, _f), _p(w, _f), _p(pr, _i), _p(pn, _i), k_in, k_out,
                                  cin, cout, kvol, _p(out, _f))
    return out


def indice_conv_bwd(feat, w, dout, pairs, pair_num):
    """-> din [K_in,Cin], dw (same shape as w)."""
    x = np.ascontiguousarray(feat, dtype=np.float32)
    w_ = np.ascontiguousarray(w, dtype=np.float32)
    g = np.ascontiguousarray(dout, dtype=np.float32)
    cin, cout = w_.shape[-2], w_.shape[-1]
    kvol = w_.size // (cin * cout)
    k_in, k_out = x.shape[0], g.shape[0]
    din = np.zeros_like(x)
    dw = np.zeros_like(w_)
    pr = np.ascontiguousarray(pairs, dtype=np.int32)
    pn = np.ascontiguousarray(pair_num, dtype=np.int32)
    if k_in and k_out:
        lib().orc_indice_conv_bwd(_p(x, _f), _p(w_, _f), _p(g, _f), _p(pr, _i), _p(pn, _i), k_in,
                                  k_out, cin, cout, kvol, _p(din, _f), _p(dw, _f))
    return din, dw


def dense(feat, coors, batch, shape):
    """-> [B,C,D,H,W] f32."""
    x = np.ascontiguousarray(feat, dtype=np.float32)
    c = np.ascontiguousarray(coors, dtype=np.int32)
    k, ch = x.shape
    sh = _i3(shape)
    out = np.empty((batch, ch, int(sh[0]), int(sh[1]), int(sh[2])), np.float32)
    lib().orc_dense(_p(x, _f), _p(c, _i), k, ch, int(batch), _p(sh, _i), _p(out, _f))
    return out


def run_sequential(seq, features, coors, spatial_shape, batch_size, collect=None):
    """Execute a SparseSequential-like container of (sparse conv | BatchNorm1d | ReLU | ...)
    modules on the oracle.  Sparse conv modules are recognised by duck typing (attributes
    ``subm``, ``kernel_size``, ``stride``, ``padding``, ``dilation``, ``weight``,
    ``indice_key``); every other module runs as a CPU torch module on the features.

    Mirrors spconv.SparseSequential.forward + SparseConvolution.forward as the reference uses
    them (second/pytorch/models/middle.py:145-192).  Returns (features, coors, shape); if
    ``collect`` is a list, appends a dict per sparse conv with its rulebook.
    """
    import copy
    import torch
    feats = np.ascontiguousarray(features, dtype=np.float32)
    coors = np.ascontiguousarray(coors, dtype=np.int32)
    shape = [int(s) for s in spatial_shape]
    cache = {}
    for m in seq:
        if hasattr(m, "subm") and hasattr(m, "kernel_size"):
            w = m.weight.detach().cpu().float().numpy()
            key = getattr(m, "indice_key", None)
            if key is not None and key in cache:
                outc, pairs, pnum, oshape = cache[key]
            elif m.subm:
                pairs, pnum = rulebook_subm(coors, batch_size, shape, m.kernel_size, m.dilation)
                outc, oshape = coors, shape
            else:
                outc, pairs, pnum, oshape = rulebook_conv(coors, batch_size, shape,
                                                         m.kernel_size, m.stride, m.padding,
                                                         m.dilation)
            if key is not None:
                cache[key] = (outc, pairs, pnum, oshape)
            feats = indice_conv_fwd(feats, w, pairs, pnum, outc.shape[0])
            if getattr(m, "bias", None) is not None:
                feats = feats + m.bias.detach().cpu().float().numpy()[None, :]
            if collect is not None:
                collect.append(dict(key=key, subm=bool(m.subm), pairs=pairs, pair_num=pnum,
                                    out_coors=outc, out_shape=oshape,
                                    k_in=coors.shape[0], k_out=outc.shape[0]))
            coors, shape = outc, oshape
        else:
            with torch.no_grad():
                mc = copy.deepcopy(m).cpu().float()
                mc.train(m.training)
                feats = mc(torch.from_numpy(feats)).numpy()
    return feats, coors, shape
