"""Generates tests/golden/lc_policy_golden.npz by RUNNING the reference's own light-curtain policy
files in the build container (the reference never travels; the fixture holds data only).

    python tests/golden/make_lc_goldens.py

What runs, verbatim from /root/reference, loaded by file path:
  * second/light_curtain/utils.py:177-237 `_get_confidence_map`.  The module's top imports
    `second.pytorch.train` (the whole training stack, needs spconv): a stub module with an unused
    `example_convert_to_torch` stands in.
  * second/light_curtain/policy.py `DPOptimizedPolicy` (:68-434): constructor, graph
    preprocessing, confidence2entropy, the dynamic programme.  Stubs: `termcolor` (absent; only
    used in debug prints), `np.int = int` (removed from NumPy 1.24 on; the file uses it as a
    dtype).
The light-curtain device (pylc/pylc.py) needs `transformations` and the compiled `pylc_lib`,
both absent: the policy gets a plain object holding CAMERA_PARAMS / TRANSFORMS computed by
oracle/lc_policy_oracle.py:device_geometry (restating pylc.py:12-27, 117-146; unpinned on its
own).  Every case stores that geometry, so the policy's outputs are the reference's for the
stored inputs.

Per case the fixture holds: inputs (anchor centres [Y,X,3], sizes, anchors_mask, cls_preds,
camera matrix/width/height, cTl, wTc, pts_per_cam_ray); the reference's confidence_map, its
graph nodes [T,N,5], one row and one column of every edge's laser-angle differences, and the
design points [T,2]; and the design points of the other registered policies
(`policy/<name>`; the random ones after `np.random.seed(77 + case index)`).
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    if not hasattr(np, "int"):
        np.int = int
    tc = types.ModuleType("termcolor")
    tc.colored = lambda s, *a, **k: s
    sys.modules["termcolor"] = tc
    second = types.ModuleType("second")
    pt = types.ModuleType("second.pytorch")
    tr = types.ModuleType("second.pytorch.train")
    tr.example_convert_to_torch = lambda *a, **k: None
    second.pytorch, pt.train = pt, tr
    sys.modules.update({"second": second, "second.pytorch": pt, "second.pytorch.train": tr})
    import matplotlib
    matplotlib.use("Agg")
    policy = _load("ref_lc_policy", "second/light_curtain/policy.py")
    utils = _load("ref_lc_utils", "second/light_curtain/utils.py")
    return policy, utils


def anchor_grid(Y, X, K, rng_xyz=(0.0, -40.0, -3.0, 70.4, 40.0, 1.0), z_centre=-1.0):
    """Anchor boxes the way the reference's AnchorGeneratorRange lays them out: K boxes per
    location (here: K rotations of one size), centres on a regular [Y, X] grid of the range."""
    x0, y0, _, x1, y1, _ = rng_xyz
    xs = np.linspace(x0, x1, X + 1, dtype=np.float32)[:-1] + np.float32((x1 - x0) / X / 2)
    ys = np.linspace(y0, y1, Y + 1, dtype=np.float32)[:-1] + np.float32((y1 - y0) / Y / 2)
    cen = np.zeros((Y, X, 3), dtype=np.float32)
    cen[:, :, 0] = xs[None, :]
    cen[:, :, 1] = ys[:, None]
    cen[:, :, 2] = z_centre
    return cen


def full_anchors(cen, K):
    Y, X = cen.shape[:2]
    a = np.zeros((K, Y, X, 7), dtype=np.float32)
    a[..., :3] = cen[None]
    a[..., 3:6] = np.array([1.6, 3.9, 1.56], dtype=np.float32)
    for k in range(K):
        a[k, :, :, 6] = np.float32(k * np.pi / K)
    return a.reshape(-1, 7)


def blobs(rng, K, Y, X, n, quant=None):
    """Class scores in [0, 1]: a low floor plus a few Gaussian bumps; optionally quantised so
    that many sums of entropies tie exactly."""
    yy, xx = np.mgrid[0:Y, 0:X].astype(np.float32)
    c = np.full((K, Y, X), 0.02, dtype=np.float32)
    for k in range(K):
        for _ in range(n):
            cy, cx = rng.uniform(0, Y), rng.uniform(0, X)
            s = rng.uniform(2.0, 8.0)
            c[k] += np.float32(rng.uniform(0.2, 0.9)) * np.exp(
                -((yy - cy) ** 2 + (xx - cx) ** 2) / np.float32(2 * s * s))
    c = np.clip(c, 0, 1).astype(np.float32)
    if quant:
        c = (np.round(c * quant) / quant).astype(np.float32)
    return c


CASES = [
    # name, K, Y, X, cam width, pts per ray, bumps, quantisation, with anchors_mask
    ("kitti", 2, 200, 176, 512, 80, 12, None, True),
    ("ties", 2, 40, 36, 64, 24, 3, 4, True),
    ("flat", 1, 50, 44, 96, 80, 0, None, False),
    ("wide", 3, 64, 56, 128, 130, 6, 8, False),
]


def main():
    from oracle import lc_policy_oracle as O
    policy_mod, utils_mod = load_reference()
    out = {}
    names = []
    for ci, (name, K, Y, X, W, N, nb, quant, with_mask) in enumerate(CASES):
        rng = np.random.default_rng(4100 + ci)
        cam, tr = O.device_geometry(width=W, height=W, fov=80.0, laser_y=-0.2)
        dev = types.SimpleNamespace(CAMERA_PARAMS=cam, TRANSFORMS=tr)
        cen = anchor_grid(Y, X, K)
        anchors = full_anchors(cen, K)
        cls = blobs(rng, K, Y, X, nb, quant)
        mask = None
        if with_mask:
            m = rng.uniform(size=(K, Y, X)) < 0.9
            m[:, : Y // 8] = rng.uniform(size=(K, Y // 8, X)) < 0.3
            mask = m.reshape(-1).astype(np.uint8)
        cmap = utils_mod._get_confidence_map(anchors, mask, cls.copy(), tr["wTc"])
        pol = policy_mod.DPOptimizedPolicy(dev, pts_per_cam_ray=N)
        dpts = pol.get_design_points(cmap)
        g = pol._graph
        nodes = np.stack(g["nodes"])
        drow = np.stack([e["Δθ_las"][0, :] for e in g["edges"]])
        dcol = np.stack([e["Δθ_las"][:, 0] for e in g["edges"]])
        names.append(name)
        p = name + "/"
        out[p + "centres"] = cen
        out[p + "K"] = np.int32(K)
        out[p + "cls"] = cls
        out[p + "mask"] = mask if mask is not None else np.zeros(0, dtype=np.uint8)
        out[p + "cam_matrix"] = cam["matrix"]
        out[p + "cam_wh"] = np.array([W, W], dtype=np.int32)
        out[p + "cTl"] = tr["cTl"]
        out[p + "wTc"] = tr["wTc"]
        out[p + "pts_per_cam_ray"] = np.int32(N)
        out[p + "max_dtheta"] = np.float64(pol._max_Δθ_las)
        out[p + "confidence_map"] = np.ascontiguousarray(cmap)
        out[p + "nodes"] = nodes
        out[p + "dtheta_row0"] = drow
        out[p + "dtheta_col0"] = dcol
        out[p + "design_points"] = np.ascontiguousarray(dpts)
        # the other registered policies (policy.py:437-684) on the same map and device
        for pname in ("FixedParallel15", "FixedParallel30", "FixedParallel45", "MaxEntropy",
                      "ParallelMaxEntropy", "RandomParallel", "GreedyDPTiesMinVel",
                      "GreedyDPTiesRandom"):
            cls_ = policy_mod.get_lc_policy_class(pname)
            pol2 = cls_(dev, pts_per_cam_ray=N) if pname.startswith("GreedyDP") else cls_(dev)
            np.random.seed(77 + ci)
            out[p + "policy/" + pname] = np.ascontiguousarray(pol2.get_design_points(cmap))
        print(name, "cmap", cmap.shape, cmap.dtype, "nodes", nodes.shape, "design", dpts.shape,
              dpts.dtype, "distinct z:", len(np.unique(np.round(dpts[:, 1], 3))))
    out["cases"] = np.array(names)
    path = os.path.join(HERE, "lc_policy_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
