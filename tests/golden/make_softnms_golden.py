"""Generates tests/golden/softnms_golden.npz by RUNNING the reference's own
second/core/non_max_suppression/nms_cpu.py:63-153 `soft_nms_jit` in the build container (numba is
absent: `numba.jit` is an identity decorator, so the function runs as the plain Python it is
written in; `spconv.utils`, which the file imports for two other functions, is the drop-in,
`second.core.box_np_ops` an empty stand-in).  The
fixture holds inputs and the reference's outputs only.

    python tests/golden/make_softnms_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def main():
    nb = types.ModuleType("numba")
    nb.jit = lambda *a, **k: (a[0] if len(a) == 1 and callable(a[0]) and not k else (lambda f: f))
    sys.modules["numba"] = nb
    import objectdetection_amd.dropin as dropin
    dropin.install()
    # second.core.box_np_ops (numba kernels, imported at the top of the file, unused by
    # soft_nms_jit): an empty stand-in, as in make_ref_goldens.py
    for name in ("second", "second.core", "second.core.box_np_ops"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["second.core"].box_np_ops = sys.modules["second.core.box_np_ops"]
    spec = importlib.util.spec_from_file_location(
        "ref_nms_cpu", os.path.join(REF, "second/core/non_max_suppression/nms_cpu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    rng = np.random.default_rng(0)
    case = 0
    for n in (1, 7, 60, 150):
        ctr = rng.uniform(0, 60, (n, 2))
        wh = rng.uniform(4, 25, (n, 2))
        boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2, rng.uniform(0.05, 1, (n, 1))],
                               1).astype(np.float32)
        if n > 5:
            boxes[3, 4] = boxes[1, 4]  # tied scores: the first maximum wins
        for method, thr in ((0, 0.001), (1, 0.001), (2, 0.001), (1, 0.2), (2, 0.3)):
            work = boxes.copy()
            keep = mod.soft_nms_jit(work, np.float32(0.5), np.float32(0.3), np.float32(thr),
                                    method)
            out[f"in_{case}"] = boxes
            out[f"arg_{case}"] = np.array([0.5, 0.3, thr, method], np.float64)
            out[f"boxes_{case}"] = work
            out[f"keep_{case}"] = np.array(keep, np.int64)
            case += 1
    out["cases"] = np.array(case)
    np.savez_compressed(os.path.join(HERE, "softnms_golden.npz"), **out)
    print("wrote", case, "cases")


if __name__ == "__main__":
    main()
