"""Make this package answer to the module names the reference imports.

    import objectdetection_amd.dropin as dropin; dropin.install()
    import spconv                                                  # -> objectdetection_amd.spconv
    from second.core.point_cloud.point_cloud_ops import points_to_voxel

After ``install()`` the reference's ``second/pytorch/models/middle.py``, ``resnet.py``,
``second/builder/voxel_builder.py`` and the viewers import unmodified.
"""
import importlib
import sys
import types


def install(voxel_encoder=False, nms=True):
    """Register ``spconv`` (+ ``spconv.utils``, ``spconv.ops``) and
    ``second.core.point_cloud.point_cloud_ops`` in ``sys.modules``.

    nms=True (default) makes ``second.core.non_max_suppression.nms_gpu`` and ``.nms_cpu`` resolve
    to ``objectdetection_amd.nms``: the reference's modules are numba-CUDA / numba-jit code and
    cannot be imported on a ROCm machine (``nms_gpu``, ``rotate_nms_gpu``, ``rotate_iou_gpu``,
    ``rotate_iou_gpu_eval``, ``nms_gpu_cc``, ``nms_cc``, ``rotate_nms_cc``, ``nms_jit`` are
    provided; ``soft_nms_jit`` is not).

    voxel_encoder=True additionally makes ``second.pytorch.models.voxel_encoder`` resolve to the
    HIP-backed encoders (same class names and registry); leave False to keep the reference's
    own torch implementation, which also runs unmodified on the tensors this package produces.
    """
    from . import point_cloud_ops, spconv
    sys.modules["spconv"] = spconv
    sys.modules["spconv.utils"] = spconv.utils
    sys.modules["spconv.ops"] = spconv.ops
    sys.modules["spconv.conv"] = importlib.import_module(spconv.__name__ + ".conv")
    sys.modules["spconv.modules"] = importlib.import_module(spconv.__name__ + ".modules")

    # second.core.point_cloud.point_cloud_ops is absent from the reference tree
    pkg_name = "second.core.point_cloud"
    try:
        core = importlib.import_module("second.core")
    except Exception:  # the reference is not on sys.path: provide bare namespace packages
        core = None
        for name in ("second", "second.core"):
            if name not in sys.modules:
                m = types.ModuleType(name)
                m.__path__ = []
                sys.modules[name] = m
        sys.modules["second"].core = sys.modules["second.core"]
        core = sys.modules["second.core"]
    pc = types.ModuleType(pkg_name)
    pc.__path__ = []
    pc.point_cloud_ops = point_cloud_ops
    sys.modules[pkg_name] = pc
    sys.modules[pkg_name + ".point_cloud_ops"] = point_cloud_ops
    core.point_cloud = pc

    if nms:
        from . import nms as _nms
        pkg = types.ModuleType("second.core.non_max_suppression")
        pkg.__path__ = []
        for name in ("nms_gpu", "rotate_nms_gpu", "rotate_iou_gpu", "rotate_iou_gpu_eval",
                     "nms_gpu_cc", "nms_cc", "rotate_nms_cc", "nms_jit"):
            setattr(pkg, name, getattr(_nms, name))
        pkg.nms_gpu_module = pkg.nms_cpu_module = _nms
        sys.modules["second.core.non_max_suppression"] = pkg
        sys.modules["second.core.non_max_suppression.nms_gpu"] = _nms
        sys.modules["second.core.non_max_suppression.nms_cpu"] = _nms
        core.non_max_suppression = pkg

    if voxel_encoder:
        from . import voxel_encoder as ve
        sys.modules["second.pytorch.models.voxel_encoder"] = ve
    return spconv
