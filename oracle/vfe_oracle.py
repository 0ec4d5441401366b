"""numpy restatement of the reference VFE (second/pytorch/models/voxel_encoder.py) -- TEST
INFRASTRUCTURE ONLY.  Pinned by tests/golden/vfe_golden.npz, which tests/golden/
make_vfe_golden.py generated from the real reference file (parity PINNED for the VFE).

`params` maps the reference's state_dict keys to arrays ("vfe1.linear.weight", ...).
"""
import numpy as np


def paddings_indicator(actual_num, max_num):
    """voxel_encoder.py:27-48: mask[k,t] = actual_num[k] > t."""
    return actual_num.astype(np.int32)[:, None] > np.arange(max_num, dtype=np.int32)[None, :]


def _linear(x, params, prefix):
    y = x @ params[prefix + ".weight"].T
    if prefix + ".bias" in params:
        y = y + params[prefix + ".bias"]
    return y


def _norm(x, params, prefix, train):
    """BatchNorm1d(eps=1e-3) over the channel (last) dim of [K,T,C]; identity if absent
    (use_norm=False -> torchplus Empty).  train=True uses biased batch statistics over K*T."""
    if prefix + ".weight" not in params:
        return x
    if train:
        flat = x.reshape(-1, x.shape[-1]).astype(np.float64)
        mean, var = flat.mean(0), flat.var(0)
    else:
        mean, var = params[prefix + ".running_mean"], params[prefix + ".running_var"]
    y = (x - mean) / np.sqrt(var + 1e-3)
    return (y * params[prefix + ".weight"] + params[prefix + ".bias"]).astype(np.float32)


def vfe_layer(x, params, prefix, train=False):
    """voxel_encoder.py:65-80."""
    p = np.maximum(_norm(_linear(x, params, prefix + "linear"), params, prefix + "norm", train), 0)
    agg = p.max(axis=1, keepdims=True)
    return np.concatenate([p, np.repeat(agg, x.shape[1], axis=1)], axis=2)


def simple_voxel(voxels, num, c=4):
    """voxel_encoder.py:220-225."""
    return voxels[:, :, :c].sum(axis=1) / num.astype(voxels.dtype)[:, None]


def simple_voxel_radius(voxels, num, c=4):
    """voxel_encoder.py:246-255."""
    m = simple_voxel(voxels, num, c)
    r = np.sqrt((m[:, :2] ** 2).sum(axis=1, keepdims=True))
    return np.concatenate([r, m[:, 2:c]], axis=1)


def voxel_feature_extractor(voxels, num, params, layer_prefixes=("vfe1.", "vfe2."),
                            with_distance=False, train=False):
    """voxel_encoder.py:113-140 (V2, :179-204, with layer_prefixes = vfe_layers.0. ...)."""
    mean = voxels[:, :, :3].sum(axis=1, keepdims=True) / num.astype(voxels.dtype)[:, None, None]
    rel = voxels[:, :, :3] - mean
    feats = [voxels, rel]
    if with_distance:
        feats.append(np.sqrt((voxels[:, :, :3] ** 2).sum(axis=2, keepdims=True)))
    x = np.concatenate(feats, axis=-1)
    mask = paddings_indicator(num, voxels.shape[1]).astype(voxels.dtype)[:, :, None]
    for pre in layer_prefixes:
        x = vfe_layer(x, params, pre, train) * mask
    x = np.maximum(_norm(_linear(x, params, "linear"), params, "norm", train), 0) * mask
    return x.max(axis=1)
