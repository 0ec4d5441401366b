"""spconv.SparseConvTensor (as constructed at second/pytorch/models/middle.py:98-99,199-200,
404-405; `.features` read/written at second/pytorch/models/resnet.py:54-64; `.dense()` at
middle.py:206)."""
import numpy as np
import torch

from .. import ops


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        """
        Args:
            features: [num_active, C] float tensor
            indices: [num_active, 4] int32 tensor, (batch, z, y, x)
            spatial_shape: 3 ints (z, y, x); list, tuple or numpy array
            batch_size: int
            grid: accepted for API compatibility (spconv's pre-allocated dense index grid);
                the hash-based rulebook builder needs none
        """
        self.features = features
        self.indices = indices
        if isinstance(spatial_shape, np.ndarray):
            spatial_shape = spatial_shape.tolist()
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = {}
        self.grid = grid

    @property
    def spatial_size(self):
        return int(np.prod(self.spatial_shape))

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key)

    def dense(self, channels_first=True):
        res = ops.dense(self.features, self.indices.int(), self.batch_size, self.spatial_shape)
        if not channels_first:
            return res.permute(0, 2, 3, 4, 1).contiguous()
        return res

    @property
    def sparity(self):
        return self.indices.shape[0] / np.prod(self.spatial_shape) / self.batch_size
