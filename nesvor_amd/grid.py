"""Host-side description of the multi-resolution hash grid (level table).

The level table is what ``tinycudann``'s GridEncoding derives from its config
(call site nesvor/nesvor/models.py:102-111): per level the fp32 grid scale, the
vertex resolution, the number of table entries and the offset into the flat
parameter tensor.  Computed once on the host in fp32 and handed to the kernels
by value (nesvor_grid_t).
"""
from dataclasses import dataclass
from typing import List

import numpy as np

from . import _lib


@dataclass
class GridLevel:
    scale: float
    res: int
    size: int
    offset: int
    hashed: bool


class HashGridSpec:
    def __init__(self, n_levels: int, n_features_per_level: int, log2_hashmap_size: int,
                 base_resolution: int, per_level_scale: float) -> None:
        if not 1 <= n_levels <= _lib.MAX_LEVELS:
            raise ValueError(f"n_levels must be in [1, {_lib.MAX_LEVELS}]")
        if n_features_per_level not in (1, 2, 4, 8):
            raise ValueError("n_features_per_level must be 1, 2, 4 or 8")
        self.n_levels = n_levels
        self.n_features = n_features_per_level
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.per_level_scale = per_level_scale
        self.levels: List[GridLevel] = []
        f32 = np.float32
        log2_scale = np.log2(f32(per_level_scale), dtype=f32)
        offset = 0
        cap = 1 << log2_hashmap_size
        for lvl in range(n_levels):
            scale = f32(np.exp2(f32(lvl) * log2_scale, dtype=f32) * f32(base_resolution) - f32(1.0))
            res = int(np.ceil(scale)) + 1
            dense = min(res**3, 2**31 - 1)
            size = min(-(-dense // 8) * 8, cap)  # 8-entry aligned, capped by the hash-map size
            self.levels.append(GridLevel(float(scale), res, size, offset, res**3 > size))
            offset += size
        self.n_entries = offset
        self.n_params = offset * n_features_per_level
        self.n_output_dims = n_levels * n_features_per_level
        g = _lib.GridT()
        g.n_levels, g.n_features = n_levels, n_features_per_level
        for i, lv in enumerate(self.levels):
            g.scale[i], g.res[i], g.size[i], g.offset[i], g.hashed[i] = lv.scale, lv.res, lv.size, lv.offset, int(lv.hashed)
        self.c_struct = g
