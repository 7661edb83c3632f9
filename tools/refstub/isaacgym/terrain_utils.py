"""Only SubTerrain is provided (needed by the in-tree generators TER:229-294).
The third-party generators (random_uniform, pyramid_sloped, ...) are NOT
available as reference code; see isaacgymloco_amd/envs/terrain.py for the
build's own restatement of the published algorithms."""
import numpy as np


class SubTerrain:
    def __init__(self, terrain_name="terrain", width=256, length=256, vertical_scale=1.0, horizontal_scale=1.0):
        self.terrain_name = terrain_name
        self.vertical_scale = vertical_scale
        self.horizontal_scale = horizontal_scale
        self.width = width
        self.length = length
        self.height_field_raw = np.zeros((self.width, self.length), dtype=np.int16)
