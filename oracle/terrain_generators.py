"""Second, independent restatement of the four third-party terrain generators the reference calls (TER:173-193):
`isaacgym.terrain_utils.random_uniform_terrain / pyramid_sloped_terrain / discrete_obstacles_terrain / stepping_stones_terrain`.

TEST INFRASTRUCTURE (like everything under oracle/): only tests/ may import this file.  The product's generators are
isaacgymloco_amd/envs/terrain.py.

Provenance.  `isaacgym` (NVIDIA Isaac Gym Preview 3/4, python/isaacgym/terrain_utils.py) is a closed third-party dependency that is not
vendored in /root/reference (legged_gym/setup.py lists `isaacgym` without a version), so there is no reference source to run or to
generate golden vectors from: **parity of these four generators is pinned to the published definition, not to reference outputs**.
This file writes that published definition down a second time, deliberately close to its published form -- physical-coordinate
interpolation through SciPy's FITPACK bilinear spline (the published code uses scipy.interpolate.interp2d(kind="linear"), removed in
SciPy 1.14; RectBivariateSpline(kx=1, ky=1) is SciPy's documented replacement on a regular grid), np.meshgrid / reshape for the pyramid,
Python ranges for the obstacle draws, both orientations of the stepping-stone sweep -- whereas envs/terrain.py computes the same
quantities with its own index arithmetic (hand-written bilinear weights on index coordinates, broadcasting, one sweep orientation).
tests/test_terrain_generators.py requires the two to produce bit-identical int16 grids for the same draws.

The published functions draw from numpy's global generator; here every function takes the numpy RandomState to draw from (`rng`),
and issues the draws through the same API calls in the same order, so that a test can give both implementations identical streams.
"""
import numpy as np
from scipy import interpolate


class SubTerrain:
    """published: class SubTerrain(terrain_name, width, length, vertical_scale, horizontal_scale) with an int16 height_field_raw"""

    def __init__(self, terrain_name="terrain", width=256, length=256, vertical_scale=1.0, horizontal_scale=1.0):
        self.terrain_name = terrain_name
        self.vertical_scale = vertical_scale
        self.horizontal_scale = horizontal_scale
        self.width = width
        self.length = length
        self.height_field_raw = np.zeros((self.width, self.length), dtype=np.int16)


def random_uniform_terrain(terrain, rng, min_height, max_height, step=1, downsampled_scale=None):
    """uniform noise on a coarse grid, bilinearly up-sampled to the height-field resolution, rounded, ADDED to the field"""
    if downsampled_scale is None:
        downsampled_scale = terrain.horizontal_scale
    min_height = int(min_height / terrain.vertical_scale)
    max_height = int(max_height / terrain.vertical_scale)
    step = int(step / terrain.vertical_scale)
    heights_range = np.arange(min_height, max_height + step, step)
    height_field_downsampled = rng.choice(heights_range, (int(terrain.width * terrain.horizontal_scale / downsampled_scale),
                                                          int(terrain.length * terrain.horizontal_scale / downsampled_scale)))
    x = np.linspace(0, terrain.width * terrain.horizontal_scale, height_field_downsampled.shape[0])
    y = np.linspace(0, terrain.length * terrain.horizontal_scale, height_field_downsampled.shape[1])
    f = interpolate.RectBivariateSpline(x, y, height_field_downsampled.astype(np.float64), kx=1, ky=1)    # interp2d(y, x, z, kind="linear")
    x_upsampled = np.linspace(0, terrain.width * terrain.horizontal_scale, terrain.width)
    y_upsampled = np.linspace(0, terrain.length * terrain.horizontal_scale, terrain.length)
    z_upsampled = np.rint(f(x_upsampled, y_upsampled))
    terrain.height_field_raw += z_upsampled.astype(np.int16)
    return terrain


def pyramid_sloped_terrain(terrain, slope=1, platform_size=1.0):
    """pyramid: product of two tent functions times the peak height, ADDED; then clipped at the height found at the platform's corner"""
    x = np.arange(0, terrain.width)
    y = np.arange(0, terrain.length)
    center_x = int(terrain.width / 2)
    center_y = int(terrain.length / 2)
    xx, yy = np.meshgrid(x, y, sparse=True)
    xx = (center_x - np.abs(center_x - xx)) / center_x
    yy = (center_y - np.abs(center_y - yy)) / center_y
    xx = xx.reshape(terrain.width, 1)
    yy = yy.reshape(1, terrain.length)
    max_height = int(slope * (terrain.horizontal_scale / terrain.vertical_scale) * (terrain.width / 2))
    terrain.height_field_raw += (max_height * xx * yy).astype(terrain.height_field_raw.dtype)
    platform_size = int(platform_size / terrain.horizontal_scale / 2)
    x1 = terrain.width // 2 - platform_size
    y1 = terrain.length // 2 - platform_size
    min_h = min(terrain.height_field_raw[x1, y1], 0)
    max_h = max(terrain.height_field_raw[x1, y1], 0)
    terrain.height_field_raw = np.clip(terrain.height_field_raw, min_h, max_h)
    return terrain


def discrete_obstacles_terrain(terrain, rng, max_height, min_size, max_size, num_rects, platform_size=1.0):
    """num_rects axis-aligned boxes of four possible heights, positions and sizes on a 4-cell raster, later boxes overwrite earlier ones"""
    max_height = int(max_height / terrain.vertical_scale)
    min_size = int(min_size / terrain.horizontal_scale)
    max_size = int(max_size / terrain.horizontal_scale)
    platform_size = int(platform_size / terrain.horizontal_scale)
    (i, j) = terrain.height_field_raw.shape
    height_range = [-max_height, -max_height // 2, max_height // 2, max_height]
    width_range = range(min_size, max_size, 4)
    length_range = range(min_size, max_size, 4)
    for _ in range(num_rects):
        width = rng.choice(width_range)
        length = rng.choice(length_range)
        start_i = rng.choice(range(0, i - width, 4))
        start_j = rng.choice(range(0, j - length, 4))
        terrain.height_field_raw[start_i:start_i + width, start_j:start_j + length] = rng.choice(height_range)
    x1 = (terrain.width - platform_size) // 2
    x2 = (terrain.width + platform_size) // 2
    y1 = (terrain.length - platform_size) // 2
    y2 = (terrain.length + platform_size) // 2
    terrain.height_field_raw[x1:x2, y1:y2] = 0
    return terrain


def stepping_stones_terrain(terrain, rng, stone_size, stone_distance, max_height, platform_size=1.0, depth=-10):
    """square stones of random height separated by holes of `depth`; rows of stones start at a random offset"""
    stone_size = int(stone_size / terrain.horizontal_scale)
    stone_distance = int(stone_distance / terrain.horizontal_scale)
    max_height = int(max_height / terrain.vertical_scale)
    platform_size = int(platform_size / terrain.horizontal_scale)
    height_range = np.arange(-max_height - 1, max_height, step=1)
    start_x = 0
    start_y = 0
    terrain.height_field_raw[:, :] = int(depth / terrain.vertical_scale)
    if terrain.length >= terrain.width:
        while start_y < terrain.length:
            stop_y = min(terrain.length, start_y + stone_size)
            start_x = rng.randint(0, stone_size)
            stop_x = max(0, start_x - stone_distance)       # fill the first hole
            terrain.height_field_raw[0:stop_x, start_y:stop_y] = rng.choice(height_range)
            while start_x < terrain.width:                   # fill the row
                stop_x = min(terrain.width, start_x + stone_size)
                terrain.height_field_raw[start_x:stop_x, start_y:stop_y] = rng.choice(height_range)
                start_x += stone_size + stone_distance
            start_y += stone_size + stone_distance
    elif terrain.width > terrain.length:
        while start_x < terrain.width:
            stop_x = min(terrain.width, start_x + stone_size)
            start_y = rng.randint(0, stone_size)
            stop_y = max(0, start_y - stone_distance)
            terrain.height_field_raw[start_x:stop_x, 0:stop_y] = rng.choice(height_range)
            while start_y < terrain.length:
                stop_y = min(terrain.length, start_y + stone_size)
                terrain.height_field_raw[start_x:stop_x, start_y:stop_y] = rng.choice(height_range)
                start_y += stone_size + stone_distance
            start_x += stone_size + stone_distance
    x1 = (terrain.width - platform_size) // 2
    x2 = (terrain.width + platform_size) // 2
    y1 = (terrain.length - platform_size) // 2
    y2 = (terrain.length + platform_size) // 2
    terrain.height_field_raw[x1:x2, y1:y2] = 0
    return terrain
