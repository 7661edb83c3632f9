"""Terrain height-grid builder (init-time data producer for the simulator).

Same public surface as legged_gym.utils.terrain.Terrain (TER:38-227): `Terrain(cfg.terrain, num_robots)`
exposes `heightsamples` (int16 [tot_rows, tot_cols]), `env_origins` ([num_rows, num_cols, 3]),
`tot_rows/tot_cols/border`, `env_length/env_width`, `in_terrain_range(pos)`.

Sub-terrain generators: flat / pyramid stairs (with border) / pit / gap restate the in-tree functions
(TER:229-294).  random-uniform, pyramid slope, discrete obstacles and stepping stones restate the
published algorithms of the third-party `isaacgym.terrain_utils` (not in the reference tree, SURVEY.md
8c: parity unpinned; numpy's bilinear interpolation replaces the removed scipy `interp2d`).
The triangle-mesh conversion of the reference (TER:72-75) is not needed: the simulator collides
against the grid directly (DESIGN.md "Terrain contact").
"""
import numpy as np


class SubTerrain:
    def __init__(self, width, length, vertical_scale, horizontal_scale):
        self.width = width
        self.length = length
        self.vertical_scale = vertical_scale
        self.horizontal_scale = horizontal_scale
        self.height_field_raw = np.zeros((width, length), dtype=np.int16)


# ---- generators restating TER:229-294 -------------------------------------------------------------
def flat_terrain(t):
    t.height_field_raw = np.zeros((t.width, t.length), dtype=np.int16)


def pyramid_stairs_terrain(t, step_width, step_height, platform_size=1.0, border_width=0.0):
    sw = round(step_width / t.horizontal_scale)
    sh = round(step_height / t.vertical_scale)
    plat = round(platform_size / t.horizontal_scale)
    bw = round(border_width / t.horizontal_scale)
    lo_x, hi_x, lo_y, hi_y = bw, t.width - bw, bw, t.length - bw
    level = 0
    while (hi_x - lo_x) > plat and (hi_y - lo_y) > plat:
        lo_x, hi_x, lo_y, hi_y = lo_x + sw, hi_x - sw, lo_y + sw, hi_y - sw
        level += sh
        t.height_field_raw[lo_x:hi_x, lo_y:hi_y] = level
    if bw > 0:
        t.height_field_raw[:bw, :] = 0
        t.height_field_raw[-bw:, :] = 0
        t.height_field_raw[:, :bw] = 0
        t.height_field_raw[:, -bw:] = 0


def pit_terrain(t, depth, platform_size=1.0):
    d = int(depth / t.vertical_scale)
    half = int(platform_size / t.horizontal_scale / 2)
    cx, cy = t.length // 2, t.width // 2
    t.height_field_raw[cx - half:cx + half, cy - half:cy + half] = -d


def gap_terrain(t, gap_size, platform_size=1.0):
    g = int(gap_size / t.horizontal_scale)
    plat = int(platform_size / t.horizontal_scale)
    cx, cy = t.length // 2, t.width // 2
    x1 = (t.length - plat) // 2
    y1 = (t.width - plat) // 2
    x2, y2 = x1 + g, y1 + g
    t.height_field_raw[cx - x2:cx + x2, cy - y2:cy + y2] = -1000
    t.height_field_raw[cx - x1:cx + x1, cy - y1:cy + y1] = 0


# ---- generators restating isaacgym.terrain_utils (published algorithms) -----------------------------
def _bilinear_upsample(coarse, out_w, out_l):
    cw, cl = coarse.shape
    xs = np.linspace(0, cw - 1, out_w)
    ys = np.linspace(0, cl - 1, out_l)
    x0 = np.clip(np.floor(xs).astype(int), 0, cw - 2) if cw > 1 else np.zeros(out_w, int)
    y0 = np.clip(np.floor(ys).astype(int), 0, cl - 2) if cl > 1 else np.zeros(out_l, int)
    fx = (xs - x0)[:, None]
    fy = (ys - y0)[None, :]
    x1 = np.minimum(x0 + 1, cw - 1)
    y1 = np.minimum(y0 + 1, cl - 1)
    c = coarse.astype(np.float64)
    return (c[x0][:, y0] * (1 - fx) * (1 - fy) + c[x1][:, y0] * fx * (1 - fy)
            + c[x0][:, y1] * (1 - fx) * fy + c[x1][:, y1] * fx * fy)


def random_uniform_terrain(t, rng, min_height, max_height, step=1.0, downsampled_scale=None):
    ds = t.horizontal_scale if downsampled_scale is None else downsampled_scale
    lo = int(min_height / t.vertical_scale)
    hi = int(max_height / t.vertical_scale)
    st = max(int(step / t.vertical_scale), 1)
    levels = np.arange(lo, hi + st, st)
    coarse = rng.choice(levels, (int(t.width * t.horizontal_scale / ds), int(t.length * t.horizontal_scale / ds)))
    t.height_field_raw = t.height_field_raw + np.rint(_bilinear_upsample(coarse, t.width, t.length)).astype(np.int16)


def pyramid_sloped_terrain(t, slope=1.0, platform_size=1.0):
    cx, cy = int(t.width / 2), int(t.length / 2)
    xx = ((cx - np.abs(cx - np.arange(t.width))) / cx).reshape(t.width, 1)
    yy = ((cy - np.abs(cy - np.arange(t.length))) / cy).reshape(1, t.length)
    peak = int(slope * (t.horizontal_scale / t.vertical_scale) * (t.width / 2))
    t.height_field_raw = t.height_field_raw + (peak * xx * yy).astype(np.int16)
    half = int(platform_size / t.horizontal_scale / 2)
    x1, y1 = t.width // 2 - half, t.length // 2 - half
    ref = t.height_field_raw[x1, y1]
    t.height_field_raw = np.clip(t.height_field_raw, min(ref, 0), max(ref, 0)).astype(np.int16)


def discrete_obstacles_terrain(t, rng, max_height, min_size, max_size, num_rects, platform_size=1.0):
    mh = int(max_height / t.vertical_scale)
    smin = int(min_size / t.horizontal_scale)
    smax = int(max_size / t.horizontal_scale)
    plat = int(platform_size / t.horizontal_scale)
    rows, cols = t.height_field_raw.shape
    heights = [-mh, -mh // 2, mh // 2, mh]
    sizes = list(range(smin, smax, 4))
    for _ in range(num_rects):
        w = rng.choice(sizes)
        l = rng.choice(sizes)
        i0 = rng.choice(list(range(0, rows - w, 4)))
        j0 = rng.choice(list(range(0, cols - l, 4)))
        t.height_field_raw[i0:i0 + w, j0:j0 + l] = rng.choice(heights)
    x1, x2 = (t.width - plat) // 2, (t.width + plat) // 2
    y1, y2 = (t.length - plat) // 2, (t.length + plat) // 2
    t.height_field_raw[x1:x2, y1:y2] = 0


def stepping_stones_terrain(t, rng, stone_size, stone_distance, max_height, platform_size=1.0, depth=-10):
    ss = int(stone_size / t.horizontal_scale)
    sd = int(stone_distance / t.horizontal_scale)      # 0 for the reference's `stone_distance=0.05` at difficulty 0 (TER:189): stones touch
    if ss < 1:
        raise ValueError("stone_size below one cell")    # (the published code fails in randint(0, 0) here)
    mh = int(max_height / t.vertical_scale)
    plat = int(platform_size / t.horizontal_scale)
    hr = np.arange(-mh - 1, mh, 1)
    t.height_field_raw[:, :] = int(depth / t.vertical_scale)
    y = 0
    while y < t.length:
        stop_y = min(t.length, y + ss)
        x = int(rng.randint(0, ss))
        t.height_field_raw[0:max(0, x - sd), y:stop_y] = rng.choice(hr)
        while x < t.width:
            stop_x = min(t.width, x + ss)
            t.height_field_raw[x:stop_x, y:stop_y] = rng.choice(hr)
            x += ss + sd
        y += ss + sd
    x1, x2 = (t.width - plat) // 2, (t.width + plat) // 2
    y1, y2 = (t.length - plat) // 2, (t.length + plat) // 2
    t.height_field_raw[x1:x2, y1:y2] = 0


class Terrain:
    """Curriculum grid of sub-terrains (TER:38-227)."""

    def __init__(self, cfg, num_robots, seed=1):
        self.cfg = cfg
        self.num_robots = num_robots
        self.type = cfg.mesh_type
        self.rng = np.random.RandomState(seed)
        if self.type in ("none", "plane"):
            return
        self.env_length = cfg.terrain_length
        self.env_width = cfg.terrain_width
        self.xSize = cfg.terrain_length * cfg.num_rows
        self.ySize = cfg.terrain_width * cfg.num_cols
        self.proportions = [float(np.sum(cfg.terrain_proportions[:i + 1])) for i in range(len(cfg.terrain_proportions))]
        self.proportions += [self.proportions[-1]] * (10 - len(self.proportions))
        self.env_origins = np.zeros((cfg.num_rows, cfg.num_cols, 3))
        self.width_per_env_pixels = int(self.env_width / cfg.horizontal_scale)
        self.length_per_env_pixels = int(self.env_length / cfg.horizontal_scale)
        self.border = int(cfg.border_size / cfg.horizontal_scale)
        self.tot_cols = int(cfg.num_cols * self.width_per_env_pixels) + 2 * self.border
        self.tot_rows = int(cfg.num_rows * self.length_per_env_pixels) + 2 * self.border
        self.height_field_raw = np.zeros((self.tot_rows, self.tot_cols), dtype=np.int16)
        if cfg.curriculum:
            self._curriculum()
        else:
            self._randomized()
        self.heightsamples = self.height_field_raw

    # TER:87-95
    def _curriculum(self):
        c = self.cfg
        for j in range(c.num_cols):
            for i in range(c.num_rows):
                self._add(self.make_terrain(j / c.num_cols + 0.001, i / c.num_rows), i, j)

    # TER:76-85
    def _randomized(self):
        c = self.cfg
        for k in range(c.num_rows * c.num_cols):
            i, j = np.unravel_index(k, (c.num_rows, c.num_cols))
            self._add(self.make_terrain(self.rng.uniform(0, 1), self.rng.choice([0.5, 0.7, 0.8])), i, j)

    # TER:111-199
    def make_terrain(self, choice, difficulty):
        c = self.cfg
        t = SubTerrain(self.width_per_env_pixels, self.width_per_env_pixels, c.vertical_scale, c.horizontal_scale)
        slope = min(difficulty * 0.5, 0.4)
        amplitude = min(0.02 + 0.1 * difficulty, 0.06)
        if difficulty < 0.5:
            step_height = 0.06 + 0.2 * difficulty
        elif difficulty in (0.5, 0.6):
            step_height = 0.16
        else:
            step_height = 0.16 + 0.15 * (difficulty - 0.6)
        p = self.proportions
        if choice < p[0]:
            flat_terrain(t)
        elif choice < p[1]:
            random_uniform_terrain(t, self.rng, -amplitude, amplitude, step=0.005, downsampled_scale=0.2)
        elif choice < p[2]:
            # (the reference's sign flip `choice < proportions[0] / 2` can never trigger here, TER:174-176)
            pyramid_sloped_terrain(t, slope=slope, platform_size=3.0)
        elif choice < p[3]:
            pyramid_sloped_terrain(t, slope=slope, platform_size=3.0)
            random_uniform_terrain(t, self.rng, -amplitude, amplitude, step=0.005, downsampled_scale=0.2)
        elif choice < p[5]:
            if choice < p[4]:
                step_height *= -1
            pyramid_stairs_terrain(t, step_width=0.30, step_height=step_height, platform_size=3.0, border_width=1.0)
        elif choice < p[6]:
            discrete_obstacles_terrain(t, self.rng, 0.06 + difficulty * 0.15, 1.0, 2.0, 20, platform_size=3.0)
        elif choice < p[7]:
            stepping_stones_terrain(t, self.rng, stone_size=1.5 * (1.05 - difficulty),
                                    stone_distance=0.05 if difficulty == 0 else 0.1, max_height=0.8, platform_size=4.0)
        elif choice < p[8]:
            pit_terrain(t, depth=min(0.5 * difficulty, 0.35), platform_size=4.0)
        else:
            gap_terrain(t, gap_size=1.0 * difficulty, platform_size=3.0)
        return t

    # TER:201-218
    def _add(self, t, row, col):
        x0 = self.border + row * self.length_per_env_pixels
        y0 = self.border + col * self.width_per_env_pixels
        self.height_field_raw[x0:x0 + self.length_per_env_pixels, y0:y0 + self.width_per_env_pixels] = t.height_field_raw
        x1 = int((self.env_length / 2.0 - 1) / t.horizontal_scale)
        x2 = int((self.env_length / 2.0 + 1) / t.horizontal_scale)
        y1 = int((self.env_width / 2.0 - 1) / t.horizontal_scale)
        y2 = int((self.env_width / 2.0 + 1) / t.horizontal_scale)
        z = np.max(t.height_field_raw[x1:x2, y1:y2]) * t.vertical_scale
        self.env_origins[row, col] = [(row + 0.5) * self.env_length, (col + 0.5) * self.env_width, z]

    # TER:220-227
    def in_terrain_range(self, pos, device="cpu"):
        import torch
        hi = torch.tensor([self.xSize + self.cfg.border_size / 2, self.ySize + self.cfg.border_size / 2], device=device)
        return torch.logical_and(pos[..., :2] >= 0, pos[..., :2] < hi).all(dim=-1)
