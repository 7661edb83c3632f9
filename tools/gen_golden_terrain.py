"""Container-only: pin the terrain generator against the reference's own `Terrain` class.

`legged_gym.utils.terrain.Terrain` (TER:38-227: curiculum / randomized_terrain / make_terrain / add_terrain_to_map and the in-tree
generators flat, pyramid_stairs, pit, gap, TER:229-294) is imported and RUN here with terrain_proportions restricted to the
generators that live in the reference tree (the rough / slope / obstacle / stepping-stone generators are third-party
`isaacgym.terrain_utils` code that is not in the tree; they stay unpinned).  mesh_type is set to "heightfield" for the run so that
the constructor does not call the third-party trimesh conversion (TER:72-75); nothing else depends on it.

Output: tests/golden/terrain_<name>.npz = {the config scalars used, height_field_raw int16, env_origins float64, tot_rows, tot_cols,
border, in_terrain_range probes}.  Only data is written -- no reference source text.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import refenv  # noqa: E402

refenv.install()
import legged_gym.envs  # noqa: E402,F401
from legged_gym.utils.terrain import Terrain as RefTerrain  # noqa: E402
from legged_gym.envs.aliengo import aliengo_config, aliengo_stairs_config  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")

# flat | (rough, slope, rough slope: absent) | stairs up | stairs down | (obstacles, stones: absent) | pit | gap
IN_TREE = [0.10, 0.0, 0.0, 0.0, 0.25, 0.25, 0.0, 0.0, 0.20, 0.20]
CASES = {
    # name: (reference config class, proportions, curriculum, numpy seed for the randomized layout)
    "stairs_geometry_curriculum": (aliengo_stairs_config.AlienGoStairsCfg, IN_TREE, True, None),
    "flat_geometry_curriculum": (aliengo_config.AlienGoRoughCfg, IN_TREE, True, None),
    "stairs_geometry_randomized": (aliengo_stairs_config.AlienGoStairsCfg, IN_TREE, False, 1),
    # the reference's own aliengo_stairs proportions hit in-tree generators only in columns 4..15 (stairs up / down); columns of the
    # third-party generators are masked out of the comparison (mask stored in the fixture)
}


def run(name, cfg_cls, proportions, curriculum, seed):
    tc = cfg_cls.terrain()
    tc.terrain_proportions = list(proportions)
    tc.curriculum = curriculum
    tc.selected = False
    tc.mesh_type = "heightfield"
    if seed is not None:
        np.random.seed(seed)           # legged_gym.utils.helpers.set_seed does exactly this before the env is built (HLP:75-84)
    t = RefTerrain(tc, 64)
    probes = np.array([[0.0, 0.0, 0.3], [-0.01, 5.0, 0.3], [5.0, -0.01, 0.3], [t.xSize + tc.border_size / 2 - 1e-3, 1.0, 0.0],
                       [t.xSize + tc.border_size / 2, 1.0, 0.0], [1.0, t.ySize + tc.border_size / 2 - 1e-3, 0.0],
                       [1.0, t.ySize + tc.border_size / 2, 0.0], [50.0, 100.0, -3.0]], dtype=np.float32)
    inside = t.in_terrain_range(torch.from_numpy(probes)).numpy()
    out = dict(
        terrain_proportions=np.array(proportions), curriculum=np.array(curriculum), np_seed=np.array(-1 if seed is None else seed),
        terrain_length=np.array(tc.terrain_length), terrain_width=np.array(tc.terrain_width), num_rows=np.array(tc.num_rows),
        num_cols=np.array(tc.num_cols), horizontal_scale=np.array(tc.horizontal_scale), vertical_scale=np.array(tc.vertical_scale),
        border_size=np.array(tc.border_size),
        height_field_raw=np.asarray(t.height_field_raw), env_origins=np.asarray(t.env_origins), tot_rows=np.array(t.tot_rows),
        tot_cols=np.array(t.tot_cols), border=np.array(t.border), probes=probes, probes_inside=inside)
    assert out["height_field_raw"].dtype == np.int16
    np.savez_compressed(os.path.join(GOLDEN, f"terrain_{name}.npz"), **out)
    h = out["height_field_raw"]
    print(f"wrote terrain_{name}.npz: grid {h.shape}, heights [{h.min()}, {h.max()}], {len(np.unique(h))} distinct, origins z max {t.env_origins[..., 2].max():.3f}")


if __name__ == "__main__":
    for name, (cls, prop, cur, seed) in CASES.items():
        run(name, cls, prop, cur, seed)
