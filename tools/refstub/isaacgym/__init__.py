"""In-container-only stand-in for the closed-source Isaac Gym Python package.

Purpose: lets tools/gen_golden.py import the *reference* pure-torch code
(/root/reference legged_gym + rsl_rl) so golden vectors can be captured.
Nothing here is shipped as a product dependency and nothing here simulates
physics: gym calls are no-ops, sim-state tensors are injected by the generator.
The helper formulas in torch_utils.py restate the published Isaac Gym
Preview 4 `isaacgym.torch_utils` (standard quaternion algebra).
"""
from . import gymapi, gymutil, gymtorch, terrain_utils, torch_utils  # noqa: F401
