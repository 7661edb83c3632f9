"""Container-only helper: make the read-only reference (/root/reference) importable.

Used ONLY by tools/gen_golden*.py to capture golden vectors.
It never runs on the GPU box (the reference does not exist there) and nothing
under isaacgymloco_amd/ imports it.

What it stubs (all import-only dependencies of the reference that are absent
from this image):
  * isaacgym            -> tools/refstub/isaacgym (no physics, gym calls are no-ops)
  * torch.utils.tensorboard.SummaryWriter -> dummy class
  * pybullet_utils.transformations        -> empty module (import-only use, ML:8)
  * ruamel.yaml                            -> empty module (import-only use, HYBR:49)
  * np.int                                -> int (removed numpy alias used at ML:210, ML:234)
"""
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"
_HERE = os.path.dirname(os.path.abspath(__file__))


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "legged_gym"))


def install():
    if not reference_available():
        raise RuntimeError("reference tree not present; golden vectors can only be regenerated in the build container")
    for p in (os.path.join(_HERE, "refstub"),
              os.path.join(REFERENCE_ROOT, "legged_gym"),
              os.path.join(REFERENCE_ROOT, "rsl_rl")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import numpy as np
    if not hasattr(np, "int"):
        np.int = int  # noqa
    import torch.utils  # noqa
    if "torch.utils.tensorboard" not in sys.modules:
        tb = types.ModuleType("torch.utils.tensorboard")

        class SummaryWriter:  # minimal dummy
            def __init__(self, *a, **k):
                pass

            def add_scalar(self, *a, **k):
                pass
        tb.SummaryWriter = SummaryWriter
        sys.modules["torch.utils.tensorboard"] = tb
    if "ruamel" not in sys.modules:
        ru = types.ModuleType("ruamel")
        ry = types.ModuleType("ruamel.yaml")
        ru.yaml = ry
        sys.modules["ruamel"] = ru
        sys.modules["ruamel.yaml"] = ry
    if "pybullet_utils" not in sys.modules:
        pu = types.ModuleType("pybullet_utils")
        tr = types.ModuleType("pybullet_utils.transformations")
        pu.transformations = tr
        sys.modules["pybullet_utils"] = pu
        sys.modules["pybullet_utils.transformations"] = tr
