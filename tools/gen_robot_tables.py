"""Container-only: write isaacgymloco_amd/robots/tables/<robot>.json (the collapsed 17-body table + collision primitives, plain numbers)
from the quadruped URDFs of the reference checkout, through robots/urdf.py.  Only derived data is written, no URDF text."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaacgymloco_amd.robots import urdf  # noqa: E402

SRC = "/root/reference/legged_gym/resources/robots"
OUT = os.path.join(ROOT, "isaacgymloco_amd", "robots", "tables")
os.makedirs(OUT, exist_ok=True)
for name in ("go1", "a1", "aliengo"):
    bodies, limits = urdf.parse(os.path.join(SRC, name, "urdf", name + ".urdf"))
    with open(os.path.join(OUT, name + ".json"), "w") as f:
        json.dump(urdf.table_to_json(bodies, limits), f, indent=0)
    print(name, "mass", sum(b["mass"] for b in bodies), "bodies", len(bodies))
