"""Container-only: dump the reference's config classes (helpers.class_to_dict, HLP:45) to
tests/golden/ref_cfg_<task>.json so tests/test_config.py can pin isaacgymloco_amd/envs/config.py."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refenv  # noqa: E402

refenv.install()
import legged_gym.envs  # noqa: E402,F401  (import order matters: envs first, avoids the circular import)
from legged_gym.utils.helpers import class_to_dict  # noqa: E402
from legged_gym.envs.aliengo import aliengo_config, aliengo_stairs_config, aliengo_amp_config, aliengo_recover_config  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def clean(d):
    if isinstance(d, dict):
        return {k: clean(v) for k, v in d.items() if k not in ("init_member_classes",)}
    if isinstance(d, (list, tuple)):
        return [clean(v) for v in d]
    if isinstance(d, (int, float, str, bool)) or d is None:
        return d
    return str(d)


for name, env, ppo in (("aliengo", aliengo_config.AlienGoRoughCfg, aliengo_config.AlienGoRoughCfgPPO),
                       ("aliengo_stairs", aliengo_stairs_config.AlienGoStairsCfg, aliengo_stairs_config.AlienGoStairsCfgPPO),
                       ("aliengo_amp", aliengo_amp_config.AlienGoRoughCfg, aliengo_amp_config.AlienGoRoughCfgPPO),
                       ("aliengo_recover", aliengo_recover_config.AlienGoRoughRecoverCfg, aliengo_recover_config.AlienGoRoughRecoverCfgPPO)):
    with open(os.path.join(OUT, f"ref_cfg_{name}.json"), "w") as f:
        json.dump({"env": clean(class_to_dict(env())), "ppo": clean(class_to_dict(ppo()))}, f, indent=1, sort_keys=True)
    print("wrote", name)
