"""lsim_robot_model of a task's asset.  A module of its own, free of torch: the CPU oracle's timing process (oracle/cpu_bench.py) builds models
too, and a torch import there brings a second OpenMP runtime into the process (see its random_policy)."""
from . import aliengo


def build_robot_model(asset):
    """lsim_robot_model for cfg.asset (LR:1135-1219): the hand-checked Aliengo table, a URDF file if `asset.file` resolves to one, or a
    stored table of robots/tables/ (go1, go2, a1) chosen by `asset.name`."""
    import os
    from . import urdf
    pats = dict(penalize_contacts_on=tuple(asset.penalize_contacts_on), terminate_after_contacts_on=tuple(asset.terminate_after_contacts_on),
                foot_name=asset.foot_name)
    if asset.name == "aliengo":
        return aliengo.build_model(pats["penalize_contacts_on"], pats["terminate_after_contacts_on"], pats["foot_name"])
    path = str(asset.file).replace("{LEGGED_GYM_ROOT_DIR}", os.environ.get("LEGGED_GYM_ROOT_DIR", ""))
    if path and os.path.isfile(path):
        return urdf.build_model(path, **pats)[0]
    return urdf.build_model_from_table(asset.name, **pats)[0]
