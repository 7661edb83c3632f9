"""The modelling-distance bounds of DESIGN.md section 4 re-measured under TRAINED policies (VERDICT r5 task 2): a trained stairs policy spends its time
where random robots rarely are (pitched trunk over a tread edge, calves brushing risers), so bounds taken under N(0, sigma) actions say little about it.

Input: the checkpoint tools/train_probe.py writes on the GPU box (actor-critic + the simulator's curriculum state after 1000 iterations).  The same policy
(mean actions, torch on the CPU) is run CLOSED LOOP through the CPU oracle's variants from identical initial states, same seeds / commands / pushes:
    shipped   64 sphere-swept collision points, at most 8 contacts, TGS-4            (liborc.so: what the product's kernels implement)
    cap32     the same points, cap lifted to 32                                      (liborc_cap32.so)
    shapes    574 points on the TRUE collision shapes (tests/dense_shapes.py), cap 32  (liborc_shapes.so)
    pgs       the shipped points on the velocity-level PGS solver
and, on the `shipped` run only, at every step the clearance of a dense sampling of the trunk's true box from the terrain (orc_body_clearance): terrain that
is inside the box while the simulated base reports no contact passed BETWEEN the shipped model's trunk sample points.
usage: python tools/trained_policy_physics.py CHECKPOINT.pt [N] [steps] [out.json]     (container: needs oracle/, tests/dense_shapes.py; no GPU)"""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dense_shapes as D  # noqa: E402
from isaacgymloco_amd.envs import config as C, lsim_config as LC, terrain as T  # noqa: E402
from isaacgymloco_amd.envs.legged_robot import build_robot_model  # noqa: E402
from isaacgymloco_amd.learn.bench_train import train_cfg_dict  # noqa: E402
from isaacgymloco_amd.learn.modules import HIMActorCritic  # noqa: E402
from oracle import oracle  # noqa: E402

ck = torch.load(sys.argv[1], weights_only=False)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 1200
out_path = sys.argv[4] if len(sys.argv) > 4 else None
task = ck["task"]
torch.set_num_threads(4)
pol = train_cfg_dict(task)["policy"]
ac = HIMActorCritic(270, 238, 45, 12, **pol)
ac.load_state_dict(ck["model_state_dict"])
ac.eval()
levels = ck["env_state"]["terrain_levels"].numpy()


def make(variant):
    cfg = C.TASKS[task][0]()
    ter = T.Terrain(cfg.terrain, N, seed=1)
    base_model = build_robot_model(cfg.asset)
    if variant == "pgs":
        os.environ["LSIM_SOLVER"] = "pgs"
    lc = LC.make_lsim_config(cfg, num_envs=N, terrain=ter, model=base_model, seed=11)
    os.environ.pop("LSIM_SOLVER", None)
    lib, model = None, base_model
    if variant == "cap32":
        lib = oracle.variant("orc_cap32")
    elif variant == "shapes":
        lib = oracle.variant("orc_shapes", D.DEFINES)
        model = D.build_dense_model(oracle.variant_structs(D.DEFINES), penalize_contacts_on=tuple(cfg.asset.penalize_contacts_on),
                                    terminate_after_contacts_on=tuple(cfg.asset.terminate_after_contacts_on), foot_name=cfg.asset.foot_name)
    sim = oracle.OracleSim(lc, model, ter.heightsamples, ter.env_origins, library=lib)
    # the robots start on the terrain levels the trained run had reached (every 16th env of its 4096), types by env index as always (LR:1234)
    lv = levels[:: max(len(levels) // N, 1)][:N].astype(np.int64)
    sim.buf["terrain_levels"][...] = lv
    ty = sim.buf["terrain_types"]
    sim.buf["env_origins"][...] = ter.env_origins[lv, ty]
    sim.reset_all()
    return sim, base_model


trunk = np.array([[*p, r] for b, p, r in D.dense_points() if b == 0], dtype=np.float32)       # the base link's true shapes: trunk box + rotor capsules


def rollout(variant):
    sim, model = make(variant)
    pen = [i for i in range(17) if (model.penalised_body_mask >> i) & 1]
    term = [i for i in range(17) if (model.termination_body_mask >> i) & 1]
    acc = dict(term=0.0, base=0.0, coll=0.0, cap=0.0, resets=0.0, unseen=0.0, unseen_1cm=0.0, deepest=0.0, nonfinite=0)
    by_body = np.zeros(17)
    clear = np.zeros(N, np.float32)
    L = sim._L
    warm = 200                                   # the first 4 s: robots settle from the reset pose
    t0 = time.time()
    with torch.inference_mode():
        sim.step(np.zeros((N, 12), np.float32))
        for t in range(STEPS):
            a = ac.act_inference(torch.from_numpy(np.array(sim.buf["obs"]))).numpy()
            sim.step(a)
            if t < warm:
                continue
            cf = np.array(sim.buf["contact_forces"])
            rst, tout = np.array(sim.buf["reset"]).astype(bool), np.array(sim.buf["time_out"]).astype(bool)
            base_hit = (np.linalg.norm(cf[:, term, :], axis=-1) > 1.0).any(1)
            acc["term"] += float((rst & ~tout).sum()); acc["resets"] += float(rst.sum())
            acc["base"] += float(base_hit.sum())
            acc["coll"] += float((np.linalg.norm(cf[:, pen, :], axis=-1) > 0.1).sum())
            by_body += (np.linalg.norm(cf, axis=-1) > 0.1).sum(0)
            acc["cap"] += float((np.array(sim.buf["contact_count"])[:, 0] > 8).sum())
            if variant == "shipped":
                # (post-step body states: envs that reset in this step show the fresh pose -- excluded)
                assert L.orc_body_clearance(sim._h, 0, trunk.ctypes.data_as(ctypes.c_void_p), len(trunk), clear.ctypes.data_as(ctypes.c_void_p)) == 0
                inside = (clear < 0.0) & ~rst & (np.linalg.norm(cf[:, 0, :], axis=-1) == 0.0)
                acc["unseen"] += float(inside.sum()); acc["unseen_1cm"] += float((inside & (clear < -0.01)).sum())
                if inside.any():
                    acc["deepest"] = max(acc["deepest"], float(-clear[inside].min()))
    es = N * (STEPS - warm)
    out = dict(variant=variant, env_steps=es, wall_s=round(time.time() - t0, 1),
               terminations_not_timeout_per_env_step=acc["term"] / es, base_contact_per_env_step=acc["base"] / es,
               collision_count_per_env_step=acc["coll"] / es, contact_cap_hits_per_env_step=acc["cap"] / es, resets_per_env_step=acc["resets"] / es,
               mean_terrain_level_end=float(np.array(sim.buf["terrain_levels"]).mean()), nonfinite_env_steps=int(np.array(sim.buf["nonfinite"])[0]),
               # |F| > 0.1 N per env-step by link kind (base; hips, thighs, calves, feet summed over the four legs): which links the collision count comes from
               contact_rate_by_link_kind={"base": by_body[0] / es, "hip": by_body[[1, 5, 9, 13]].sum() / es, "thigh": by_body[[2, 6, 10, 14]].sum() / es,
                                          "calf": by_body[[3, 7, 11, 15]].sum() / es, "foot": by_body[[4, 8, 12, 16]].sum() / es})
    if variant == "shipped":
        out.update(trunk_inside_terrain_unseen_per_env_step=acc["unseen"] / es, trunk_inside_deeper_than_1cm_per_env_step=acc["unseen_1cm"] / es,
                   deepest_unseen_trunk_penetration_m=acc["deepest"], dense_trunk_points=int(len(trunk)))
    sim.close()
    return out


res = {"task": task, "checkpoint_iterations": ck.get("iterations"), "num_envs": N, "steps": STEPS, "eval_hip_4096": ck.get("eval_hip"), "variants": []}
for v in (sys.argv[5].split(",") if len(sys.argv) > 5 else ("shipped", "cap32", "shapes", "pgs")):
    r = rollout(v)
    res["variants"].append(r)
    print(json.dumps(r), flush=True)
if out_path:
    json.dump(res, open(out_path, "w"), indent=1)
