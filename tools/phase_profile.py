"""Per-phase shader-clock profile of kernels A and B (diagnostics; not part of the product build).

Builds isaacgymloco_amd/csrc/variants/liblsim_phasetiming.so with -DLS_PHASE_TIMING (every LS_PHASE site adds its elapsed
s_memtime ticks, lane 0 of each wave, to a device-side accumulator), runs the env-only loop and prints the mean ticks per wave
per step for each phase site (source line of ls_kernels.h).  Run on the GPU box:  python tools/phase_profile.py [task] [N]
The instrumentation itself costs ~10 % (MI355X_MICROARCH.md), so read the numbers as shares, not absolutes.
"""
import ctypes
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "isaacgymloco_amd", "csrc")
OUT = os.path.join(CSRC, "variants", "liblsim_phasetiming.so")


def build():
    from isaacgymloco_amd.csrc import build as B
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    return B.build_variant(OUT, ["-DLS_PHASE_TIMING"])


def sites():
    """line number -> phase text for every LS_PHASE / LS_COLLECTIVE site of the two drivers"""
    src = open(os.path.join(CSRC, "ls_kernels.h")).read().splitlines()
    base = {}
    out = {}
    for i, l in enumerate(src, 1):
        if "constexpr int ls_line0 = __LINE__" in l:
            base[len(base)] = i - (96 if "- 96" in l else 0)
    def base_for(i):
        cands = [b for k, b in base.items() if (b if k == 0 else b + 96) <= i]
        return cands[-1]
    for i, l in enumerate(src, 1):
        m = re.search(r"LS_(PHASE|COLLECTIVE)\((.*)\);", l) or re.search(r"(LS_)(TORQUES_KINEMATICS|KINEMATICS)\(\);", l)
        if m and not l.lstrip().startswith("#") and base and i > min(base.values()):
            out[(i - base_for(i)) & 127] = m.group(2)[:70]
    return out


# lanes with work in each phase, by construction of the lane roles (ls_physics.h / ls_post.h headers); "R" = one lane per constraint row
# (3 per contact + 1 per joint-limit row: 12-36), "P" = one lane per collision point of the model (64 for Aliengo)
NOMINAL_LANES = [("KINEMATICS", "48 (12 matrix elements x 4 legs)"), ("ph_torques", "12"), ("ph_body_inertia", "17"), ("ph_leg_composite", "24"),
                 ("ph_leg_block", "36"), ("ph_leg_schur", "24"), ("ph_base_assemble", "42"), ("ph_base_factor", "6"), ("ph_free_leg", "4 + 6"),
                 ("ph_free_base", "6"), ("ph_free_finish", "18 + P"), ("ph_collide", "P"), ("wc_compact_contacts", "P + 12"), ("ph_rows", "R"),
                 ("wc_delassus_pgs", "R"), ("ph_apply_impulses", "18 + 17"), ("ph_contact_forces", "17"), ("ph_integrate", "13"),
                 ("ph_body_states_all", "17"), ("ph_store_sim_state", "51"), ("ph_load_a", "64"), ("ph_post_state", "8"), ("ph_callback", "3"),
                 ("ph_heights", "64 (187 + 63 points)"), ("ph_termination", "1"), ("ph_reward_terms", "one per active term (21)"),
                 ("ph_reward_total", "1"), ("ph_build_obs", "64"), ("ph_term_outputs", "64"), ("ph_load_b", "64"), ("ph_b_store", "64"),
                 ("ph_b_reset_state", "14"), ("ph_b_reset_store", "29"), ("ph_b_episode_stats", "51"), ("ph_b_housekeeping", "53 (env 0 only)")]


def nominal_lanes(text):
    for key, lanes in NOMINAL_LANES:
        if key in text:
            return lanes
    return "?"


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    if "--build-only" in sys.argv:
        print(build()); sys.exit(0)
    from isaacgymloco_amd.csrc.build import variant_is_stale
    if variant_is_stale(OUT):
        build()
    os.environ["LSIM_LIB"] = OUT
    import torch
    from isaacgymloco_amd.envs import config as C
    from isaacgymloco_amd.envs.legged_robot import LeggedRobot
    task = sys.argv[1] if len(sys.argv) > 1 else "aliengo"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    cfg = C.TASKS[task][0]()
    cfg.env.num_envs = N
    env = LeggedRobot(cfg, sim_device="cuda:0", seed=1)
    env.reset()
    env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
    acts = [torch.randn(N, 12, device="cuda:0") for _ in range(16)]
    L = ctypes.CDLL(OUT)
    t = (ctypes.c_ulonglong * 128)(); c = (ctypes.c_ulonglong * 128)(); by = (ctypes.c_ulonglong * (3 * 129))()
    for i in range(50):
        env.step_device(acts[i % 16])
    L.lsim_debug_read_phase_ticks(t, c)
    L.lsim_debug_read_phase_ticks_by(by)
    K = 200
    for i in range(K):
        env.step_device(acts[i % 16])
    L.lsim_debug_read_phase_ticks(t, c)
    L.lsim_debug_read_phase_ticks_by(by)
    names = sites()
    tot_a = sum(t[s] for s in range(96)); tot_b = sum(t[s] for s in range(96, 128))
    print(f"task {task} N {N}: mean ticks per wave per step: kernel A {tot_a / (K * N):.0f}, kernel B {tot_b / (K * N):.0f}")
    for s in range(128):
        if c[s]:
            tot = tot_a if s < 96 else tot_b
            nm = names.get(s, "?")
            kinds = " ".join(f"{by[k * 129 + s] / max(by[k * 129 + 128], 1):7.0f}" for k in range(3)) if s < 96 else ""
            print(f"site {s:2d} calls/step {c[s] / (K * N):4.1f} ticks/step {t[s] / (K * N):8.0f} {100.0 * t[s] / tot:5.1f}%  [<=3 contacts / >=6 / resetting: {kinds}]  busy lanes {nominal_lanes(nm):<28s} {nm}")
    tot_by = [sum(by[k * 129 + s] for s in range(96)) / max(by[k * 129 + 128], 1) for k in range(3)]
    print(f"kernel A by kind of wave: at most 3 contacts {tot_by[0]:.0f} ticks ({by[128] / K:.0f} waves per step), 6 or more {tot_by[1]:.0f} ({by[129 + 128] / K:.0f}), "
          f"resetting {tot_by[2]:.0f} ({by[2 * 129 + 128] / K:.0f})")
