"""Marginal cost of every (idempotent) phase of kernel A: one build per phase with -DLS_EXP_TWICE=<source line of the phase site>, in which
that phase runs twice; the change of the kernel time against the product build is what the phase costs.  (Shader-clock instrumentation
perturbs the schedule and static counts ignore stalls; running a phase a second time on the same inputs does neither.)
  python tools/phase_cost.py --build      (CPU: compiles the variants into isaacgymloco_amd/csrc/variants/, ~40 s each)
  python tools/phase_cost.py              (GPU box: env-only bench per variant, table of differences)"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaacgymloco_amd.csrc import build as B   # noqa: E402

VAR = os.path.join(B.HERE, "variants")
PAT = re.compile(r"LS_TORQUES_KINEMATICS\(\);|LS_PHASE\(ph_(body_inertia|leg_composite|leg_block|leg_schur|base_assemble|base_factor|free_leg|free_base|"
                 r"free_finish|rows|apply_impulses|body_states_all|termination|post_state)|LS_COLLECTIVE\(wc_compact|LS_PHASE\(wc_delassus|LS_KINEMATICS\(\);|"
                 r"ph_heights\(cx, sh, lane, env, true\); ph_base")


def sites():
    out = []
    src = open(os.path.join(B.HERE, "ls_kernels.h")).read().splitlines()
    start = next(i for i, l in enumerate(src) if "LS_WAVE_FN void ls_wave_step_a" in l)
    for i, l in enumerate(src[start:], start + 1):
        if l.startswith("}"):
            break
        if PAT.search(l) and not l.lstrip().startswith("#") and not l.lstrip().startswith("//"):
            out.append((i, l.strip()[:90]))
    # phases that are not idempotent as written have explicit probes in ls_kernels.h (second pass with a zero step / harmless double write)
    out += [(9001, "ph_integrate (probe: second pass with dt = 0)"), (9002, "ph_reward_terms (probe)"), (9003, "ph_callback (probe)")]
    return out


def bench(lib):
    env = dict(os.environ)
    if lib:
        env["LSIM_LIB"] = lib
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "env", "--steps", "300", "--warmup", "50", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True)
    return json.loads(r.stdout.strip().splitlines()[-1])["kernel_a_ms"]


if __name__ == "__main__":
    os.makedirs(VAR, exist_ok=True)
    if "--build" in sys.argv:
        for line, text in sites():
            out = os.path.join(VAR, f"liblsim_twice{line}.so")
            B.build_variant(out, [f"-DLS_EXP_TWICE={line}"])
            print("built", out, flush=True)
        sys.exit(0)
    base = min(bench(None), bench(None))
    print(f"product build: kernel A {base * 1e3:.1f} us")
    tot = 0.0
    for line, text in sites():
        lib = os.path.join(VAR, f"liblsim_twice{line}.so")
        if not os.path.exists(lib):
            continue
        t = min(bench(lib), bench(lib))
        tot += t - base
        print(f"line {line:3d}  +{(t - base) * 1e3:6.2f} us  {100 * (t - base) / base:5.1f} %  {text}", flush=True)
    print(f"sum of the listed phases: {tot * 1e3:.1f} us of {base * 1e3:.1f}")
