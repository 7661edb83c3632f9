"""Static vector-instruction count of every phase of kernels A / B (diagnostics).  Compiles lsim_hip.hip to assembly with -DLS_PHASE_MARKS
(every LS_PHASE / LS_COLLECTIVE site leaves a `; LS_MARK <line>` comment behind it) and counts the v_* / ds_* / global_* instructions laid out
between consecutive marks.  Layout order is not execution order and loop bodies count once, so read it as "how much code does this phase
hold", next to tools/phase_profile.py's measured time shares.   usage: python tools/phase_static.py"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaacgymloco_amd.csrc import build as B   # noqa: E402

out = "/tmp/lsim_marks.s"
flags = [f for f in B.FLAGS + B.SIM_FLAGS if f not in ("-shared", "-fPIC")]
subprocess.check_call([os.environ.get("HIPCC", "hipcc")] + flags + ["-DLS_PHASE_MARKS", "--cuda-device-only", "-S",
                                                                     os.path.join(B.HERE, "lsim_hip.hip"), "-o", out])
src = open(os.path.join(B.HERE, "ls_kernels.h")).read().splitlines()
kernel = None
counts = []
cur = {"v": 0, "ds": 0, "g": 0, "s": 0}
for line in open(out):
    m = re.match(r"^(_Z\d+lsim_k_step_[ab])\w*:", line)
    if m:
        kernel = m.group(1)[-1]
        cur = {"v": 0, "ds": 0, "g": 0, "s": 0}
        continue
    if kernel is None:
        continue
    if ".amdhsa_kernel" in line:
        kernel = None
        continue
    t = line.strip()
    m = re.match(r"; LS_MARK (\d+)", t)
    if m:
        counts.append((kernel, int(m.group(1)), cur))
        cur = {"v": 0, "ds": 0, "g": 0, "s": 0}
    elif t.startswith("v_"):
        cur["v"] += 1
    elif t.startswith("ds_"):
        cur["ds"] += 1
    elif t.startswith("global_") or t.startswith("buffer_"):
        cur["g"] += 1
    elif t.startswith("s_") and not t.startswith("s_waitcnt") and not t.startswith("s_nop"):
        cur["s"] += 1
print("kernel line  valu   lds  vmem  salu  phase")
for k, ln, c in counts:
    print(f"{k:>6s} {ln:4d} {c['v']:5d} {c['ds']:5d} {c['g']:5d} {c['s']:5d}  {src[ln - 1].strip()[:110]}")
